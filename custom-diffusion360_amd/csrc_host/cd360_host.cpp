// Host-side glue ABOVE the C ABI (include/cd360_hip.h), in C++ where the Python glue was the bottleneck: the autograd node of nn.Linear /
// F.linear on the fine-tuning path (sgm/modules/attention.py:89-115,323-329,368-372,422,515-516,748,786 of the reference train through
// torch's own Linear autograd).  cd360/grad.py::LinearFn is the same node in Python -- ~90 us of interpreter time per Linear forward +
// backward (Function.apply, shape bookkeeping, ctypes marshalling), 704 Linears per SDXL-size step, which left an eagerly launched step
// host-bound on slower hosts.  This node does the same launches (cd360_gemm_bf16 forward and data gradient, cd360_gemm_tn_bf16 weight
// gradient) through function pointers taken from the ALREADY LOADED libcd360_hip.so -- no kernel lives here, nothing is computed on the host,
// and without the library the module refuses to initialise.  Built by __graft_entry__.build() with g++ against torch's headers.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>

#include <mutex>
#include <unordered_map>

namespace {

using gemm_fn = int (*)(const void*, const void*, void*, int64_t, int, int, int64_t, int64_t, int64_t, const void*, const void*, int64_t, const void*, int, int,
                        float, const void*, void*, int, void*);
using gemm_tn_fn = int (*)(const void*, const void*, void*, int64_t, int, int, int64_t, int64_t, int, void*, void*);
using ws_fn = int64_t (*)(int64_t, int, int);

gemm_fn p_gemm = nullptr;
gemm_tn_fn p_gemm_tn = nullptr;
ws_fn p_tn_ws = nullptr;

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

bool gemm_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw) {
  return M > 0 && K > 0 && N > 0 && K % 64 == 0 && N % 16 == 0 && lda % 8 == 0 && ldw % 8 == 0 && M < (1LL << 31) && (M + 256) * lda * 2 < (1LL << 32) &&
         (N + 256) * ldw * 2 < (1LL << 32);
}

// frozen tensors -> derived copies (fp32 bias, transposed weight), keyed on (impl, version) like cd360/ops.py::_cached
struct Derived {
  uint32_t version;
  c10::weak_intrusive_ptr<c10::TensorImpl> owner;
  at::Tensor value;
};
std::mutex g_mu;
std::unordered_map<const void*, Derived> g_f32, g_wt;

template <class Make>
at::Tensor cached(std::unordered_map<const void*, Derived>& cache, const at::Tensor& t, Make make) {
  if (t.requires_grad() || t.is_view()) return make(t);
  const void* key = t.unsafeGetTensorImpl();
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = cache.find(key);
  if (it != cache.end()) {
    auto live = it->second.owner.lock();
    if (live && live.get() == t.unsafeGetTensorImpl() && it->second.version == t._version()) return it->second.value;
  }
  if (cache.size() > 4096) {
    for (auto i = cache.begin(); i != cache.end();) i = i->second.owner.expired() ? cache.erase(i) : std::next(i);
  }
  at::Tensor v = make(t);
  cache.insert_or_assign(key, Derived{(uint32_t)t._version(), c10::weak_intrusive_ptr<c10::TensorImpl>(t.getIntrusivePtr()), v});
  return v;
}

at::Tensor run_gemm(const at::Tensor& a2, const at::Tensor& w, const at::Tensor& bias32, const at::Tensor& res2) {  // a2 [M, K], w [N, K], res2 [M, N] | undefined
  const int64_t M = a2.size(0), K = a2.size(1), N = w.size(0);
  at::Tensor out = at::empty({M, N}, a2.options());
  const int rc = p_gemm(a2.data_ptr(), w.data_ptr(), out.data_ptr(), M, (int)N, (int)K, a2.stride(0), w.stride(0), N, bias32.defined() ? bias32.data_ptr() : nullptr,
                        res2.defined() ? res2.data_ptr() : nullptr, res2.defined() ? res2.stride(0) : 0, nullptr, 0, 0, 0.f, nullptr, nullptr, 0, cur_stream());
  TORCH_CHECK(rc == 0, "cd360_gemm_bf16 failed with code ", rc);
  return out;
}

struct LinearFn : public torch::autograd::Function<LinearFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor x, at::Tensor weight, c10::optional<at::Tensor> bias_, c10::optional<at::Tensor> res_) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const at::Tensor bias = bias_.has_value() ? *bias_ : at::Tensor(), res = res_.has_value() ? *res_ : at::Tensor();
    const int64_t K = weight.size(1), N = weight.size(0);
    at::Tensor xc = x.detach();
    if (!xc.is_contiguous()) xc = xc.contiguous();
    at::Tensor w = weight.detach();
    if (w.stride(1) != 1 || w.stride(0) % 8) w = w.contiguous();
    at::Tensor b32;
    if (bias.defined()) {
      if (bias.scalar_type() == at::kFloat && bias.is_contiguous() && !bias.requires_grad()) b32 = bias;
      else b32 = cached(g_f32, bias, [](const at::Tensor& t) { return t.detach().to(at::kFloat).contiguous(); });
    }
    at::Tensor r2;
    if (res.defined()) {
      r2 = res.detach();
      if (r2.scalar_type() != at::kBFloat16) r2 = r2.to(at::kBFloat16);
      if (!r2.is_contiguous()) r2 = r2.contiguous();
      r2 = r2.reshape({-1, N});
    }
    at::Tensor out = run_gemm(xc.reshape({-1, K}), w, b32, r2);
    std::vector<int64_t> shape(x.sizes().begin(), x.sizes().end());
    shape.back() = N;
    ctx->save_for_backward({x, weight});
    // (which inputs want a gradient, recorded here: with optional inputs absent the edge indices of needs_input_grad() shift)
    ctx->saved_data["gx"] = x.requires_grad();
    ctx->saved_data["gw"] = weight.requires_grad();
    ctx->saved_data["gb"] = bias.defined() && bias.requires_grad();
    ctx->saved_data["gr"] = res.defined() && res.requires_grad();
    ctx->saved_data["has_bias"] = bias.defined();
    if (bias.defined()) ctx->saved_data["bias_dtype"] = (int64_t)bias.scalar_type();
    return out.reshape(shape);
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const at::Tensor& x = saved[0];
    const at::Tensor& weight = saved[1];
    const int64_t N = weight.size(0), K = weight.size(1);
    at::Tensor dy = grads[0].contiguous();
    at::Tensor dy2 = dy.reshape({-1, N});
    at::Tensor dx, dw, db, dres;
    if (ctx->saved_data["gx"].toBool()) {
      if (gemm_ok(dy2.size(0), K, N, N, N)) {
        at::Tensor wt = cached(g_wt, weight, [](const at::Tensor& t) { return t.detach().t().contiguous(); });
        dx = run_gemm(dy2, wt, at::Tensor(), at::Tensor()).reshape(x.sizes());
      } else {
        dx = at::mm(dy2, weight.detach()).reshape(x.sizes());
      }
    }
    if (ctx->saved_data["gw"].toBool()) {
      at::Tensor x2 = x.detach().reshape({-1, K});
      if (!x2.is_contiguous()) x2 = x2.contiguous();
      const int64_t M = dy2.size(0);
      const bool ok = N % 8 == 0 && K % 8 == 0 && ((uintptr_t)dy2.data_ptr() % 16 == 0) && ((uintptr_t)x2.data_ptr() % 16 == 0) && (M + 64) * N * 2 < (1LL << 31) &&
                      (M + 64) * K * 2 < (1LL << 31);
      if (ok) {
        dw = at::empty({N, K}, x2.options());
        at::Tensor ws = at::empty({std::max<int64_t>(16, p_tn_ws(M, (int)N, (int)K))}, x2.options().dtype(at::kByte));
        const int rc = p_gemm_tn(dy2.data_ptr(), x2.data_ptr(), dw.data_ptr(), M, (int)N, (int)K, N, K, 1, ws.data_ptr(), cur_stream());
        TORCH_CHECK(rc == 0, "cd360_gemm_tn_bf16 failed with code ", rc);
        dw = dw.to(weight.scalar_type());
      } else {
        dw = at::mm(dy2.t(), x2).to(weight.scalar_type());
      }
    }
    if (ctx->saved_data["gb"].toBool())
      db = at::sum(dy2, at::IntArrayRef{0}, false, at::kFloat).to((at::ScalarType)ctx->saved_data["bias_dtype"].toInt());
    if (ctx->saved_data["gr"].toBool()) dres = dy;
    return {dx, dw, db, dres};
  }
};

void init(const std::string& lib_path) {
  void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_NOLOAD);  // the library cd360._lib has loaded: never a second copy, never a fallback
  TORCH_CHECK(h != nullptr, "cd360 host glue: ", lib_path, " is not loaded in this process");
  p_gemm = (gemm_fn)dlsym(h, "cd360_gemm_bf16");
  p_gemm_tn = (gemm_tn_fn)dlsym(h, "cd360_gemm_tn_bf16");
  p_tn_ws = (ws_fn)dlsym(h, "cd360_gemm_tn_workspace_bytes");
  TORCH_CHECK(p_gemm && p_gemm_tn && p_tn_ws, "cd360 host glue: the library lacks the GEMM entry points");
}

at::Tensor linear(const at::Tensor& x, const at::Tensor& weight, const c10::optional<at::Tensor>& bias, const c10::optional<at::Tensor>& res) {
  TORCH_CHECK(p_gemm != nullptr, "cd360 host glue: init(lib_path) first");
  return LinearFn::apply(x, weight, bias, res);
}

uint64_t current_stream() { return (uint64_t)(uintptr_t)cur_stream(); }

}  // namespace

PYBIND11_MODULE(_cd360_host, m) {
  m.def("init", &init);
  m.def("linear", &linear, py::arg("x"), py::arg("weight"), py::arg("bias") = py::none(), py::arg("res") = py::none());
  m.def("current_stream", &current_stream);
}
