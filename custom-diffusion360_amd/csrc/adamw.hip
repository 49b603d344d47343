// AdamW on fp32 master weights for the fine-tuning step (configs/train_co3d_concept.yaml: optimizer_config AdamW; the reference's loop is
// Lightning's optimizer.step(), main.py): ONE pass over the trainable parameters instead of torch's per-tensor casts + ten multi-tensor
// passes -- reads the bf16 gradient, the fp32 master / exp_avg / exp_avg_sq, writes the three fp32 states and the refreshed bf16 parameter.
// HBM-bound: 2 (grad) + 3 x (4 + 4) (states) + 2 (bf16 parameter) = 28 bytes per element; 66.7 M trainable values at SDXL size = 1.87 GB.
// The trainable tensors are scattered (gradients are whatever autograd allocated), so a launch takes up to 64 (gradient, parameter) pointer
// pairs by value together with each tensor's end offset in the flat state buffers; a block finds its tensor by bisection.
// The step count lives on the device (hipGraph replays advance it): cd360_adamw_tick increments it, the update kernel reads it.
#include "cd360_common.h"
#include "cd360_adamw.h"

namespace {

struct AdamwParams {
  const uint16_t* grad[CD360_ADAMW_MAX_TENSORS];
  uint16_t* param[CD360_ADAMW_MAX_TENSORS];
  long begin[CD360_ADAMW_MAX_TENSORS];  // offset of tensor t in the flat state buffers (multiple of 8)
  long numel[CD360_ADAMW_MAX_TENSORS];
  long vec_end[CD360_ADAMW_MAX_TENSORS];  // cumulative number of 8-element vectors up to and including tensor t (this launch's index space)
  float lr[CD360_ADAMW_MAX_TENSORS], wd[CD360_ADAMW_MAX_TENSORS];
  float* master;
  float* exp_avg;
  float* exp_avg_sq;
  const float* step;
  float beta1, beta2, eps;
  int n;
};

__global__ void adamw_tick_kernel(float* step) { *step += 1.f; }

__global__ __launch_bounds__(256) void adamw_kernel(const AdamwParams p) {
  const float t = *p.step;
  const float bc1 = 1.f - powf(p.beta1, t), bc2_sqrt = sqrtf(1.f - powf(p.beta2, t));
  const long total = p.vec_end[p.n - 1];
  for (long vid = (long)blockIdx.x * blockDim.x + threadIdx.x; vid < total; vid += (long)gridDim.x * blockDim.x) {
    int lo = 0, hi = p.n - 1;  // first tensor whose vec_end > vid
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p.vec_end[mid] > vid) hi = mid; else lo = mid + 1;
    }
    const long e0 = (vid - (lo ? p.vec_end[lo - 1] : 0)) * 8;  // first element of this vector inside tensor lo
    const long left = p.numel[lo] - e0;
    const int cnt = left < 8 ? (int)left : 8;
    const float lr = p.lr[lo], decay = 1.f - lr * p.wd[lo], step_size = lr / bc1;
    const uint16_t* g = p.grad[lo] + e0;
    uint16_t* w = p.param[lo] + e0;
    const long s = p.begin[lo] + e0;
    float gv[8], mv[8], vv[8], xv[8];
    if (cnt == 8 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
      const u32x4 gq = *reinterpret_cast<const u32x4*>(g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gv[2 * e] = bf16lo_to_f32(gq[e]);
        gv[2 * e + 1] = bf16hi_to_f32(gq[e]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(p.exp_avg + s + 4 * h), v4 = *reinterpret_cast<const f32x4*>(p.exp_avg_sq + s + 4 * h),
                    x4 = *reinterpret_cast<const f32x4*>(p.master + s + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          mv[4 * h + e] = m4[e];
          vv[4 * h + e] = v4[e];
          xv[4 * h + e] = x4[e];
        }
      }
    } else {
      for (int e = 0; e < cnt; ++e) {
        gv[e] = bf16lo_to_f32((uint32_t)g[e]);
        mv[e] = p.exp_avg[s + e];
        vv[e] = p.exp_avg_sq[s + e];
        xv[e] = p.master[s + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (e < cnt) {  // torch.optim.adamw._single_tensor_adamw, in its order
        xv[e] *= decay;
        mv[e] += (gv[e] - mv[e]) * (1.f - p.beta1);
        vv[e] = vv[e] * p.beta2 + (1.f - p.beta2) * gv[e] * gv[e];
        const float denom = sqrtf(vv[e]) / bc2_sqrt + p.eps;
        xv[e] -= step_size * (mv[e] / denom);
      }
    }
    if (cnt == 8 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 m4, v4, x4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m4[e] = mv[4 * h + e];
          v4[e] = vv[4 * h + e];
          x4[e] = xv[4 * h + e];
        }
        *reinterpret_cast<f32x4*>(p.exp_avg + s + 4 * h) = m4;
        *reinterpret_cast<f32x4*>(p.exp_avg_sq + s + 4 * h) = v4;
        *reinterpret_cast<f32x4*>(p.master + s + 4 * h) = x4;
      }
      u32x4 wq;
#pragma unroll
      for (int e = 0; e < 4; ++e) wq[e] = pack_bf16x2(xv[2 * e], xv[2 * e + 1]);
      *reinterpret_cast<u32x4*>(w) = wq;
    } else {
      for (int e = 0; e < cnt; ++e) {
        p.exp_avg[s + e] = mv[e];
        p.exp_avg_sq[s + e] = vv[e];
        p.master[s + e] = xv[e];
        w[e] = (uint16_t)(pack_bf16x2(xv[e], 0.f) & 0xffffu);
      }
    }
  }
}

}  // namespace

// *step += 1 (one thread): call once per optimisation step, before the cd360_adamw_bf16 launches of that step
extern "C" int cd360_adamw_tick(void* step, void* stream) {
  if (!step) return CD360_ERR_ARG;
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (float*)step);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// One AdamW update (torch.optim.AdamW semantics: decoupled weight decay, bias correction with the step count *step, no amsgrad) of
// n <= CD360_ADAMW_MAX_TENSORS tensors: for tensor t, grads[t] / params[t] are bf16 device pointers to numel[t] values (gradient read,
// parameter OVERWRITTEN with the rounded new master value), begin[t] (a multiple of 8) its offset in the flat fp32 state buffers
// master / exp_avg / exp_avg_sq (16-byte aligned), lr[t] / wd[t] its learning rate and weight decay.  All arrays are HOST arrays.
extern "C" int cd360_adamw_bf16(int n, const void* const* grads, void* const* params, const int64_t* begin, const int64_t* numel, const float* lr,
                                const float* wd, void* master, void* exp_avg, void* exp_avg_sq, const void* step, float beta1, float beta2, float eps,
                                void* stream) {
  if (n <= 0 || n > CD360_ADAMW_MAX_TENSORS || !grads || !params || !begin || !numel || !lr || !wd || !master || !exp_avg || !exp_avg_sq || !step)
    return CD360_ERR_ARG;
  if (((uintptr_t)master | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 || (uintptr_t)step % 4) return CD360_ERR_ARG;
  AdamwParams p;
  long vecs = 0;
  for (int t = 0; t < n; ++t) {
    if (!grads[t] || !params[t] || numel[t] <= 0 || begin[t] < 0) return CD360_ERR_ARG;
    if (begin[t] % 8 || ((uintptr_t)grads[t] | (uintptr_t)params[t]) % 2) return CD360_ERR_SHAPE;
    p.grad[t] = (const uint16_t*)grads[t];
    p.param[t] = (uint16_t*)params[t];
    p.begin[t] = begin[t];
    p.numel[t] = numel[t];
    vecs += (numel[t] + 7) / 8;
    p.vec_end[t] = vecs;
    p.lr[t] = lr[t];
    p.wd[t] = wd[t];
  }
  p.master = (float*)master; p.exp_avg = (float*)exp_avg; p.exp_avg_sq = (float*)exp_avg_sq; p.step = (const float*)step;
  p.beta1 = beta1; p.beta2 = beta2; p.eps = eps; p.n = n;
  const long blocks = (vecs + 255) / 256;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(blocks > 256L * 16 ? 256L * 16 : blocks)), dim3(256), 0, (hipStream_t)stream, p);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
