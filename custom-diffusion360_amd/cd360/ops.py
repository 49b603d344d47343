"""Torch-tensor front end of the C ABI.  torch owns memory and the stream; every call goes to libcd360_hip.so.

No operator here has a PyTorch/CPU implementation: a CPU tensor (or a missing library) raises."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _host, _lib, routes
from ._lib import Cd360Error, check

_I64x3 = ctypes.c_int64 * 3


# ----------------------------------------------------------------------------------------------- per-kernel timing
# bench.py brackets every C-ABI launch with HIP events ON THE LAUNCH STREAM (torch's current stream is the stream every
# kernel here is enqueued on) and reads them back after the timed region; off by default (zero overhead).
_PROF = None
_PROF_SHAPES = False


def profile_start(shapes: bool = False) -> None:
    """shapes=True: the GEMM entries are keyed per problem shape ("gemm8p[MxNxK]", "gemm_tn[MxNxK]") instead of per kernel."""
    global _PROF, _PROF_SHAPES
    _PROF, _PROF_SHAPES = [], bool(shapes)


def _shape_tag(name: str, M: int, N: int, K: int) -> str:
    return f"{name}[{M}x{N}x{K}]" if _PROF_SHAPES else name


def profile_stop() -> dict:
    """-> {kernel: {"ms": total, "n": launches, "flops": algorithmic flops, "bytes": algorithmic HBM bytes}} (synchronises)."""
    global _PROF
    rec, _PROF = _PROF or [], None
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1, flops, nbytes in rec:
        d = out.setdefault(name, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
        d["ms"] += e0.elapsed_time(e1)
        d["n"] += 1
        d["flops"] += flops
        d["bytes"] += nbytes
    return out


class _timed:
    __slots__ = ("name", "flops", "nbytes", "e0")

    def __init__(self, name, flops=0.0, nbytes=0.0):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        if _PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if _PROF is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _PROF.append((self.name, self.e0, e1, self.flops, self.nbytes))
        return False


# Both helpers return plain Python ints (None for a null pointer): every pointer / stream parameter of the C ABI is typed c_void_p in
# cd360/_lib.py, and ctypes converts an int itself -- building a c_void_p object per argument cost 1.5 us each, 13 000 times per eagerly
# launched fine-tuning step, and `torch.cuda.current_stream().cuda_stream` 9 us per launch (tools/probe/train_host_profile.py: 20 + 21 ms
# of a 147 ms step).  The raw-stream query below is what torch's own extensions use; it follows `torch.cuda.stream(...)` contexts and
# graph captures exactly like current_stream().
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need_gpu(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise Cd360Error("cd360 operators run only on the GPU (HIP extension); got a CPU tensor")
        if t.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("this cd360 HIP operator has no backward; call it under torch.no_grad()")


def _wants_grad(*ts) -> bool:
    """True when the call must be recorded by autograd: the differentiable operators then go through cd360/grad.py
    (torch.autograd.Function around the forward and backward HIP kernels)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# ----------------------------------------------------------------------------------------------- attention
ATTN_PRESCALE = 64 ** -0.5 * 1.4426950408889634  # softmax scale x log2(e): what attention(..., prescaled=True) expects folded into q


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, nk: Optional[int] = None,
              out: Optional[torch.Tensor] = None, want_lse: bool = False, prescaled: bool = False):
    """softmax(q k^T / 8) v per head, projection layouts consumed in place.
    q [b, Nq, H*64], k, v [b, >=Nk, H*64] (last dim contiguous; row / batch strides free, e.g. slices of one merged q|k|v
    projection) -> [b, Nq, H*64].  `nk` limits the keys when k / v are padded.  want_lse=True returns (out, lse [b*H, Nq] fp32),
    the training forward (cd360_attn_fwd_lse_bf16).  Differentiable: under autograd the call is recorded (grad.AttentionFn).
    prescaled=True: q already carries ATTN_PRESCALE (folded into the q projection by the caller; cd360_attn_fwd_prescaled_bf16) --
    forward only."""
    if _wants_grad(q, k, v):
        from . import grad
        assert out is None and not want_lse and not prescaled
        return grad.AttentionFn.apply(q, k, v, heads, nk)
    _need_gpu(q, k, v)
    b, nq, inner = q.shape
    assert inner == heads * 64 and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1 and k.shape[-1] == inner and v.shape[-1] == inner
    nk = k.shape[1] if nk is None else nk
    assert v.shape[1] >= nk and k.shape[1] >= nk
    if out is None:
        out = torch.empty(b, nq, inner, dtype=torch.bfloat16, device=q.device)
    lib = _lib.load()
    strides = (_I64x3(q.stride(0), 64, q.stride(1)), _I64x3(k.stride(0), 64, k.stride(1)),
               _I64x3(v.stride(0), 64, v.stride(1)), _I64x3(out.stride(0), 64, out.stride(1)))
    # self-attention (tiled kernel) and the <= 96-key cross-attention (register-resident kernel) are different kernels: timed apart
    with _timed("attn_self" if nk > 96 else "attn_smallk", 4.0 * b * heads * nq * nk * 64, 2.0 * (2 * b * nq * inner + 2 * b * nk * inner)):
        if prescaled:
            assert not want_lse
            check(lib.cd360_attn_fwd_prescaled_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), b, heads, nq, nk, *strides, _stream()),
                  "cd360_attn_fwd_prescaled_bf16")
            return out
        if want_lse:
            lse = torch.empty(b * heads, nq, dtype=torch.float32, device=q.device)
            check(lib.cd360_attn_fwd_lse_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), b, heads, nq, nk, *strides, 64 ** -0.5, _stream()),
                  "cd360_attn_fwd_lse_bf16")
            return out, lse
        check(lib.cd360_attn_fwd_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), b, heads, nq, nk, *strides, 64 ** -0.5, _stream()),
              "cd360_attn_fwd_bf16")
    return out


def attention_bwd(q, k, v, o, dout, lse, heads: int, nk: Optional[int] = None, need_dq: bool = True, need_dkv: bool = True, out=None):
    """Backward of attention(..., want_lse=True): (dq | None, dk | None, dv | None), bf16, contiguous, rows >= nk of dk / dv zero.
    out = (dq, dk, dv) writes into caller-provided (possibly strided, last dim contiguous) tensors instead, e.g. the three column
    slices of one d(q|k|v) buffer."""
    _need_gpu(q, k, v, o, dout, lse)
    b, nq, inner = q.shape
    nk = k.shape[1] if nk is None else nk
    if dout.stride(2) != 1 or dout.stride(0) % 8 or dout.stride(1) % 8 or dout.data_ptr() % 16:
        dout = dout.contiguous()
    assert dout.dtype == torch.bfloat16 and dout.shape == q.shape and lse.shape == (b * heads, nq) and lse.dtype == torch.float32
    if out is not None:
        dq, dk, dv = out
        assert all(t.dtype == torch.bfloat16 and t.stride(2) == 1 and t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0 for t in out)
        assert dq.shape == q.shape and dk.shape[1] >= nk and dv.shape[1] >= nk
    else:
        dq = torch.empty(b, nq, inner, dtype=torch.bfloat16, device=q.device) if need_dq else None
        dk = torch.zeros(b, k.shape[1], inner, dtype=torch.bfloat16, device=q.device) if need_dkv else None
        dv = torch.zeros(b, v.shape[1], inner, dtype=torch.bfloat16, device=q.device) if need_dkv else None
    ws = torch.empty(b * heads * nq, dtype=torch.float32, device=q.device)
    s3 = lambda t: None if t is None else _I64x3(t.stride(0), 64, t.stride(1))
    flops = 4.0 * b * heads * nq * nk * 64 * ((1.5 if need_dq else 0.0) + (2.0 if need_dkv else 0.0))
    with _timed("attn_bwd", flops, 2.0 * (4 * b * nq * inner + 4 * b * nk * inner)):
        check(_lib.load().cd360_attn_bwd_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(dout), _ptr(lse), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws),
                                             b, heads, nq, nk, s3(q), s3(k), s3(v), s3(o), s3(dout), s3(dq), s3(dk), s3(dv), 64 ** -0.5, _stream()),
              "cd360_attn_bwd_bf16")
    return dq, dk, dv


def self_attention_qkv(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """Self-attention on the merged projection output qkv [b, N, 3*H*64] (q | k | v column slices, read in place).  Under autograd
    the backward kernel writes dq, dk, dv straight into the three slices of ONE d(qkv) buffer (grad.SelfAttentionFn), so the
    projection's backward is a single GEMM and no slice-gradient fills / adds are launched."""
    inner = heads * 64
    assert qkv.shape[-1] == 3 * inner
    if _wants_grad(qkv):
        from . import grad
        return grad.SelfAttentionFn.apply(qkv, heads)
    return attention(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], heads, qkv.shape[1])


def attention_fp8mfma(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, nk: Optional[int] = None, amax=None) -> torch.Tensor:
    """`attention` with both contractions on fp8 (e4m3) MFMA -- BASELINE configs[4], Nk <= 96 only.  Same bf16 tensors; the
    per-tensor scales come from `amax` = (max|q|, max|k|, max|v|), measured here (one host sync) when not given."""
    _need_gpu(q, k, v)
    b, nq, inner = q.shape
    assert inner == heads * 64 and q.dtype == k.dtype == v.dtype == torch.bfloat16
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    nk = k.shape[1] if nk is None else nk
    if amax is None:
        amax = torch.stack([q.abs().amax(), k[:, :nk].abs().amax(), v[:, :nk].abs().amax()]).float().tolist()
    out = torch.empty(b, nq, inner, dtype=torch.bfloat16, device=q.device)
    arr = (ctypes.c_float * 3)(*[float(a) for a in amax])
    with _timed("attn_fp8mfma", 4.0 * b * heads * nq * nk * 64, 2.0 * (2 * b * nq * inner + 2 * b * nk * inner)):
      check(
        _lib.load().cd360_attn_fwd_fp8mfma_bf16(
            _ptr(q), _ptr(k), _ptr(v), _ptr(out), b, heads, nq, nk,
            _I64x3(q.stride(0), 64, q.stride(1)), _I64x3(k.stride(0), 64, k.stride(1)),
            _I64x3(v.stride(0), 64, v.stride(1)), _I64x3(out.stride(0), 64, out.stride(1)),
            64 ** -0.5, arr, _stream()),
        "cd360_attn_fwd_fp8mfma_bf16")
    return out


def memory_efficient_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_bias=None, op=None) -> torch.Tensor:
    """Same contract as xformers.ops.memory_efficient_attention for the layout the reference uses:
    q, k, v contiguous [B*H, N, 64] (sgm/modules/attention.py:393-408)."""
    if attn_bias is not None:
        raise NotImplementedError("attn_bias is not used on this path (attention.py:403)")
    _need_gpu(q, k, v)
    bh, nq, d = q.shape
    nk = k.shape[1]
    if d != 64:
        raise Cd360Error(f"head dim {d} unsupported (SDXL uses 64)")
    dt = q.dtype
    q, k, v = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
    out = torch.empty_like(q)
    check(_lib.load().cd360_attn_fwd_xformers_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), bh, nq, nk, 64 ** -0.5, _stream()),
          "cd360_attn_fwd_xformers_bf16")
    return out.to(dt)


# ----------------------------------------------------------------------------------------------- rays / projection
def patch_rays(cams: torch.Tensor, xs: torch.Tensor, ys: torch.Tensor) -> torch.Tensor:
    _need_gpu(cams, xs, ys)
    b, n1, _ = cams.shape
    r = xs.numel()
    rays = torch.empty(b, n1, r * r, 6, dtype=torch.float32, device=cams.device)
    check(_lib.load().cd360_patch_rays(_ptr(cams), _ptr(xs), _ptr(ys), _ptr(rays), b, n1 - 1, r, _stream()), "cd360_patch_rays")
    return rays


def ray_project_index(cams, xs, ys, t, want_points=True, want_grid=True, want_index=True):
    """t: [S] or [hw, S] fp32.  Returns dict(points, grid, x0, y0, mask)."""
    _need_gpu(cams, xs, ys, t)
    b, n1, _ = cams.shape
    n, r = n1 - 1, xs.numel()
    hw, S = r * r, t.shape[-1]
    stride = 0 if t.dim() == 1 else S
    dev = cams.device
    pts = torch.empty(b, hw, S, 3, dtype=torch.float32, device=dev) if want_points else None
    grid = torch.empty(b, n, hw, S, 2, dtype=torch.float32, device=dev) if want_grid else None
    x0 = torch.empty(b, n, hw, S, dtype=torch.int32, device=dev) if want_index else None
    y0 = torch.empty_like(x0) if want_index else None
    mask = torch.empty_like(x0) if want_index else None
    check(_lib.load().cd360_ray_project_index(_ptr(cams), _ptr(xs), _ptr(ys), _ptr(t.contiguous()), stride, b, n, r, S, _ptr(pts), _ptr(grid),
                                             _ptr(x0), _ptr(y0), _ptr(mask), _stream()), "cd360_ray_project_index")
    return dict(points=pts, grid=grid, x0=x0, y0=y0, mask=mask)


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, u: torch.Tensor, eps: float = 1e-5, want_dists: bool = False, inplace: bool = False):
    """Inverse-CDF depth sampling (cd360_sample_pdf; replaces pytorch3d._C.sample_pdf at nerfsd_pytorch3d.py:300-305).
    bins [..., n_bins + 1], weights [..., n_bins], u [..., n_samples] in [0, 1), fp32 -> samples [..., n_samples]
    (inplace=True writes them over u, as pytorch3d does) and, with want_dists, the gaps to the next sample (:306)."""
    _need_gpu(bins, weights, u)
    n_bins, n_samples = weights.shape[-1], u.shape[-1]
    rows = weights.numel() // n_bins
    assert bins.shape[-1] == n_bins + 1 and bins.numel() == rows * (n_bins + 1) and u.numel() == rows * n_samples
    assert bins.dtype == weights.dtype == u.dtype == torch.float32
    if inplace and (want_dists or not u.is_contiguous()):
        raise ValueError("sample_pdf(inplace=True) needs a contiguous u and cannot return dists")
    bins, weights, uc = bins.contiguous(), weights.contiguous(), u.contiguous()
    out = uc if inplace else torch.empty_like(uc)
    dists = torch.empty_like(uc) if want_dists else None
    check(_lib.load().cd360_sample_pdf(_ptr(bins), _ptr(weights), _ptr(uc), _ptr(out), _ptr(dists), float(eps), rows, n_bins, n_samples,
                                       _stream()), "cd360_sample_pdf")
    return (out, dists) if want_dists else out


def feature_gather(xref: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """xref [n_img, r*r, C] (fp32|bf16), grid [n_img, P, 2] fp32 -> [n_img, P, C]."""
    _need_gpu(xref, grid)
    n_img, hw, C = xref.shape
    r = int(round(hw ** 0.5))
    assert r * r == hw and grid.dtype == torch.float32
    xref, grid = xref.contiguous(), grid.contiguous()
    P = grid.shape[1]
    out = torch.empty(n_img, P, C, dtype=xref.dtype, device=xref.device)
    dt = {torch.float32: 0, torch.bfloat16: 1}[xref.dtype]
    check(_lib.load().cd360_feature_gather(_ptr(xref), _ptr(grid), _ptr(out), n_img, P, r, C, dt, _stream()), "cd360_feature_gather")
    return out


# ----------------------------------------------------------------------------------------------- FeatureNeRF
def plucker_features(cams, xs, ys) -> torch.Tensor:
    _need_gpu(cams, xs, ys)
    b, n1, _ = cams.shape
    r = xs.numel()
    out = torch.empty(b, n1 - 1, r * r, 104, dtype=torch.float32, device=cams.device)
    with _timed("plucker_features", 0.0, 4.0 * 104 * b * (n1 - 1) * r * r):
      check(_lib.load().cd360_plucker_features(_ptr(cams), _ptr(xs), _ptr(ys), _ptr(out), b, n1 - 1, r, _stream()), "cd360_plucker_features")
    return out


def plucker_features_bf16(cams, xs, ys) -> torch.Tensor:
    """plucker_features as bf16 rows of 128 (99 values + zero pad): the A operand of the zP table GEMM on gemm()."""
    _need_gpu(cams, xs, ys)
    b, n1, _ = cams.shape
    r = xs.numel()
    out = torch.empty(b, n1 - 1, r * r, 128, dtype=torch.bfloat16, device=cams.device)
    with _timed("plucker_features", 0.0, 2.0 * 128 * b * (n1 - 1) * r * r):
        check(_lib.load().cd360_plucker_features_bf16(_ptr(cams), _ptr(xs), _ptr(ys), _ptr(out), b, n1 - 1, r, _stream()), "cd360_plucker_features_bf16")
    return out


def nerf_k_padded() -> int:
    return _lib.load().cd360_nerf_k_padded()


def nerf_mlp_aggregate(cams, xs, ys, t, Y, zP, lv, cview, Wk, want_logits=False, img_map=None):
    """img_map (int32 [b*n], optional): Y / lv hold only distinct table images [n_tab, hw, ...]; (batch, view) reads image img_map[b*n_idx].
    Differentiable with respect to Y, zP, lv, cview and Wk (grad.NerfAggregateFn; logits and lse are then always returned)."""
    if _wants_grad(Y, zP, lv, cview, Wk):
        from . import grad
        return grad.NerfAggregateFn.apply(cams, xs, ys, t, Y, zP, lv, cview, Wk, img_map)
    _need_gpu(cams, xs, ys, t, Y, zP, lv, cview, Wk, img_map)
    b, n1, _ = cams.shape
    n, r = n1 - 1, xs.numel()
    hw, S = r * r, t.shape[-1]
    C = Y.shape[-1]
    ntab = b * n if img_map is None else Y.shape[0]
    assert Y.shape == (ntab, hw, C) and zP.shape == (b * n, hw, C) and Y.dtype == torch.bfloat16 and zP.dtype == torch.bfloat16
    assert lv.shape == (ntab, hw) and lv.dtype == torch.float32 and cview.shape == (b, n) and cview.dtype == torch.float32
    assert img_map is None or (img_map.dtype == torch.int32 and img_map.numel() == b * n and img_map.is_contiguous())
    assert Wk.shape == (C, nerf_k_padded()) and Wk.dtype == torch.bfloat16
    for x in (Y, zP, lv, cview, Wk):
        assert x.is_contiguous()
    stride = 0 if t.dim() == 1 else S
    g = torch.empty(b, hw * S, C, dtype=torch.bfloat16, device=Y.device)
    logits = torch.empty(b, n, hw * S, dtype=torch.float32, device=Y.device) if want_logits else None
    lse = torch.empty(b, hw * S, 2, dtype=torch.float32, device=Y.device) if want_logits else None
    # two passes (cd360_nerf_mlp_aggregate_ws): the per-(view, sample) geometry, view logit and softmax statistics once, into `ws`, then
    # the per-channel-slice gather / MLP / aggregate reading 32-byte records (the library falls back to the one-pass kernels outside the
    # two-pass envelope or under cd360_tuning.nerf_kernel = 0 | 1)
    lib = _lib.load()
    ws = torch.empty(int(lib.cd360_nerf_ws_bytes(b, n, r, S)), dtype=torch.uint8, device=Y.device)
    with _timed("nerf_mlp_aggregate", 2.0 * b * n * hw * S * 99 * C, 2.0 * C * (2 * b * n * hw + b * hw * S)):
      check(lib.cd360_nerf_mlp_aggregate_ws(_ptr(cams), _ptr(xs), _ptr(ys), _ptr(t.contiguous()), stride, _ptr(Y), _ptr(zP), _ptr(lv),
                                            _ptr(cview), _ptr(Wk), _ptr(img_map), _ptr(g), _ptr(logits), _ptr(lse), b, n, r, S, C, ntab,
                                            _ptr(ws), ws.numel(), _stream()),
          "cd360_nerf_mlp_aggregate_ws")
    return g, logits, lse


def nerf_mlp_aggregate_bwd(cams, xs, ys, t, Y, zP, lv, cview, Wk, img_map, g, lse, dg, scatter: bool = True):
    """Backward kernel of nerf_mlp_aggregate -> (dz [b,n,hw*S,C] bf16, F [b,n,hw*S,112] bf16, dY [tables,hw,C] fp32,
    dlv [tables,hw] fp32, dcview [b,n] fp32, dlogit [b,n,hw*S] fp32).  scatter=False skips the atomic table scatters (dY, dlv and
    dcview come back None; the caller reduces dz / dlogit itself)."""
    _need_gpu(cams, xs, ys, t, Y, zP, lv, cview, Wk, img_map, g, lse, dg)
    b, n1, _ = cams.shape
    n, r = n1 - 1, xs.numel()
    hw, S = r * r, t.shape[-1]
    C = Y.shape[-1]
    dev = Y.device
    dg = dg.contiguous()
    assert dg.dtype == torch.bfloat16 and dg.shape == (b, hw * S, C) and g.shape == dg.shape and lse.shape == (b, hw * S, 2)
    kp = nerf_k_padded()
    dz = torch.empty(b, n, hw * S, C, dtype=torch.bfloat16, device=dev)
    F = torch.empty(b, n, hw * S, kp, dtype=torch.bfloat16, device=dev)
    dY = torch.zeros(Y.shape, dtype=torch.float32, device=dev) if scatter else None
    # dlogit: every 64-channel chunk stores its term into `parts`, the library sums them in chunk order -- no atomics, so the view-logit
    # gradients (and with them a whole fine-tuning step) are bit-reproducible run to run (cd360_nerf_mlp_aggregate_bwd_det)
    dlogit = torch.empty(b, n, hw * S, dtype=torch.float32, device=dev)
    parts = torch.empty(C // 64, b, n, hw * S, dtype=torch.float32, device=dev)
    dlv = torch.zeros(lv.shape, dtype=torch.float32, device=dev) if scatter else None
    dcview = torch.zeros(b, n, dtype=torch.float32, device=dev) if scatter else None
    stride = 0 if t.dim() == 1 else S
    with _timed("nerf_mlp_aggregate_bwd", 2.0 * b * n * hw * S * 99 * C, 2.0 * C * (2 * b * n * hw + (2 + n) * b * hw * S)):
        check(_lib.load().cd360_nerf_mlp_aggregate_bwd_det(_ptr(cams), _ptr(xs), _ptr(ys), _ptr(t.contiguous()), stride, _ptr(Y), _ptr(zP), _ptr(lv),
                                                          _ptr(cview), _ptr(Wk), _ptr(img_map), _ptr(g), _ptr(lse), _ptr(dg), _ptr(dz), _ptr(F),
                                                          _ptr(dY), _ptr(dlogit), _ptr(parts), _ptr(dlv), _ptr(dcview), b, n, r, S, C, _stream()),
              "cd360_nerf_mlp_aggregate_bwd_det")
    return dz, F, dY, dlv, dcview, dlogit


# ----------------------------------------------------------------------------------------------- volume rendering
def volrender(feats, sigma_raw, dists, rgb_raw=None, want_weights=False, sigma_is_raw=True, rgb_is_raw=True):
    """feats [b,hw,S,C] (fp32|bf16), sigma_raw [b,hw,S] fp32, dists [S] or [hw,S] fp32, rgb_raw [b,hw,S,3] fp32|None.
    Differentiable with respect to feats, sigma_raw and rgb_raw (grad.VolRenderFn)."""
    if _wants_grad(feats, sigma_raw, rgb_raw):
        from . import grad
        return grad.VolRenderFn.apply(feats, sigma_raw, dists, rgb_raw, want_weights, sigma_is_raw, rgb_is_raw)
    _need_gpu(feats, sigma_raw, dists, rgb_raw)
    b, hw, S, C = feats.shape
    feats, sigma_raw, dists = feats.contiguous(), sigma_raw.contiguous().float(), dists.contiguous().float()
    if rgb_raw is not None:
        rgb_raw = rgb_raw.contiguous().float()
    dev = feats.device
    rendered = torch.empty(b, hw, C, dtype=feats.dtype, device=dev)
    fg = torch.empty(b, hw, 1, dtype=torch.float32, device=dev)
    alphas = torch.empty(b, hw, S, 1, dtype=torch.float32, device=dev)
    weights = torch.empty(b, hw, S, 1, dtype=torch.float32, device=dev) if want_weights else None
    rgb = torch.empty(b, hw, 3, dtype=torch.float32, device=dev) if rgb_raw is not None else None
    dt = {torch.float32: 0, torch.bfloat16: 1}[feats.dtype]
    stride = 0 if dists.dim() == 1 else S
    with _timed("volrender", 0.0, feats.element_size() * C * b * hw * (S + 1.0)):
      check(_lib.load().cd360_volrender(_ptr(feats), _ptr(sigma_raw), _ptr(rgb_raw), _ptr(dists), stride, _ptr(rendered), _ptr(fg),
                                     _ptr(alphas), _ptr(weights), _ptr(rgb), b, hw, S, C, dt, (0 if sigma_is_raw else 1) | (0 if rgb_is_raw else 2),
                                     _stream()), "cd360_volrender")
    return rendered, fg, alphas, weights, rgb


def volrender_bwd(feats, sigma_raw, dists, rgb_raw, d_rendered, d_fg, d_alphas, d_weights, d_rgb, sigma_is_raw=True, rgb_is_raw=True):
    """Backward of volrender: (d_feats [b,hw,S,C], d_sigma_raw [b,hw,S] fp32, d_rgb_raw [b,hw,S,3] fp32 | None).  Gradients that
    did not arrive are None."""
    _need_gpu(feats, sigma_raw, dists, rgb_raw, d_rendered, d_fg, d_alphas, d_weights, d_rgb)
    b, hw, S, C = feats.shape
    f32 = lambda t, shape: None if t is None else t.reshape(shape).contiguous().float()
    feats, sigma_raw, dists = feats.contiguous(), sigma_raw.contiguous().float(), dists.contiguous().float()
    rgb_raw = None if rgb_raw is None else rgb_raw.contiguous().float()
    if d_rendered is None:
        d_rendered = torch.zeros(b, hw, C, dtype=feats.dtype, device=feats.device)
    d_rendered = d_rendered.contiguous().to(feats.dtype)
    d_fg, d_alphas, d_weights, d_rgb = f32(d_fg, (b, hw)), f32(d_alphas, (b, hw, S)), f32(d_weights, (b, hw, S)), f32(d_rgb, (b, hw, 3))
    d_feats = torch.empty_like(feats)
    d_sigma = torch.empty(b, hw, S, dtype=torch.float32, device=feats.device)
    d_rgb_raw = torch.empty(b, hw, S, 3, dtype=torch.float32, device=feats.device) if rgb_raw is not None else None
    dt = {torch.float32: 0, torch.bfloat16: 1}[feats.dtype]
    stride = 0 if dists.dim() == 1 else S
    with _timed("volrender_bwd", 0.0, feats.element_size() * C * b * hw * (2.0 * S + 1.0)):
        check(_lib.load().cd360_volrender_bwd(_ptr(feats), _ptr(sigma_raw), _ptr(rgb_raw), _ptr(dists), stride, _ptr(d_rendered), _ptr(d_fg),
                                             _ptr(d_alphas), _ptr(d_weights), _ptr(d_rgb), _ptr(d_feats), _ptr(d_sigma), _ptr(d_rgb_raw), b, hw, S, C,
                                             dt, (0 if sigma_is_raw else 1) | (0 if rgb_is_raw else 2), _stream()), "cd360_volrender_bwd")
    return d_feats, d_sigma, d_rgb_raw


def rowdot4(h: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """h [..., C] bf16, w [4, C] fp32 -> [..., 4] fp32 (FeatureNeRF decoder).  Differentiable (grad.RowDot4Fn)."""
    if _wants_grad(h, w):
        from . import grad
        return grad.RowDot4Fn.apply(h, w)
    _need_gpu(h, w)
    C = h.shape[-1]
    assert h.dtype == torch.bfloat16 and h.is_contiguous() and w.shape == (4, C) and w.dtype == torch.float32 and w.is_contiguous()
    rows = h.numel() // C
    out = torch.empty(*h.shape[:-1], 4, dtype=torch.float32, device=h.device)
    with _timed("rowdot4", 0.0, 2.0 * rows * C):
      check(_lib.load().cd360_rowdot4_bf16(_ptr(h), _ptr(w), _ptr(out), rows, C, _stream()), "cd360_rowdot4_bf16")
    return out


def rowdot1(h: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """h [..., C] bf16, w [C] fp32 -> [...] fp32 (cd360_rowdot1_bf16: the view-logit column lv = xref . vf of the reference tables).
    Not differentiable: the training path differentiates vf through grad.NerfRenderFn's own reductions."""
    _need_gpu(h, w)
    C = h.shape[-1]
    assert h.dtype == torch.bfloat16 and h.is_contiguous() and w.shape == (C,) and w.dtype == torch.float32 and w.is_contiguous()
    rows = h.numel() // C
    out = torch.empty(h.shape[:-1], dtype=torch.float32, device=h.device)
    with _timed("rowdot1", 0.0, 2.0 * rows * C):
        check(_lib.load().cd360_rowdot1_bf16(_ptr(h), _ptr(w), _ptr(out), rows, C, _stream()), "cd360_rowdot1_bf16")
    return out


def render_loss_ok(fg, alphas, rgb, op, mask_, want) -> bool:
    """Can cd360_render_loss_f32 serve these tensors (fp32, on the GPU, a one-channel mask)?"""
    ts = [t for t in (fg, alphas, rgb, op, mask_, want) if t is not None]
    from . import routes
    return (not routes.no_train_fusions and all(t.is_cuda and t.dtype == torch.float32 for t in ts) and (mask_ is None or mask_.shape[1] == 1)
            and (rgb is None or (mask_ is not None and want is not None)))


def render_loss(fg, alphas, rgb, op, bgw, mask_, want, den) -> torch.Tensor:
    """The fg / bg / rgb loss terms of one pose block (loss.py:188-207 of the reference) -> [b, 3] fp32; differentiable with respect
    to fg [b, hw, 1], alphas [b, hw, S, 1] and rgb [b, hw, 3] | None (grad.RenderLossFn).  op, bgw [b, hw]; mask_ [b, 1, r, r];
    want [b, 3, r, r]; den [b]."""
    if _wants_grad(fg, alphas, rgb):
        from . import grad
        return grad.RenderLossFn.apply(fg, alphas, rgb, op, bgw, mask_, want, den)
    return _render_loss_fwd(fg, alphas, rgb, op, bgw, mask_, want, den)


def _render_loss_fwd(fg, alphas, rgb, op, bgw, mask_, want, den):
    _need_gpu(fg, alphas, rgb, op, bgw, mask_, want, den)
    b, hw = op.shape
    S = alphas.numel() // (b * hw)
    c = lambda t: None if t is None else t.detach().contiguous()
    fg, alphas, rgb, op, bgw, mask_, want, den = map(c, (fg, alphas, rgb, op, bgw, mask_, want, den))
    assert fg.numel() == b * hw and (rgb is None or rgb.numel() == b * hw * 3)
    out = torch.empty(b, 3, dtype=torch.float32, device=op.device)
    check(_lib.load().cd360_render_loss_f32(_ptr(fg), _ptr(alphas), _ptr(rgb), _ptr(op), _ptr(bgw), _ptr(mask_), _ptr(want), _ptr(den), _ptr(out),
                                            b, hw, S, _stream()), "cd360_render_loss_f32")
    return out


def render_loss_bwd(fg, alphas, rgb, op, bgw, mask_, want, den, g):
    """-> (d_fg, d_alphas, d_rgb | None) in the shapes of fg, alphas, rgb."""
    _need_gpu(fg, alphas, rgb, op, bgw, mask_, want, den, g)
    b, hw = op.shape
    S = alphas.numel() // (b * hw)
    c = lambda t: None if t is None else t.detach().contiguous()
    fgc, alc, rgbc, op, bgw, mask_, want, den = map(c, (fg, alphas, rgb, op, bgw, mask_, want, den))
    d_fg, d_al = torch.empty_like(fgc), torch.empty_like(alc)
    d_rgb = None if rgbc is None else torch.empty_like(rgbc)
    check(_lib.load().cd360_render_loss_bwd_f32(_ptr(fgc), _ptr(alc), _ptr(rgbc), _ptr(op), _ptr(bgw), _ptr(mask_), _ptr(want), _ptr(den),
                                                _ptr(g.contiguous().float()), _ptr(d_fg), _ptr(d_al), _ptr(d_rgb), b, hw, S, _stream()),
          "cd360_render_loss_bwd_f32")
    return d_fg, d_al, d_rgb


def nerf_pack_weights(W1, b1, b2, wv, bv, Wd, kcol):
    """cd360_nerf_pack_weights_bf16: bf16 parameters of one FeatureNeRFEncoding -> (bf16 arena Wf | Wk | Wp, fp32 arena b1 | b2 | vf |
    v_cam | bv | Wd); the caller slices (see the header)."""
    _need_gpu(W1, b1, b2, wv, bv, Wd, kcol)
    C, NK = W1.shape[0], kcol.numel()
    assert all(t.dtype == torch.bfloat16 and t.is_contiguous() for t in (W1, b1, b2, wv, bv, Wd)) and kcol.dtype == torch.int32
    assert W1.shape == (C, C + 198) and wv.numel() == C + 198 and Wd.shape == (4, C) and b1.numel() == C and b2.numel() == C
    wb = torch.empty(C * (C + NK + 128), dtype=torch.bfloat16, device=W1.device)
    wf = torch.empty(7 * C + 104, dtype=torch.float32, device=W1.device)
    check(_lib.load().cd360_nerf_pack_weights_bf16(_ptr(W1), _ptr(b1), _ptr(b2), _ptr(wv), _ptr(bv), _ptr(Wd), _ptr(kcol), _ptr(wb), _ptr(wf),
                                                   C, NK, _stream()), "cd360_nerf_pack_weights_bf16")
    return wb, wf


def nerf_unpack_grads(grads, kpos, C: int, NK: int):
    """cd360_nerf_unpack_grads_bf16: the nine gradients of the packed operands (None = none arrived) -> (dW1 [C, C + 198] bf16, small
    bf16 arena db1 | db2 | dwv | dbv | dWd)."""
    import ctypes
    some = next(g for g in grads if g is not None)
    gs = [None if g is None else g.contiguous() for g in grads]
    _need_gpu(kpos, *gs)
    for g, n_ in zip(gs, (C * C, C * NK, C * 128, C, C, C, 99, 1, 4 * C)):
        assert g is None or (g.numel() == n_ and g.dtype in (torch.float32, torch.bfloat16)), "nerf_unpack_grads: unexpected gradient"
    ptrs = (ctypes.c_void_p * 9)(*[None if g is None else g.data_ptr() for g in gs])
    dts = (ctypes.c_int * 9)(*[1 if (g is not None and g.dtype == torch.bfloat16) else 0 for g in gs])
    dW1 = torch.empty(C, C + 198, dtype=torch.bfloat16, device=some.device)
    small = torch.empty(7 * C + 200, dtype=torch.bfloat16, device=some.device)
    check(_lib.load().cd360_nerf_unpack_grads_bf16(ptrs, dts, _ptr(kpos), _ptr(dW1), _ptr(small), C, NK, _stream()), "cd360_nerf_unpack_grads_bf16")
    return dW1, small


def rowdot4_bwd(d_out: torch.Tensor, h: torch.Tensor, w: torch.Tensor, need_dh: bool = True, need_dw: bool = True):
    """Backward of rowdot4: d_out [..., 4] fp32 -> (dh [..., C] bf16 | None, dw [4, C] fp32 | None) (cd360_rowdot4_bwd_bf16)."""
    _need_gpu(d_out, h, w)
    C = h.shape[-1]
    rows = h.numel() // C
    d2 = d_out.reshape(rows, 4).float().contiguous()
    assert h.dtype == torch.bfloat16 and h.is_contiguous() and w.shape == (4, C) and w.dtype == torch.float32 and w.is_contiguous()
    lib = _lib.load()
    dh = torch.empty_like(h) if need_dh else None
    part = torch.empty(lib.cd360_rowdot4_bwd_slabs(rows), 4, C, dtype=torch.float32, device=h.device) if need_dw else None
    with _timed("rowdot4_bwd", 0.0, 2.0 * rows * C * (int(need_dh) + int(need_dw))):
        check(lib.cd360_rowdot4_bwd_bf16(_ptr(d2), _ptr(h), _ptr(w), _ptr(dh), _ptr(part), rows, C, _stream()), "cd360_rowdot4_bwd_bf16")
    return dh, (None if part is None else part.sum(0))


# ----------------------------------------------------------------------------------------------- GroupNorm (+SiLU)
def gn_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
            out: Optional[torch.Tensor] = None, tile_stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: channels-last bf16 viewed as [N, P, C] (contiguous).  gamma/beta fp32 [C].
    tile_stats: the fp32 [N, slabs, C, 2] per-slab channel sums conv_igemm(..., want_stats=True) returned for this very tensor
    (the statistics read pass is skipped).  Differentiable with respect to x (grad.GroupNormSiluFn)."""
    if _wants_grad(x, gamma, beta):
        from . import grad
        assert out is None
        return grad.GroupNormSiluFn.apply(x, gamma, beta, groups, eps, silu, tile_stats)
    _need_gpu(x, gamma, beta, tile_stats)
    N, P, C = x.shape
    slabs = 0
    if tile_stats is not None:
        assert tile_stats.dtype == torch.float32 and tile_stats.is_contiguous() and tile_stats.shape[0] == N and tile_stats.shape[2:] == (C, 2)
        slabs = tile_stats.shape[1]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    lib = _lib.load()
    ws = torch.empty(lib.cd360_gn_workspace_bytes(N, P, C), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.empty_like(x)
    with _timed("gn_silu", 0.0, 2.0 * (3 if tile_stats is None else 2) * N * P * C):
      check(lib.cd360_gn_silu_bf16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(ws), N, P, C, groups, float(eps), int(silu),
                                   _ptr(tile_stats), slabs, _stream()), "cd360_gn_silu_bf16")
    return out


def gn_silu_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool) -> torch.Tensor:
    """Backward of gn_silu with respect to x: x, dy [N, P, C] bf16 -> dx."""
    _need_gpu(x, dy, gamma, beta)
    N, P, C = x.shape
    dy = dy.contiguous()
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and dy.dtype == torch.bfloat16 and dy.shape == x.shape
    lib = _lib.load()
    ws = torch.empty(lib.cd360_gn_bwd_workspace_bytes(N, P, C), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x)
    with _timed("gn_silu_bwd", 0.0, 2.0 * 6 * N * P * C):
        check(lib.cd360_gn_silu_bwd_bf16(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), _ptr(dx), _ptr(ws), N, P, C, groups, float(eps), int(silu),
                                         _stream()), "cd360_gn_silu_bwd_bf16")
    return dx


# ----------------------------------------------------------------------------------------------- epilogues
def geglu(proj: torch.Tensor) -> torch.Tensor:
    """proj [..., 2*inner] bf16 = [x | gate] -> [..., inner] = x * gelu(gate).  Differentiable (grad.GegluFn)."""
    if _wants_grad(proj):
        from . import grad
        return grad.GegluFn.apply(proj)
    _need_gpu(proj)
    inner = proj.shape[-1] // 2
    assert proj.dtype == torch.bfloat16 and proj.is_contiguous()
    rows = proj.numel() // (2 * inner)
    out = torch.empty(*proj.shape[:-1], inner, dtype=torch.bfloat16, device=proj.device)
    with _timed("geglu", 0.0, 2.0 * 3 * rows * inner):
        check(_lib.load().cd360_geglu_bf16(_ptr(proj), _ptr(out), rows, inner, _stream()), "cd360_geglu_bf16")
    return out


def geglu_bwd(proj: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """Backward of geglu: d(proj) [..., 2*inner] = [dy gelu(gate) | dy x gelu'(gate)]."""
    _need_gpu(proj, dy)
    inner = proj.shape[-1] // 2
    dy = dy.contiguous()
    assert proj.dtype == torch.bfloat16 and proj.is_contiguous() and dy.dtype == torch.bfloat16 and dy.shape == (*proj.shape[:-1], inner)
    rows = proj.numel() // (2 * inner)
    din = torch.empty_like(proj)
    with _timed("geglu_bwd", 0.0, 2.0 * 5 * rows * inner):
        check(_lib.load().cd360_geglu_bwd_bf16(_ptr(proj), _ptr(dy), _ptr(din), rows, inner, _stream()), "cd360_geglu_bwd_bf16")
    return din


def concat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=1) for channels-last bf16 images [N, C, H, W]; returns a channels-last [N, Ca+Cb, H, W].
    Differentiable (grad.ConcatChannelsFn: the backward is two channel slices)."""
    if _wants_grad(a, b):
        from . import grad
        return grad.ConcatChannelsFn.apply(a, b)
    _need_gpu(a, b)
    N, ca, H, W = a.shape
    cb = b.shape[1]
    at, bt = a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and at.is_contiguous() and bt.is_contiguous() and b.shape[0] == N
    out = torch.empty(N, H, W, ca + cb, dtype=torch.bfloat16, device=a.device)
    with _timed("concat_channels", 0.0, 2.0 * 2 * N * H * W * (ca + cb)):
        check(_lib.load().cd360_concat_channels_bf16(_ptr(at), _ptr(bt), _ptr(out), N * H * W, ca, cb, _stream()), "cd360_concat_channels_bf16")
    return out.permute(0, 3, 1, 2)


def concat_gn_stats(sa: torch.Tensor, sb: torch.Tensor) -> torch.Tensor:
    """torch.cat([sa, sb], dim=2) of two GroupNorm slab-statistics tensors [N, slabs, C, 2] fp32 (the statistics of a channel concatenation
    are the concatenated per-channel statistics) on cd360_concat_channels_bf16: a row of C (sum, sumsq) pairs is a row of 4 C bf16 words
    to that kernel -- no torch-issued cat kernel inside a captured step."""
    _need_gpu(sa, sb)
    N, slabs, ca, two = sa.shape
    cb = sb.shape[2]
    assert two == 2 and sb.shape == (N, slabs, cb, 2) and sa.dtype == sb.dtype == torch.float32 and sa.is_contiguous() and sb.is_contiguous()
    out = torch.empty(N, slabs, ca + cb, 2, dtype=torch.float32, device=sa.device)
    check(_lib.load().cd360_concat_channels_bf16(_ptr(sa), _ptr(sb), _ptr(out), N * slabs, 4 * ca, 4 * cb, _stream()), "cd360_concat_channels_bf16")
    return out


# ----------------------------------------------------------------------------------------------- conv3x3 / GEMM (implicit GEMM)
def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight [Cout, Cin, k, k] (k = 3 | 1) or a Linear weight [Cout, Cin] -> the bf16 [Cout, taps*Cin] matrix
    cd360_conv_igemm_bf16 reads.  K order: G = cd360_conv_k_order(Cin, taps) 64-channel chunks per group; group outer, tap middle,
    chunk-in-group inner: k = ((cg*taps + ky*3+kx)*G + j)*64 + ci%64 with ci//64 = cg*G + j."""
    if w.dim() == 2:
        return w.detach().to(torch.bfloat16).contiguous()
    cout, cin, kh, kw = w.shape
    assert cin % 64 == 0 and kh == kw and kh in (1, 3)
    g = _lib.load().cd360_conv_k_order(cin, kh * kw)
    wp = w.detach().reshape(cout, cin // (64 * g), g, 64, kh * kw).permute(0, 1, 4, 2, 3)  # [co, group, tap, chunk-in-group, 64]
    return wp.reshape(cout, -1).to(torch.bfloat16).contiguous()


def conv_igemm(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], N: int, H: int, W: int, taps: int = 9,
               emb: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, want_stats: bool = False, stride: int = 1,
               alg_channels: Optional[tuple] = None, w_dgrad=None):
    """x [N*H*W, Cin] (or [N, H*W, Cin]) channels-last bf16; w_packed [Cout, taps*Cin] bf16; bias fp32 [Cout]; emb bf16 [N, Cout];
    res bf16 [N*Ho*Wo, Cout] -> [N, Ho*Wo, Cout] bf16 = conv3x3 (taps=9; stride 1, or 2 with Ho = H/2, Wo = W/2) or x @ w^T (taps=1)
    + bias + emb[n] + res.
    want_stats=True (H*W % 128 == 0): returns (out, tile_stats) with tile_stats fp32 [N, slabs, Cout, 2] = per pixel slab the channel
    sums / sums of squares of `out`, for gn_silu(out, ..., tile_stats=tile_stats).
    Differentiable with respect to x, emb and res (grad.ConvIgemmFn) when `w_dgrad` is given: a callable returning the packed weight
    of the data-gradient convolution (taps flipped, channels transposed: sgm...util.packed_conv_dgrad).  No weight gradient."""
    if _wants_grad(x, emb, res, w_packed, bias):
        from . import grad
        return grad.conv_igemm(x, w_packed, bias, N, H, W, taps, emb, res, want_stats, stride, alg_channels, w_dgrad)
    _need_gpu(x, w_packed, bias, emb, res)
    cin = x.shape[-1]
    cout = w_packed.shape[0]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() == N * H * W * cin
    assert w_packed.dtype == torch.bfloat16 and w_packed.is_contiguous() and w_packed.shape[1] == taps * cin
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())
    assert emb is None or (emb.dtype == torch.bfloat16 and emb.shape == (N, cout) and emb.stride(1) == 1)  # rows may be strided
    ho, wo = H // stride, W // stride
    assert res is None or (res.dtype == torch.bfloat16 and res.is_contiguous() and res.numel() == N * ho * wo * cout)
    out = torch.empty(N, ho * wo, cout, dtype=torch.bfloat16, device=x.device)
    m = N * ho * wo
    lib = _lib.load()
    stats = None
    if want_stats:
        _lib.query_stream(_stream())
        rows = lib.cd360_conv_stats_rows(N, H, W, cin, cout, taps, stride)  # pixels per slab: the kernel serving this shape decides
        if rows <= 0 or (ho * wo) % rows or (rows < 64 and (ho * wo) % 128):  # slabs must not straddle images (register-staged kernel: 128-pixel tiles)
            raise Cd360Error(f"conv_igemm(want_stats=True): Ho*Wo = {ho * wo} is not a whole number of the kernel's {rows}-pixel slabs")
        stats = torch.empty(N, (ho * wo) // rows, cout, 2, dtype=torch.float32, device=x.device)
    acin, acout = alg_channels or (cin, cout)  # un-padded channel counts for the algorithmic FLOP / byte accounting
    with _timed("conv_igemm", 2.0 * m * taps * acin * acout, 2.0 * (N * H * W * acin + m * acout + taps * acin * acout)):
        check(lib.cd360_conv_igemm_bf16(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(emb), 0 if emb is None else emb.stride(0), _ptr(res), _ptr(out),
                                       N, H, W, cin, cout, taps, stride, _ptr(stats), _stream()), "cd360_conv_igemm_bf16")
    return (out, stats) if want_stats else out


def pack_upsample_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight [Cout, Cin, 3, 3] of an Upsample block -> [4, Cout, 4*Cin] bf16 for conv_up2x: phase (a, b) = (row, column parity of
    the output pixel) sees the source rows {i + a - 1, i + a}; of the three kernel rows ky, those that land on the same source row are
    summed (a = 0: {0}, {1, 2}; a = 1: {0, 1}, {2}), likewise the columns -- fp32 sums, one rounding.  K order as pack_conv_weight with
    tap slot t = 2 ty + tx."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cin % 64 == 0
    w32 = w.detach().float()
    sets = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    g = _lib.load().cd360_conv_k_order(cin, 9)
    phases = []
    for a in (0, 1):
        for b in (0, 1):
            taps = [sum(w32[:, :, ky, kx] for ky in sets[a][ty] for kx in sets[b][tx]) for ty in (0, 1) for tx in (0, 1)]  # 4 x [Cout, Cin]
            we = torch.stack(taps, -1)  # [Cout, Cin, 4]
            wp = we.reshape(cout, cin // (64 * g), g, 64, 4).permute(0, 1, 4, 2, 3)  # [co, group, tap, chunk-in-group, 64]
            phases.append(wp.reshape(cout, -1))
    return torch.stack(phases, 0).to(torch.bfloat16).contiguous()


def conv_up2x(x: torch.Tensor, w_phases: torch.Tensor, bias: Optional[torch.Tensor], N: int, H: int, W: int) -> torch.Tensor:
    """Upsample.forward: nearest 2x + conv3x3 in one launch (cd360_conv_up2x_bf16).  x [N, H*W, Cin] channels-last bf16 (the SOURCE
    image), w_phases from pack_upsample_conv_weight -> [N, 4*H*W, Cout] (the 2H x 2W image).  Forward only."""
    _need_gpu(x, w_phases, bias)
    cin = x.shape[-1]
    cout = w_phases.shape[1]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() == N * H * W * cin
    assert w_phases.dtype == torch.bfloat16 and w_phases.is_contiguous() and w_phases.shape == (4, cout, 4 * cin)
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout)
    out = torch.empty(N, 4 * H * W, cout, dtype=torch.bfloat16, device=x.device)
    m = N * H * W
    # algorithmic accounting = the operator replaced: a 3 x 3 convolution over the 2H x 2W image (what the reference computes)
    with _timed("conv_igemm", 2.0 * 4 * m * 9 * cin * cout, 2.0 * (m * cin + 4 * m * cout + 9 * cin * cout)):
        check(_lib.load().cd360_conv_up2x_bf16(_ptr(x), _ptr(w_phases), _ptr(bias), _ptr(out), N, H, W, cin, cout, _stream()), "cd360_conv_up2x_bf16")
    return out


# ----------------------------------------------------------------------------------------------- Linear layers (cd360_gemm_bf16)
def _rows2d(t: torch.Tensor):
    """(rows, row stride) of a bf16 tensor [..., C] whose leading dims collapse to uniformly strided rows (last dim contiguous)."""
    assert t.dtype == torch.bfloat16 and t.stride(-1) == 1, "bf16 with a contiguous last dim"
    rows, ld = 1, None
    for size, stride in zip(reversed(t.shape[:-1]), reversed(t.stride()[:-1])):
        if size == 1:
            continue
        if ld is None:
            ld = stride
        elif stride != rows * ld:
            raise Cd360Error("cd360 gemm: the leading dims do not collapse to uniformly strided rows")
        rows *= size
    return rows, (t.shape[-1] if ld is None else ld)


def gemm_tile_n(M: int, N: int) -> int:
    _lib.query_stream(_stream())
    return _lib.load().cd360_gemm_tile_n(M, N)


def pack_ln_linear(weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm(gamma, beta) followed by Linear(weight, bias), folded for gemm(..., ln=...): -> (w' bf16 = weight * gamma,
    wsum fp32 = rowsum(w') of the ROUNDED w', cb fp32 = weight @ beta + bias)."""
    w32 = weight.detach().float()
    wp = (w32 * gamma.detach().float()[None, :]).to(torch.bfloat16).contiguous()
    cb = w32 @ beta.detach().float()
    if bias is not None:
        cb = cb + bias.detach().float()
    return wp, wp.float().sum(1).contiguous(), cb.contiguous()


def geglu_row_order(inner: int, device=None) -> torch.Tensor:
    """Row permutation of a GEGLU projection [2*inner, K] (value rows | gate rows) for gemm(..., geglu=True): per 32 output columns
    the 32 value rows then the 32 gate rows."""
    assert inner % 32 == 0
    g = torch.arange(inner // 32, device=device)[:, None] * 32 + torch.arange(32, device=device)[None, :]
    return torch.cat([g, g + inner], 1).reshape(-1)


def row_stats(x: torch.Tensor) -> torch.Tensor:
    """x [..., C] bf16 -> fp32 [rows, 1, 2] (sum, sum of squares) per row: LayerNorm statistics input of gemm(..., ln=...)."""
    _need_gpu(x)
    rows, ld = _rows2d(x)
    C = x.shape[-1]
    st = torch.empty(rows, 1, 2, dtype=torch.float32, device=x.device)
    with _timed("row_stats", 0.0, 2.0 * rows * C):
        check(_lib.load().cd360_row_stats_bf16(_ptr(x), _ptr(st), rows, C, ld, _stream()), "cd360_row_stats_bf16")
    return st


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, ln=None,
         want_stats: bool = False, geglu: bool = False, out: Optional[torch.Tensor] = None):
    """a [..., K] @ w[N, K]^T with the fused epilogues of cd360_gemm_bf16 -> out [..., N] (N / 2 with geglu), bf16.
    bias fp32 [N]; res bf16 [..., N]; ln = (stats fp32 [rows, parts, 2], wsum fp32 [N], eps) with w / bias from pack_ln_linear;
    want_stats=True returns (out, stats fp32 [rows, parts_out, 2]) for the next LayerNorm fold.  Not recorded by autograd: the
    differentiable form is linear() (grad.LinearFn)."""
    _need_gpu(a, w, bias, res)
    M, lda = _rows2d(a)
    K = a.shape[-1]
    N = w.shape[0]
    assert w.dtype == torch.bfloat16 and w.dim() == 2 and w.shape[1] == K and w.stride(1) == 1
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N)
    nout = N // 2 if geglu else N
    if out is None:
        out = torch.empty(*a.shape[:-1], nout, dtype=torch.bfloat16, device=a.device)
    mo, ldo = _rows2d(out)
    assert mo == M and out.shape[-1] == nout
    ldr = 0
    if res is not None:
        mr, ldr = _rows2d(res)
        assert mr == M and res.shape[-1] == N
    stats_in = wsum = None
    parts = ln_dim = 0
    eps = 0.0
    if ln is not None:
        stats_in, wsum, eps = ln
        assert stats_in.dtype == torch.float32 and stats_in.is_contiguous() and stats_in.shape[0] == M and stats_in.shape[2] == 2
        assert wsum.dtype == torch.float32 and wsum.is_contiguous() and wsum.numel() == N
        parts, ln_dim = stats_in.shape[1], K
    lib = _lib.load()
    stats_out = None
    if want_stats:
        _lib.query_stream(_stream())
        tn = lib.cd360_gemm_tile_n(M, N)
        stats_out = torch.empty(M, (N + tn - 1) // tn, 2, dtype=torch.float32, device=a.device)
    with _timed(_shape_tag("gemm8p", M, N, K), 2.0 * M * N * K, 2.0 * (M * K + N * K + M * nout + (M * N if res is not None else 0))):
        check(lib.cd360_gemm_bf16(_ptr(a), _ptr(w), _ptr(out), M, N, K, lda, w.stride(0), ldo, _ptr(bias), _ptr(res), ldr, _ptr(stats_in), parts,
                                  ln_dim, float(eps), _ptr(wsum), _ptr(stats_out), 1 if geglu else 0, _stream()), "cd360_gemm_bf16")
    return (out, stats_out) if want_stats else out


def gemm_ok(M: int, N: int, K: int, lda: Optional[int] = None, ldw: Optional[int] = None) -> bool:
    """Shape envelope of cd360_gemm_bf16 (K % 64, N % 16, 32-bit buffer offsets): callers outside it keep their tensors on torch."""
    lda, ldw = (K if lda is None else lda), (K if ldw is None else ldw)
    return (M > 0 and K > 0 and N > 0 and K % 64 == 0 and N % 16 == 0 and lda % 8 == 0 and ldw % 8 == 0 and M < 2 ** 31
            and (M + 256) * lda * 2 < 2 ** 32 and (N + 256) * ldw * 2 < 2 ** 32)


_F32_CACHE = {}  # id(tensor) -> (weakref, version, fp32 copy): biases of frozen Linears as the fp32 vectors the GEMM epilogue reads
_WT_CACHE = {}   # id(weight) -> (weakref, version, weight^T contiguous): the data-gradient GEMM's operand


def _cached(cache: dict, t: torch.Tensor, make):
    import weakref
    # a view (`w[:, :c]` of pose_emb_layers is a fresh tensor object on every call) or a trainable tensor gains nothing from an id-keyed
    # cache and would leave one dead entry per call behind: compute directly
    if t.requires_grad or t._base is not None:
        return make(t)
    ent = cache.get(id(t))
    if ent is not None and ent[0]() is t and ent[1] == t._version:
        return ent[2]
    if len(cache) > 4096:  # entries of tensors that died (ids are reused): drop them
        for k in [k for k, e in cache.items() if e[0]() is None]:
            del cache[k]
    val = make(t)
    cache[id(t)] = (weakref.ref(t), t._version, val)
    return val


def bias_f32(b: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if b is None or (b.dtype == torch.float32 and b.is_contiguous() and not b.requires_grad):
        return b
    return _cached(_F32_CACHE, b, lambda t: t.detach().float().contiguous())


def weight_t(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] -> w^T [K, N] contiguous bf16, cached per (tensor, version): dX = dY W is cd360_gemm_bf16(dY, W^T)."""
    return _cached(_WT_CACHE, w, lambda t: t.detach().t().contiguous())


def linear_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """True when F.linear(x, weight) can run on cd360_gemm_bf16 (bf16 on the GPU, inside the kernel's shape envelope)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 2 and x.shape[-1] == weight.shape[1]):
        return False
    N, K = weight.shape
    M = x.numel() // max(K, 1)
    return gemm_ok(M, N, K)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.linear(x, weight, bias) (+ res) on the hand-written MFMA GEMM, in every mode: no tape -> one cd360_gemm_bf16 launch; under
    autograd -> grad.LinearFn (forward the same launch; data gradient the same kernel on W^T; weight gradient cd360_gemm_tn_bf16).
    The one Linear of every route: the fused inference blocks, the plain module route (sample.py's patched forwards, hooks) and the
    fine-tuning step all end here.  x [..., K] bf16, weight [N, K] bf16 (nn.Linear layout); bias any float dtype."""
    assert linear_ok(x, weight), "cd360 linear: bf16 GPU tensors with K % 64 == 0 and N % 16 == 0 (use linear_ok() to test)"
    if x.stride(-1) != 1:
        x = x.contiguous()
    try:
        _rows2d(x)
    except Cd360Error:
        x = x.contiguous()
    if res is not None and (res.dtype != torch.bfloat16 or res.stride(-1) != 1):
        res = res.to(torch.bfloat16).contiguous()
    if _wants_grad(x, weight, bias, res):
        # the autograd node in C++ (cd360/_host.py: the same launches without ~90 us of interpreter time per Linear forward + backward);
        # the Python node serves profiling runs (per-launch events), the A/B switch, and trees where the glue was not built
        if _PROF is None and not routes.no_host_glue:
            host = _host.get()
            if host is not None:
                return host.linear(x, weight, bias, res)
        from . import grad
        return grad.LinearFn.apply(x, weight, bias, res)
    w = weight.detach()
    if w.stride(1) != 1 or w.stride(0) % 8:
        w = w.contiguous()
    return gemm(x.detach(), w, bias=bias_f32(bias), res=None if res is None else res.detach())


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out_dtype=torch.bfloat16) -> torch.Tensor:
    """a [M, N]^T @ b [M, K] -> [N, K] (fp32 accumulation; cd360_gemm_tn_bf16): the weight gradient dW = dY^T X of a Linear.
    a, b bf16 with contiguous last dims and uniform row strides (multiples of 8); N, K multiples of 8.  Deterministic."""
    _need_gpu(a, b)
    M, lda = _rows2d(a)
    Mb, ldb = _rows2d(b)
    N, K = a.shape[-1], b.shape[-1]
    assert M == Mb and out_dtype in (torch.bfloat16, torch.float32)
    lib = _lib.load()
    out = torch.empty(N, K, dtype=out_dtype, device=a.device)
    ws = torch.empty(max(16, lib.cd360_gemm_tn_workspace_bytes(M, N, K)), dtype=torch.uint8, device=a.device)
    with _timed(_shape_tag("gemm_tn", M, N, K), 2.0 * M * N * K, 2.0 * (M * N + M * K) + out.element_size() * N * K):
        check(lib.cd360_gemm_tn_bf16(_ptr(a), _ptr(b), _ptr(out), M, N, K, lda, ldb, 1 if out_dtype == torch.bfloat16 else 0, _ptr(ws), _stream()),
              "cd360_gemm_tn_bf16")
    return out


ADAMW_MAX_TENSORS = 64  # CD360_ADAMW_MAX_TENSORS of include/cd360_hip.h


class AdamwPlan:
    """Host-side argument arrays of cd360_adamw_bf16 for a fixed list of bf16 parameters whose fp32 master / exp_avg / exp_avg_sq live
    at offsets `begin` of three flat buffers: offsets, sizes, learning rates and decays are built once; the gradient AND parameter
    pointers are refreshed on every step (a parameter whose storage was replaced -- `p.data = ...` -- must not be written through a
    stale address)."""

    def __init__(self, params, begin, lr, wd):
        self.n = len(params)
        self.params = list(params)
        assert all(p.dtype == torch.bfloat16 and p.is_contiguous() for p in self.params)
        self.chunks = []
        for c0 in range(0, self.n, ADAMW_MAX_TENSORS):
            idx = list(range(c0, min(self.n, c0 + ADAMW_MAX_TENSORS)))
            k = len(idx)
            self.chunks.append(dict(
                idx=idx, k=k, grads=(ctypes.c_void_p * k)(), params=(ctypes.c_void_p * k)(*[params[i].data_ptr() for i in idx]),
                begin=(ctypes.c_int64 * k)(*[int(begin[i]) for i in idx]), numel=(ctypes.c_int64 * k)(*[params[i].numel() for i in idx]),
                lr=(ctypes.c_float * k)(*[float(lr[i]) for i in idx]), wd=(ctypes.c_float * k)(*[float(wd[i]) for i in idx])))


def adamw_step(plan: AdamwPlan, grads, master: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: torch.Tensor,
               beta1: float, beta2: float, eps: float) -> None:
    """One AdamW step of every tensor of `plan` (cd360_adamw_tick + cd360_adamw_bf16): `grads` are the bf16 gradients in plan order
    (contiguous), `step` the fp32 device scalar holding the number of steps taken so far."""
    _need_gpu(master, exp_avg, exp_avg_sq, step, *grads)
    lib = _lib.load()
    assert len(grads) == plan.n and all(g.dtype == torch.bfloat16 and g.is_contiguous() for g in grads)
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (master, exp_avg, exp_avg_sq, step))
    check(lib.cd360_adamw_tick(_ptr(step), _stream()), "cd360_adamw_tick")
    total = sum(g.numel() for g in grads)
    with _timed("adamw", 0.0, 28.0 * total):
        for c in plan.chunks:
            for j, i in enumerate(c["idx"]):
                c["grads"][j] = grads[i].data_ptr()
                c["params"][j] = plan.params[i].data_ptr()
                assert grads[i].numel() == c["numel"][j] == plan.params[i].numel()
            check(lib.cd360_adamw_bf16(c["k"], c["grads"], c["params"], c["begin"], c["numel"], c["lr"], c["wd"], _ptr(master), _ptr(exp_avg),
                                       _ptr(exp_avg_sq), _ptr(step), beta1, beta2, eps, _stream()), "cd360_adamw_bf16")


def gemm_tn_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    if not (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.stride(-1) == 1 and b.stride(-1) == 1):
        return False
    try:
        M, lda = _rows2d(a)
        Mb, ldb = _rows2d(b)
    except Cd360Error:
        return False
    N, K = a.shape[-1], b.shape[-1]
    return (M == Mb and N % 8 == 0 and K % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and (M + 64) * lda * 2 < 2 ** 31 and (M + 64) * ldb * 2 < 2 ** 31)


def gemm_cstats(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None):
    """gemm(a, w, bias, res) for an output a GroupNorm reads next: -> (out, cstats) with cstats fp32 [rows / S, N, 2] = per slab of S = 64
    or 32 rows (cd360_gemm_cstats_rows: the tiling decides) and channel the (sum, sum of squares) of the stored outputs (gn_silu's
    tile_stats after a reshape to [images, slabs, N, 2]); cstats is None when the tiling chosen for this shape writes no slabs or
    rows % 64 != 0 (plain gemm then)."""
    _need_gpu(a, w, bias, res)
    M, lda = _rows2d(a)
    K, N = a.shape[-1], w.shape[0]
    lib = _lib.load()
    _lib.query_stream(_stream())
    slab = lib.cd360_gemm_cstats_rows(M, N)
    if M % 64 or slab <= 0:
        return gemm(a, w, bias=bias, res=res), None
    assert w.dtype == torch.bfloat16 and w.dim() == 2 and w.shape[1] == K and w.stride(1) == 1
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N)
    out = torch.empty(*a.shape[:-1], N, dtype=torch.bfloat16, device=a.device)
    ldr = 0
    if res is not None:
        mr, ldr = _rows2d(res)
        assert mr == M and res.shape[-1] == N
    cstats = torch.empty(M // slab, N, 2, dtype=torch.float32, device=a.device)
    with _timed(_shape_tag("gemm8p", M, N, K), 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N + (M * N if res is not None else 0))):
        check(lib.cd360_gemm_cstats_bf16(_ptr(a), _ptr(w), _ptr(out), M, N, K, lda, w.stride(0), N, _ptr(bias), _ptr(res), ldr, _ptr(cstats), _stream()),
              "cd360_gemm_cstats_bf16")
    return out, cstats


def qproj_attention_ok(a: torch.Tensor, nk: int) -> bool:
    """Shape envelope of qproj_attention: [b, Nq, K] queries with Nq % 128 == 0 (a token tile stays inside one batch element),
    K % 64 == 0, at most 96 keys."""
    return a.dim() == 3 and a.shape[1] % 128 == 0 and a.shape[2] % 64 == 0 and nk <= 96


def kv_pack_fp8(k: torch.Tensor, v: torch.Tensor, nk: int, heads: int, out=None):
    """k, v bf16 [B, >= nk, heads*64] (last dim contiguous) -> (kv8 uint8 [B, heads, 14336], scales fp32 [B, heads, 2]): the OCP e4m3
    image of K and V^T that cd360_qproj_attn_fp8_bf16 copies into its LDS (cd360_kv_pack_fp8; BASELINE configs[4]).  Once per image:
    the text context is constant over a trajectory.  `out` = (kv8, scales) buffers to write into."""
    _need_gpu(k, v)
    B = k.shape[0]
    assert k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and k.shape[-1] == heads * 64 and v.shape[-1] == heads * 64
    assert k.stride(2) == 1 and v.stride(2) == 1 and k.shape[1] >= nk and v.shape[1] >= nk and v.shape[0] == B and nk <= 96
    if out is None:
        kv8 = torch.empty(B, heads, 96 * 64 + 64 * 128, dtype=torch.uint8, device=k.device)
        scales = torch.empty(B, heads, 2, dtype=torch.float32, device=k.device)
    else:  # buffers that stay put (a captured graph re-packs into them on every replay of the render step)
        kv8, scales = out
        assert kv8.shape == (B, heads, 96 * 64 + 64 * 128) and kv8.dtype == torch.uint8 and scales.shape == (B, heads, 2) and scales.dtype == torch.float32
    assert kv8.numel() == _lib.load().cd360_kv_fp8_bytes(B, heads)
    check(_lib.load().cd360_kv_pack_fp8(_ptr(k), _ptr(v), _ptr(kv8), _ptr(scales), B, heads, nk, k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                        _stream()), "cd360_kv_pack_fp8")
    return kv8, scales


def qproj_attention(a: torch.Tensor, w: torch.Tensor, k: torch.Tensor, v: torch.Tensor, nk: int, heads: int,
                    bias: Optional[torch.Tensor] = None, ln=None, dup: int = 0, tag: str = "qproj_attn", fp8=None) -> torch.Tensor:
    """softmax((a w^T [+ LayerNorm fold, + bias]) k^T / 8) v per head with the query projection and the attention in ONE kernel
    (cd360_qproj_attn_bf16): a [b, Nq, K] bf16, w [heads*64, K] bf16, k / v [b, >= nk, heads*64] (last dim contiguous, e.g. the two
    halves of the merged k|v projection), nk <= 96 -> [b, Nq, heads*64].  bias / ln as gemm().  Forward only.
    dup > 0 (cd360_qproj_attn_dedup_bf16): k / v hold b + dup batch elements; the last `dup` query elements attend to the keys of their own
    batch index AND of index + dup -> [b + dup, Nq, heads*64] (the de-duplicated third of a 3-way CFG batch: q projected once)."""
    _need_gpu(a, w, bias)
    b, nq, K = a.shape
    N = heads * 64
    M, lda = _rows2d(a)
    assert w.dtype == torch.bfloat16 and w.shape == (N, K) and w.stride(1) == 1
    assert 0 <= dup <= b and qproj_attention_ok(a, nk)
    if fp8 is None:
        _need_gpu(k, v)
        assert k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and k.shape[0] == b + dup and v.shape[0] == b + dup and k.shape[-1] == N and v.shape[-1] == N
        assert k.stride(2) == 1 and v.stride(2) == 1 and k.shape[1] >= nk and v.shape[1] >= nk
    else:  # (kv8, scales) of kv_pack_fp8: both attention contractions on fp8 MFMA (cd360_qproj_attn_fp8_bf16); k, v are not read
        kv8, kvs = fp8
        _need_gpu(kv8, kvs)
        assert kv8.dtype == torch.uint8 and kv8.is_contiguous() and kv8.shape == (b + dup, heads, 96 * 64 + 64 * 128)
        assert kvs.dtype == torch.float32 and kvs.is_contiguous() and kvs.shape == (b + dup, heads, 2)
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N)
    stats_in = wsum = None
    parts = ln_dim = 0
    eps = 0.0
    if ln is not None:
        stats_in, wsum, eps = ln
        assert stats_in.dtype == torch.float32 and stats_in.is_contiguous() and stats_in.shape[0] == M and stats_in.shape[2] == 2
        assert wsum.dtype == torch.float32 and wsum.is_contiguous() and wsum.numel() == N
        parts, ln_dim = stats_in.shape[1], K
    out = torch.empty(b + dup, nq, N, dtype=torch.bfloat16, device=a.device)
    flops = 2.0 * M * N * K + 4.0 * (M + dup * nq) * nk * N
    if fp8 is not None:
        with _timed(tag, flops, 2.0 * (M * K + N * K + M * N) + 2.0 * b * nk * N):
            check(_lib.load().cd360_qproj_attn_fp8_bf16(_ptr(a), _ptr(w), _ptr(out), M, N, K, lda, w.stride(0), N, _ptr(bias), _ptr(stats_in), parts,
                                                       ln_dim, float(eps), _ptr(wsum), _ptr(fp8[0]), _ptr(fp8[1]), nq, nk, 64 ** -0.5, dup,
                                                       _stream()), "cd360_qproj_attn_fp8_bf16")
        return out
    with _timed(tag, flops, 2.0 * (M * K + N * K + M * N + 2 * b * nk * N)):  # per-kernel timing name: A3 "qproj_attn", A2 "qproj_attn_text"
        check(_lib.load().cd360_qproj_attn_dedup_bf16(_ptr(a), _ptr(w), _ptr(out), M, N, K, lda, w.stride(0), N, _ptr(bias), _ptr(stats_in), parts,
                                                     ln_dim, float(eps), _ptr(wsum), _ptr(k), _ptr(v), k.stride(0), k.stride(1), v.stride(0),
                                                     v.stride(1), nq, nk, 64 ** -0.5, dup, _stream()), "cd360_qproj_attn_dedup_bf16")
    return out


def pose_embed(x: torch.Tensor, xref: torch.Tensor, wa: torch.Tensor, wb: torch.Tensor) -> torch.Tensor:
    """pose_emb_layers(cat[x, xref]) (attention.py:634) as x wa^T + xref wb^T through the C ABI (cd360_pose_embed_bf16); wa = W[:, :C],
    wb = W[:, C:], contiguous.  Forward-only operator-level entry (the modules run the same two products on the library GEMM, which
    torch differentiates)."""
    _need_gpu(x, xref, wa, wb)
    C = x.shape[-1]
    for t in (x, xref, wa, wb):
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
    assert xref.shape == x.shape and wa.shape == (C, C) and wb.shape == (C, C)
    out = torch.empty_like(x)
    rows = x.numel() // C
    with _timed("pose_embed", 4.0 * rows * C * C, 2.0 * (3 * rows * C + 2 * C * C)):
        check(_lib.load().cd360_pose_embed_bf16(_ptr(x), _ptr(xref), _ptr(wa), _ptr(wb), _ptr(out), rows, C, _stream()), "cd360_pose_embed_bf16")
    return out


def add_layernorm(a: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, eps: float, want_sum: bool = True,
                  alias: bool = False):
    """(a + b, LayerNorm(a + b) * gamma + beta) in one pass; b None -> (None, LayerNorm(a)).  All bf16, last dim C.
    Differentiable with respect to a and b (grad.AddLayerNormFn).  alias (only with b None): the first result is `a` again -- under
    autograd an alias whose gradient the LayerNorm backward kernel adds in its own pass; use it as the residual stream from here on."""
    if _wants_grad(a, b, gamma, beta):
        from . import grad
        return grad.add_layernorm(a, b, gamma, beta, eps, want_sum, alias)
    _need_gpu(a, b, gamma, beta)
    C = a.shape[-1]
    assert a.dtype == torch.bfloat16 and a.is_contiguous() and gamma.dtype == torch.bfloat16 and beta.dtype == torch.bfloat16
    assert b is None or (b.dtype == torch.bfloat16 and b.is_contiguous() and b.shape == a.shape)
    rows = a.numel() // C
    s = torch.empty_like(a) if (b is not None and want_sum) else None
    ln = torch.empty_like(a)
    with _timed("add_layernorm", 0.0, 2.0 * rows * C * (2 + (b is not None) + (s is not None))):
        check(_lib.load().cd360_add_layernorm_bf16(_ptr(a), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(s), _ptr(ln), rows, C, float(eps), _stream()),
              "cd360_add_layernorm_bf16")
    return (a if (alias and b is None) else s), ln


def add_layernorm_bwd(x: torch.Tensor, gamma: torch.Tensor, d_ln: torch.Tensor, d_sum: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """Backward of add_layernorm with respect to the summed input x (= a + b): LayerNorm backward of d_ln plus d_sum."""
    _need_gpu(x, gamma, d_ln, d_sum)
    C = x.shape[-1]
    d_ln = d_ln.contiguous()
    d_sum = None if d_sum is None else d_sum.contiguous()
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and gamma.dtype == torch.bfloat16 and d_ln.dtype == torch.bfloat16 and d_ln.shape == x.shape
    assert d_sum is None or (d_sum.dtype == torch.bfloat16 and d_sum.shape == x.shape)
    rows = x.numel() // C
    dx = torch.empty_like(x)
    with _timed("add_layernorm_bwd", 0.0, 2.0 * rows * C * (3 + (d_sum is not None))):
        check(_lib.load().cd360_add_layernorm_bwd_bf16(_ptr(x), _ptr(gamma), _ptr(d_ln), _ptr(d_sum), _ptr(dx), rows, C, float(eps), _stream()),
              "cd360_add_layernorm_bwd_bf16")
    return dx


def cfg_euler_step(x: torch.Tensor, eps: torch.Tensor, sigma: torch.Tensor, sigma_next: torch.Tensor, scale: float, scale_im: float):
    """x [n,...] fp32, eps [3n,...] fp32 (u | ic | c), sigma / sigma_next 0-d fp32 device tensors -> Euler-updated x (one kernel)."""
    _need_gpu(x, eps, sigma, sigma_next)
    assert x.dtype == torch.float32 and eps.dtype == torch.float32 and eps.numel() == 3 * x.numel() and x.is_contiguous() and eps.is_contiguous()
    assert sigma.dtype == torch.float32 and sigma_next.dtype == torch.float32 and sigma.numel() == 1 and sigma_next.numel() == 1
    out = torch.empty_like(x)
    check(_lib.load().cd360_cfg_euler_step_f32(_ptr(x), _ptr(eps), _ptr(sigma), _ptr(sigma_next), float(scale), float(scale_im), _ptr(out),
                                              x.numel(), _stream()), "cd360_cfg_euler_step_f32")
    return out


def unet_stage_in(x: torch.Tensor, step_tab: torch.Tensor, step: torch.Tensor, w_k36: torch.Tensor, bias: torch.Tensor, temb_tab: torch.Tensor,
                  lab: torch.Tensor, h: torch.Tensor, emb_act: torch.Tensor):
    """Head of a captured sampling step (cd360_unet_stage_in): x [bs, 4, H, W] fp32 -> h [rep bs, H W, Cout] bf16 = the UNet's input
    convolution of bf16(c_in x) written to every CFG branch, emb_act [rep bs, E] bf16 = silu(temb_tab[step] + lab); c_in = step_tab[step][2]."""
    _need_gpu(x, step_tab, step, w_k36, bias, temb_tab, lab, h, emb_act)
    bs, four, H, W = x.shape
    rep = lab.shape[0] // bs
    cout, E = w_k36.shape[1], temb_tab.shape[1]
    assert four == 4 and x.dtype == torch.float32 and x.is_contiguous() and step.dtype == torch.int32 and step_tab.dtype == torch.float32
    assert w_k36.shape[0] == 36 and w_k36.dtype == torch.float32 and bias.dtype == torch.float32 and w_k36.is_contiguous()
    assert temb_tab.dtype == lab.dtype == h.dtype == emb_act.dtype == torch.bfloat16 and lab.shape == emb_act.shape == (rep * bs, E)
    assert h.shape == (rep * bs, H * W, cout) and h.is_contiguous() and temb_tab.is_contiguous() and lab.is_contiguous() and emb_act.is_contiguous()
    check(_lib.load().cd360_unet_stage_in(_ptr(x), _ptr(step_tab), _ptr(step), _ptr(w_k36), _ptr(bias), _ptr(h), _ptr(temb_tab), _ptr(lab),
                                         _ptr(emb_act), bs, rep, H, W, cout, E, _stream()), "cd360_unet_stage_in")
    return h, emb_act


def cfg_euler_step_cl(x: torch.Tensor, eps_cl: torch.Tensor, step_tab: torch.Tensor, step: torch.Tensor, scale: float, scale_im: float):
    """Tail of a captured sampling step, IN PLACE on x [bs, 4, H, W] fp32: eps_cl [3 bs, H W, >= 4] bf16 channels-last rows (a channel slice
    of wider rows is fine: the row stride is passed), sigma / sigma_next = step_tab[step][0 / 1]  (cd360_cfg_euler_step_cl)."""
    _need_gpu(x, eps_cl, step_tab, step)
    bs = x.shape[0]
    hw = x.shape[2] * x.shape[3]
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == 4 and eps_cl.dtype == torch.bfloat16
    assert eps_cl.shape[0] == 3 * bs and eps_cl.shape[1] == hw and eps_cl.shape[2] >= 4 and eps_cl.stride(2) == 1
    ld = eps_cl.stride(1)
    assert eps_cl.stride(0) == hw * ld
    check(_lib.load().cd360_cfg_euler_step_cl(_ptr(x), _ptr(eps_cl), _ptr(step_tab), _ptr(step), float(scale), float(scale_im), bs, hw, ld,
                                             _stream()), "cd360_cfg_euler_step_cl")
    return x


def out_conv4_ok(H: int, W: int, cin: int) -> bool:
    return W in (32, 64, 128) and H % 2 == 0 and cin % 64 == 0 and H * W * cin * 2 < 2 ** 31


def pack_out_conv4_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight [4, Cin, 3, 3] -> [64, Cin] bf16, row tap * 4 + co (tap = 3 ky + kx), rows 36 .. 63 zero (cd360_out_conv4_bf16)."""
    co, cin, kh, kw = w.shape
    assert co == 4 and kh == 3 and kw == 3
    out = torch.zeros(64, cin, dtype=torch.bfloat16, device=w.device)
    out[:36] = w.detach().permute(2, 3, 0, 1).reshape(36, cin).to(torch.bfloat16)
    return out.contiguous()


def out_conv4(x: torch.Tensor, w36: torch.Tensor, bias: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
    """The UNet's 3 x 3 output convolution to FOUR channels: x [N, H W, Cin] bf16 channels-last -> [N, H W, 4] bf16 (cd360_out_conv4_bf16)."""
    _need_gpu(x, w36, bias)
    cin = x.shape[-1]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() == N * H * W * cin and out_conv4_ok(H, W, cin)
    assert w36.shape == (64, cin) and w36.dtype == torch.bfloat16 and w36.is_contiguous() and bias.dtype == torch.float32 and bias.numel() == 4
    out = torch.empty(N, H * W, 4, dtype=torch.bfloat16, device=x.device)
    with _timed("conv_igemm", 2.0 * N * H * W * 9 * cin * 4, 2.0 * (N * H * W * (cin + 4) + 36 * cin)):
        check(_lib.load().cd360_out_conv4_bf16(_ptr(x), _ptr(w36), _ptr(bias), _ptr(out), N, H, W, cin, _stream()), "cd360_out_conv4_bf16")
    return out
