"""Host side of the fused FeatureNeRF render (the caller of cd360_nerf_mlp_aggregate).

Restructures FeatureNeRFEncoding.forward (reference sgm/modules/nerfsd_pytorch3d.py:53-161) so that the
`[b, n, hw, S, C+198]` concat is never formed.  With W1 = plane_coefs.0.weight split by input group
  W1 = [ Wf (C) | Wx (96+3: enc16(q_i), q_i) | Wp (96+3: enc8(plucker_i), dir_i) ]
and w_v = nviews.weight split as [ vf (C) | enc16(q_0), q_0 (99) | o_i^tgt (3) | enc16(o_i^tgt) (96) ]:

  z_i(sample)  = bilinear(Y_i)(sample) + zP_i(ray) + Wx . [enc16(q_i), q_i]        Y_i = xref_i Wf^T
  logit_i      = bilinear(lv_i)(sample) + c_i  (+ terms common to all views)        lv_i = xref_i vf,  zP_i = Wp.feat + b1
  h(sample)    = W2 . sum_i softmax_i(logit) SiLU(z_i) + b2                          (softmax weights sum to 1)

The three tables (Y, zP, lv) are GEMMs over n*hw rows (S = 24 times fewer rows than the reference's per-sample
Linear) on cd360_gemm_bf16 (ops.linear, forward and backward) and the rest happens inside one HIP kernel (csrc/nerf_fused.hip).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import memo, ops, routes

NUM_FREQS = 16


def xyz_k_columns(C: int) -> list:
    """Column of plane_coefs.0.weight feeding each of the kernel's 112 per-sample inputs (-1 = zero pad).
    Kernel order: k = ks*16 + h*8 + j, pair = j>>1, w = ks*4 + pair; w < 24: (comp = w%3, freq = 2*(w//3) + h),
    j even = sin, odd = cos; w == 24: raw q (h=0: q0, q1; h=1: q2).  Reference order (utils_cameraray.py:222-242):
    enc = [sin f0 (xyz) ... sin f15 (xyz) | cos f0 (xyz) ... cos f15 (xyz)], then q (3)."""
    cols = []
    for ks in range(7):
        for h in range(2):
            for j in range(8):
                w, is_cos = ks * 4 + (j >> 1), j & 1
                if w < 24:
                    comp, kf = w % 3, 2 * (w // 3) + h
                    cols.append(C + (48 if is_cos else 0) + kf * 3 + comp)
                elif w == 24:
                    raw = {(0, 0): 0, (0, 1): 1, (1, 0): 2}.get((h, j), None)
                    cols.append(-1 if raw is None else C + 96 + raw)
                else:
                    cols.append(-1)
    return cols


def positional_encoding(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """utils_cameraray.py:222-242 (only used here for the per-view constants: O(b*n) values)."""
    start = -1 * (n_freqs / 2)
    freqs = 2.0 ** torch.arange(start, start + n_freqs, device=x.device) * np.pi
    xf = (x[..., None, :] * freqs[:, None]).flatten(-2)  # [..., n_freqs * D], frequency-major as the reference's concatenation
    return torch.cat([torch.sin(xf), torch.cos(xf)], dim=-1)


_grid_cache = {}


def _const(key, make):
    """Host-built constants, uploaded once per (what, device): a training step then issues no host-to-device copy for them."""
    if key not in _grid_cache:
        _grid_cache[key] = make()
    return _grid_cache[key]


class _LiveNerfWeights(torch.autograd.Function):
    """The derived operands of the fused render for TRAINABLE parameters as ONE autograd node: slicing W1 into its gather / encoding /
    Plucker column groups, the column permutation of the encoding part, the zero pads and the dtype casts are ~50 tiny torch kernels when
    autograd records them one by one (and as many again in the backward pass); twelve pose blocks made that ~1200 launches of a
    fine-tuning step.  forward: the same values; backward: the column groups' gradients written back into one dW1, the three slices of
    nviews.weight concatenated."""

    @staticmethod
    def forward(ctx, W1, b1, W2, b2, wv, bv, Wd):
        C, dev = W2.shape[0], W1.device
        cols = xyz_k_columns(C)
        ps = (W1, b1, W2, b2, wv, bv, Wd)
        if (all(p.is_cuda and p.dtype == torch.bfloat16 and p.is_contiguous() for p in ps) and Wd.shape[0] == 4 and C % 4 == 0
                and not routes.no_train_fusions):
            # every parameter bf16 (the fine-tuning step: bf16 parameters beside fp32 masters): one kernel each way
            NK = len(cols)
            kcol, kpos = _const(("xyz_k_tables", C, str(dev)), lambda: (
                torch.tensor(cols, dtype=torch.int32, device=dev),
                torch.tensor([cols.index(c) if c in cols else -1 for c in range(C + 198)], dtype=torch.int32, device=dev)))
            wb, wf = ops.nerf_pack_weights(W1, b1, b2, wv.reshape(-1), bv.reshape(-1), Wd, kcol)
            ctx.fast = (C, NK, kpos, wv.shape, bv.shape)
            nWf, nWk = C * C, C * NK
            return (wb[:nWf].view(C, C), wb[nWf:nWf + nWk].view(C, NK), wb[nWf + nWk:].view(C, 128), wf[:C], W2, wf[C:2 * C],
                    wf[2 * C:3 * C], wf[3 * C:3 * C + 99], wf[3 * C + 100].reshape(()), wf[3 * C + 104:].view(4, C))
        ctx.fast = None
        idx, pos = _const(("xyz_k", C, str(dev)), lambda: (torch.tensor([c for c in cols if c >= 0], device=dev),
                                                           torch.tensor([i for i, c in enumerate(cols) if c >= 0], device=dev)))
        bf = torch.bfloat16
        Wk = torch.zeros(C, len(cols), dtype=bf, device=dev)
        Wk[:, pos] = W1[:, idx].to(bf)
        Wp = torch.zeros(C, 128, dtype=bf, device=dev)
        Wp[:, :99] = W1[:, C + 99:C + 198]
        wvf = wv.reshape(-1).float()
        ctx.meta = (C, len(cols), idx, pos, tuple(t.dtype for t in (W1, b1, W2, b2, wv, bv, Wd)), wv.shape, bv.shape)
        return (W1[:, :C].to(bf).contiguous(), Wk, Wp, b1.float(), W2.to(bf).contiguous(), b2.float(), wvf[:C].contiguous(),
                wvf[C + 99:C + 198].contiguous(), bv.float().reshape(()), Wd.float().contiguous())

    @staticmethod
    def backward(ctx, dWf, dWk, dWp, db1, dW2, db2, dvf, dvc, dbv, dWd):
        if ctx.fast is not None:
            C, NK, kpos, wv_shape, bv_shape = ctx.fast
            grads = (dWf, dWk, dWp, db1, db2, dvf, dvc, dbv, dWd)
            if all(g is None for g in grads):
                return None, None, dW2, None, None, None, None
            dW1, sm = ops.nerf_unpack_grads(grads, kpos, C, NK)
            some_w1 = dWf is not None or dWk is not None or dWp is not None
            return (dW1 if some_w1 else None, None if db1 is None else sm[:C], None if dW2 is None else dW2.to(torch.bfloat16),
                    None if db2 is None else sm[C:2 * C], sm[2 * C:3 * C + 198].view(wv_shape) if (dvf is not None or dvc is not None) else None,
                    None if dbv is None else sm[3 * C + 198].reshape(bv_shape), None if dWd is None else sm[3 * C + 200:].view(4, C))
        C, nk, idx, pos, dts, wv_shape, bv_shape = ctx.meta
        some = next(g for g in (dWf, dWk, dWp, db1, dW2, db2, dvf, dvc, dbv, dWd) if g is not None)
        dev = some.device
        dW1 = None
        if dWf is not None or dWk is not None or dWp is not None:
            dW1 = torch.zeros(C, C + 198, dtype=dts[0], device=dev)
            if dWf is not None:
                dW1[:, :C] = dWf
            if dWk is not None:
                dW1[:, idx] = dWk[:, pos].to(dts[0])
            if dWp is not None:
                dW1[:, C + 99:C + 198] = dWp[:, :99]
        dwv = None
        if dvf is not None or dvc is not None:
            z = lambda g, n_: torch.zeros(n_, dtype=torch.float32, device=dev) if g is None else g.reshape(-1).float()
            dwv = torch.cat([z(dvf, C), torch.zeros(99, dtype=torch.float32, device=dev), z(dvc, 99)]).reshape(wv_shape).to(dts[4])
        cast = lambda g, i: None if g is None else g.to(dts[i])
        return (dW1, cast(db1, 1), cast(dW2, 2), cast(db2, 3), dwv, None if dbv is None else dbv.reshape(bv_shape).to(dts[5]), cast(dWd, 6))


class FusedNerfWeights:
    """Derived, cached forms of one FeatureNeRFEncoding's parameters."""

    def __init__(self, W1, b1, W2, b2, wv, bv, Wd, dtype=torch.bfloat16, live: bool = False):
        """live=True (training): the derived tensors stay connected to the parameters by autograd (slicing, transposes and casts
        are recorded), so gradients of the tables and of Wk flow back into plane_coefs / nviews / decoder; nothing is cached."""
        C = W2.shape[0]
        self.C = C
        dev = W1.device
        if live and W1.is_cuda and dtype == torch.bfloat16 and C % 64 == 0 and not routes.library_linear:  # the predicate of `fused` below
            # what the fused render reads, as one autograd node (the torch-GEMM route below keeps its transposed / fp32 forms)
            (self.Wf, self.Wk, self.Wp, self.b1, self.W2, self.b2_f32, self.vf, self.v_cam, self.bv,
             self.Wd) = _LiveNerfWeights.apply(W1, b1, W2, b2, wv, bv, Wd)
            self.live = True
            return
        if not live:
            W1, b1, W2, b2, wv, bv, Wd = (p.detach() for p in (W1, b1, W2, b2, wv, bv, Wd))
        W1f = W1.float()
        self.Wf_t = W1f[:, :C].t().contiguous().to(dtype)  # [C, C]: Y = xref @ Wf_t
        cols = xyz_k_columns(C)
        Wk = torch.zeros(C, len(cols), dtype=torch.float32, device=dev)
        idx, pos = _const(("xyz_k", C, str(dev)), lambda: (torch.tensor([c for c in cols if c >= 0], device=dev),
                                                           torch.tensor([i for i, c in enumerate(cols) if c >= 0], device=dev)))
        Wk[:, pos] = W1f[:, idx]
        self.Wk_f32 = Wk  # [C, 112] fp32 (kept for tests / re-quantisation)
        self.Wk = Wk.to(torch.bfloat16).contiguous()
        Wp = torch.zeros(104, C, dtype=torch.float32, device=dev)
        Wp[:99] = W1f[:, C + 99:C + 198].t()
        self.Wp_t = Wp.contiguous()  # [104, C] fp32
        self.b1 = b1.float().contiguous()
        self.W2_t = W2.t().contiguous().to(dtype)
        self.b2 = b2.to(dtype).contiguous()
        wvf = wv.float().reshape(-1)
        self.vf = wvf[:C].contiguous()
        self.v_cam = wvf[C + 99:C + 198].contiguous()  # [o_tgt (3) | enc16(o_tgt) (96)] rows of nviews.weight
        self.bv = bv.float().reshape(())
        self.Wd = Wd.float().contiguous()  # [4, C]
        # Linear-layout bf16 operands of the table GEMMs on cd360_gemm_bf16 (ops.linear).  live: slices / pads / casts are recorded by
        # autograd, so the GEMMs' weight gradients (grad.LinearFn -> cd360_gemm_tn_bf16) flow back into plane_coefs
        self.Wf = W1f[:, :C].contiguous().to(torch.bfloat16)                                              # [C, C]:   Y = xref Wf^T
        self.Wp = torch.cat([W1f[:, C + 99:C + 198], W1f.new_zeros(C, 29)], 1).to(torch.bfloat16)        # [C, 128]: zP = plucker Wp^T + b1
        self.W2 = W2.to(torch.bfloat16) if live else W2.detach().to(torch.bfloat16).contiguous()          # [C, C]:   h = g W2^T + b2
        self.b2_f32 = b2.float() if live else b2.detach().float().contiguous()
        self.live = live


def patch_positions(r: int, device, jitter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """utils_cameraray.py:106-147: r patch centres in [1, -1]; with `jitter` (r + 1 uniforms, stratified training) a uniform point of
    each patch.  Eval: computed on the host once and cached.  A host `jitter` (the reference's CPU generator) is combined on the host and
    uploaded; a device `jitter` is combined on the device against cached bounds (no host tensor: graph capture)."""
    key = (r, str(device))
    if jitter is None:
        return _const(key, lambda: ((lambda e: (e[:-1] + e[1:]) / 2)(torch.linspace(1, -1, r + 1))).to(device))

    def bounds():
        edges = torch.linspace(1, -1, r + 1)
        center = (edges[1:] + edges[:-1]) / 2.0
        return torch.cat([edges[:1], center], -1), torch.cat([center, edges[-1:]], -1)

    if jitter.is_cuda:
        lower, upper = _const(("bounds",) + key, lambda: tuple(t.to(device) for t in bounds()))
        return (lower + (upper - lower) * jitter)[:-1].contiguous()
    lower, upper = bounds()
    return _upload_small((lower + (upper - lower) * jitter)[:-1], device)


_SYNC_UPLOAD = bool(__import__("os").environ.get("CD360_SYNC_UPLOAD"))  # A/B: the pageable, host-blocking copy of rounds 1-5
_PIN_RINGS = {}  # (numel, dtype) -> [pinned [32, numel] buffer, 32 events, next slot]


def _upload_small(t: torch.Tensor, device) -> torch.Tensor:
    """A small host tensor to the device WITHOUT stalling the host: a pageable `.to(device)` is a synchronous copy, and inside an eagerly
    launched step every one of them (24 per fine-tuning step: the stratified patch jitter of 12 pose blocks on the reference's CPU
    generator) parked the Python thread until the GPU had caught up -- 7 ms of a 130 ms step (tools/probe/train_host_profile.py, round 6).
    Rows of a pinned ring + non-blocking copies; a slot is reused only after the event behind its last copy has completed."""
    if not torch.cuda.is_available() or torch.device(device).type != "cuda" or _SYNC_UPLOAD:
        return t.to(device)
    key = (t.numel(), t.dtype)
    ring = _PIN_RINGS.get(key)
    if ring is None:
        ring = _PIN_RINGS[key] = [torch.empty(32, t.numel(), dtype=t.dtype).pin_memory(), [None] * 32, 0]
    buf, events, i = ring
    if events[i] is not None:
        events[i].synchronize()
    buf[i].copy_(t.reshape(-1))
    out = buf[i].to(device, non_blocking=True).reshape(t.shape)
    ev = torch.cuda.Event()
    ev.record()
    events[i] = ev
    ring[2] = (i + 1) % 32
    return out


_depth_cache = {}


def depth_samples(num_samples: int, far: float, near: float, device, num_rays: int, jitter: Optional[torch.Tensor] = None):
    """Raymarcher buffers + stratified_sampling (nerfsd_pytorch3d.py:248-259,308-330).
    Returns (t, dists): [S] each in eval mode, [hw, S] with jitter."""
    key = (int(num_samples), float(far), float(near), str(device))
    if jitter is None and key in _depth_cache:  # eval mode: constants -- no host-to-device copy per call (hipGraph-capturable)
        return _depth_cache[key]
    if jitter is None:
        l = torch.linspace(near, near + (near + far), num_samples + 1)
        _depth_cache[key] = (((l[1:] + l[:-1]) / 2.0).to(device), (l[1:] - l[:-1]).to(device))
        return _depth_cache[key]

    def bounds():
        l = torch.linspace(near, near + (near + far), num_samples + 1)
        center = (l[1:] + l[:-1]) / 2.0
        return torch.cat([l[:1], center], -1).to(device)[None], torch.cat([center, l[-1:]], -1).to(device)[None]

    lower, upper = _const(("depth_bounds",) + key, bounds)
    j = lower + (upper - lower) * jitter.to(device)
    return ((j[..., :-1] + j[..., 1:]) / 2.0).contiguous(), (j[..., 1:] - j[..., :-1]).contiguous()


def view_constants(fw: FusedNerfWeights, cams: torch.Tensor) -> torch.Tensor:
    """c_i = w_v[o_tgt].o_i^tgt + w_v[enc].enc16(o_i^tgt) + b_v, with o_i^tgt the reference camera centre in the
    target view frame (nerfsd_pytorch3d.py:116-123,146-147).  cams [b, n+1, 16] -> [b, n] fp32."""
    return ((_camera_constants(cams) * fw.v_cam).sum(-1) + fw.bv).contiguous()


_CAM_CONSTS = memo.Memo()  # (packed cameras, version) -> camera constants: the twelve pose blocks of a forward share one packed camera tensor
_PLUCKER = memo.Memo(2)    # ... -> Plucker features of a level (one entry per resolution in flight)


def _camera_constants(cams: torch.Tensor):
    """Reference camera centres in the target view frame and their positional encoding: camera-only values, computed once per packed
    camera tensor instead of once per pose block (a dozen tiny kernels each).  Memoised under cd360.memo's capture rule."""
    def make():
        with torch.no_grad():
            R = cams[..., :9].reshape(*cams.shape[:-1], 3, 3)
            T = cams[..., 9:12]
            center = -(T[..., None, :] * R).sum(-1)  # -T @ R^T
            o = (center[:, 1:, :, None] * R[:, :1]).sum(-2) + T[:, :1]  # centre_i @ R_0 + T_0
            return torch.cat([o, positional_encoding(o, NUM_FREQS)], -1).contiguous()  # [b, n, 3 + 96]
    return _CAM_CONSTS.get(cams, make)


def _plucker_rows(cams: torch.Tensor, xs: torch.Tensor, ys: torch.Tensor) -> torch.Tensor:
    """ops.plucker_features_bf16 -- a function of the cameras and the patch grid only, so the blocks of one resolution share it within a
    forward when the grid is the cached eval-mode one (a stratified training step draws a grid per block: always recomputed)."""
    if torch.is_grad_enabled() or xs.requires_grad or ys.requires_grad:
        return ops.plucker_features_bf16(cams, xs, ys)
    # The entry HOLDS xs / ys and is matched by identity: a jittered grid drawn per block (stratified mode under no_grad) is freed when its
    # block returns, and the next block's fresh grid may be handed the same id() -- an id-keyed entry would then serve the previous
    # block's features for another jitter.
    key = _GridKey(xs, ys)
    return _PLUCKER.get(cams, lambda: ops.plucker_features_bf16(cams, xs, ys), extra=key)


class _GridKey:
    """(xs, ys) of a patch grid as a memo key: equal only to a key made of the SAME tensor objects at the same versions; the key keeps
    both tensors alive, so their ids cannot be recycled while the entry exists."""
    __slots__ = ("xs", "ys", "vx", "vy")

    def __init__(self, xs, ys):
        self.xs, self.ys, self.vx, self.vy = xs, ys, xs._version, ys._version

    def __eq__(self, other):
        return isinstance(other, _GridKey) and other.xs is self.xs and other.ys is self.ys and other.vx == self.vx and other.vy == self.vy

    __hash__ = None


def fused_feature_nerf(fw: FusedNerfWeights, cams: torch.Tensor, xref: Optional[torch.Tensor], num_samples: int, far: float,
                       near: float = 0.0, xy_jitter=None, depth_jitter=None, want_view_weights: bool = False, tables=None, dims=None,
                       depths=None):
    """cams [b, n+1, 16] fp32 (device), xref [b, n, hw, C] -> (h [b,hw,S,C] bf16, dec [b,hw,S,4] fp32, dists, view_weights|None).
    With precomputed `tables` = (Y, lv, img_map) xref may be None and `dims` = (b, n, hw, C).  `depths` = (t, dists), [S] or [hw, S]
    each: sample depths chosen by the caller (importance sampling) instead of the uniform / jittered ones."""
    b, n, hw, C = xref.shape if dims is None else dims
    r = int(math.isqrt(hw))
    dev = cams.device
    xs = patch_positions(r, dev, None if xy_jitter is None else xy_jitter[0])
    ys = patch_positions(r, dev, None if xy_jitter is None else xy_jitter[1])
    t, dists = depths if depths is not None else depth_samples(num_samples, far, near, dev, hw, depth_jitter)
    # the three table GEMMs run on cd360_gemm_bf16 in every mode (ops.linear: recorded by autograd when the weights are live) -- Plucker
    # features written as bf16 rows of 128 by their kernel (no fp32 intermediate, no cast pass), bias fused; CD360_LIBRARY_LINEAR=1 = the
    # round-1 torch GEMMs (A/B)
    fused = cams.is_cuda and C % 64 == 0 and fw.Wp.dtype == torch.bfloat16 and not routes.library_linear
    if fused:
        zP = ops.linear(_plucker_rows(cams, xs, ys).reshape(b * n * hw, 128), fw.Wp, fw.b1).reshape(b * n, hw, C)
    else:
        pf = ops.plucker_features(cams, xs, ys).reshape(b * n * hw, 104)
        zP = torch.addmm(fw.b1, pf, fw.Wp_t).to(torch.bfloat16).reshape(b * n, hw, C)
    cview = view_constants(fw, cams)
    if tables is None and torch.is_grad_enabled() and any(w.requires_grad for w in (fw.Wf, fw.vf, fw.Wk, zP, cview)):
        # training: live weights; the render and its backward go through grad.NerfRenderFn (no table scatters)
        from . import grad
        g, logits, lse = grad.NerfRenderFn.apply(cams, xs, ys, t, xref, fw.Wf, fw.vf, zP, cview, fw.Wk)
    else:
        if tables is None:
            tables = reference_tables(fw, xref)
        Y, lv = tables[0], tables[1]
        img_map = tables[2] if len(tables) > 2 else None
        g, logits, lse = ops.nerf_mlp_aggregate(cams, xs, ys, t, Y, zP, lv, cview, fw.Wk, want_logits=want_view_weights, img_map=img_map)
    if fused:
        h = ops.linear(g.reshape(-1, C), fw.W2, fw.b2_f32).reshape(b, hw, num_samples, C)
    else:
        h = torch.addmm(fw.b2, g.reshape(-1, C), fw.W2_t).reshape(b, hw, num_samples, C)
    dec = ops.rowdot4(h, fw.Wd)
    vw = None
    if want_view_weights:
        vw = torch.softmax(logits, dim=1).reshape(b, n, hw, num_samples, 1)
    return h, dec, dists, vw


def _view_logit_column(x2: torch.Tensor, vf: torch.Tensor) -> torch.Tensor:
    """lv = x2 . vf in fp32 (one read of the bf16 features on cd360_rowdot1_bf16; torch.mv elsewhere, and whenever autograd has to
    differentiate the product with respect to vf -- the precomputed-tables route under grad; the training path proper forms lv inside
    grad.NerfRenderFn and differentiates it there)."""
    if torch.is_grad_enabled() and vf.requires_grad:
        return torch.mv(x2.float(), vf.float())
    vf = vf.detach()
    if x2.is_cuda and x2.dtype == torch.bfloat16 and x2.shape[-1] % 8 == 0 and x2.is_contiguous() and not routes.library_linear:
        return ops.rowdot1(x2, vf.float().contiguous())
    return torch.mv(x2.float(), vf.float())


def reference_tables(fw: FusedNerfWeights, xref: torch.Tensor):
    """Per-pixel tables that depend only on the reference features (cacheable across target poses and steps):
    Y = xref @ Wf^T (bf16) and lv = xref @ vf (fp32)."""
    b, n, hw, C = xref.shape
    x2 = xref.reshape(b * n * hw, C)
    if x2.is_cuda and x2.dtype == torch.bfloat16 and C % 64 == 0 and not routes.library_linear:
        Y = ops.linear(x2, fw.Wf).reshape(b * n, hw, C)
    else:
        Y = torch.mm(x2.to(fw.Wf_t.dtype), fw.Wf_t).reshape(b * n, hw, C)
    lv = _view_logit_column(x2, fw.vf).reshape(b * n, hw)
    return Y.contiguous(), lv.contiguous()
