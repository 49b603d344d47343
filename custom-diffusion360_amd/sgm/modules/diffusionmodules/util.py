"""nn utilities of the UNet (reference sgm/modules/diffusionmodules/util.py:153-344), HIP-backed where it matters.

GroupNorm32 runs the fused channels-last GroupNorm(+SiLU) kernel (cd360_gn_silu_bf16); the thin factories
(conv_nd, linear, zero_module, timestep_embedding) keep the reference's names because openaimodel/attention
import them by name."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from cd360 import ops, routes


def checkpoint(func, inputs, params, flag):
    """Activation checkpointing is disabled on this path: the shipped config sets use_checkpoint False and the pose
    blocks cannot be checkpointed at all (non-tensor `pose` argument).  Kept for signature parity
    (util.py:153-168)."""
    if flag:
        raise NotImplementedError("use_checkpoint=True is not supported on the HIP path (and breaks pose blocks in the reference)")
    return func(*inputs)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """Sinusoidal embedding [N, dim] = [cos | sin] (util.py:206-231)."""
    if repeat_only:
        return timesteps[:, None].expand(-1, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def scale_module(module, scale):
    for p in module.parameters():
        p.detach().mul_(scale)
    return module


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def _fp32_affine(norm: nn.GroupNorm):
    """fp32 copies of gamma/beta for the kernel, refreshed when the parameters change."""
    key = (norm.weight.data_ptr(), norm.weight._version, norm.bias._version, norm.weight.device)
    cache = getattr(norm, "_cd360_affine", None)
    if cache is None or cache[0] != key:
        cache = (key, norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous())
        norm._cd360_affine = cache
    return cache[1], cache[2]


def tag_gn_stats(img: torch.Tensor, stats) -> torch.Tensor:
    """Attach the conv epilogue's GroupNorm slab statistics to the image VIEW a module returns, so that a GroupNorm in the NEXT
    module (SpatialTransformer.norm, the following ResBlock's in_layers, UNet.out) can skip its statistics pass.  The tag is a
    plain attribute of this Python object: it does not survive any tensor op, and the consumer re-checks address, shape and
    version counter before trusting it."""
    if stats is not None:
        img._cd360_gn_stats = (stats, img.data_ptr(), img._version, tuple(img.shape))
    return img


def _tagged_gn_stats(x: torch.Tensor):
    tag = getattr(x, "_cd360_gn_stats", None)
    if tag is None:
        return None
    stats, ptr, version, shape = tag
    return stats if (x.data_ptr() == ptr and x._version == version and tuple(x.shape) == shape) else None


def group_norm_tokens(norm: nn.GroupNorm, x: torch.Tensor, silu: bool, stats=None) -> torch.Tensor:
    """GroupNorm(+SiLU) of an image tensor [N, C, H, W]; returns channels-last tokens [N, H*W, C] in x's dtype.
    The token tensor aliases a channels-last image: `.reshape(N, H, W, C).permute(0, 3, 1, 2)` is free.
    `stats`: per-slab channel sums the producing conv already computed (conv_tokens(..., want_stats=True)); also picked up from
    a tag left by tag_gn_stats."""
    N, C, H, W = x.shape
    if stats is None:
        stats = _tagged_gn_stats(x)
    xt = x.permute(0, 2, 3, 1)
    if not xt.is_contiguous():
        xt = xt.contiguous()  # NCHW-contiguous input: one layout change, then everything stays channels-last
    xt = xt.reshape(N, H * W, C)
    dt = xt.dtype
    if dt != torch.bfloat16:
        xt = xt.to(torch.bfloat16)
        stats = None  # the statistics were taken of the bf16 tensor the conv wrote; x is something else
    if torch.is_grad_enabled() and (norm.weight.requires_grad or norm.bias.requires_grad):
        g, b = norm.weight.float(), norm.bias.float()  # `trainkeys: all`: live casts, the affine gradients come back through GroupNormSiluFn
    else:
        g, b = _fp32_affine(norm)
    y = ops.gn_silu(xt, g, b, norm.num_groups, norm.eps, silu, tile_stats=stats)
    return y if dt == torch.bfloat16 else y.to(dt)


def tokens_to_image(t: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """[N, H*W, C] -> [N, C, H, W] view in channels_last memory format (no copy)."""
    N, _, C = t.shape
    return t.reshape(N, H, W, C).permute(0, 3, 1, 2)


def packed_conv(conv: nn.Conv2d):
    """(w_packed [Cout_p, taps*Cin_p] bf16 in the kernel's K order (ops.pack_conv_weight), bias fp32 [Cout_p] | None) for
    cd360_conv_igemm_bf16, cached on the module and rebuilt when the parameters change.  Cin is zero-padded to a multiple of 64 and
    Cout to a multiple of 16 when needed (the UNet's 4 -> 320 input conv and 320 -> 4 output conv, openaimodel.py:663-670,967-973):
    conv_tokens pads the input channels / slices the output channels accordingly.  Envelope: 3x3 pad 1 (stride 1 | 2) or 1x1
    (stride 1), no dilation, no groups; returns None outside it."""
    w = conv.weight
    k = conv.kernel_size
    ok3 = k == (3, 3) and conv.padding == (1, 1) and conv.stride in ((1, 1), (2, 2))
    ok1 = k == (1, 1) and conv.padding == (0, 0) and conv.stride == (1, 1)
    if not (w.is_cuda and (ok3 or ok1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"):
        return None
    key = (w.data_ptr(), w._version, w.dtype, w.device, None if conv.bias is None else conv.bias._version)
    cache = getattr(conv, "_cd360_packed", None)
    if cache is None or cache[0] != key:
        cout, cin = w.shape[:2]
        cin_p, cout_p = -(-cin // 64) * 64, -(-cout // 16) * 16
        wd = w.detach()
        if (cin_p, cout_p) != (cin, cout):
            wd = torch.nn.functional.pad(wd, (0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout))
        bias = None
        if conv.bias is not None:
            bias = torch.nn.functional.pad(conv.bias.detach().float(), (0, cout_p - cout)).contiguous()
        cache = (key, ops.pack_conv_weight(wd), bias)
        conv._cd360_packed = cache
    return cache[1], cache[2]


def packed_conv_dgrad(conv: nn.Conv2d):
    """Packed weight of the DATA-GRADIENT convolution of `conv` for cd360_conv_igemm_bf16 (cd360.grad.ConvIgemmFn): channels
    transposed, taps flipped -- dx = conv_stride1(dy [zero-inserted for stride 2], w_d), w_d[ci, co, ky, kx] = w[co, ci, 2-ky, 2-kx].
    As an igemm weight its input channels (= Cout, padded like packed_conv's outputs) are zero-padded to a multiple of 64 and its
    output channels (= Cin padded to 64 by packed_conv) kept.  Cached on the module like packed_conv."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.dtype, w.device)
    cache = getattr(conv, "_cd360_packed_dgrad", None)
    if cache is None or cache[0] != key:
        cout, cin = w.shape[:2]
        cin_p, cout_p = -(-cin // 64) * 64, -(-(-(-cout // 16) * 16) // 64) * 64
        wt = w.detach().flip(2, 3).transpose(0, 1)  # [cin, cout, k, k]
        wt = torch.nn.functional.pad(wt, (0, 0, 0, 0, 0, cout_p - cout, 0, cin_p - cin))
        cache = (key, ops.pack_conv_weight(wt.contiguous()))
        conv._cd360_packed_dgrad = cache
    return cache[1]


def _frozen(conv: nn.Conv2d) -> bool:
    """True when the conv's own parameters need no gradient (always, for trainkeys pose / poseattn: diffusion.py:117-150); the
    implicit-GEMM kernel has a data gradient but no weight gradient, so a trainable conv stays on torch's own convolution."""
    return not (torch.is_grad_enabled() and (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)))


def conv_tokens(conv: nn.Conv2d, tokens: torch.Tensor, N: int, H: int, W: int, emb=None, res=None, want_stats: bool = False):
    """conv(3x3 pad 1, stride 1 | 2; or 1x1) on channels-last tokens [N, H*W, Cin] -> [N, Ho*Wo, Cout], with the per-image addend
    `emb` [N, Cout] and the residual `res` [N, Ho*Wo, Cout] fused into the epilogue (cd360_conv_igemm_bf16); MIOpen only outside the
    kernel's envelope (none of the SDXL UNet's convs).  want_stats=True returns (tokens, stats): the GroupNorm slab statistics
    of the output, or None when the kernel cannot give them."""
    pk = packed_conv(conv) if (tokens.dtype == torch.bfloat16 and _frozen(conv)) else None
    if pk is None:
        y = conv(tokens_to_image(tokens, H, W))
        if emb is not None:
            y = y + emb[:, :, None, None]
        y = y.permute(0, 2, 3, 1).reshape(N, y.shape[2] * y.shape[3], -1)
        y = y if res is None else y + res
        return (y, None) if want_stats else y
    taps = 9 if conv.kernel_size == (3, 3) else 1
    stride = conv.stride[0]
    cin, cout = conv.in_channels, conv.out_channels
    cin_p = pk[0].shape[1] // taps
    cout_p = pk[0].shape[0]
    if cin_p != cin:
        tokens = torch.nn.functional.pad(tokens, (0, cin_p - cin))  # zero channels against zero weights
    padded_out = cout_p != cout
    if padded_out and (emb is not None or res is not None):
        raise NotImplementedError("emb / res epilogue with a padded channel count")
    stats_ok = want_stats and not padded_out and ((H // stride) * (W // stride)) % 128 == 0 and not routes.no_gn_stats
    out = ops.conv_igemm(tokens, pk[0], pk[1], N, H, W, taps, emb, res, want_stats=stats_ok, stride=stride, alg_channels=(cin, cout),
                         w_dgrad=lambda: packed_conv_dgrad(conv))
    y, stats = out if stats_ok else (out, None)
    if padded_out:
        y = y[..., :cout]
    return (y, stats) if want_stats else y


def conv_image(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """A plain nn.Conv2d call site (the UNet's input conv, Downsample.op, the output conv) on the implicit-GEMM kernel: image in,
    image out, channels-last in between; CPU / non-bf16 tensors go through the module itself."""
    if not (x.is_cuda and x.dtype == torch.bfloat16) or routes.edge_convs_miopen or not _frozen(conv) or packed_conv(conv) is None:
        return conv(x)  # (the environment knob keeps these four convs on MIOpen, for A/B runs)
    N, _, H, W = x.shape
    xt = x.permute(0, 2, 3, 1)
    xt = (xt if xt.is_contiguous() else xt.contiguous()).reshape(N, H * W, -1)
    s = conv.stride[0]
    return tokens_to_image(conv_tokens(conv, xt, N, H, W), H // s, W // s)


class GroupNorm32(nn.GroupNorm):
    """fp32-statistics GroupNorm (util.py:309-311) on the HIP kernel."""

    def forward(self, x):
        N, C, H, W = x.shape
        return tokens_to_image(group_norm_tokens(self, x, silu=False), H, W)


def normalization(channels):
    return GroupNorm32(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims == 2:
        return nn.Conv2d(*args, **kwargs)
    if dims == 1:
        return nn.Conv1d(*args, **kwargs)
    if dims == 3:
        return nn.Conv3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


class HipLinear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose bf16 GPU forward is the hand-written MFMA GEMM in EVERY mode: no tape ->
    one cd360_gemm_bf16 launch with the bias fused; under autograd -> cd360.grad.LinearFn (data gradient on the same kernel, weight
    gradient on cd360_gemm_tn_bf16).  Whatever route reaches the module -- the fused inference blocks read its weight directly, but
    sample.py's patched forwards (sample.py:247-262), hooked blocks (diffusion.py:151-163) and the fine-tuning step call it -- runs the
    same kernel.  Other dtypes / devices / shapes outside the kernel's envelope (K % 64, N % 16) take torch's F.linear."""

    def forward(self, x):
        if ops.linear_ok(x, self.weight) and not routes.library_linear:
            return ops.linear(x, self.weight, self.bias)
        return torch.nn.functional.linear(x, self.weight, self.bias)


class HipLayerNorm(nn.LayerNorm):
    """nn.LayerNorm whose bf16 GPU forward is cd360_add_layernorm_bf16 (differentiable: cd360_add_layernorm_bwd_bf16), so that a route
    that calls `self.norm1(x)` itself (sample.py:33-80) stays on the HIP kernels."""

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.bfloat16 and self.weight is not None and self.weight.dtype == torch.bfloat16 and self.bias is not None
                and len(self.normalized_shape) == 1 and x.shape[-1] <= 2048 and x.shape[-1] % 8 == 0 and not routes.library_linear):
            return ops.add_layernorm(x.contiguous(), None, self.weight, self.bias, self.eps)[1]
        return super().forward(x)


def linear(*args, **kwargs):
    return HipLinear(*args, **kwargs)


def avg_pool_nd(dims, *args, **kwargs):
    if dims == 2:
        return nn.AvgPool2d(*args, **kwargs)
    if dims == 1:
        return nn.AvgPool1d(*args, **kwargs)
    if dims == 3:
        return nn.AvgPool3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")
