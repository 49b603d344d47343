"""SDXL UNet with a reference stream and pose-conditioned transformer blocks (reference
sgm/modules/diffusionmodules/openaimodel.py:73-376,525-1093), HIP-backed.

Constructor arguments, forward signature, the 4-tuple it returns and every state_dict key match the reference, so SDXL
safetensors and the delta checkpoint load unchanged.  Differences underneath: activations are kept channels-last in the
parameter dtype (bf16), GroupNorm+SiLU is one fused kernel in front of each conv, the transformers run on the HIP
attention / FeatureNeRF kernels, and there is no autocast (the reference's fp16 autocast region, :992, becomes "compute in
the parameters' dtype").  Convolutions and plain Linear layers stay on MIOpen / hipBLASLt through torch.
Unused reference options (dims != 2, resblock_updown, scale-shift norm, AttentionBlock, fairscale checkpointing,
integer / "continuous" / "timestep" class embeddings) raise NotImplementedError.
"""
from __future__ import annotations


from abc import abstractmethod
from typing import List, Optional, Tuple, Union

import torch
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from cd360 import ops, routes
from ...modules.attention import SpatialTransformer
from ...modules.diffusionmodules.util import (
    conv_image,
    conv_nd,
    conv_tokens,
    group_norm_tokens,
    linear,
    normalization,
    tag_gn_stats,
    _tagged_gn_stats,
    timestep_embedding,
    tokens_to_image,
    zero_module,
)
from ...util import default, exists


def _cat_channels(a, b):
    """th.cat([a, b], dim=1) (openaimodel.py:1074-1076); channels-last bf16 goes through cd360_concat_channels_bf16."""
    if a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape[1] % 8 == 0 and b.shape[1] % 8 == 0:
        cl = torch.channels_last
        out = ops.concat_channels(a.contiguous(memory_format=cl), b.contiguous(memory_format=cl))
        # GroupNorm statistics of a concatenation are the concatenated per-channel statistics: when both inputs still carry the slab sums
        # their producers' epilogues took (tag_gn_stats), the in_layers GroupNorm of the ResBlock that reads `out` needs no pass over it
        sa, sb = _tagged_gn_stats(a), _tagged_gn_stats(b)
        if sa is not None and sb is not None and not torch.is_grad_enabled() and not (routes.no_gn_stats or routes.no_concat_stats):
            na, nb = sa.shape[1], sb.shape[1]  # [N, slabs, C, 2]: bring both to the coarser slab count (slabs are consecutive pixel runs)
            if na > nb and na % nb == 0:
                sa = sa.reshape(sa.shape[0], nb, na // nb, sa.shape[2], 2).sum(2)
            elif nb > na and nb % na == 0:
                sb = sb.reshape(sb.shape[0], na, nb // na, sb.shape[2], 2).sum(2)
            if sa.shape[1] == sb.shape[1]:
                sa, sb = sa.contiguous(), sb.contiguous()
                tag_gn_stats(out, ops.concat_gn_stats(sa, sb) if sa.dtype == sb.dtype == torch.float32 else torch.cat([sa, sb], dim=2))
        return out
    return th.cat([a, b], dim=1)


class TimestepBlock(nn.Module):
    @abstractmethod
    def forward(self, x, emb):
        """Apply the module to `x` given `emb` timestep embeddings."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Threads (x, emb) and the no-grad reference stream (xr, embr) through its children (openaimodel.py:73-111)."""

    def forward(self, x, emb, context=None, xr=None, embr=None, contextr=None, pose=None, mask_ref=None, prev_weights=None):
        weights = fg_mask = alphas = predicted_rgb = None
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
                if xr is not None:
                    with torch.no_grad():
                        xr = layer(xr, embr).detach()
            elif isinstance(layer, SpatialTransformer):
                x, xr, fg_mask, weights, alphas, predicted_rgb = layer(x, xr, context, contextr, pose, mask_ref, prev_weights=prev_weights)
            else:
                call = (lambda v: conv_image(layer, v)) if isinstance(layer, nn.Conv2d) else layer  # (conv_image itself defers to MIOpen for a trainable conv)
                x = call(x)
                if xr is not None:
                    with torch.no_grad():
                        xr = conv_image(layer, xr).detach() if isinstance(layer, nn.Conv2d) else layer(xr).detach()
        return x, xr, fg_mask, weights, alphas, predicted_rgb


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_up=False, kernel_size=3, scale_factor=2):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("2-D only")
        self.channels, self.out_channels, self.use_conv, self.dims, self.scale_factor = channels, out_channels or channels, use_conv, dims, scale_factor
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, kernel_size, padding=padding)

    def _phase_weights(self):
        """The conv's weight as four 2 x 2-tap phase kernels (ops.pack_upsample_conv_weight), rebuilt when the parameters change."""
        w, b = self.conv.weight, self.conv.bias
        key = (w.data_ptr(), w._version, None if b is None else b._version)
        if getattr(self, "_up_pack", None) is None or self._up_pack[0] != key:
            self._up_pack = (key, ops.pack_upsample_conv_weight(w), None if b is None else b.detach().float().contiguous())
        return self._up_pack[1], self._up_pack[2]

    def forward(self, x):
        assert x.shape[1] == self.channels
        conv = self.conv if self.use_conv else None
        if (conv is not None and x.is_cuda and x.dtype == torch.bfloat16 and self.scale_factor == 2 and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
                and conv.stride == (1, 1) and conv.weight.dtype == torch.bfloat16 and self.channels % 64 == 0 and self.out_channels % 16 == 0
                and not torch.is_grad_enabled() and not routes.no_upsample_fold):
            # nearest 2x + conv3x3 as four 2 x 2-tap phase convolutions of the source image: no upsampled intermediate, 4 / 9 of the MACs
            N, _, H, W = x.shape
            xt = x.permute(0, 2, 3, 1)
            xt = (xt if xt.is_contiguous() else xt.contiguous()).reshape(N, H * W, -1)
            wp, bias = self._phase_weights()
            return tokens_to_image(ops.conv_up2x(xt, wp, bias, N, H, W), 2 * H, 2 * W)
        x = F.interpolate(x, scale_factor=self.scale_factor, mode="nearest")
        if not self.use_conv:
            return x
        N, _, H, W = x.shape
        xt = x.permute(0, 2, 3, 1)
        xt = (xt if xt.is_contiguous() else xt.contiguous()).reshape(N, H * W, -1)
        return tokens_to_image(conv_tokens(self.conv, xt, N, H, W), H, W)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_down=False):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("2-D only")
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if use_conv:
            self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return conv_image(self.op, x) if isinstance(self.op, nn.Conv2d) else self.op(x)


class ResBlock(TimestepBlock):
    """GN -> SiLU -> conv3x3 (+ emb) -> GN -> SiLU -> conv3x3 + skip (openaimodel.py:233-376); GN+SiLU is one HIP kernel."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False, kernel_size=3, exchange_temb_dims=False, skip_t_emb=False):
        super().__init__()
        if use_scale_shift_norm or up or down or exchange_temb_dims or skip_t_emb or dims != 2 or use_checkpoint:
            raise NotImplementedError("ResBlock option not used by the SDXL config")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint, self.use_scale_shift_norm = use_conv, use_checkpoint, use_scale_shift_norm
        padding = kernel_size // 2
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, kernel_size, padding=padding))
        self.updown = False
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, kernel_size, padding=padding)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, kernel_size, padding=padding)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    @staticmethod
    def _gn_silu(norm, x):
        return tokens_to_image(group_norm_tokens(norm, x, silu=True), x.shape[2], x.shape[3])

    def forward(self, x, emb):
        return self._forward(x, emb)

    def _forward(self, x, emb):
        """Works on channels-last tokens: GN+SiLU kernel -> implicit-GEMM conv with `+ emb` fused -> GN+SiLU -> conv with the skip
        connection fused as residual.  No elementwise kernels are left between the four launches."""
        N, _, H, W = x.shape
        t = group_norm_tokens(self.in_layers[0], x, silu=True)  # [N, HW, Cin]
        pre = getattr(emb, "_cd360_emb_outs", None)  # UNetModel: every ResBlock's emb_layers(emb) from ONE GEMM (column slices)
        emb_out = pre[id(self)] if (pre is not None and id(self) in pre) else self.emb_layers(emb).type(t.dtype)
        if emb_out.stride(-1) != 1 or emb_out.dtype != t.dtype:
            emb_out = emb_out.to(t.dtype).contiguous()
        h, h_stats = conv_tokens(self.in_layers[2], t, N, H, W, emb=emb_out, want_stats=True)
        t2 = group_norm_tokens(self.out_layers[0], tokens_to_image(h, H, W), silu=True, stats=h_stats)  # statistics from the conv epilogue
        xt = x.permute(0, 2, 3, 1)
        xt = (xt if xt.is_contiguous() else xt.contiguous()).reshape(N, H * W, -1)
        if isinstance(self.skip_connection, nn.Identity):
            skip = xt
        else:
            skip = conv_tokens(self.skip_connection, xt, N, H, W)
        out, out_stats = conv_tokens(self.out_layers[3], t2, N, H, W, res=skip if skip.dtype == t2.dtype else skip.to(t2.dtype), want_stats=True)
        return tag_gn_stats(tokens_to_image(out, H, W), out_stats)  # the next module's GroupNorm (if any) reuses them


class UNetModel(nn.Module):
    def __init__(self, in_channels: int, model_channels: int, out_channels: int, num_res_blocks: int, attention_resolutions,
                 dropout: float = 0.0, channel_mult: Union[List, Tuple] = (1, 2, 4, 8), conv_resample: bool = True, dims: int = 2,
                 num_classes: Optional[Union[int, str]] = None, use_checkpoint: bool = False, num_heads: int = -1,
                 num_head_channels: int = -1, num_heads_upsample: int = -1, use_scale_shift_norm: bool = False,
                 resblock_updown: bool = False, transformer_depth=1, context_dim: Optional[int] = None,
                 disable_self_attentions: Optional[List[bool]] = None, num_attention_blocks: Optional[List[int]] = None,
                 disable_middle_self_attn: bool = False, use_linear_in_transformer: bool = False,
                 spatial_transformer_attn_type: str = "softmax", adm_in_channels: Optional[int] = None, use_fairscale_checkpoint=False,
                 offload_to_cpu=False, transformer_depth_middle: Optional[int] = None,
                 # pose-conditioning arguments (openaimodel.py:584-599)
                 image_cross_blocks=None, rgb: bool = False, far: float = 2.0, num_samples: float = 32,
                 not_add_context_in_triplane: bool = False, rgb_predict: bool = False, add_lora: bool = False, mode: str = "feature-nerf",
                 average: bool = False, num_freqs: int = 16, use_prev_weights_imp_sample: bool = False, stratified: bool = False,
                 poscontrol_interval: int = 4, imp_sampling_percent: float = 0.9, near_plane: float = 0.0):
        super().__init__()
        if resblock_updown or use_scale_shift_norm or dims != 2 or use_fairscale_checkpoint or use_checkpoint:
            raise NotImplementedError("UNet option not used by the SDXL config")
        if num_classes != "sequential":
            raise NotImplementedError("only num_classes='sequential' (SDXL vector conditioning) is built")
        if num_head_channels == -1:
            raise NotImplementedError("set num_head_channels (SDXL uses 64)")
        if disable_self_attentions is not None or num_attention_blocks is not None or disable_middle_self_attn:
            raise NotImplementedError("attention-ablation options are not used by the SDXL config")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.rgb, self.rgb_predict = rgb, rgb_predict
        image_cross_blocks = list(image_cross_blocks) if image_cross_blocks is not None else []
        channel_mult = list(channel_mult)
        attention_resolutions = list(attention_resolutions)
        transformer_depth = len(channel_mult) * [transformer_depth] if isinstance(transformer_depth, int) else list(transformer_depth)
        transformer_depth_middle = default(transformer_depth_middle, transformer_depth[-1])
        self.num_res_blocks = len(channel_mult) * [num_res_blocks] if isinstance(num_res_blocks, int) else list(num_res_blocks)
        self.attention_resolutions, self.dropout, self.channel_mult, self.conv_resample = attention_resolutions, dropout, channel_mult, conv_resample
        self.num_classes, self.use_checkpoint = num_classes, use_checkpoint
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), nn.SiLU(), linear(time_embed_dim, time_embed_dim))
        assert adm_in_channels is not None
        self.label_emb = nn.Sequential(nn.Sequential(linear(adm_in_channels, time_embed_dim), nn.SiLU(), linear(time_embed_dim, time_embed_dim)))

        st_kw = dict(context_dim=context_dim, use_linear=use_linear_in_transformer, attn_type=spatial_transformer_attn_type,
                     use_checkpoint=use_checkpoint, rgb_predict=rgb_predict, far=far, num_samples=num_samples, add_lora=add_lora, mode=mode,
                     average=average, num_freqs=num_freqs, use_prev_weights_imp_sample=use_prev_weights_imp_sample, stratified=stratified,
                     poscontrol_interval=poscontrol_interval, imp_sampling_percent=imp_sampling_percent, near_plane=near_plane)
        id_attention = 0

        def transformer(ch, depth):
            nonlocal id_attention
            st = SpatialTransformer(ch, ch // num_head_channels, num_head_channels, depth=depth, disable_self_attn=False,
                                    image_cross=(id_attention in image_cross_blocks), **st_kw)
            id_attention += 1
            return st

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        input_block_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(transformer(ch, transformer_depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                input_block_chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, dims=dims), transformer(ch, transformer_depth_middle), ResBlock(ch, time_embed_dim, dropout, dims=dims))

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, time_embed_dim, dropout, out_channels=model_channels * mult, dims=dims)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(transformer(ch, transformer_depth[level]))
                if level and i == self.num_res_blocks[level]:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))

        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        self._emb_cat = None  # (key, concatenated emb_layers weights, biases) of _tag_emb_outs
        self._last_pose_out = None  # index of the last output block holding a pose block (set on first forward)

    @property
    def dtype(self):
        return self.out[2].weight.dtype

    def _tag_emb_outs(self, emb: torch.Tensor) -> torch.Tensor:
        """`emb_layers` of all ResBlocks (SiLU -> Linear(4*model_channels, out_channels), openaimodel.py:296-303) applied to the
        same `emb` as ONE GEMM against the concatenated weights: 17 SiLU launches and 17 GEMMs with M = batch (3) rows -- 17 us
        each on hipBLASLt -- become one of each.  The result rides on the `emb` tensor object as column slices keyed by block;
        a ResBlock that does not find its slice computes its own projection as before."""
        if not (emb.is_cuda and emb.dtype == torch.bfloat16) or torch.is_grad_enabled() or routes.no_emb_merge:
            return emb
        blocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        key = tuple((b.emb_layers[1].weight.data_ptr(), b.emb_layers[1].weight._version, b.emb_layers[1].bias._version) for b in blocks)
        if self._emb_cat is None or self._emb_cat[0] != key:
            w = torch.cat([b.emb_layers[1].weight.detach() for b in blocks], 0).contiguous()
            bias = torch.cat([b.emb_layers[1].bias.detach() for b in blocks], 0).contiguous()
            self._emb_cat = (key, w, bias)
        return self._emb_outs_from_act(torch.nn.functional.silu(emb), emb, blocks)

    def _emb_outs_from_act(self, act: torch.Tensor, emb: torch.Tensor, blocks=None) -> torch.Tensor:
        """The merged emb_layers GEMM on act = silu(emb); the per-block column slices ride on `emb` (see _tag_emb_outs)."""
        if blocks is None:
            blocks = [m for m in self.modules() if isinstance(m, ResBlock)]
            key = tuple((b.emb_layers[1].weight.data_ptr(), b.emb_layers[1].weight._version, b.emb_layers[1].bias._version) for b in blocks)
            if self._emb_cat is None or self._emb_cat[0] != key:
                w = torch.cat([b.emb_layers[1].weight.detach() for b in blocks], 0).contiguous()
                bias = torch.cat([b.emb_layers[1].bias.detach() for b in blocks], 0).contiguous()
                self._emb_cat = (key, w, bias)
        if ops.linear_ok(act, self._emb_cat[1]) and not routes.library_linear:
            allp = ops.linear(act, self._emb_cat[1], self._emb_cat[2])  # [b, sum(out_channels)] on the hand-written GEMM (M = batch rows)
        else:
            allp = torch.nn.functional.linear(act, self._emb_cat[1], self._emb_cat[2])
        outs, off = {}, 0
        for blk in blocks:
            c = blk.out_channels
            outs[id(blk)] = allp[:, off:off + c]
            off += c
        emb._cd360_emb_outs = outs
        return emb

    def forward(self, x, timesteps=None, context=None, y=None, timesteps2=None, **kwargs):
        """x [b,4,L,L]; context [b (+ b*n), 77, ctx]; y [b (+ b*n), adm]; kwargs: pose, mask_ref, input_ref [b,n,4,L,L], sigmas_ref.
        -> (eps [b,4,L,L], fg_mask_list, alphas_list, predicted_rgb_list)   (openaimodel.py:975-1093)"""
        dt = self.dtype
        b = x.size(0)
        pose, mask_ref = kwargs.get("pose"), kwargs.get("mask_ref")
        reference_image = "input_ref" in kwargs and kwargs["input_ref"] is not None
        contextr = embr = hr = None
        if "input_ref" in kwargs:
            contextr = context[b:]
            yr = y[b:] if y is not None else None
            xr = kwargs["input_ref"]
            if xr is not None:
                b, n = xr.shape[:2]
        context = context[:b].to(dt)
        assert y is not None, "must specify y: the model is class-conditional (num_classes='sequential')"
        y = y[:b]
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels).to(dt))
        assert y.shape[0] == x.shape[0]
        emb = self._tag_emb_outs(emb + self.label_emb(y.to(dt)))

        h = x.to(dt).contiguous(memory_format=torch.channels_last)
        if reference_image:
            with torch.no_grad():
                if "sigmas_ref" in kwargs and kwargs["sigmas_ref"] is not None:
                    t_embr = timestep_embedding(kwargs["sigmas_ref"], self.model_channels)
                elif timesteps2 is not None:
                    t_embr = timestep_embedding(timesteps2, self.model_channels)
                else:
                    t_embr = timestep_embedding(torch.zeros_like(timesteps), self.model_channels)
                embr = self.time_embed(t_embr.to(dt))[:, None].expand(-1, n, -1).reshape(b * n, -1)
                embr = self._tag_emb_outs(embr + self.label_emb(yr.reshape(b * n, -1).to(dt)))
                contextr = contextr.to(dt)
                hr = xr.reshape(b * n, *xr.shape[2:]).to(dt).contiguous(memory_format=torch.channels_last)

        h, fg_mask_list, alphas_list, predicted_rgb_list = self._trunk(h, emb, context, hr, embr, contextr, pose, mask_ref, reference_image)
        if h.is_cuda and h.dtype == torch.bfloat16 and not torch.is_grad_enabled():
            out = tokens_to_image(self._out_conv_tokens(group_norm_tokens(self.out[0], h, silu=True), h.shape[0], h.shape[2], h.shape[3]), h.shape[2], h.shape[3])
        else:
            out = conv_image(self.out[2], tokens_to_image(group_norm_tokens(self.out[0], h, silu=True), h.shape[2], h.shape[3]))
        return out.type(x.dtype), fg_mask_list, alphas_list, predicted_rgb_list

    def _trunk(self, h, emb, context, hr, embr, contextr, pose, mask_ref, reference_image, first_done: bool = False):
        """input_blocks -> middle_block -> output_blocks (openaimodel.py:1032-1084).  first_done: `h` already is the output of
        input_blocks[0] (the input convolution: the sampling job's stage-in kernel computes it, forward_staged)."""
        fg_mask_list, alphas_list, predicted_rgb_list = [], [], []

        def collect(fg, al, rgb):
            if fg is not None:
                fg_mask_list.extend(fg)
            if al is not None:
                alphas_list.extend(al)
            if rgb is not None:
                predicted_rgb_list.extend(rgb)

        hs, hrs = [], []
        for bi_, module in enumerate(self.input_blocks):
            if bi_ == 0 and first_done:
                hs.append(h)
                hrs.append(hr)
                continue
            h, hr, fg, _, al, rgb = module(h, emb, context, hr, embr, contextr, pose, mask_ref=mask_ref, prev_weights=None)
            collect(fg, al, rgb)
            hs.append(h)
            hrs.append(hr)
        h, hr, fg, _, al, rgb = self.middle_block(h, emb, context, hr, embr, contextr, pose, mask_ref=mask_ref, prev_weights=None)
        collect(fg, al, rgb)
        # The reference stream only exists to feed the pose blocks (their `context_ref`); past the last one its activations are
        # never read again (the reference still computes them, openaimodel.py:1071-1084).  It is dropped there: three level-0
        # ResBlocks on b*n images in the SDXL layout.
        if self._last_pose_out is None:
            self._last_pose_out = max((i for i, m in enumerate(self.output_blocks)
                                       if any(isinstance(l, SpatialTransformer) and l.image_cross for l in m)), default=-1)
        for i, module in enumerate(self.output_blocks):
            h = _cat_channels(h, hs.pop())
            hrp = hrs.pop()
            if i > self._last_pose_out:
                hr = None
            if reference_image and hr is not None:
                hr = _cat_channels(hr, hrp)
            h, hr, fg, _, al, rgb = module(h, emb, context, hr, embr, contextr, pose, mask_ref=mask_ref, prev_weights=None)
            collect(fg, al, rgb)
        return h, fg_mask_list, alphas_list, predicted_rgb_list

    @torch.no_grad()
    def forward_staged(self, h_tokens, emb_act, context, pose, H: int, W: int):
        """The sampling job's entry (cd360/job.py::Sampler, captured steps): `h_tokens` [b, H W, model_channels] bf16 = the input
        convolution of the scaled latent and `emb_act` [b, 4 model_channels] bf16 = silu(time_embed(..) + label_emb(y)), both written by
        cd360_unet_stage_in; no reference stream (sample.py's mode).  Returns the output convolution's channels-last rows
        [b, H W, out_channels] bf16 (a channel slice of the kernel's 16-wide rows) for cd360_cfg_euler_step_cl -- forward()'s arithmetic
        between those two points, launch for launch."""
        emb = self._emb_outs_from_act(emb_act, emb_act)
        h = tokens_to_image(h_tokens, H, W)
        h, _, _, _ = self._trunk(h, emb, context.to(self.dtype), None, None, None, pose, None, False, first_done=True)
        t = group_norm_tokens(self.out[0], h, silu=True)
        return self._out_conv_tokens(t, h.shape[0], H, W)

    def _out_conv_tokens(self, t, N: int, H: int, W: int):
        """self.out[2] (Conv2d(model_channels -> out_channels, 3 x 3, padding 1), openaimodel.py:967-973) on activated tokens [N, H W, C] ->
        [N, H W, out_channels]: four output channels at an image width the small kernel serves go to cd360_out_conv4_bf16 (one pass over the
        activation instead of a 256-column GEMM tile with 252 empty columns); everything else through conv_tokens."""
        conv = self.out[2]
        if (conv.out_channels == 4 and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1) and t.is_cuda
                and t.dtype == torch.bfloat16 and conv.weight.dtype == torch.bfloat16 and ops.out_conv4_ok(H, W, conv.in_channels)
                and not torch.is_grad_enabled() and not routes.edge_convs_miopen and not routes.no_out_conv4 and conv.groups == 1 and conv.dilation == (1, 1)):
            w, b = conv.weight, conv.bias
            key = (w.data_ptr(), w._version, None if b is None else b._version)
            if getattr(self, "_out4_pack", None) is None or self._out4_pack[0] != key:
                bias = b.detach().float().contiguous() if b is not None else torch.zeros(4, device=w.device)
                self._out4_pack = (key, ops.pack_out_conv4_weight(w), bias)
            return ops.out_conv4(t.contiguous(), self._out4_pack[1], self._out4_pack[2], N, H, W)
        return conv_tokens(conv, t, N, H, W)
