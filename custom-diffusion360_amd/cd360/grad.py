"""Autograd recording of the differentiable cd360 operators (BASELINE config 4: the fine-tuning loop differentiates the pose
path; the reference relies on torch autograd through xformers, grid_sample, nn.Linear ... -- diffusion.py:226-241, main.py).

Every class is a torch.autograd.Function whose forward AND backward are HIP kernels behind the C ABI (ops.py), the Linear layers
included (LinearFn: cd360_gemm_bf16 forward and data gradient, cd360_gemm_tn_bf16 weight gradient).  ops.* dispatch here when an input requires grad and grad mode is
on; under torch.no_grad() (sampling) nothing in this file runs.  Parameters that the shipped configs never train through these
operators (norm affines, conv weights: trainkeys in {pose, poseattn}, diffusion.py:117-150) get no wgrad kernel: asking for one
raises instead of silently returning None."""
from __future__ import annotations

import torch

from . import ops


def _no_wgrad(name: str, *params):
    for p in params:
        if p is not None and p.requires_grad:
            raise NotImplementedError(f"{name}: no weight-gradient kernel (the reference's trainkeys pose / poseattn never train these)")


class LinearFn(torch.autograd.Function):
    """ops.linear: y = x W^T + b (+ res) (nn.Linear / F.linear at attention.py:89-115,323-329,368-372,422,515-516,748,786; openaimodel.py
    time / label / ResBlock embeddings) on cd360_gemm_bf16.  Backward: dX = dY W is the SAME kernel on the transposed weight (cached for
    frozen weights); dW = dY^T X is cd360_gemm_tn_bf16 (fp32 accumulation, only for weights that train: trainkeys pose -> pose_emb_layers,
    plane_coefs); db = column sums of dY; d_res = dY."""

    @staticmethod
    def forward(ctx, x, weight, bias, res):
        w = weight.detach()
        if w.stride(1) != 1 or w.stride(0) % 8:
            w = w.contiguous()
        out = ops.gemm(x.detach(), w, bias=ops.bias_f32(bias), res=None if res is None else res.detach())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = None if bias is None else bias.dtype
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, K = weight.shape
        dy = dy.contiguous()
        dy2, x2 = dy.reshape(-1, N), x.reshape(-1, K)
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            if ops.gemm_ok(dy2.shape[0], K, N):
                dx = ops.gemm(dy2, ops.weight_t(weight)).reshape(x.shape)
            else:  # outside the kernel's envelope (N % 64 != 0): not a shape of the shipped configs
                dx = torch.mm(dy2, weight.detach()).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            if ops.gemm_tn_ok(dy2, x2):
                dw = ops.gemm_tn(dy2, x2, out_dtype=torch.bfloat16).to(weight.dtype)
            else:
                dw = torch.mm(dy2.t(), x2).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0, dtype=torch.float32).to(ctx.bias_dtype)
        if ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres


class AttentionFn(torch.autograd.Function):
    """ops.attention (attention.py:393-408) with cd360_attn_fwd_lse_bf16 / cd360_attn_bwd_bf16."""

    @staticmethod
    def forward(ctx, q, k, v, heads, nk):
        out, lse = ops.attention(q, k, v, heads, nk, want_lse=True)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.heads, ctx.nk = heads, nk
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        need_dq = ctx.needs_input_grad[0]
        need_dkv = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dq, dk, dv = ops.attention_bwd(q, k, v, out, dout, lse, ctx.heads, ctx.nk, need_dq, need_dkv)
        return dq, dk, dv, None, None


class SelfAttentionFn(torch.autograd.Function):
    """ops.self_attention_qkv: q, k, v are column slices of one projection output; d(q|k|v) is written in place by the kernel."""

    @staticmethod
    def forward(ctx, qkv, heads):
        inner = heads * 64
        out, lse = ops.attention(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], heads, qkv.shape[1], want_lse=True)
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        inner = ctx.heads * 64
        dqkv = torch.empty_like(qkv)
        sl = lambda t: (t[..., :inner], t[..., inner:2 * inner], t[..., 2 * inner:])
        q, k, v = sl(qkv)
        ops.attention_bwd(q, k, v, out, dout, lse, ctx.heads, qkv.shape[1], out=sl(dqkv))
        return dqkv, None


class VolRenderFn(torch.autograd.Function):
    """ops.volrender (_TruncExp + VolRender, attention.py:192-208, nerfsd_pytorch3d.py:170-231) with cd360_volrender_bwd."""

    @staticmethod
    def forward(ctx, feats, sigma_raw, dists, rgb_raw, want_weights, sigma_is_raw, rgb_is_raw):
        rendered, fg, alphas, weights, rgb = ops.volrender(feats, sigma_raw, dists, rgb_raw, want_weights, sigma_is_raw, rgb_is_raw)
        ctx.save_for_backward(feats, sigma_raw, dists, rgb_raw)
        ctx.set_materialize_grads(False)
        ctx.flags = (sigma_is_raw, rgb_is_raw)
        return rendered, fg, alphas, weights, rgb

    @staticmethod
    def backward(ctx, d_rendered, d_fg, d_alphas, d_weights, d_rgb):
        feats, sigma_raw, dists, rgb_raw = ctx.saved_tensors
        d_feats, d_sigma, d_rgb_raw = ops.volrender_bwd(feats, sigma_raw, dists, rgb_raw, d_rendered, d_fg, d_alphas, d_weights, d_rgb, *ctx.flags)
        d_sigma = d_sigma.reshape(sigma_raw.shape).to(sigma_raw.dtype)
        if d_rgb_raw is not None:
            d_rgb_raw = d_rgb_raw.reshape(rgb_raw.shape).to(rgb_raw.dtype)
        return d_feats, d_sigma, None, d_rgb_raw, None, None, None


class GroupNormSiluFn(torch.autograd.Function):
    """ops.gn_silu (GroupNorm32 + SiLU, diffusionmodules/util.py:309-311, openaimodel.py:280-328) with cd360_gn_silu_bwd_bf16."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, tile_stats):
        y = ops.gn_silu(x, gamma, beta, groups, eps, silu, tile_stats=tile_stats)
        ctx.save_for_backward(x, gamma, beta)
        ctx.cfg = (groups, eps, silu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        groups, eps, silu = ctx.cfg
        dx = ops.gn_silu_bwd(x, dy, gamma, beta, groups, eps, silu) if ctx.needs_input_grad[0] else None
        dgamma = dbeta = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # affine gradients (`trainkeys: all`, diffusion.py:145-147 -- not a set the shipped configs train): fp32 torch reductions over
            # the recomputed normalised activations; the hot path (trainkeys pose / poseattn) never asks for them
            C = x.shape[-1]
            xf = x.float().reshape(x.shape[0], -1, groups, C // groups)
            mu = xf.mean((1, 3), keepdim=True)
            xhat = ((xf - mu) * torch.rsqrt(xf.var((1, 3), unbiased=False, keepdim=True) + eps)).reshape(x.shape[0], -1, C)
            du = dy.float().reshape(x.shape[0], -1, C)
            if silu:
                u = xhat * gamma.float() + beta.float()
                sg = torch.sigmoid(u)
                du = du * (sg * (1 + u * (1 - sg)))
            dgamma = (du * xhat).sum((0, 1)).to(gamma.dtype) if ctx.needs_input_grad[1] else None
            dbeta = du.sum((0, 1)).to(beta.dtype) if ctx.needs_input_grad[2] else None
        return dx, dgamma, dbeta, None, None, None, None


class GegluFn(torch.autograd.Function):
    """ops.geglu (GEGLU.forward, attention.py:89-96) with cd360_geglu_bwd_bf16."""

    @staticmethod
    def forward(ctx, proj):
        ctx.save_for_backward(proj)
        return ops.geglu(proj)

    @staticmethod
    def backward(ctx, dy):
        (proj,) = ctx.saved_tensors
        return ops.geglu_bwd(proj, dy)


class AddLayerNormFn(torch.autograd.Function):
    """ops.add_layernorm (residual add + nn.LayerNorm, attention.py:609-636) with cd360_add_layernorm_bwd_bf16.  Returns
    (sum, ln) -- only ln when b is None (the caller keeps using `a` as the residual stream).  The backward adds the gradient that
    arrived on `sum` to the LayerNorm gradient in the same pass; a and b receive the same tensor."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps, alias=False):
        s, ln = ops.add_layernorm(a, b, gamma, beta, eps, want_sum=True)
        x = a if b is None else s
        ctx.save_for_backward(x, gamma)
        ctx.set_materialize_grads(False)
        ctx.eps, ctx.has_b, ctx.alias = eps, b is not None, alias and b is None
        if b is None:
            # alias: `a` itself comes back as the "sum" output, so a caller that keeps using the residual stream uses THIS tensor and the
            # gradient arriving on it joins the LayerNorm gradient inside the backward kernel (otherwise autograd adds the two in a pass
            # of its own over the whole tensor)
            return (a.view_as(a), ln) if alias else ln
        return s, ln

    @staticmethod
    def backward(ctx, *grads):
        x, gamma = ctx.saved_tensors
        d_sum, d_ln = (grads if (ctx.has_b or ctx.alias) else (None, grads[0]))
        if d_ln is None:
            dx = d_sum
        else:
            dx = ops.add_layernorm_bwd(x, gamma, d_ln, d_sum, ctx.eps)
        dgamma = dbeta = None
        if d_ln is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]):
            # affine gradients (`trainkeys: all`): fp32 torch reductions, off the hot path (trainkeys pose / poseattn freeze the norms)
            C = x.shape[-1]
            xf = x.float().reshape(-1, C)
            xhat = (xf - xf.mean(-1, keepdim=True)) * torch.rsqrt(xf.var(-1, unbiased=False, keepdim=True) + ctx.eps)
            g2 = d_ln.float().reshape(-1, C)
            dgamma = (g2 * xhat).sum(0).to(gamma.dtype) if ctx.needs_input_grad[2] else None
            dbeta = g2.sum(0).to(gamma.dtype) if ctx.needs_input_grad[3] else None
        return dx, (dx if ctx.has_b else None), dgamma, dbeta, None, None


def add_layernorm(a, b, gamma, beta, eps, want_sum=True, alias=False):
    """ops.add_layernorm's (sum | None, ln) contract under autograd.  alias (b None): (a as a fresh autograd alias, ln), see AddLayerNormFn."""
    if b is None:
        if alias:
            return AddLayerNormFn.apply(a, None, gamma, beta, eps, True)
        return None, AddLayerNormFn.apply(a, None, gamma, beta, eps)
    s, ln = AddLayerNormFn.apply(a, b, gamma, beta, eps)
    return (s if want_sum else None), ln


class NerfAggregateFn(torch.autograd.Function):
    """ops.nerf_mlp_aggregate (FeatureNeRFEncoding.forward, nerfsd_pytorch3d.py:53-158) with cd360_nerf_mlp_aggregate_bwd.
    The kernel leaves dz = softmax_i dg SiLU'(z_i) and the generated inputs F in HBM; the parameter-side reductions are one GEMM
    (dWk = dz^T F) and one sum over the depth samples (dzP)."""

    @staticmethod
    def forward(ctx, cams, xs, ys, t, Y, zP, lv, cview, Wk, img_map):
        g, logits, lse = ops.nerf_mlp_aggregate(cams, xs, ys, t, Y, zP, lv, cview, Wk, want_logits=True, img_map=img_map)
        ctx.save_for_backward(cams, xs, ys, t, Y, zP, lv, cview, Wk, g, lse)
        ctx.img_map = img_map
        ctx.mark_non_differentiable(logits, lse)
        return g, logits, lse

    @staticmethod
    def backward(ctx, dg, _dlogits, _dlse):
        cams, xs, ys, t, Y, zP, lv, cview, Wk, g, lse = ctx.saved_tensors
        dz, F, dY, dlv, dcview, _ = ops.nerf_mlp_aggregate_bwd(cams, xs, ys, t, Y, zP, lv, cview, Wk, ctx.img_map, g, lse, dg)
        b, n, npts, C = dz.shape
        S = t.shape[-1]
        dzP = dz.reshape(b * n, npts // S, S, C).sum(2, dtype=torch.float32).to(zP.dtype)
        dWk = ops.gemm_tn(dz.reshape(-1, C), F.reshape(-1, F.shape[-1])).to(Wk.dtype)
        return None, None, None, None, dY.to(Y.dtype), dzP, dlv, dcview, dWk, None


class NerfRenderFn(torch.autograd.Function):
    """The fused render from the reference FEATURES and the live weights (the training path): Y = xref Wf^T and lv = xref . vf are
    formed here, and the backward never scatters into those tables.  With xg = bilinear_gather(xref) at the sample positions
    (cd360_feature_gather, the same corner arithmetic as the fused kernel),
        dWf = dz^T xg          dvf = dlogit^T xg          dzP = sum_s dz          dcview = sum dlogit          dWk = dz^T F
    are three weight-gradient GEMMs (cd360_gemm_tn_bf16) and two reductions over tensors the backward kernel wrote once; the 4 x C atomic
    adds per (sample, view) of the table form (NerfAggregateFn) measured 22 ms per render at the training shapes, this form a fraction
    of it.  xref itself gets no gradient (the reference stream runs under no_grad: attention.py:845-857).  Wf [C_out, C_in] is the
    Linear-layout slice plane_coefs.0.weight[:, :C]."""

    @staticmethod
    def forward(ctx, cams, xs, ys, t, xref, Wf, vf, zP, cview, Wk):
        if xref.requires_grad:
            raise NotImplementedError("gradients with respect to the reference features are not provided (the reference stream is no_grad)")
        b, n, hw, C = xref.shape
        x2 = xref.reshape(b * n * hw, C)
        if ops.linear_ok(x2, Wf):
            Y = ops.linear(x2.detach(), Wf.detach()).reshape(b * n, hw, C)
        else:
            Y = torch.mm(x2.to(Wf.dtype), Wf.detach().t()).reshape(b * n, hw, C).contiguous()
        from .nerf import _view_logit_column
        lv = _view_logit_column(x2.detach(), vf).reshape(b * n, hw).contiguous()
        g, logits, lse = ops.nerf_mlp_aggregate(cams, xs, ys, t, Y, zP, lv, cview, Wk, want_logits=True)
        ctx.save_for_backward(cams, xs, ys, t, xref, Y, lv, zP, cview, Wk, g, lse)
        ctx.wf_dtype = Wf.dtype
        ctx.mark_non_differentiable(logits, lse)
        return g, logits, lse

    @staticmethod
    def backward(ctx, dg, _dlogits, _dlse):
        cams, xs, ys, t, xref, Y, lv, zP, cview, Wk, g, lse = ctx.saved_tensors
        dz, F, _, _, _, dlogit = ops.nerf_mlp_aggregate_bwd(cams, xs, ys, t, Y, zP, lv, cview, Wk, None, g, lse, dg, scatter=False)
        b, n, npts, C = dz.shape
        S = t.shape[-1]
        hw = npts // S
        grid = ops.ray_project_index(cams, xs, ys, t, want_points=False, want_index=False)["grid"].reshape(b * n, npts, 2)
        xg = ops.feature_gather(xref.reshape(b * n, hw, C).to(dz.dtype), grid).reshape(b * n * npts, C)
        dz2 = dz.reshape(b * n * npts, C)
        if ops.gemm_tn_ok(dz2, xg):
            dWf = ops.gemm_tn(dz2, xg).to(ctx.wf_dtype)                                 # [C_out, C_in], the layout of Wf
            # dlogit as columns 0 and 1 of an [M, 8] bf16 operand: high part and rounding remainder.  The view-logit gradients of a sample
            # sum to zero over its views (softmax), so dvf is what is left after heavy cancellation: with dlogit rounded ONCE to bf16 its
            # 2^-9 relative steps came back as 4.5e-2 ... 5.3e-2 of max |dvf| against the reference's autograd (round 6); the remainder
            # column costs nothing (the operand has eight columns either way) and leaves fp32-level rounding
            dl8 = torch.zeros(b * n * npts, 8, dtype=xg.dtype, device=xg.device)
            dflat = dlogit.reshape(-1).float()
            hi = dflat.to(xg.dtype)
            dl8[:, 0] = hi
            dl8[:, 1] = dflat - hi.float()
            dv2 = ops.gemm_tn(dl8, xg, out_dtype=torch.float32)
            dvf = (dv2[0] + dv2[1]).contiguous()
            dWk = ops.gemm_tn(dz2, F.reshape(-1, F.shape[-1])).to(Wk.dtype)
        else:
            dWf = torch.mm(dz2.t(), xg).to(ctx.wf_dtype)
            dflat = dlogit.reshape(1, -1).float()
            hi = dflat.to(xg.dtype)  # (high part + rounding remainder, as above)
            dvf = (torch.mm(hi, xg).float() + torch.mm((dflat - hi.float()).to(xg.dtype), xg).float()).reshape(C)
            dWk = torch.mm(dz2.t(), F.reshape(-1, F.shape[-1])).to(Wk.dtype)
        dzP = dz.reshape(b * n, hw, S, C).sum(2, dtype=torch.float32).to(zP.dtype)
        dcview = dlogit.reshape(b, n, npts).sum(-1)
        return None, None, None, None, None, dWf, dvf, dzP, dcview, dWk


class RenderLossFn(torch.autograd.Function):
    """ops.render_loss: the foreground / background / rgb terms of one pose block (loss.py:188-207 of the reference) as one kernel each way
    (cd360_render_loss_f32, cd360_render_loss_bwd_f32); the resized targets are constants of the step."""

    @staticmethod
    def forward(ctx, fg, alphas, rgb, op, bgw, mask_, want, den):
        ctx.save_for_backward(fg, alphas, rgb, op, bgw, mask_, want, den)
        return ops._render_loss_fwd(fg, alphas, rgb, op, bgw, mask_, want, den)

    @staticmethod
    def backward(ctx, g):
        fg, alphas, rgb, op, bgw, mask_, want, den = ctx.saved_tensors
        d_fg, d_al, d_rgb = ops.render_loss_bwd(fg, alphas, rgb, op, bgw, mask_, want, den, g)
        return d_fg, d_al, d_rgb, None, None, None, None, None


class RowDot4Fn(torch.autograd.Function):
    """ops.rowdot4 (FeatureNeRFEncoding.decoder, Linear(C -> 4, no bias), nerfsd_pytorch3d.py:49-51,160); backward =
    cd360_rowdot4_bwd_bf16 (dh = d w in one pass, dw = d^T h as a slab-wise column reduction)."""

    @staticmethod
    def forward(ctx, h, w):
        ctx.save_for_backward(h, w)
        return ops.rowdot4(h, w)

    @staticmethod
    def backward(ctx, d_out):
        h, w = ctx.saved_tensors
        return ops.rowdot4_bwd(d_out, h, w, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


class ConvIgemmFn(torch.autograd.Function):
    """ops.conv_igemm (conv3x3 / 1x1 + bias + emb + residual, openaimodel.py:280-376) differentiated with respect to its
    activations.  The data gradient is the SAME implicit-GEMM kernel run on dy with the transposed, tap-flipped weight
    (cd360_conv_igemm_bf16 again, no new kernel); a stride-2 convolution's data gradient is that stride-1 convolution applied to
    dy with zeros inserted between the pixels.  d_res = dy, d_emb = sum of dy over the pixels of each image."""

    @staticmethod
    def forward(ctx, x, w_packed, bias, emb, res, N, H, W, taps, want_stats, stride, alg_channels, w_dgrad):
        _no_wgrad("conv_igemm", w_packed, bias)
        if w_dgrad is None:
            raise NotImplementedError("conv_igemm under autograd needs w_dgrad (the packed data-gradient weight)")
        out = ops.conv_igemm(x, w_packed, bias, N, H, W, taps, emb, res, want_stats, stride, alg_channels)
        y, stats = out if want_stats else (out, None)
        ctx.cfg = (N, H, W, taps, stride, x.shape[-1], tuple(x.shape))
        ctx.w_dgrad = w_dgrad
        ctx.set_materialize_grads(False)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        N, H, W, taps, stride, cin, xshape = ctx.cfg
        if dy is None:
            return (None,) * 13
        dy = dy.contiguous()
        cout = dy.shape[-1]
        dx = d_emb = d_res = None
        if ctx.needs_input_grad[4]:
            d_res = dy
        if ctx.needs_input_grad[3]:
            d_emb = dy.reshape(N, -1, cout).sum(1, dtype=torch.float32).to(dy.dtype)
        if ctx.needs_input_grad[0]:
            wd = ctx.w_dgrad() if callable(ctx.w_dgrad) else ctx.w_dgrad  # [Cin_p16, taps * Cout_p64]
            cin_d = wd.shape[1] // taps
            g = dy.reshape(N, (H // stride) * (W // stride), cout)
            if stride == 2:
                z = torch.zeros(N, H, W, cout, dtype=dy.dtype, device=dy.device)
                z[:, ::2, ::2] = g.reshape(N, H // 2, W // 2, cout)
                g = z.reshape(N, H * W, cout)
            if cin_d != cout:
                g = torch.nn.functional.pad(g, (0, cin_d - cout))
            dx = ops.conv_igemm(g.contiguous(), wd, None, N, H, W, taps)[..., :cin].reshape(xshape)
        return dx, None, None, d_emb, d_res, None, None, None, None, None, None, None, None


def conv_igemm(x, w_packed, bias, N, H, W, taps, emb, res, want_stats, stride, alg_channels, w_dgrad):
    y, stats = ConvIgemmFn.apply(x, w_packed, bias, emb, res, N, H, W, taps, want_stats, stride, alg_channels, w_dgrad)
    return (y, stats) if want_stats else y


class ConcatChannelsFn(torch.autograd.Function):
    """ops.concat_channels (the skip-connection concat, openaimodel.py:1074-1076)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[1]
        return ops.concat_channels(a, b)

    @staticmethod
    def backward(ctx, d):
        return d[:, :ctx.ca], d[:, ctx.ca:]
