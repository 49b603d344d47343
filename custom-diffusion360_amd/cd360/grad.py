"""Autograd recording of the differentiable cd360 operators (BASELINE config 4: the fine-tuning loop differentiates the pose
path; the reference relies on torch autograd through xformers, grid_sample, nn.Linear ... -- diffusion.py:226-241, main.py).

Every class is a torch.autograd.Function whose forward AND backward are HIP kernels behind the C ABI (ops.py); plain library
GEMMs (nn.Linear, torch.mm) stay with torch's own autograd.  ops.* dispatch here when an input requires grad and grad mode is
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
