"""Parity of the BACKWARD HIP kernels (BASELINE config 4: the fine-tuning loop) against torch autograd of the fp32 CPU oracle.

The reference has no hand-written backward: it trains through torch autograd of its own forward (plus _TruncExp's clamped
derivative, attention.py:192-208).  The oracle is that forward restated in torch, so `torch.autograd.grad` of the oracle IS the
reference gradient.  Tolerance: bf16 kernels within 2e-2 of the oracle gradient relative to its max magnitude (two bf16 roundings
on the way: the forward's outputs and the packed P / dS operands), inputs rounded to bf16 for both sides."""
import pytest
import torch

from oracle import pose_path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


def bf(x):
    return x.to(torch.bfloat16).float()


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,Nq,Nk,kv_grad", [(2, 2, 128, 128, True), (1, 3, 200, 77, True), (1, 1, 1024, 1024, True), (2, 2, 96, 40, False),
                                               (1, 2, 333, 333, True), (1, 2, 50, 20, True), (2, 5, 2048, 77, False), (1, 2, 100, 97, True)])
def test_attention_backward(B, H, Nq, Nk, kv_grad):
    """dq, dk, dv of softmax(q k^T / 8) v through ops.attention under autograd: tiled and small-Nk forward kernels (lse from both),
    ragged tiles on both sides, k / v as NaN-padded slices of a merged projection, and the dq-only launch (text context: k, v
    without grad)."""
    from cd360 import ops
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk)
    q = bf(torch.randn(B, Nq, H * 64, generator=g))
    k = bf(torch.randn(B, Nk, H * 64, generator=g))
    v = bf(torch.randn(B, Nk, H * 64, generator=g))
    do = bf(torch.randn(B, Nq, H * 64, generator=g))

    def split(t):
        return t.reshape(B, t.shape[1], H, 64).permute(0, 2, 1, 3).reshape(B * H, t.shape[1], 64)

    qo, ko, vo = (t.clone().requires_grad_(True) for t in (q, k, v))
    want = O.attention_core(split(qo), split(ko), split(vo)).reshape(B, H, Nq, 64).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    gq, gk, gv = torch.autograd.grad(want, (qo, ko, vo), do)

    nkp = (Nk + 7) // 8 * 8
    kv = torch.full((B, nkp, 2 * H * 64 + 64), float("nan"))
    kv[:, :Nk, :H * 64] = k
    kv[:, :Nk, H * 64 + 64:] = v
    kvd = kv.to(DEV, torch.bfloat16)
    qd = q.to(DEV, torch.bfloat16).requires_grad_(True)
    kd = kvd[..., :H * 64].detach().requires_grad_(kv_grad)
    vd = kvd[..., H * 64 + 64:].detach().requires_grad_(kv_grad)
    out = ops.attention(qd, kd, vd, H, nk=Nk)
    assert rel(out, want) < 1e-2
    out.backward(do.to(DEV, torch.bfloat16))
    assert rel(qd.grad, gq) < 2e-2
    if kv_grad:
        assert rel(kd.grad[:, :Nk], gk) < 2e-2 and rel(vd.grad[:, :Nk], gv) < 2e-2
        assert (kd.grad[:, Nk:] == 0).all() and (vd.grad[:, Nk:] == 0).all()
    else:
        assert kd.grad is None and vd.grad is None
