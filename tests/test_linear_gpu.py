"""ONE Linear for every route (`-m gpu`): cd360.ops.linear / grad.LinearFn (cd360_gemm_bf16 forward and data gradient,
cd360_gemm_tn_bf16 weight gradient) against torch autograd of F.linear in fp32 on the same bf16-rounded tensors; the weight-gradient
kernel on its own (ragged M, strided operands, narrow outputs); and the routes: a block whose `forward` was rebound the way
sample.py:247-262 does it, a hooked block (diffusion.py:151-163) and a fine-tuning step of the reduced UNet must not reach a library
GEMM (F.linear / torch.mm / addmm / matmul with more than 8 rows) on bf16 GPU tensors."""
import contextlib
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

import weights as W
from cd360.cameras import unpack_cameras

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


@pytest.mark.parametrize("M,N,K,lda,ldb", [(4096, 640, 1280, None, None), (1000, 128, 64, None, None), (24576, 1280, 112, None, None),
                                           (5000, 8, 640, None, None), (2048, 640, 640, 1920, 1280), (64, 320, 2560, None, None),
                                           (100000, 640, 640, None, None)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn_weight_gradient_kernel(M, N, K, lda, ldb, out_dtype):
    """a^T b with both operands row-major over the contraction index: ragged M (not a multiple of the 64-row K-tile), K = 112 (rows of
    224 bytes: the FeatureNeRF F matrix), N = 8 (the view-logit column), column slices of wider matrices (lda / ldb), M = 10^5 (many
    slabs); fp32 and bf16 outputs; repeat launches bit-identical (fixed summation order)."""
    from cd360 import ops
    a_full = rnd(M, lda or N, seed=M + N).to(BF)
    b_full = rnd(M, ldb or K, seed=M + K + 1).to(BF)
    a, b = a_full[:, :N], b_full[:, :K]
    assert ops.gemm_tn_ok(a, b)
    got = ops.gemm_tn(a, b, out_dtype=out_dtype)
    want = a.float().t() @ b.float()
    assert got.dtype == out_dtype and rel(got, want) < (6e-3 if out_dtype == torch.bfloat16 else 2e-4)
    assert torch.equal(got, ops.gemm_tn(a, b, out_dtype=out_dtype))


@pytest.mark.parametrize("shape,N,K,bias,res", [((3, 1024, ), 1280, 1280, True, False), ((2, 100), 640, 128, False, True), ((4096, ), 320, 640, True, True),
                                                ((3, 77), 2560, 2048, False, False), ((4, ), 1280, 320, True, False)])
def test_linear_forward_and_gradients_match_torch(shape, N, K, bias, res):
    """ops.linear under autograd (grad.LinearFn) against torch's F.linear in fp32 on the same bf16 values: output, dX (the same GEMM on
    W^T), dW (cd360_gemm_tn_bf16), db, d_res; and the no-grad call equals the recorded one bit for bit."""
    from cd360 import ops
    x = rnd(*shape, K, seed=K).to(BF).requires_grad_(True)
    w = rnd(N, K, seed=N + 1, scale=K ** -0.5).to(BF).requires_grad_(True)
    b = rnd(N, seed=3).to(BF).requires_grad_(True) if bias else None
    r = rnd(*shape, N, seed=4).to(BF).requires_grad_(True) if res else None
    cot = rnd(*shape, N, seed=5).to(BF)
    y = ops.linear(x, w, b, res=r)
    y.backward(cot)
    with torch.no_grad():
        assert torch.equal(y, ops.linear(x, w, b, res=r))
    xf, wf = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True)
    bf_ = b.detach().float().requires_grad_(True) if bias else None
    rf = r.detach().float().requires_grad_(True) if res else None
    yf = F.linear(xf, wf, bf_)
    if res:
        yf = yf + rf
    yf.backward(cot.float())
    assert rel(y, yf) < 6e-3
    assert rel(x.grad, xf.grad) < 8e-3 and rel(w.grad, wf.grad) < 8e-3
    if bias:
        assert rel(b.grad, bf_.grad) < 8e-3
    if res:
        assert rel(r.grad, rf.grad) < 1e-6


def test_linear_weight_slices_share_one_parameter_gradient():
    """pose_emb_layers(cat[x, xref]) as two GEMMs over the two column halves of ONE [C, 2C] parameter read in place (attention.py:634):
    forward equals the concatenated Linear, and the two weight gradients land in the halves of the parameter's gradient."""
    from cd360 import ops
    C, rows = 640, 2048
    x, xr = rnd(rows, C, seed=1).to(BF), rnd(rows, C, seed=2).to(BF)
    w = rnd(C, 2 * C, seed=3, scale=(2 * C) ** -0.5).to(BF).requires_grad_(True)
    cot = rnd(rows, C, seed=4).to(BF)
    y = ops.linear(x, w[:, :C], None, res=ops.linear(xr, w[:, C:]))
    y.backward(cot)
    wf = w.detach().float().requires_grad_(True)
    yf = F.linear(torch.cat([x, xr], -1).float(), wf)
    yf.backward(cot.float())
    assert rel(y, yf) < 8e-3 and rel(w.grad, wf.grad) < 8e-3


# ------------------------------------------------------------------------------------------------ routes
@contextlib.contextmanager
def no_library_gemm(max_rows: int = 8):
    """Inside: F.linear / torch.mm / torch.addmm / torch.matmul / Tensor.addmm_ on bf16 GPU matrices with more than `max_rows` rows on
    both sides raise -- the tensor-level statement of "no Cijk_* row except M <= 8 GEMVs" in a kernel trace."""
    saved = (F.linear, torch.mm, torch.addmm, torch.matmul, torch.Tensor.addmm_, torch.nn.functional.linear)

    def guard(name, fn, idx):
        def wrapped(*args, **kw):
            ts = [args[i] for i in idx if i < len(args) and isinstance(args[i], torch.Tensor)]
            if len(ts) == len(idx) and all(t.is_cuda and t.dtype == BF and t.dim() >= 2 for t in ts):
                if all(t.numel() // t.shape[-1] > max_rows for t in ts):
                    raise AssertionError(f"library GEMM {name} reached with {[tuple(t.shape) for t in ts]}")
            return fn(*args, **kw)
        return wrapped

    F.linear = torch.nn.functional.linear = guard("F.linear", saved[0], (0, 1))
    torch.mm = guard("torch.mm", saved[1], (0, 1))
    torch.addmm = guard("torch.addmm", saved[2], (1, 2))
    torch.matmul = guard("torch.matmul", saved[3], (0, 1))
    torch.Tensor.addmm_ = guard("addmm_", saved[4], (1, 2))
    try:
        yield
    finally:
        F.linear, torch.mm, torch.addmm, torch.matmul, torch.Tensor.addmm_, torch.nn.functional.linear = saved


def make_block(C=128, heads=2, cd=64, image_cross=False):
    from sgm.modules.attention import BasicTransformerBlock
    blk = BasicTransformerBlock(C, heads, 64, context_dim=cd, checkpoint=False, attn_mode="softmax-xformers", image_cross=image_cross, far=2,
                                num_samples=6, rgb_predict=True, mode="feature-nerf").eval()
    W.load_into(blk, seed=9)
    return blk.to(DEV, BF)


def _patched_forward(self, x, context=None, context_ref=None, pose=None, mask_ref=None, prev_weights=None, **kw):
    """The call sequence of sample.py's rebound BasicTransformerBlock.forward (sample.py:33-80) for a block without pose conditioning:
    the submodules are called one by one through the module protocol."""
    x = self.attn1(self.norm1(x), context=None) + x
    x = self.attn2(self.norm2(x), context=context) + x
    x = self.ff(self.norm3(x)) + x
    return x, None, None, None, None


@torch.no_grad()
def test_rebound_forward_and_hooked_blocks_stay_on_the_hand_written_gemm():
    """sample.py:247-262 rebinds `forward` on every block; diffusion.py:151-163 hooks blocks for the references harvest.  Both routes
    must (a) give the fused route's result within rounding and (b) never reach a library GEMM."""
    blk = make_block()
    x, ctx = rnd(3, 256, 128, seed=1).to(BF), rnd(3, 77, 64, seed=2).to(BF)
    fused = blk(x, context=ctx)[0]
    blk.forward = types.MethodType(_patched_forward, blk)
    with no_library_gemm():
        patched = blk(x, context=ctx)[0]
    del blk.forward
    seen = []
    h = blk.register_forward_hook(lambda m, i, o: seen.append(o[0].shape))
    from sgm.modules.attention import SpatialTransformer
    st = SpatialTransformer(128, 2, 64, depth=2, context_dim=64, use_linear=True, attn_type="softmax-xformers", use_checkpoint=False,
                            image_cross=False).eval()
    W.load_into(st, seed=5)
    st = st.to(DEV, BF)
    img = rnd(3, 128, 16, 16, seed=6).to(BF).contiguous(memory_format=torch.channels_last)
    want = st(img, None, context=ctx)[0]
    hk = st.transformer_blocks[1].attn2.to_q.register_forward_hook(lambda m, i, o: seen.append(o.shape))  # a hook deep inside the tree
    with no_library_gemm():
        got = st(img, None, context=ctx)[0]
    hk.remove()
    h.remove()
    assert len(seen) == 1 and seen[0] == (3, 256, 128), "the submodule hook must fire (the fused route would have skipped it)"
    assert rel(patched, fused) < 2e-2 and rel(got, want) < 2e-2


@torch.no_grad()
def test_block_level_hooks_keep_the_fused_internals_submodule_hooks_do_not():
    """A forward hook on a block (the references harvest, diffusion.py:151-163) sees the block's inputs and outputs around `__call__`: the
    block keeps its fused internals and returns the un-hooked result bit for bit.  A hook on a submodule needs that submodule called
    through the module protocol: the strict route (checked against the fused result within rounding; the hook must fire)."""
    blk = make_block()
    x, ctx = rnd(3, 256, 128, seed=1).to(BF), rnd(3, 77, 64, seed=2).to(BF)
    fused = blk(x, context=ctx)[0]
    seen = []
    h = blk.register_forward_hook(lambda m, i, o: seen.append(("block", i[0].data_ptr() == x.data_ptr(), o[0].shape)))
    assert torch.equal(blk(x, context=ctx)[0], fused) and seen == [("block", True, (3, 256, 128))]
    h.remove()
    h = blk.norm2.register_forward_hook(lambda m, i, o: seen.append(("norm2", o.shape)))
    with no_library_gemm():
        strict = blk(x, context=ctx)[0]
    h.remove()
    assert seen[-1] == ("norm2", (3, 256, 128)) and not torch.equal(strict, fused) and rel(strict, fused) < 2e-2


def test_finetune_step_reaches_no_library_gemm():
    """One optimisation step of the reduced UNet (trainkeys = pose; forward of both streams, four-term loss, backward, AdamW) with every
    library GEMM entry guarded: Linear forward / dgrad / wgrad, the FeatureNeRF table GEMMs and their parameter gradients all run on
    cd360_gemm_bf16 / cd360_gemm_tn_bf16.  Gradients stay pinned by tests/test_backward_gpu.py against the reference's autograd."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cd360 import synth
    from make_golden_params import UNET_TINY
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    cfg = dict(UNET_TINY, context_dim=64, adm_in_channels=64)  # every Linear inside the GEMM's envelope (K % 64 == 0), as in the SDXL config
    b, n = 2, 2
    net = UNetModel(**cfg).train()
    W.load_into(net, seed=5)
    net = net.to(DEV, BF)
    g = {"x": rnd(b, 4, 16, 16, seed=1), "t": torch.tensor([0.7, 0.2], device=DEV), "ctx": rnd(b + b * n, 77, 64, seed=2), "y": rnd(b + b * n, 64, seed=3),
         "input_ref": rnd(b, n, 4, 16, 16, seed=4), "sigmas_ref": torch.tensor([0.05, 0.05], device=DEV)}
    pose = synth.pose_batch(b, n, seed=11)
    for k, p in net.named_parameters():
        p.requires_grad = "pose" in k
    with no_library_gemm():
        out, fgs, alphas, rgbs = net(g["x"], timesteps=g["t"], context=g["ctx"], y=g["y"], pose=pose, input_ref=g["input_ref"],
                                     sigmas_ref=g["sigmas_ref"], mask_ref=None)
        loss = out.float().square().mean() + sum(f.float().mean() for f in fgs) + sum(r.float().square().mean() for r in rgbs)
        loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters() if p.requires_grad}
    assert grads and all(v is not None and torch.isfinite(v.float()).all() for k, v in grads.items() if not k.endswith("nviews.bias"))
    assert any(v.float().abs().max() > 0 for v in grads.values())


def test_fused_adamw_matches_torch_adamw_on_fp32_masters():
    """cd360_adamw_bf16 (MasterAdamW's fused path) against torch.optim.AdamW on fp32 masters, step for step: ragged tensor sizes (tails
    of fewer than 8 values, an unaligned view), two groups with their own lr / weight decay, a tensor without a gradient in one step;
    masters and moments agree to fp32 rounding, the bf16 parameters are the rounded masters, more than 64 tensors take two launches."""
    from cd360 import finetune
    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = [(1280, 2560), (640,), (3, 5), (1,), (4, 1280), (77,), (640, 1280)] + [(9 + i,) for i in range(70)]
    base = [torch.randn(*s, generator=g, device="cuda").mul_(0.1).to(torch.bfloat16) for s in shapes]
    base[5] = torch.randn(78, generator=g, device="cuda").to(torch.bfloat16)[1:]  # a 2-byte-aligned view: the kernel's scalar path
    mk = lambda: [torch.nn.Parameter(b.clone() if b.is_contiguous() and b.storage_offset() == 0 else b) for b in base]
    pa, pb = mk(), mk()
    pb[5] = torch.nn.Parameter(torch.cat([base[5][:1], base[5]])[1:])  # same values, own storage, same misalignment
    groups = lambda ps: [{"params": ps[:4], "lr": 1e-3, "weight_decay": 0.1}, {"params": ps[4:], "lr": 3e-4}]
    fused = finetune.MasterAdamW(groups(pa), lr=1e-4, betas=(0.9, 0.95), eps=1e-6)
    plain = finetune.MasterAdamW(groups(pb), lr=1e-4, betas=(0.9, 0.95), eps=1e-6, fused=False)
    assert fused.fused and not plain.fused and fused.capturable and len(fused._fused_plan(list(range(len(pa)))).chunks) == 2
    for it in range(6):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if it == 3 and i == 2:
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g, device="cuda").to(torch.bfloat16)
            a.grad, b.grad = gr.clone(), gr.clone()
        fused.step()
        plain.step()
    assert float(fused.steps) == 6
    rel = lambda x, y: float((x.float() - y.float()).abs().max() / y.float().abs().max().clamp_min(1e-12))
    for i, (ma, mb, a, b) in enumerate(zip(fused.master, plain.master, pa, pb)):
        if i == 2:
            continue  # skipped once: torch's per-tensor step count differs from the one-per-optimiser count by design
        assert rel(ma, mb) < 2e-6, (i, rel(ma, mb))
        assert torch.equal(a.detach(), ma.to(torch.bfloat16)) and rel(a, b) < 1e-2
    st = plain.opt.state[plain.master[0]]
    n0 = pa[0].numel()
    assert rel(fused.exp_avg[:n0].view_as(st["exp_avg"]), st["exp_avg"]) < 2e-6 and rel(fused.exp_avg_sq[:n0].view_as(st["exp_avg_sq"]), st["exp_avg_sq"]) < 2e-6


def test_fused_adamw_updates_are_seen_by_every_weight_cache():
    """The fused optimiser (and a hipGraph replay of it) writes the bf16 parameters through raw pointers; MasterAdamW.bump_versions makes
    the update visible to the caches keyed on `_version` (LayerNorm-folded packs, W^T copies, fp32 biases, the FeatureNeRF's fused weights,
    the pose-projection split).  The reference's periodic image logging is exactly this sequence: train, sample under no_grad, train,
    sample.  One pose block, trainable set `poseattn` (pose parameters + attn1 / attn2, diffusion.py:121-138): after every burst of fused
    optimiser steps the no_grad (fused-route) forward must equal, bit for bit, the forward of a freshly built block that holds the
    current parameter values -- and must differ from the forward before the burst."""
    import weights as W
    from cd360 import finetune, synth
    from sgm.modules.attention import BasicTransformerBlock
    dev, bf = "cuda", torch.bfloat16

    def build():
        return BasicTransformerBlock(128, 2, 64, context_dim=64, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2, num_samples=6,
                                     rgb_predict=True, mode="feature-nerf", stratified=True).eval()

    blk = build()
    W.load_into(blk, seed=51)
    blk = blk.to(dev, bf)
    train = lambda k: "pose" in k or k.startswith(("attn1.", "attn2."))
    for k, p in blk.named_parameters():
        p.requires_grad = train(k)
    b, n, hw = 2, 3, 64
    x, ctx = W.tensor("x", (b, hw, 128), seed=51).to(dev, bf), W.tensor("ctx", (b, 77, 64), seed=51).to(dev, bf)
    cref = W.tensor("cref", (b * n, hw, 128), seed=51).to(dev, bf)
    pose = synth.pose_batch(b, n, seed=4)
    opt = finetune.MasterAdamW([p for p in blk.parameters() if p.requires_grad], lr=2e-2)
    assert opt.fused

    def sample(m):
        with torch.no_grad():
            out = m(x, context=ctx, context_ref=cref, pose=pose)
        return out[0].clone(), out[1].clone(), out[4].clone()

    prev = sample(blk)
    for burst in range(2):
        for _ in range(2):
            opt.zero_grad()
            out, fg, _, _, rgb = blk(x, context=ctx, context_ref=cref, pose=pose)
            (out.float().pow(2).mean() + fg.float().mean() + rgb.float().mean()).backward()
            opt.step()
        now = sample(blk)
        fresh = build()
        fresh.load_state_dict({k: v.detach().float().cpu() for k, v in blk.state_dict().items()}, strict=False)
        want = sample(fresh.to(dev, bf))
        for a, w_, p_ in zip(now, want, prev):
            assert torch.equal(a, w_), f"burst {burst}: a cache survived the fused optimiser step"
        assert not torch.equal(now[0], prev[0]) and not torch.equal(now[1], prev[1])
        prev = now


@pytest.mark.parametrize("trainable", [False, True])
def test_cpp_autograd_node_of_linear_equals_the_python_node(trainable):
    """cd360/_host.py (csrc_host/cd360_host.cpp): the fine-tuning Linear's autograd node as C++ host glue issues exactly the launches
    grad.LinearFn issues from Python -- output, dX, dW, db and the residual's gradient bit-identical, frozen and trainable weights, 3-D
    inputs, a strided weight view (pose_emb_layers' column halves)."""
    from cd360 import _host, ops, routes
    assert _host.get() is not None, "the host glue must be built on a GPU box (__graft_entry__.build)"
    g = torch.Generator(device=DEV).manual_seed(11)
    x0 = torch.randn(3, 80, 256, generator=g, device=DEV).to(torch.bfloat16)
    wfull = (torch.randn(128, 512, generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    b0 = torch.randn(128, generator=g, device=DEV).to(torch.bfloat16)
    r0 = torch.randn(3, 80, 128, generator=g, device=DEV).to(torch.bfloat16)
    cot = torch.randn(3, 80, 128, generator=g, device=DEV).to(torch.bfloat16)

    def run(glue):
        x = x0.clone().requires_grad_(True)
        wf = wfull.clone().requires_grad_(trainable)
        w = wf[:, 256:]  # a column-half view: row stride 512
        b = b0.clone().requires_grad_(trainable)
        r = r0.clone().requires_grad_(True)
        with routes.override(no_host_glue=not glue):
            y = ops.linear(x, w, b, r)
            y.backward(cot)
        return [y.detach(), x.grad, r.grad] + ([wf.grad, b.grad] if trainable else [])

    a, p = run(True), run(False)
    assert len(a) == len(p)
    for u, v in zip(a, p):
        assert u is not None and v is not None and torch.equal(u, v)
