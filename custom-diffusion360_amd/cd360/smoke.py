"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (used by __graft_entry__.smoke())."""
from __future__ import annotations

import torch


def run(verbose: bool = True) -> dict:
    import weights as W  # tests/golden/weights.py (deterministic synthetic weights)
    from oracle import pose_path as O
    from sgm.modules.attention import BasicTransformerBlock

    from . import synth
    from .cameras import pack_cameras

    torch.manual_seed(0)
    C, heads, r, n, S, b, T, cd = 128, 2, 8, 3, 6, 2, 77, 64
    blk = BasicTransformerBlock(C, heads, 64, context_dim=cd, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2,
                                num_samples=S, rgb_predict=True, mode="feature-nerf", stratified=True).eval()
    sd = W.load_into(blk, seed=9)
    pose = synth.pose_batch(b, n, seed=11)
    x = W.tensor("x", (b, r * r, C), seed=9)
    ctx = W.tensor("ctx", (b, T, cd), seed=9)
    cref = W.tensor("cref", (b * n, r * r, C), seed=9)
    ref = O.transformer_block(sd, x, ctx, heads, context_ref=cref, cams=pack_cameras(pose), num_samples=S, far=2.0)
    dev = torch.device("cuda:0")
    blk = blk.to(dev, torch.bfloat16)
    with torch.no_grad():
        out = blk(x.to(dev, torch.bfloat16), context=ctx.to(dev, torch.bfloat16), context_ref=cref.to(dev, torch.bfloat16), pose=pose)
    torch.cuda.synchronize()
    res = {}
    for name, got, want in (("x", out[0], ref[0]), ("fg", out[1], ref[1]), ("alphas", out[3], ref[2]), ("rgb", out[4], ref[3])):
        got = got.float().cpu()
        err = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        res[name] = err
        if verbose:
            print(f"smoke {name}: max rel err {err:.3e}")
        assert torch.isfinite(got).all() and err < 3e-2, (name, err)
    return res
