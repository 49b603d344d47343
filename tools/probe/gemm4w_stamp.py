#!/usr/bin/env python
"""Phases of one launch of the four-wave 256 x 256 arrangement (gemm_cfg = 9) on the 100 MHz counter: entry -> loop -> behind the loop ->
staging image written -> stores issued, per workgroup round (probe build: tools/probe/gemm_stamp.sh)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))
import torch
from cd360 import _lib
_lib.LIB_PATH = os.path.join(ROOT, "custom-diffusion360_amd", "lib", "libcd360_stamp.so")
from cd360 import ops
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
CFG = int(os.environ.get("STAMP_CFG", "9"))
NWV = {9: 4, 7: 16, 3: 8}.get(CFG, 12)  # waves per workgroup of the tiling (-1: the dispatch's own choice; 12 = 8 + 4 movers on the narrow shapes)
SHAPES = ((3072, 10240, 64, False), (3072, 10240, 1280, False), (3072, 10240, 1280, True), (3072, 3840, 1280, False))
if CFG < 0:
    SHAPES = ((3072, 1280, 64, False), (3072, 1280, 1280, False), (3072, 1280, 5120, False), (3072, 3840, 1280, False))
for (M, N, K, geglu) in SHAPES:
    a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    _lib.set_tuning(gemm_cfg=CFG)
    kw = dict(bias=bias, geglu=geglu)
    for _ in range(3):
        ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    nwg = ((M + 63) // 64) * ((N + 63) // 64)  # (an upper bound on the workgroups of any tiling)
    buf = torch.zeros(nwg * NWV * 8, dtype=torch.int32, device=dev)
    t = _lib.Tuning()
    _lib.load().cd360_get_tuning(ctypes.byref(t))
    ptr = buf.data_ptr()
    t.reserved[0], t.reserved[1] = ctypes.c_int32(ptr & 0xFFFFFFFF).value, ctypes.c_int32(ptr >> 32).value
    _lib.load().cd360_set_tuning(ctypes.byref(t))
    ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    t.reserved[0] = t.reserved[1] = -1
    _lib.load().cd360_set_tuning(ctypes.byref(t))
    st = (buf.cpu().numpy().astype("int64") & 0xFFFFFFFF).reshape(nwg, NWV, 8)
    st = st[st[:, 0, 0] != 0]
    nwg = st.shape[0]
    t0 = st[:, :, 0].min()
    rel = (st - t0) / 100.0  # us
    first = rel[:, 0, 0] < (rel[:, 0, 0].min() + 3.0)  # workgroups of the first round
    names = ["entry", "loop entered", "loop left", "past the barrier", "image written", "stores issued"]
    for rnd, sel in (("round 1", first), ("round 2", ~first)):
        if sel.sum() == 0:
            continue
        r = rel[sel]
        print(f"{M}x{N}x{K} geglu={geglu} {rnd} ({int(sel.sum())} workgroups): " + " | ".join(f"{n} {r[:, :, i].mean():6.2f}" for i, n in enumerate(names))
              + f" | last stamp {r[:, :, 5].max():6.2f}")
    _lib.set_tuning(gemm_cfg=-1)
