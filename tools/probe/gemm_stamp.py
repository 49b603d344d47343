"""Where a K-tile of the GEMM core spends its cycles: s_memtime stamps taken by every wave at four points of every K-tile
(A loop top, B k-steps 0..2 issued, D past the rendezvous, E last k-step issued) in the probe build (tools/probe/gemm_stamp.sh 1).
    python tools/probe/gemm_stamp.py [M N K]          (default 3072 1280 1280; env CD360_GEMM_* select the tiling as usual)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))
import torch

from cd360 import _lib

_lib.LIB_PATH = os.path.join(ROOT, "custom-diffusion360_amd", "lib", "libcd360_stamp.so")
from cd360 import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (3072, 1280, 1280)
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(torch.bfloat16)
bias = torch.randn(N, generator=g).to(dev)
res = torch.randn(M, N, generator=g).to(dev).to(torch.bfloat16)
NWG, NW, NT = 4096, 8, 64
buf = torch.zeros(NWG * NW * NT * 8, dtype=torch.int32, device=dev)
for _ in range(5):
    ops.gemm(a, w, bias=bias, res=res, want_stats=True)
torch.cuda.synchronize()
# the stamp buffer's device pointer travels in cd360_tuning.reserved[0..1] (probe build only)
import ctypes

t = _lib.Tuning()
_lib.load().cd360_get_tuning(ctypes.byref(t))
ptr = buf.data_ptr()
t.reserved[0], t.reserved[1] = ctypes.c_int32(ptr & 0xFFFFFFFF).value, ctypes.c_int32(ptr >> 32).value
_lib.load().cd360_set_tuning(ctypes.byref(t))
ops.gemm(a, w, bias=bias, res=res, want_stats=True)
torch.cuda.synchronize()
t.reserved[0] = t.reserved[1] = -1
_lib.load().cd360_set_tuning(ctypes.byref(t))
nk = K // 64
st = buf.cpu().numpy().astype("int64").reshape(NWG, NW, NT, 8) & 0xFFFFFFFF
used = [i for i in range(NWG) if st[i].any()]
print(f"M={M} N={N} K={K}: {len(used)} workgroups stamped, {nk} K-tiles")
nk = min(nk, NT)
st = st[used][:, :, :nk]
A, B, D, E, C1, C2 = (st[..., i] for i in range(6))
mid = slice(4, nk - 4) if nk > 10 else slice(1, nk - 1)
d = lambda x, y: ((x - y) & 0xFFFFFFFF).astype("float64")
print("per K-tile, cycles (s_memtime ticks), tiles %d..%d, mean over all waves | wave 0 of the first workgroup:" % (mid.start, mid.stop - 1))
segs = {"A->B  k-steps 0..2 (reads, MFMAs, spread DMA pieces)": d(B, A), "B->C1 wait for the wave's own fragment reads (lgkmcnt 0)": d(C1, B),
        "C1->C2 wait for the next tile's DMA pieces (counted vmcnt)": d(C2, C1), "C2->D workgroup rendezvous (s_barrier)": d(D, C2),
        "D->E  next tile's first reads, last k-step + DMA pieces": d(E, D), "E->A' loop back": d(A[..., 1:], E[..., :-1])}
for name, v in segs.items():
    vv = v[..., mid] if v.shape[-1] == nk else v[..., slice(mid.start, mid.stop - 1)]
    print(f"  {name:66s} {vv.mean():8.1f} | {vv[0, 0].mean():8.1f}")
tile = d(A[..., 1:], A[..., :-1])[..., slice(mid.start, mid.stop - 1)]
print(f"  whole K-tile (A -> A')                                             {tile.mean():8.1f} | {tile[0, 0].mean():8.1f}")
first = d(A[:, :, 0], A[:, :1, 0].min(axis=1, keepdims=True))
print(f"  loop entry skew between the waves of a workgroup (max - min of A at tile 0): {first.max(axis=1).mean():8.1f}")
total = d(E[:, :, nk - 1], A[:, :, 0])
print(f"  K loop, first A to last E: {total.mean():8.1f} cycles")
Bw = d(B, A)[..., mid]
print("  A->B by wave of the workgroup (mean):", " ".join(f"{Bw[:, i].mean():7.1f}" for i in range(Bw.shape[1])))
Dw = d(D, C2)[..., mid]
print("  C2->D by wave of the workgroup (mean):", " ".join(f"{Dw[:, i].mean():7.1f}" for i in range(Dw.shape[1])))
