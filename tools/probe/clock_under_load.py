#!/usr/bin/env python
"""Average shader clock while (a) a chain of 4096^3 GEMMs, (b) the FF1 + GEGLU GEMM of the 1280 level, (c) captured steady denoise steps run:
s_memtime / s_memrealtime stamps on the stream around each stretch.  The MFMA peak the rooflines are priced against assumes 2.4 GHz."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd")]
import torch  # noqa: E402

so = "/tmp/clock_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools/probe/clock_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.cd360_probe_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
from cd360 import ops  # noqa: E402


def stamp():
    t = torch.zeros(2, dtype=torch.int64, device=dev)
    lib.cd360_probe_stamp(t.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return t


def measure(name, fn, reps, flops=None):
    fn(); torch.cuda.synchronize()
    a = stamp()
    for _ in range(reps):
        fn()
    b = stamp()
    torch.cuda.synchronize()
    d = (b - a).tolist()
    secs = d[1] / 100e6
    msg = f"{name}: {d[0] / secs / 1e6:7.0f} MHz shader clock over {secs * 1e3:8.2f} ms"
    if flops:
        tf = flops * reps / secs / 1e12
        peak_at_clock = 2.5e3 * (d[0] / secs) / 2.4e9
        msg += f"; {tf:7.1f} TF/s = {tf / 2.5e3:.3f} of the 2.4 GHz peak, {tf / peak_at_clock:.3f} of the peak at the measured clock"
    print(msg, flush=True)


g = torch.Generator(device=dev).manual_seed(1)
def graphed(fn, n):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    return gr.replay

for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192)):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    measure(f"cd360 gemm {M}^3 x20", graphed(lambda: ops.gemm(a, w), 20), 10, 20 * 2.0 * M * N * K)
    measure(f"hipBLASLt  {M}^3 x20", graphed(lambda: torch.nn.functional.linear(a, w), 20), 10, 20 * 2.0 * M * N * K)
M, N, K = 3072, 10240, 1280
a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
measure("FF1 + GEGLU 3072 x 10240 x 1280 x50", graphed(lambda: ops.gemm(a, w, geglu=True), 50), 10, 50 * 2.0 * M * N * K)
measure("FF1 plain   3072 x 10240 x 1280 x50", graphed(lambda: ops.gemm(a, w), 50), 10, 50 * 2.0 * M * N * K)
measure("hipBLASLt   3072 x 10240 x 1280 x50", graphed(lambda: torch.nn.functional.linear(a, w), 50), 10, 50 * 2.0 * M * N * K)
x = torch.randn(1 << 28, device=dev)
measure("HBM copy 1 GiB x20 (no MFMA)", graphed(lambda: x.clone(), 20), 5)

import bench  # noqa: E402
from cd360 import synth  # noqa: E402
from cd360.job import Sampler  # noqa: E402
net = bench.build_model(128, 50, 50, dev)
pose = [synth.pose_batch(1, 50, seed=100, n_train=50)[0]] * 3
ctx = torch.randn(3, 77, 2048, generator=g, device=dev).to(torch.bfloat16)
y = torch.randn(3, 2816, generator=g, device=dev).to(torch.bfloat16)
x0 = torch.randn(1, 4, 128, 128, generator=g, device=dev)
smp = Sampler(net, pose, ctx, y, 50, use_graph=True)
smp.prepare(x0)
xx = smp.step(x0, 0, alias=True)
state = {"i": 1}
def one():
    smp.step(smp.gx, state["i"], alias=True)
    state["i"] = state["i"] % 48 + 1
measure("steady denoise step x40", one, 40, 2.03e13)
