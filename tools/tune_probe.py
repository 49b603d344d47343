"""Does PyTorch TunableOp (hipBLASLt + rocBLAS solution search) find faster kernels than the default heuristic for the Linear shapes
of the steady sampling step?  Prints default vs tuned time per shape; writes the tuned table to gpurun_out/tunableop.csv.
Measured on MI355X (round 1): no -- tuned == default within noise on every shape (L2 out 22.8 us / 441 TF/s both ways), so bench.py
leaves TunableOp off."""
import os
import sys
import time

import torch
import torch.nn.functional as F

dev, BF = "cuda", torch.bfloat16


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [("L2 out", 3072, 1280, 1280, True), ("L2 qkv", 3072, 1280, 3840, False), ("L2 ff1", 3072, 1280, 10240, True),
          ("L2 ff2", 3072, 5120, 1280, True), ("L1 out", 12288, 640, 640, True), ("L1 qkv", 12288, 640, 1920, False),
          ("L1 ff1", 12288, 640, 5120, True), ("L1 ff2", 12288, 2560, 640, True)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
import torch.cuda.tunable as tun
tun.enable(False)
data = []
for tag, M, K, N, bias in shapes:
    x = torch.randn(3, M // 3, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
    b = torch.randn(N, device=dev).to(BF) if bias else None
    data.append((tag, M, K, N, x, w, b, timeit(lambda: F.linear(x, w, b))))
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(8)
tun.set_max_tuning_iterations(20)
tun.set_filename(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tunableop.csv"))
for tag, M, K, N, x, w, b, us0 in data:
    t0 = time.time()
    F.linear(x, w, b)  # tunes this shape
    torch.cuda.synchronize()
    tt = time.time() - t0
    us1 = timeit(lambda: F.linear(x, w, b))
    fl = 2.0 * M * K * N
    print(f"{tag}: M{M} K{K} N{N} bias={b is not None}: default {us0:7.1f} us {fl / us0 / 1e6:7.1f} TF/s | tuned {us1:7.1f} us {fl / us1 / 1e6:7.1f} TF/s  (tuning took {tt:.1f} s)", flush=True)
if hasattr(tun, "write_file"):
    tun.write_file()
