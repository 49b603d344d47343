"""Does an RCCL all-reduce capture into a hipGraph on this stack?  One process, world size 1 (the only configuration a 1-GPU box offers):
the collective still goes through ncclAllReduce on the capture stream.  python tools/probe/rccl_graph_probe.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)  # communicator set-up outside the capture
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        y = x * 2
        dist.all_reduce(y)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):  # the process-group watchdog thread polls events while we capture
    y = x * 2
    dist.all_reduce(y)
    z = y + 1
for i in range(3):
    x.fill_(float(i))
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, float(z[0]), "expected", 2.0 * i + 1)
dist.destroy_process_group()
