import os, sys
sys.path.insert(0, "/root/repo"); os.chdir("/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-profile", "--no-graph", "--steps", "2", "--warmup", "2"]
import bench, torch
from torch.profiler import profile, ProfilerActivity
orig_step = bench.Sampler.step
state = {"n": 0}
def step(self, x, i):
    state["n"] += 1
    if state["n"] == 6:  # a steady step in the timed region
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            out = orig_step(self, x, i)
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        from collections import Counter
        c = Counter(e.name[:70] for e in evs)
        print("TOTAL GPU KERNELS", len(evs))
        for k, v in c.most_common(40):
            print(v, k)
        # attribute tiny copy kernels to python callers
        ka = prof.key_averages(group_by_stack_n=6)
        rows = [r for r in ka if ("copy" in r.key or "to" == r.key or "aten::_to_copy" in r.key or "aten::mul" == r.key) and r.count >= 10]
        for r in sorted(rows, key=lambda r: -r.count)[:12]:
            print("OP", r.key, r.count, [s for s in r.stack if "repo" in s][:4])
        return out
    return orig_step(self, x, i)
bench.Sampler.step = step
bench.main()
