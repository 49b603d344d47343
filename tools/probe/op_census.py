"""Which source lines of the package issue the torch ops of one fine-tune step (BASELINE config 4)?  A TorchDispatchMode counts every
aten op by the innermost repo frame on the Python stack (autograd-engine ops of stock torch nodes have none: bucket '<autograd>').
    python tools/probe/op_census.py [--top 60]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

PKG = os.path.join(ROOT, "custom-diffusion360_amd")
VIEW_OPS = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten._reshape_alias", "aten.expand", "aten.slice", "aten.select", "aten.t.", "aten.transpose",
            "aten.permute", "aten.as_strided", "aten.detach", "aten.alias", "aten.unsqueeze", "aten.squeeze", "aten.empty", "aten.new_empty", "aten.split",
            "aten.chunk", "aten.unbind", "aten.lift_fresh", "aten.unfold", "aten.narrow", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten._local_scalar_dense")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_line = collections.Counter()
        self.by_op = collections.Counter()
        self.by_shape = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        where = "<autograd>"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if fr.filename.startswith(PKG):
                where = f"{os.path.relpath(fr.filename, PKG)}:{fr.lineno}"
                break
        name = str(func)
        if not any(v in name for v in VIEW_OPS):  # count what launches a kernel: views / allocations / metadata ops do not
            self.by_line[where] += 1
            self.by_op[(where, name)] += 1
            if where == "<autograd>":  # no repo frame on the engine's stack: the operand shapes say which forward op this is the backward of
                shapes = tuple((tuple(a.shape), str(a.dtype).replace("torch.", "")) for a in args if isinstance(a, torch.Tensor))[:2]
                self.by_shape[(name, shapes)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    import bench_train
    from cd360 import finetune
    census = Census()
    real = finetune.train_step
    calls = [0]

    def counted(*args, **kw):
        calls[0] += 1
        if calls[0] == 2:  # the second step: caches are warm
            with census:
                return real(*args, **kw)
        return real(*args, **kw)

    finetune.train_step = counted
    bench_train.measure(steps=1, warmup=1, profile=False)
    print(f"ops in one step: {sum(census.by_line.values())}")
    for where, c in census.by_line.most_common(a.top):
        ops = sorted(((n, op) for (w, op), n in census.by_op.items() if w == where), reverse=True)[:4]
        print(f"{c:6d}  {where:55s} " + "  ".join(f"{op.replace('aten.', '')}x{n}" for n, op in ops))
    print("\n<autograd> ops by operand shapes:")
    for (name, shapes), c in census.by_shape.most_common(a.top):
        print(f"{c:6d}  {name.replace('aten.', ''):28s} {shapes}")


if __name__ == "__main__":
    main()
