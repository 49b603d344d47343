"""Find reads of uninitialised memory behind the C ABI: every torch.empty / empty_like / new_empty in the process returns NaN-filled
(floating point) or 0x7f7f...-filled (integer) memory, every public function of cd360.ops is wrapped to check -- synchronously -- whether
its outputs contain NaN although none of its tensor arguments did, and one fine-tuning step (eval-mode raymarchers) plus one sampling
forward are run.  A kernel that writes all of its outputs and reads only what was written is unaffected; one that reads a workspace or an
output it has not (fully) written shows up by name."""
import functools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
DEV = "cuda"

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64, torch.uint8, torch.int16):
            t.fill_(0x7f)
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


def tensors_of(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (list, tuple)):
        for v in x:
            yield from tensors_of(v)
    elif isinstance(x, dict):
        for v in x.values():
            yield from tensors_of(v)


def has_nan(x):
    return any(t.is_cuda and t.dtype.is_floating_point and t.numel() and bool(torch.isnan(t).any()) for t in tensors_of(x))


FOUND = []


def wrap(name, fn):
    @functools.wraps(fn)
    def inner(*a, **k):
        bad_in = has_nan(a) or has_nan(k)
        out = fn(*a, **k)
        if not bad_in and has_nan(out):
            shapes = [tuple(t.shape) for t in tensors_of(a)][:6]
            which = [i for i, t in enumerate(tensors_of(out)) if t.is_cuda and t.dtype.is_floating_point and t.numel() and bool(torch.isnan(t).any())]
            FOUND.append((name, shapes, which))
            print(f"  !! {name}: NaN in output(s) {which} with clean inputs; arg shapes {shapes}", flush=True)
        return out
    return inner


def main():
    import types
    from cd360 import ops
    for name, fn in list(vars(ops).items()):
        if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == ops.__name__:
            setattr(ops, name, wrap(name, fn))
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    from test_modules_gpu import _sdxl_net
    net, g = _sdxl_net(seed=43)
    net.eval()
    finetune.select_trainable(net, "pose")
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    b, n, L = 2, 2, 32
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    batch = dict(noised=rn(b, 4, L, L), timesteps=torch.full((b,), 500.0, device=DEV), context=rn(b + b * n, 77, 2048), y=rn(b + b * n, 2816),
                 pose=synth.pose_batch(b, n, seed=3), input_ref=rn(b, n, 4, L, L), sigmas_ref=torch.full((b,), 3.0, device=DEV),
                 target=rn(b, 4, L, L), target_rgb=rn(b, 3, 8 * L, 8 * L).clamp(-1, 1), w=torch.full((b, 1, 1, 1), 0.7, device=DEV),
                 mask=torch.ones(b, 1, L, L, device=DEV), opacity=torch.sigmoid(3 * rn(b, 1, 8 * L, 8 * L)))
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
    torch.empty, torch.empty_like = empty, empty_like  # from here on (the model and the batch are built)
    for step in range(2):
        print(f"train step {step} (poisoned torch.empty):", flush=True)
        total, _ = finetune.train_step(net, loss_fn, opt, **batch)
        torch.cuda.synchronize()
        bad = [k for k, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad.float()).all()]
        print(f"   loss {float(total):.6f}; non-finite gradients: {len(bad)} {bad[:4]}", flush=True)
    print("sampling forward (no_grad, fused route):", flush=True)
    with torch.no_grad():
        out = net(batch["noised"].to(torch.bfloat16), timesteps=batch["timesteps"], context=batch["context"].to(torch.bfloat16),
                  y=batch["y"].to(torch.bfloat16), pose=batch["pose"], input_ref=batch["input_ref"].to(torch.bfloat16), sigmas_ref=batch["sigmas_ref"])
    torch.cuda.synchronize()
    print("   eps finite:", bool(torch.isfinite(out[0].float()).all()), flush=True)
    print("offenders:", sorted(set(f[0] for f in FOUND)) or "none", flush=True)


if __name__ == "__main__":
    main()
