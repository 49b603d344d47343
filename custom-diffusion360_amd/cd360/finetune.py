"""Fine-tuning glue around the pose path: what the reference's Lightning engine and callbacks do to the UNet between training
steps and the sampling path.  pytorch_lightning is not part of the drop-in; these are plain functions on the
UNet module that an engine (or a test) calls.

  select_trainable          diffusion.py:117-150   which parameters train for trainkeys in {poseattn, pose, all}
  combine_losses            diffusion.py:226-241   lambda-weighted total of the four loss terms, gated by drop_im
  register_reference_hooks  diffusion.py:28-41,151-163   forward hooks that keep the pose blocks' outputs on reference images
  harvest_references        main.py:596-607        concatenate, all-gather over ranks (RCCL `nccl` / gloo), interleave, register
                                                   the `references` buffer the sampling path reads (sample.py:91)
  delta_state_dict          main.py:611-624        the delta checkpoint: pose parameters (no raymarcher buffers) + references
  load_delta_state_dict     sgm/util.py:227-240    its inverse on a freshly built UNet
  optimizer_param_groups    diffusion.py:310-361   the optimiser groups (pose parameters at lr, attention / remaining weights at multiplier * lr)
  MasterAdamW / train_step  diffusion.py:226-262   one optimisation step of the fine-tuning loop: UNet forward (HIP kernels), the four
                                                   loss terms, backward (HIP backward kernels via cd360/grad.py), AdamW on fp32 master
                                                   copies of the trainable (bf16) parameters

The all-gather in harvest_references is the one exchange step of the training side: each rank holds the features of its share
of the reference images ([N_local, hw, C] per pose block, 12 blocks) and every rank needs all of them in dataset order
(DistributedSampler gives sample i to rank i % world, hence the transpose before flattening).  It is issued once per
validation epoch on ~2 GB of bf16 features at 1024^2, as ONE collective per block over xGMI.
"""
from __future__ import annotations

import collections
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .sampling import pose_blocks


# ---------------------------------------------------------------- which parameters train
def select_trainable(unet: torch.nn.Module, trainkeys: str = "pose") -> List[str]:
    """Sets requires_grad exactly as diffusion.py:117-150 and returns the names left trainable.  Call it BEFORE any grad-enabled forward:
    freshly constructed modules have every parameter trainable, and the HIP path refuses (NotImplementedError) to run a trainable
    convolution / norm affine it cannot differentiate."""
    named = list(unet.named_parameters())
    if trainkeys == "poseattn":
        blocks = []
        for name, p in named:
            if not ("pose" in name or "transformer_blocks" in name):
                p.requires_grad = False
            elif "pose" in name:
                p.requires_grad = True
                blocks.append(name.split(".pose")[0])
        blocks = set(blocks)
        for name, p in named:
            if "transformer_blocks" in name:
                hit = any(b in name and ("attn1" in name or "attn2" in name or "pose" in name) for b in blocks)
                p.requires_grad = bool(hit)
    elif trainkeys == "pose":
        for name, p in named:
            p.requires_grad = "pose" in name
    elif trainkeys == "all":
        # the reference accepts it (diffusion.py:117-150) and so does this function -- off the hand-written path: trainable convolutions
        # run on torch's own convolution, the GroupNorm / LayerNorm affine gradients are fp32 torch reductions
        import warnings
        warnings.warn("trainkeys='all': convolution weights train through torch's convolution and the GroupNorm / LayerNorm affine gradients "
                      "through torch reductions -- correct, but not the hand-written path the shipped configs ('pose' / 'poseattn') run on",
                      stacklevel=2)
        for _, p in named:
            p.requires_grad = True
    else:
        raise ValueError(f"unknown trainkeys {trainkeys!r}")
    return [n for n, p in named if p.requires_grad]


def optimizer_param_groups(unet: torch.nn.Module, trainkeys: str = "pose", lr: float = 1e-4, multiplier: float = 0.05) -> List[dict]:
    """The optimiser groups of DiffusionEngine.configure_optimizers (diffusion.py:310-361) for the UNet: the pose parameters at
    `lr`; for 'poseattn' the attn1 / attn2 weights of the pose blocks, and for 'all' every other parameter, at `multiplier * lr`
    (configs/train_co3d_concept.yaml:2,7-8: lr 1e-4, multiplier 0.05).  Each group also lists its parameter `names`."""
    named = list(unet.named_parameters())
    main = [(n, p) for n, p in named if "pose" in n]
    low = []
    if trainkeys == "poseattn":
        blocks = {n.split(".pose")[0] for n, _ in main}
        low = [(n, p) for n, p in named if "transformer_blocks" in n and "pose" not in n and ("attn1" in n or "attn2" in n)
               and any(b in n for b in blocks)]
    elif trainkeys == "all":
        low = [(n, p) for n, p in named if "pose" not in n]
    elif trainkeys != "pose":
        raise ValueError(f"unknown trainkeys {trainkeys!r}")
    groups = [{"params": [p for _, p in main], "lr": lr, "names": [n for n, _ in main]}]
    if low:
        groups.append({"params": [p for _, p in low], "lr": multiplier * lr, "names": [n for n, _ in low]})
    return groups


# ---------------------------------------------------------------- loss weighting
def combine_losses(loss, loss_fg, loss_bg, loss_rgb, drop_im, *, rgb: bool = True, rgb_predict: bool = True, global_step: int = 1,
                   loss_fg_lambda: float = 10.0, loss_bg_lambda: float = 10.0, loss_rgb_lambda: float = 5.0, as_tensors: bool = False):
    """DiffusionEngine.forward (diffusion.py:226-241).  `drop_im` [b] is 1 where the sample kept its reference images; the
    render losses only count those samples.  Defaults are configs/train_co3d_concept.yaml:9-11.
    as_tensors=True: the logged terms stay 0-d device tensors and the `loss_rgb.mean() > 0` test of the reference becomes a device-side
    select (torch.where: a zero or NaN rgb term contributes nothing to the total, as skipping it does) -- no host synchronisation between
    the forward and the backward pass, and the step can be captured into a hipGraph.  The skipped term's GRADIENT is an exact zero on the
    product route too: `cd360_render_loss_bwd_f32` writes 0 (not 0 * NaN) where the upstream gradient of the rgb term is 0
    (tests/test_backward_gpu.py::test_render_loss_skipped_rgb_term_has_zero_gradient).  The op-by-op torch route kept for A/B
    (`routes.no_train_fusions`) has torch's semantics there: 0 * NaN = NaN."""
    total = loss.mean()
    out = {"loss": total.detach()}
    den = drop_im.sum() + 1e-12
    if rgb and global_step > 0:
        fg = (loss_fg.mean(1) * drop_im.reshape(-1)).sum() / den
        bg = (loss_bg.mean(1) * drop_im.reshape(-1)).sum() / den
        total = total + loss_fg_lambda * fg + loss_bg_lambda * bg
        out["loss_fg"], out["loss_bg"] = fg.detach(), bg.detach()
    if rgb_predict and (as_tensors or loss_rgb.mean() > 0):
        lr = (loss_rgb.mean(1) * drop_im.reshape(-1)).sum() / den
        if as_tensors:  # the reference's `if loss_rgb.mean() > 0` without a host read: a NaN (or zero) rgb term is skipped, not added
            lr = torch.where(loss_rgb.mean() > 0, lr, torch.zeros_like(lr))
        total = total + loss_rgb_lambda * lr
        out["loss_rgb"] = lr.detach()
    if not as_tensors:
        out = {k: float(v) for k, v in out.items()}
    return total, out


# ---------------------------------------------------------------- references harvest
def _is_pose_block_name(name: str) -> bool:
    parts = name.split(".")
    return len(parts) > 1 and parts[-2] == "transformer_blocks"


def register_reference_hooks(unet: torch.nn.Module):
    """-> (activations, handles).  A pose block called WITHOUT a pose (the `onlyref` validation pass over the reference images)
    returns `(x, None, None, None, None)`; the hook keeps `x` (diffusion.py:28-41: only when out[1] is None)."""
    activations: Dict[str, list] = collections.defaultdict(list)
    handles = []

    def hook(name, _module, _inp, out):
        if isinstance(out, tuple) and out[1] is None:
            activations[name].append(out[0].detach())

    for name, module in unet.named_modules():
        if _is_pose_block_name(name) and hasattr(module, "pose_emb_layers"):
            handles.append(module.register_forward_hook(lambda m, i, o, name=name: hook(name, m, i, o)))
    return activations, handles


def remove_hooks(handles) -> None:
    for h in handles:
        h.remove()


def harvest_references(unet: torch.nn.Module, activations: Dict[str, list], group: Optional[dist.ProcessGroup] = None) -> Dict[str, torch.Tensor]:
    """main.py:596-607.  Every rank must have run the same number of reference images.  Returns {block name: references}."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    out = {}
    for name, module in unet.named_modules():
        if not (_is_pose_block_name(name) and hasattr(module, "pose_emb_layers")):
            continue
        if not activations.get(name):
            raise RuntimeError(f"no reference activations were recorded for {name} (run the reference images with pose=None first)")
        local = torch.cat(activations[name]).contiguous()
        if world > 1:
            gathered = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(gathered, local, group=group)
            refs = torch.stack(gathered).transpose(0, 1).reshape(-1, *local.shape[1:])  # "(b n) ..." with n = rank
        else:
            refs = local
        if "references" in module._buffers:
            module._buffers["references"] = refs
        else:
            module.register_buffer("references", refs)
        out[name] = refs
    return out


# ---------------------------------------------------------------- delta checkpoint
def delta_state_dict(state_dict: Dict[str, torch.Tensor], embeds: Optional[list] = None) -> Dict[str, object]:
    """main.py:611-624: keys containing 'pose' (but not the raymarcher buffers) and the `references` buffers; `embed` carries
    the two new-token embedding rows of the text encoders when the caller has them (they live outside the UNet)."""
    delta = {k: v for k, v in state_dict.items() if ("pose" in k and "raymarcher" not in k) or "references" in k}
    if embeds is not None:
        delta["embed"] = list(embeds)
    return delta


def load_delta_state_dict(unet: torch.nn.Module, sd_delta: Dict[str, object], prefix: str = "model.diffusion_model.") -> List[str]:
    """sgm/util.py:227-240 for the UNet part: registers each pose block's `references` buffer, loads the pose parameters,
    returns the unexpected keys.  `sd_delta` is not modified."""
    sd = {k: v for k, v in sd_delta.items() if k != "embed"}
    for name, module in unet.named_modules():
        if _is_pose_block_name(name) and hasattr(module, "pose_emb_layers"):
            key = f"{prefix}{name}.references"
            if key not in sd:
                raise KeyError(f"delta checkpoint has no {key}")
            ref = sd.pop(key).to(next(module.parameters()).device)
            if "references" in module._buffers:
                module._buffers["references"] = ref
            else:
                module.register_buffer("references", ref)
    stripped = {k[len(prefix):] if k.startswith(prefix) else k: v for k, v in sd.items()}
    _missing, unexpected = unet.load_state_dict(stripped, strict=False)
    return list(unexpected)


# ---------------------------------------------------------------- one optimisation step
class MasterAdamW:
    """AdamW (configs/train_co3d_concept.yaml: optimizer_config AdamW) on fp32 master copies of the trainable parameters: the HIP
    path keeps the UNet in bf16, and a bf16 parameter cannot absorb updates of lr ~ 1e-5 (its spacing near 1 is 8e-3), so the
    optimiser state and the accumulated weights live in fp32 and the bf16 parameters are refreshed from them after every step.
    bf16 parameters on the GPU take the fused path: masters and both moments in three flat fp32 buffers, the whole update (gradient
    read, decay, moments, bias-corrected step, bf16 write-back) in one pass of cd360_adamw_bf16 with the step count on the device
    (graph-capturable as it is).  Anything else (CPU, fp32 parameters, amsgrad / maximize) runs torch.optim.AdamW on the masters."""

    def __init__(self, params, lr: float = 1e-5, fused: Optional[bool] = None, **kw):
        """`params`: an iterable of parameters, or optimiser groups as returned by optimizer_param_groups (per-group `lr`)."""
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [{**{k: v for k, v in g.items() if k not in ("params", "names")}, "params": [p for p in g["params"] if p.requires_grad]}
                      for g in params]
        else:
            groups = [{"params": [p for p in params if p.requires_grad]}]
        self.params = [p for g in groups for p in g["params"]]
        can_fuse = (bool(self.params) and all(p.is_cuda and p.dtype == torch.bfloat16 and p.is_contiguous() for p in self.params)
                    and not kw.get("amsgrad") and not kw.get("maximize") and not (set(kw) - {"betas", "eps", "weight_decay", "capturable", "amsgrad",
                                                                                            "maximize", "foreach"}))
        if fused and not can_fuse:
            raise ValueError("MasterAdamW(fused=True) needs bf16 parameters on the GPU and plain AdamW options")
        self.fused = can_fuse if fused is None else bool(fused)
        if self.fused:
            from . import ops
            dev = self.params[0].device
            begin, total = [], 0
            for p in self.params:
                begin.append(total)
                total += (p.numel() + 7) // 8 * 8  # every tensor starts on a 32-byte boundary of the flat buffers
            self._master = torch.zeros(total, dtype=torch.float32, device=dev)
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self._master), torch.zeros_like(self._master)
            self.steps = torch.zeros((), dtype=torch.float32, device=dev)  # steps taken; on the device so that graph replays advance it
            self.master = [self._master[b:b + p.numel()].view(p.shape) for b, p in zip(begin, self.params)]
            for m, p in zip(self.master, self.params):
                m.copy_(p.detach())
            self.betas, self.eps = tuple(kw.get("betas", (0.9, 0.999))), float(kw.get("eps", 1e-8))
            self.param_groups = [{"lr": g.get("lr", lr), "weight_decay": g.get("weight_decay", kw.get("weight_decay", 1e-2)), "n": len(g["params"])}
                                 for g in groups]  # edit lr / weight_decay here (a captured graph keeps the values it was captured with)
            self._begin, self._plan, self._plan_key, self.opt, self.capturable = begin, None, None, None, True
            self._ops = ops
            return
        self.master = [p.detach().float().clone() for p in self.params]
        it = iter(self.master)
        kw.pop("foreach", None)
        self.opt = torch.optim.AdamW([{**g, "params": [next(it) for _ in g["params"]]} for g in groups], lr=lr, **kw)
        self.param_groups = self.opt.param_groups
        self.capturable = all(g.get("capturable") for g in self.opt.param_groups)

    def bump_versions(self, active=None) -> None:
        """The fused kernel (and a hipGraph replay of it) writes the bf16 parameters through raw pointers, which torch's version counter
        does not see.  Every derived-weight cache of the package (LayerNorm-folded packs, W^T copies, fp32 biases, the FeatureNeRF's fused
        weights, the pose-projection split) is keyed on `_version`, so the update is made visible here -- no kernel is launched."""
        for i in (range(len(self.params)) if active is None else active):
            torch.autograd.graph.increment_version(self.params[i])

    allreduce_single = False  # True: run the gradient all-reduce even in a group of one (tests of the collective path on a 1-GPU box)

    def zero_grad(self) -> None:
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def allreduce_grads(self, group: Optional[dist.ProcessGroup] = None) -> None:
        """Data-parallel fine-tuning (the reference trains under Lightning DDP, main.py): average the gradients of the trainable
        parameters over the ranks as ONE flat fp32 all-reduce (66.7 M values = 267 MB at SDXL size: a single RCCL ring over xGMI,
        per-link bound, instead of 96 small collectives).  No-op without an initialised process group."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not self.allreduce_single):
            return
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("allreduce_grads: a trainable parameter has no gradient on this rank (ranks would disagree on the buffer layout)")
        flat = torch.cat([g.reshape(-1).float() for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
        off = 0
        for p, g in zip(self.params, grads):
            n = g.numel()
            p.grad = flat[off:off + n].reshape(g.shape).to(g.dtype)
            off += n

    def _fused_plan(self, active):
        lrs = [g["lr"] for g in self.param_groups for _ in range(g["n"])]
        wds = [g["weight_decay"] for g in self.param_groups for _ in range(g["n"])]
        key = (tuple(active), tuple(lrs), tuple(wds))
        if self._plan_key != key:
            self._plan = self._ops.AdamwPlan([self.params[i] for i in active], [self._begin[i] for i in active], [lrs[i] for i in active],
                                             [wds[i] for i in active])
            self._plan_key = key
        return self._plan

    @torch.no_grad()
    def step(self, group: Optional[dist.ProcessGroup] = None) -> None:
        self.allreduce_grads(group)
        if self.fused:
            # parameters without a gradient are skipped as torch does; the step count is one per optimiser, not one per tensor
            active = [i for i, p in enumerate(self.params) if p.grad is not None]
            if active:
                grads = [self.params[i].grad if self.params[i].grad.is_contiguous() else self.params[i].grad.contiguous() for i in active]
                self._ops.adamw_step(self._fused_plan(active), grads, self._master, self.exp_avg, self.exp_avg_sq, self.steps, self.betas[0],
                                     self.betas[1], self.eps)
                self.bump_versions(active)
            return
        for m, p in zip(self.master, self.params):
            m.grad = None if p.grad is None else p.grad.float()
        self.opt.step()
        for m, p in zip(self.master, self.params):
            p.copy_(m)


def train_step(unet: torch.nn.Module, loss_fn, optimizer, *, noised, timesteps, context, y, pose, input_ref, sigmas_ref, target, target_rgb,
               w, mask, opacity, drop_im=None, mask_ref=None, as_tensors: bool = False, **loss_kw):
    """One step of the fine-tuning loop on already-noised inputs (the engine's denoiser / conditioner / data loading stay outside
    the path): forward, StandardDiffusionLossImgRef.get_loss (loss.py:177-209), the lambda-weighted total
    (diffusion.py:226-241), backward, optimiser step.  Returns (total loss tensor, dict of logged terms).
    The logged terms are read back AFTER the optimiser step has been queued (one host synchronisation per step, at its end, instead of
    four between the forward and the backward pass); as_tensors=True leaves them on the device (no synchronisation at all)."""
    optimizer.zero_grad()
    out, fgs, alphas, rgbs = unet(noised, timesteps=timesteps, context=context, y=y, pose=pose, input_ref=input_ref, sigmas_ref=sigmas_ref,
                                  mask_ref=mask_ref)
    l2, lfg, lbg, lrgb = loss_fn.get_loss(out, fgs, rgbs, target, target_rgb, w, mask, mask_ref, opacity, alphas)
    if drop_im is None:
        drop_im = torch.ones(noised.shape[0], device=noised.device)
    total, logged = combine_losses(l2, lfg, lbg, lrgb, drop_im, as_tensors=True, **loss_kw)
    total.backward()
    optimizer.step()
    if not as_tensors:
        logged = {k: float(v) for k, v in logged.items()}
    return total.detach(), logged


class GraphedTrainStep:
    """train_step captured ONCE into a hipGraph and replayed: forward, loss, backward and the optimiser step of BASELINE config 4 are
    ~6400 kernel launches and ~16000 torch operator calls per step -- host work that takes longer than the kernels run;
    a replay costs one launch.  Conditions: fixed shapes (the batch is copied into static buffers), a MasterAdamW on its fused
    path (or built with capturable=True), the raymarchers on device_rng=True (the stratified jitter of patch x / y is then drawn by the device generator, which
    a graph advances on every replay; the reference draws those two on the CPU generator).  Under an initialised process group the
    gradient all-reduce is part of the graph (RCCL collectives capture; tools/probe/rccl_graph_probe.py).  The returned loss terms are 0-d device tensors that the next replay overwrites.
    Side effects of construction, all deliberate: (1) `warmup` REAL optimisation steps run on the construction batch before the capture
    (weights, moments and the device step counter advance by `warmup`; pass warmup_restore=True to snapshot and restore parameters,
    masters, moments and the step counter around them); (2) `device_rng = True` stays set on every raymarcher, so later eager steps draw
    their patch jitter from the device generator too; (3) lr and weight_decay are frozen into the graph at capture time -- an LR scheduler
    that edits `optimizer.param_groups` has no effect on replays (re-capture to change them)."""

    def __init__(self, unet: torch.nn.Module, loss_fn, optimizer: "MasterAdamW", batch: dict, warmup: int = 3, warmup_restore: bool = False,
                 weight_prefetch: bool = False, **loss_kw):
        if not optimizer.capturable:
            raise ValueError("GraphedTrainStep: the torch fall-back of MasterAdamW must be built with capturable=True (the fused path is as it is)")
        self.unet, self.loss_fn, self.optimizer, self.loss_kw = unet, loss_fn, optimizer, loss_kw
        for m in unet.modules():
            if hasattr(m, "device_rng"):
                m.device_rng = True
        clone = lambda v: v.clone() if torch.is_tensor(v) else ({k: clone(x) for k, x in v.items()} if isinstance(v, dict) else v)
        self.static = {k: clone(v) for k, v in batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # allocator, library workspaces, weight packs, optimiser state: all settled before the capture
            saved = None
            if warmup_restore and optimizer.fused:
                saved = ([p.detach().clone() for p in optimizer.params], optimizer._master.clone(), optimizer.exp_avg.clone(),
                         optimizer.exp_avg_sq.clone(), optimizer.steps.clone())
            for _ in range(warmup):
                train_step(unet, loss_fn, optimizer, as_tensors=True, **self.static, **loss_kw)
            if saved is not None:
                with torch.no_grad():
                    for p, q in zip(optimizer.params, saved[0]):
                        p.copy_(q)
                    optimizer._master.copy_(saved[1]); optimizer.exp_avg.copy_(saved[2]); optimizer.exp_avg_sq.copy_(saved[3]); optimizer.steps.copy_(saved[4])
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        from . import memo
        memo.new_epoch()  # nothing memoised before this point -- by the warm-up steps or by an earlier capture -- may feed this graph
        # with a process group the gradient all-reduce (one flat RCCL all-reduce, MasterAdamW.allreduce_grads) is captured with the step; the
        # group's watchdog thread polls events meanwhile, which only the thread-local capture mode tolerates.  Every rank captures the same
        # sequence and must replay in lock-step, as with any collective.
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        import contextlib
        pf = contextlib.nullcontext()
        if weight_prefetch:  # (cd360/prefetch.py; measured SLOWER here, 121.3 vs 117.9 ms: 3400 short launches leave the touches no room -- off)
            from .prefetch import WeightPrefetcher
            pf = WeightPrefetcher(next(unet.parameters()).device)
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            with pf:
                self.total, self.logged = train_step(unet, loss_fn, optimizer, as_tensors=True, **self.static, **loss_kw)
        memo.new_epoch()

    @staticmethod
    def _load(dst, src):
        if torch.is_tensor(dst):
            dst.copy_(src)
        elif isinstance(dst, dict):
            for k in dst:
                GraphedTrainStep._load(dst[k], src[k])

    def __call__(self, **batch):
        """Copies `batch` (same keys and shapes as at construction; omitted keys keep their values) into the static buffers, replays."""
        for k, v in batch.items():
            self._load(self.static[k], v)
        self.graph.replay()
        self.optimizer.bump_versions()  # the replay rewrote the parameters behind torch's back: invalidate every cache keyed on their version
        return self.total, self.logged
