"""The sampling job of the pose path: one captured sampler per GPU walking its share of the target poses, one all-gather of the final
latents at the end (BASELINE.json configs[1] and [2]).

The reference samples its target poses one after the other on one GPU (sample.py:331-349: for every pose, `sampler(denoiser, randn, cond,
uc)` with the patched UNet, `clear_rendered_feat` between images).  Here every rank takes the contiguous share `shard.assign_poses` gives
it, samples those poses with ONE `Sampler` -- whose render step and steady-state step are each captured once into a hipGraph and re-pointed
at the next pose by `Sampler.retarget` (camera buffer and conditioning rewritten in place) -- and the ranks exchange nothing until
`shard.gather_latents` collects the final latents (RCCL all-gather over xGMI; gloo in the CPU tests).  `bench.py` times exactly this loop.
"""
from __future__ import annotations

import sys
from typing import Callable, List, Sequence

import torch

from . import shard


class Sampler:
    """Minimal Euler (DDIM-equivalent, EpsScaling) step with the 3-way image/text CFG of guiders.py:102-133.
    With `use_graph` the steady-state step (cached render) and the render step are each captured once into a hipGraph and
    replayed: ~3000 launches per step are then issued by the GPU front end instead of the Python interpreter."""

    def __init__(self, net, pose, ctx, y, n_steps, scale=7.5, scale_im=3.5, use_graph=False, prefetch=None, graph_render=True):
        from cd360 import sampler as S
        self.net, self.pose, self.n_steps = net, pose, n_steps
        if use_graph:  # the graphs read the cameras through ONE buffer this sampler owns (retarget rewrites it in place)
            from sgm.modules.utils_cameraray import PoseBuffer
            self.pose = PoseBuffer(pose, ctx.device)
        self.scale, self.scale_im = scale, scale_im
        dev = ctx.device
        # the reference's own stack (cd360/sampler.py mirrors sampling.py / guiders.py / denoiser.py; parity: tests/test_sampler_cpu.py)
        self.denoiser = S.DiscreteDenoiser().to(dev)
        self.guider = S.ScheduledCFGImgTextRef(scale, scale_im)
        self.sigmas = S.LegacyDDPMDiscretization()(n_steps, device=dev)  # n_steps + 1 values, last = 0
        # conditioning is constant over a trajectory: the guider's (uc, uc, c) batch is assembled once per image, not per step
        self.bs = bs = ctx.shape[0] // 3  # diffusion samples (target poses) per replay: ctx / y hold [uc x bs | . | c x bs]
        c = {"crossattn": ctx[2 * bs:], "vector": y[2 * bs:]}
        uc = {"crossattn": ctx[:bs], "vector": y[:bs]}
        _, _, cond3 = self.guider.prepare_inputs(ctx.new_zeros(bs, 1), ctx.new_zeros(bs), c, uc)
        self.ctx, self.y = cond3["crossattn"].contiguous(), cond3["vector"].contiguous()
        self.prefetch = prefetch  # cd360.prefetch.WeightPrefetcher or None: armed around both captures
        self.use_graph, self.graph = use_graph, None
        self.graph_render, self.rgraph = use_graph and graph_render, None  # graph_render=False: the render step launched eagerly (A/B)
        # staged steps (round 6): the captured graphs read every per-step scalar from tables through a device-side step index and start /
        # end on cd360_unet_stage_in / cd360_cfg_euler_step_cl, so a replay holds no torch-issued kernel (cd360.routes.no_stage: the A/B partner);
        # eager launches (use_graph=False) run the same staged step
        from cd360 import routes
        self.staged = bool(not routes.no_stage and self._stageable())

    def retarget(self, pose, ctx, y):
        """Point the sampler at another target pose / conditioning (the next pose of this rank's share).  The captured graphs read both
        through fixed device buffers, so the new values are copied INTO them: the CFG conditioning batch, and the packed
        [3, n+1, 16] camera tensor this sampler owns (sgm/modules/utils_cameraray.py::PoseBuffer)."""
        bs = self.bs
        c = {"crossattn": ctx[2 * bs:], "vector": y[2 * bs:]}
        uc = {"crossattn": ctx[:bs], "vector": y[:bs]}
        _, _, cond3 = self.guider.prepare_inputs(ctx.new_zeros(bs, 1), ctx.new_zeros(bs), c, uc)
        self.ctx.copy_(cond3["crossattn"])
        self.y.copy_(cond3["vector"])
        if self.use_graph:
            self.pose.rewrite(pose)
        else:
            self.pose = pose
        if self.staged and getattr(self, "lab", None) is not None:
            self.lab.copy_(self.net.label_emb(self.y.to(self.net.dtype)))

    # ------------------------------------------------------------------------------------------------ staged steps
    def _stageable(self) -> bool:
        """The staged step serves the UNet this package builds (4 -> C input convolution, 4-channel output, merged emb projections)."""
        from cd360 import routes
        net = self.net
        try:
            conv = net.input_blocks[0][0]
            return (isinstance(conv, torch.nn.Conv2d) and conv.in_channels == 4 and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
                    and conv.stride == (1, 1) and conv.out_channels % 8 == 0 and conv.weight.dtype == torch.bfloat16 and conv.weight.is_cuda
                    and net.out[2].out_channels == 4 and hasattr(net, "forward_staged") and not routes.no_emb_merge)
        except (AttributeError, IndexError, TypeError):
            return False

    @torch.no_grad()
    def _build_stage(self, x):
        """Per-schedule tables and static buffers of the staged steps: row i = what step i of the trajectory derives from sigma_i alone,
        computed with the denoiser's / the UNet's own modules exactly as the un-staged step computes them inside its graph."""
        from sgm.modules.diffusionmodules.util import timestep_embedding
        net, dev, dt = self.net, x.device, self.net.dtype
        rows, tembs = [], []
        for i in range(self.n_steps):
            sq = self.denoiser.possibly_quantize_sigma(self.sigmas[i].reshape(1))
            _, _, c_in, c_noise = self.denoiser.scaling(sq)
            c_noise = self.denoiser.possibly_quantize_c_noise(c_noise)
            rows.append(torch.stack([self.sigmas[i], self.sigmas[i + 1], c_in[0], torch.zeros_like(c_in[0])]))
            tembs.append(net.time_embed(timestep_embedding(c_noise.expand(3), net.model_channels).to(dt))[0])
        self.step_tab = torch.stack(rows).float().contiguous()
        self.temb_tab = torch.stack(tembs).to(dt).contiguous()
        self._iota = torch.arange(self.n_steps, dtype=torch.int32, device=dev)
        self.gi = torch.zeros(1, dtype=torch.int32, device=dev)
        conv = net.input_blocks[0][0]
        self.w36 = conv.weight.detach().float().permute(2, 3, 1, 0).reshape(36, conv.out_channels).contiguous()
        self.b_in = (conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=dev)).contiguous()
        self.lab = net.label_emb(self.y.to(dt)).contiguous()
        bs3, (H, W) = self.y.shape[0], x.shape[2:]
        self.h0 = torch.empty(bs3, H * W, conv.out_channels, dtype=dt, device=dev)
        self.emb_act = torch.empty_like(self.lab)

    def _math_staged(self):
        """One sampler step on the static buffers, in place on self.gx: stage-in kernel -> UNet trunk -> fused [c_out, 3-way CFG, to_d,
        Euler] kernel reading the output convolution's rows as they lie."""
        from cd360 import ops
        H, W = self.gx.shape[2:]
        ops.unet_stage_in(self.gx, self.step_tab, self.gi, self.w36, self.b_in, self.temb_tab, self.lab, self.h0, self.emb_act)
        eps_cl = self.net.forward_staged(self.h0, self.emb_act, self.ctx, self.pose, H, W)
        return ops.cfg_euler_step_cl(self.gx, eps_cl, self.step_tab, self.gi, self.scale, self.scale_im)

    def _math(self, x, s, s_next, t_unused=None):
        """One sampler step = guider.prepare_inputs -> DiscreteDenoiser (sigma -> table index, c_in) -> UNet -> fused
        [c_out, 3-way CFG, to_d, Euler] kernel."""
        from cd360.sampler import fused_cfg3_euler_step
        unet = lambda x_in, c_noise: self.net(x_in, timesteps=c_noise, context=self.ctx, y=self.y, pose=self.pose)[0]  # noqa: E731
        return fused_cfg3_euler_step(self.denoiser, unet, x, s, s_next, self.scale, self.scale_im)

    @torch.no_grad()
    def eps(self, x, i):
        """The UNet's output for step i of the schedule (the three CFG branches), launched eagerly: what --fp8-attn's tolerance report compares."""
        x3 = x.expand(3, -1, -1, -1) if x.shape[0] == 1 else torch.cat([x] * 3)
        x_in, c_noise, _, _, _ = self.denoiser.network_inputs(x3, self.sigmas[i].expand(x3.shape[0]), {})
        return self.net(x_in, timesteps=c_noise, context=self.ctx, y=self.y, pose=self.pose)[0].float()

    def _pin_rendered(self):
        """Keep every block's cached render in a fixed buffer so a captured graph keeps reading the current image's render."""
        from cd360 import sampling
        for _, blk in sampling.pose_blocks(self.net):
            blk.pin_rendered()  # render + its pose_emb_layers half (rendered_feat @ Wb^T) in buffers that stay put across images
        for att in sampling._cross_attentions(self.net):  # same for the per-image context K / V^T cache
            if att._kv_cache is None:
                continue
            key, (k, vt, nk) = att._kv_cache[:2]
            st = getattr(att, "_static_kv", None)
            if st is None or st[0].shape != k.shape:
                att._static_kv = (k.clone(), vt.clone())
            else:
                st[0].copy_(k)
                st[1].copy_(vt)
            att._kv_cache = (key, (att._static_kv[0], att._static_kv[1], nk)) + tuple(att._kv_cache[2:])
            from cd360 import routes
            if routes.fp8_attn and 64 < nk <= 96:  # --fp8-attn: the e4m3 image of the pinned K / V, re-packed by every (captured) render step
                sk = att._static_kv[0]
                if getattr(att, "_static_kv8", None) is None or att._static_kv8[0].shape[0] != sk.shape[0]:
                    att._static_kv8 = ops_kv8_buffers(sk, att.heads)
                from cd360 import ops
                ops.kv_pack_fp8(sk, att._static_kv[1], nk, att.heads, out=att._static_kv8)
                att._kv8_cache = (sk, sk._version, att._static_kv8)

    def _capture(self, fn):
        """Warm `fn` on a side stream (allocator / library workspaces), then capture it into a hipGraph."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # under torch.distributed the process group's watchdog thread may still poll the events of finished collectives (the launch
        # barrier) while this thread captures: only the thread-local capture mode tolerates that (tools/probe/rccl_graph_probe.py)
        import torch.distributed as dist
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        import contextlib
        from . import memo
        memo.new_epoch()  # values memoised by the eager warm-up (or by the other capture) must be re-derived inside this graph
        with torch.cuda.graph(graph, capture_error_mode=mode):
            # every GEMM / convolution launch of the captured step also enqueues, on a forked side stream, a touch of its weights that runs
            # beside the preceding launches (one step streams 5.3 GB of weights through a 256 MB Infinity Cache: cd360/prefetch.py)
            with (self.prefetch if self.prefetch is not None else contextlib.nullcontext()):
                out = fn()
        memo.new_epoch()
        return graph, out

    def _render(self):
        """Step 0 of an image: clear the cached render, run the full step (all 12 FeatureNeRF renders), re-pin the caches."""
        from cd360 import sampling
        sampling.clear_rendered_feat(self.net)
        out = self._math_staged() if self.staged else self._math(self.gx, self.gs[0], self.gs[1], self.gt)
        self._pin_rendered()
        return out

    @torch.no_grad()
    def prepare(self, x):
        """Untimed set-up of the graph mode: one eager render (builds the per-image tables and the static cache buffers), then the
        steady-state step and the render step are each captured once.  Both graphs read / write the same static buffers."""
        if not self.use_graph or self.graph is not None:
            return
        s, s_next, t = self.sigmas[0], self.sigmas[1], self.sigmas[0:1]
        self.gx, self.gs, self.gt = x.clone(), torch.stack([s, s_next]), t.clone()
        if self.staged:
            self._build_stage(x)
        self._render()
        if self.staged:
            self.gx.copy_(x)  # (the staged step updates gx in place: captures and warm-ups below start from a sane latent again)
        self.graph, self.gout = self._capture(self._math_staged if self.staged else (lambda: self._math(self.gx, self.gs[0], self.gs[1], self.gt)))
        if self.graph_render:
            try:
                self.rgraph, self.rout = self._capture(self._render)
                self._pins = self._snapshot_pins()
            except Exception as e:  # a host synchronisation inside the render path would make it uncapturable: stay eager
                print(f"[bench] render step not captured ({type(e).__name__}: {e}); launching it eagerly", file=sys.stderr)
                self.rgraph = None
                torch.cuda.synchronize()

    def _snapshot_pins(self):
        from cd360 import sampling
        return ([(blk, blk.rendered_feat, blk._rendered_proj) for _, blk in sampling.pose_blocks(self.net)],
                [(att, att._kv_cache) for att in sampling._cross_attentions(self.net)])

    def _restore_pins(self):
        for blk, r, proj in self._pins[0]:
            blk.rendered_feat, blk._rendered_proj = r, proj
        for att, kv in self._pins[1]:
            att._kv_cache = kv

    @torch.no_grad()
    def step(self, x, i, alias: bool = False):
        """Step i of the schedule on latent x.  alias=True (the job loop): the result may be the sampler's own latent buffer, valid until
        the next call, and passing it straight back skips the copy-in -- a staged replay is then ONE 4-byte index copy plus the graph."""
        i = i % self.n_steps
        s, s_next, t = self.sigmas[i], self.sigmas[i + 1], self.sigmas[i:i + 1]
        if self.staged:
            # (eager launches run the SAME staged step: the stage-in kernel's input convolution rounds a few outputs to the other bf16
            # neighbour than the implicit-GEMM kernel does, and 70 random-init blocks amplify that to 1e-2 of eps -- graph and eager
            # must not differ by it; tests/test_f_rows_gpu.py holds the staged ends against the module arithmetic op by op)
            if self.use_graph:
                self.prepare(x)
            elif getattr(self, "step_tab", None) is None:
                self.gx = x.clone()
                self._build_stage(x)
            if x is not self.gx:
                self.gx.copy_(x)
            self.gi.copy_(self._iota[i:i + 1])
            if not self.use_graph:
                if i == 0:
                    from cd360 import sampling
                    sampling.clear_rendered_feat(self.net)
                self._math_staged()
            elif i == 0:
                if self.rgraph is not None:
                    self.rgraph.replay()
                    self._restore_pins()
                else:
                    self._render()
            else:
                self.graph.replay()
            return self.gx if alias else self.gx.clone()
        if not self.use_graph:
            if i == 0:
                from cd360 import sampling
                sampling.clear_rendered_feat(self.net)  # new image: the render runs again
            return self._math(x, s, s_next, t)
        self.prepare(x)
        self.gx.copy_(x)
        self.gs[0].copy_(s)
        self.gs[1].copy_(s_next)
        self.gt.copy_(t)
        if i == 0:
            if self.rgraph is not None:
                self.rgraph.replay()
                self._restore_pins()
                return self.rout.clone()
            return self._render().clone()
        self.graph.replay()
        return self.gout.clone()


def ops_kv8_buffers(k, heads):
    return (torch.empty(k.shape[0], heads, 96 * 64 + 64 * 128, dtype=torch.uint8, device=k.device),
            torch.empty(k.shape[0], heads, 2, dtype=torch.float32, device=k.device))


def sample_assigned(sampler, jobs: Sequence[tuple], steps: int) -> List[torch.Tensor]:
    """The per-rank loop of the job: `jobs` = [(pose, ctx, y, x0), ...] for the poses of this rank, in `shard.assign_poses` order; every
    pose walks `steps` steps of the sampler's schedule from its own start latent (step 0 renders).  `sampler` needs `retarget(pose, ctx, y)`
    and `step(x, i)` -- `Sampler` above; the gloo tests drive the same loop with a CPU stand-in."""
    finals = []
    for j, (pose, ctx, y, x0) in enumerate(jobs):
        if j > 0:
            sampler.retarget(pose, ctx, y)
        x = x0.clone()
        alias = isinstance(sampler, Sampler)  # (the CPU stand-ins of the gloo tests take (x, i) only)
        for i in range(steps):
            x = sampler.step(x, i, alias=True) if alias else sampler.step(x, i)
        finals.append(x.clone() if alias else x)
    return finals


def sample_poses(make_sampler: Callable, make_job: Callable[[int], tuple], num_poses: int, steps: int, world: int = 1, rank: int = 0):
    """BASELINE configs[2] as a function: `num_poses` target poses over `world` ranks.  `make_job(p)` builds pose p's (pose, ctx, y, x0);
    `make_sampler(pose, ctx, y)` the rank's sampler for its first pose.  Returns (latents of ALL poses [num_poses, 4, L, L] in pose
    order, identical on every rank; this rank's pose indices).  No collective on the data path; one all-gather at the end."""
    if num_poses < world:  # decided from the arguments alone, so EVERY rank raises (a rank-local check would leave the others in the all-gather)
        raise ValueError("more ranks than target poses: every rank needs at least one pose (the all-gather is sized per rank)")
    mine = shard.assign_poses(num_poses, world, rank)
    jobs = [make_job(p) for p in mine]
    sampler = make_sampler(*jobs[0][:3])
    finals = sample_assigned(sampler, jobs, steps)
    return shard.gather_latents(torch.cat(finals, 0), num_poses), mine
