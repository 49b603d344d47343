#!/usr/bin/env python
"""Headline benchmark: UNet denoise steps/s at SDXL 1024^2 with 50 reference views (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 50 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload ("sample.py 50-step sampling, SDXL 1024^2, 50-view synthetic cameras"): random-init SDXL UNet from the reference's
network_config (bf16), latent 128^2, 3-way CFG batch (uncond / image / image+text, sample.py:166-171), 50 reference views
taken from synthetic per-block `references` buffers, one target pose per GPU.  A "step" is one UNet denoise step over the
CFG batch plus the CFG combine and Euler update.  The timed region walks the sampler's own schedule: step 0 of every 50-step
trajectory runs the 12 FeatureNeRF renders, the other 49 use the cached render (sample.py:122-133), so with --steps 50 the
timed region is exactly one image.  Multi-GPU = independent target poses (weak scaling), one RCCL all-gather of the final
latents at the end of the job (outside the per-step path).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def build_model(latent: int, n_ref: int, n_train: int, device, seed: int = 0, native_sampling: bool = True):
    from cd360.configs import SDXL_NETWORK_CONFIG
    from sgm.util import instantiate_from_config
    from cd360 import sampling

    torch.manual_seed(seed)
    with torch.device(device):
        net = instantiate_from_config(SDXL_NETWORK_CONFIG)
    net = net.to(torch.bfloat16).eval()
    g = torch.Generator(device=device).manual_seed(seed + 1)
    with torch.no_grad():
        for name, blk in sampling.pose_blocks(net):
            c = blk.pose_emb_layers.weight.shape[0]
            # the stock init makes the pose path a no-op: perturb pose_emb_layers / decoder / proj_out
            blk.pose_emb_layers.weight.add_(torch.randn(c, 2 * c, generator=g, device=device, dtype=torch.float32).mul_(0.02).to(torch.bfloat16))
            blk.pose_featurenerf.model.decoder.weight.copy_(torch.randn(4, c, generator=g, device=device).mul_(0.02))
        for m in net.modules():
            if m.__class__.__name__ == "SpatialTransformer":
                m.proj_out.weight.copy_(torch.randn(m.proj_out.weight.shape, generator=g, device=device).mul_(0.02))
        # (the 320 -> 4 output convolution is zero-initialised too, openaimodel.py:967-971: eps would be exactly 0 and every comparison of it vacuous)
        net.out[2].weight.copy_(torch.randn(net.out[2].weight.shape, generator=g, device=device).mul_(0.02))
        refs = {}
        for name, blk in sampling.pose_blocks(net):
            c = blk.pose_emb_layers.weight.shape[0]
            r = latent // 2 if c == 640 else latent // 4
            refs[name] = torch.randn(n_train + 1, r * r, c, generator=g, device=device).to(torch.bfloat16)
        sampling.set_references(net, refs)
        choices = [int(x) for x in torch.linspace(0, n_train - n_train / n_ref, n_ref)]
        if native_sampling:  # (--route sample_py leaves this to the rebinding sample.py itself performs)
            sampling.enable_reference_sampling(net, choices)
    return net


from cd360.job import Sampler, ops_kv8_buffers, sample_assigned  # noqa: E402,F401  (the sampling job lives in the package: cd360/job.py)


def cpu_baseline(net, latent: int, threads: int):
    """CPU oracle (fp32 restatement of the reference) on a bounded sample of the same workload: ONE of the three CFG branches of
    ONE steady-state denoise step (cached render) at the bench's latent size; scaled by 1/3 to the metric's unit."""
    from oracle import pose_path as O
    from cd360 import sampling

    torch.set_num_threads(threads)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items() if "references" not in k and "raymarcher" not in k}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, latent, latent, generator=g)
    ctx = torch.randn(1, 77, 2048, generator=g)
    y = torch.randn(1, 2816, generator=g)
    rendered = {}
    for name, blk in sampling.pose_blocks(net):
        st_key, _, d = name.rpartition(".transformer_blocks.")
        c = blk.pose_emb_layers.weight.shape[0]
        r = latent // 2 if c == 640 else latent // 4
        rendered.setdefault(st_key, {})[int(d)] = torch.randn(1, r * r, c, generator=g) * 0.1
    cams = torch.zeros(1, 2, 16)  # unused on the cached path, but marks the blocks as pose blocks
    t0 = time.perf_counter()
    with torch.no_grad():
        O.unet_forward(sd, x, torch.tensor([500.0]), ctx, y, cams=cams, st_state={"rendered": rendered})
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / (3.0 * dt), 6), "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle/pose_path.py unet_forward, fp32: 1 of 3 CFG branches of one steady-state step (cached render), latent {latent}^2, "
                      f"{dt:.1f} s measured, value = 1/(3*t)"}



def _baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim (the file travels with the repository); the literal is the fall-back."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:  # noqa: BLE001
        return "UNet denoise steps/sec @ SDXL 1024\u00b2, 50 ref views, 1/2/4/8 MI355X"


def self_launch_argv(n: int, argv, port: int = 0):
    """argv of the N-rank launch of this same command: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py <the caller's arguments>` (the form the driver uses for N > 1).  port = 0 picks a free one."""
    if not port:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(argv[0])] + list(argv[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--latent", type=int, default=128, help="latent side (128 = 1024^2 image)")
    ap.add_argument("--refs", type=int, default=50)
    ap.add_argument("--traj", type=int, default=50, help="sampler steps per image (render on step 0 of each)")
    ap.add_argument("--poses", type=int, default=0, help="target poses of the whole job (default: one per GPU); P > GPUs gives every rank "
                    "its cd360.shard.assign_poses share and runs them one after the other (BASELINE configs[2] on fewer than 8 GPUs)")
    ap.add_argument("--poses-per-replay", type=int, default=1, help="target poses batched into ONE denoise step (CFG batch 3 x this): what a rank "
                    "with several poses to sample gains from batching them (the 1280-level GEMMs then run 256-row tiles).  A separate line -- "
                    "`config.poses_per_replay`, metric suffixed -- never the headline; --poses must be a multiple of it per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying the steady-state step from a hipGraph")
    ap.add_argument("--route", choices=["fused", "hooked", "module", "sample_py"], default="fused",
                    help="hooked: a no-op forward hook on every transformer BLOCK (what the references harvest registers, diffusion.py:151-163): "
                    "the SpatialTransformer leaves its fused route, the blocks keep their fused internals.  module: a no-op hook on a SUBMODULE "
                    "of every block, so the blocks take the strict module route a patched sample.py (sample.py:247-262) takes -- every "
                    "submodule through the module protocol, same kernels, un-fused.  sample_py: `forward` rebound on every SpatialTransformer / block "
                    "instance the way the UNCHANGED sample.py does it (sample.py:247-278; stand-in functions of tests/golden/sample_py_stub.py), "
                    "no call into cd360.sampling: the rebinding is recognised and served by the fused route")
    ap.add_argument("--no-train-step", action="store_true", help="skip the fine-tuning step measurement appended after the timed region (BASELINE configs[3])")
    ap.add_argument("--no-weight-prefetch", action="store_true", help="capture the steps without the weight prefetcher (cd360/prefetch.py): the A/B partner")
    ap.add_argument("--prefetch-wgs", type=int, default=256)
    ap.add_argument("--prefetch-lag", type=int, default=2)
    ap.add_argument("--prefetch-min-mb", type=float, default=0.25)
    ap.add_argument("--fp8-attn", action="store_true", help="BASELINE configs[4]: the text / pose-token cross-attention of every block with q K^T and P V on "
                    "fp8 MFMA (cd360_qproj_attn_fp8_bf16) for the whole run; adds an `fp8_tolerance` object (rendered features and eps against the bf16 run)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher.  The same command line is re-run as N ranks (one per GPU) under
        # torch.distributed.run on this node, rendezvous on 127.0.0.1 at a free port; rank 0 prints the one JSON line.
        os.execv(sys.executable, self_launch_argv(args.gpus, sys.argv))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started a different number of ranks"
    # CD360_BENCH_ONE_GPU=1: every rank on GPU 0 over gloo -- a SANITY run of the N > 1 control path (sharding, barriers, captures under a
    # process group, per-rank times, the final all-gather) on a 1-GPU box; RCCL refuses two ranks on one device.  Never a measurement.
    one_gpu = bool(os.environ.get("CD360_BENCH_ONE_GPU"))
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or "RANK" in os.environ:  # launched through torch.distributed.run (also with a single rank)
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from cd360 import ops, routes, synth
    if args.fp8_attn:
        routes.set(fp8_attn=True)

    from cd360 import shard
    n_poses = args.poses or world
    mine = shard.assign_poses(n_poses, world, rank)  # indices of the target poses this rank samples (independent trajectories)
    assert len(mine) >= 1, "--poses must be >= --gpus"
    net = build_model(args.latent, args.refs, 50, dev, native_sampling=args.route != "sample_py")
    if args.route == "sample_py":
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import sample_py_stub
        sample_py_stub.register(net, [int(x) for x in torch.linspace(0, 50 - 50 / args.refs, args.refs)])  # sample.py:274-278
    elif args.route != "fused":
        from sgm.modules.attention import BasicTransformerBlock
        for m in net.modules():
            if isinstance(m, BasicTransformerBlock):
                (m if args.route == "hooked" else m.norm1).register_forward_hook(lambda mod, inp, out: None)
    jobs = []
    ppr = max(1, args.poses_per_replay)
    assert len(mine) % ppr == 0, "--poses-per-replay must divide the poses of every rank"
    for j0 in range(0, len(mine), ppr):  # one job = `ppr` target poses = one CFG-3 batch of 3 * ppr with their own latents; the 50 reference cameras are shared
        group = mine[j0:j0 + ppr]
        one = [synth.pose_batch(1, args.refs, seed=100 + pi, n_train=50)[0] for pi in group]
        pose = one * 3  # [null image x ppr | image x ppr | image + text x ppr]: the last two thirds are the SAME camera objects (de-duplicated render)
        g = torch.Generator(device=dev).manual_seed(7 + group[0])
        ctx = torch.randn(3 * ppr, 77, 2048, generator=g, device=dev).to(torch.bfloat16)
        y = torch.randn(3 * ppr, 2816, generator=g, device=dev).to(torch.bfloat16)
        jobs.append((pose, ctx, y, torch.randn(ppr, 4, args.latent, args.latent, generator=g, device=dev)))
    pose, ctx, y, x = jobs[0]
    pf = None
    if not args.no_graph and not args.no_weight_prefetch:
        from cd360.prefetch import WeightPrefetcher
        pf = WeightPrefetcher(dev, lag=args.prefetch_lag, wgs=args.prefetch_wgs, min_bytes=int(args.prefetch_min_mb * (1 << 20)))
    smp = Sampler(net, pose, ctx, y, args.traj, use_graph=not args.no_graph, prefetch=pf, graph_render=not os.environ.get("CD360_BENCH_EAGER_RENDER"))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed set-up, reported: the FeatureNeRF reference tables (first eager render) and the two hipGraph captures
    sync(); t0 = time.perf_counter(); smp.prepare(x); sync(); prepare_ms = (time.perf_counter() - t0) * 1e3
    xw = x.clone()
    for i in range(args.warmup):
        xw = smp.step(xw, i)
    # individually timed render / steady steps (reported in config, not the headline)
    sync(); t0 = time.perf_counter(); xw = smp.step(x.clone(), 0); sync(); render_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); xw = smp.step(xw, 1); sync(); steady_ms = (time.perf_counter() - t0) * 1e3

    # timed region: K steps of EVERY pose of this rank, one pose after the other (the graphs read the pose / conditioning of the job
    # through the sampler's static buffers: switching pose = pointing the sampler at the next job and re-rendering on its step 0)
    sync()
    t0 = time.perf_counter()
    finals = sample_assigned(smp, jobs, args.steps)  # cd360/job.py: retarget -> K steps, pose after pose
    sync()
    elapsed = time.perf_counter() - t0
    steps_done = args.steps * len(jobs)  # replays of this rank; each advances `ppr` poses by one denoise step
    if len(jobs) > 1:
        smp.retarget(*jobs[0][:3])

    # Per-kernel HIP-event timing: the SAME K steps replayed eagerly right after the timed region (events cannot bracket kernels
    # inside a hipGraph replay, and ~800 event records per step would otherwise sit inside the headline number).
    prof = {}
    if not args.no_profile and rank == 0:
        smp.use_graph = False
        ops.profile_start()
        xp = x.clone()
        for i in range(args.steps):
            xp = smp.step(xp, i)
        prof = ops.profile_stop()
        smp.use_graph = not args.no_graph

    # What a ONE-image job pays on top of the timed render step: the FeatureNeRF reference tables (Y, lv of the 51 distinct reference
    # images per pose block) are built once per job in Sampler.prepare(), outside the timed region.  Two eager render steps after the
    # timed region, the second with every block's tables dropped first, give their cost on this box.
    tables_ms = None
    if rank == 0 and not args.no_profile:
        from cd360 import sampling as _smp
        smp.use_graph = False
        sync_ = torch.cuda.synchronize
        smp.step(x.clone(), 0); sync_()
        t0 = time.perf_counter(); smp.step(x.clone(), 0); sync_(); r_plain = (time.perf_counter() - t0) * 1e3
        kept = [(blk, blk._ref_tables) for _, blk in _smp.pose_blocks(net)]
        for blk, _ in kept:
            blk._ref_tables = None
        t0 = time.perf_counter(); smp.step(x.clone(), 0); sync_(); r_tables = (time.perf_counter() - t0) * 1e3
        for blk, tab in kept:  # the captured graphs read the ORIGINAL table buffers: put them back
            blk._ref_tables = tab
        tables_ms = max(r_tables - r_plain, 0.0)
        smp.use_graph = not args.no_graph

    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    rank_ms = [elapsed / steps_done * 1e3]
    if world > 1:
        every = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(every, tmax)  # per-rank elapsed: the line shows the spread, `value` uses the slowest rank
        rank_ms = [float(t.item()) / steps_done * 1e3 for t in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gathered = shard.gather_latents(torch.cat(finals, 0), n_poses)  # the job's one exchange: final latents of every pose
        assert gathered.shape[0] == n_poses and torch.isfinite(gathered).all()
    elapsed = float(tmax.item())
    assert all(torch.isfinite(f).all() for f in finals)

    sum_steps = args.steps * n_poses  # pose-steps of the whole job
    if rank == 0:
        # ---- rooflines.  `achieved` = ALGORITHMIC flops (MFMA-bound kernels) or bytes (HBM-bound) of the launches / their HIP-event time on
        # the launch stream (ops._timed: SURVEY.md section 8d formulas, stated per kernel in DESIGN.md section 7b) ----
        MFMA_KERNELS = {"gemm8p", "qproj_attn", "qproj_attn_text", "attn_self", "conv_igemm", "nerf_mlp_aggregate"}
        try:
            from cd360 import _lib
            with open(_lib.LIB_PATH, "rb") as f:
                lib_bytes = f.read()
        except Exception:  # noqa: BLE001
            lib_bytes = b""

        def roofline_of(name):
            e = prof[name]
            if name in MFMA_KERNELS and e["flops"] > 0:
                ach = e["flops"] / (e["ms"] * 1e-3) / 1e12
                r = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TF, 4)}
            else:
                ach = e["bytes"] / (e["ms"] * 1e-3) / 1e9
                r = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
            r.update(traffic=None, launches=e["n"], avg_us=round(e["ms"] * 1e3 / e["n"], 2), alg_bytes_per_launch=round(e["bytes"] / e["n"]),
                     alg_flops_per_launch=round(e["flops"] / e["n"]))
            # HBM traffic per launch comes from the committed rocprofv3 PMC passes over this same command (profiles/*_pmc_traffic.json:
            # FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 1/2-FETCH correction applied)
            # -- and is reported only while every kernel symbol that pass measured still exists in the library that is running: a
            # traffic figure of a kernel that has since been rewritten is refused (traffic = null, `traffic_stale` says which symbol)
            try:
                import glob
                pm_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
                with open(pm_file) as f:
                    ent = json.load(f)["kernels"][name]
                gone = [sym for sym in ent.get("symbols", ["<no symbol list: profile predates the check>"]) if sym.encode() not in lib_bytes]
                if gone:
                    r["traffic_stale"] = f"{os.path.relpath(pm_file, ROOT)} measured {gone[0]}, which the loaded library no longer contains"
                else:
                    r["traffic"] = ent["hbm_bytes_per_launch"]
                    r["traffic_source"] = os.path.relpath(pm_file, ROOT)
            except Exception:  # noqa: BLE001
                r["traffic"] = None
            return r

        roof, roofs = None, {}
        if prof:
            roof = roofline_of(max(prof, key=lambda k: prof[k]["ms"]))  # the dominant kernel of the timed steps
            # the kernels north_star names: attention on MFMA, the FeatureNeRF render on HBM / VALU, and the whole step
            notes = {"gemm8p": "every Linear of the transformer blocks (cd360_gemm_bf16: LayerNorm fold / GEGLU / residual epilogues)",
                     "qproj_attn": "pose-token cross-attention A3, FUSED form: q projection + softmax(q k^T) v over 77 keys in one kernel; flops = 2 M C^2 + 4 M 77 C "
                                   "(the out projection runs as a gemm8p launch with the residual fused); render step only",
                     "qproj_attn_text": "text cross-attention A2 of every block on the same fused kernel (128 x 128 tile, mover waves): LayerNorm fold + "
                                        "q projection + softmax(q k^T) v over 77 keys, q never in HBM; flops = 2 M C^2 + 4 M 77 C",
                     "attn_self": "self-attention attn1 (tiled flash kernel), core form 4 B H N^2 64, projections in gemm8p",
                     "attn_smallk": "cross-attention core over <= 96 keys on the register-resident kernel (shapes the fused kernel does not serve)",
                     "nerf_mlp_aggregate": "FeatureNeRF gather + per-sample MLP + view softmax (A5-A8) after the algebraic restructuring, in two passes "
                                           "(geometry / view logits / softmax statistics once per (view, sample), then gather + 99-input MFMA slice + SiLU + "
                                           "weighted sum per 64-channel slice): full-line gathers staged through LDS; VALU-bound (75 % of the SIMD cycles: "
                                           "transcendentals of sin / cos and SiLU, the fp32 bilinear blend); flops = 2 M 99 C MFMA part only; render step only",
                     "volrender": "volume render scan (A10), HBM-bound; render step only",
                     "conv_igemm": "all 51 convolutions (implicit GEMM)"}
            for k in notes:
                if k in prof:
                    roofs[k] = roofline_of(k)
                    roofs[k]["what"] = notes[k]
            if "qproj_attn" in roofs:  # the north star's target kernel: say how far it is
                roofs["qproj_attn"]["target"] = 0.8
                roofs["qproj_attn"]["gap"] = round(0.8 - roofs["qproj_attn"]["frac"], 4)
                # What fraction of the MFMA peak the kernel's OWN structure allows if every MFMA it executes issued back to back with the softmax
                # VALU (7.1 per MFMA, half an MFMA's issue time) fully hidden: algorithmic FLOP / executed MFMA FLOP x occupied fraction of the tiles.
                # Executed per token and head: q K^T over 96 key slots (three 32-key blocks) and P V over 80 (five 16-key groups) for 77 keys; the
                # de-duplicated launch does 2 q-projections + 3 attentions; C = 640 is 2.5 tiles of 256 columns (a sixth of the third tile idles).
                def _ceil(C):
                    q, a_alg, a_exe = 2.0 * C * C, 4.0 * 77 * C, 2.0 * (96 + 80) * C
                    return (2 * q + 3 * a_alg) / (2 * q + 3 * a_exe) * (C / (256.0 * -(-C // 256)))
                roofs["qproj_attn"]["ceiling"] = {"C640": round(_ceil(640), 3), "C1280": round(_ceil(1280), 3),
                                                  "what": "algorithmic / executed MFMA FLOP x tile occupancy, softmax VALU assumed fully hidden; priced at the 2.4 GHz "
                                                          "peak (under MFMA load the chip sustains 1.9-2.0 GHz on random operands: x 0.8 on top)"}
            # whole steady-state step: 2.03e13 FLOP per CFG-3 step at 1024^2 (FlopCounterMode on the plain UNet)
            if args.latent == 128:
                roofs["steady_step"] = {"bound": "mfma", "achieved": round(ppr * 2.03e13 / (steady_ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TF,
                                        "unit": "TFLOP/s", "frac": round(ppr * 2.03e13 / (steady_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4),
                                        "what": "2.03e13 algorithmic FLOP of one CFG-3 UNet step / steady_step_ms (hipGraph replay wall time)"}
        out = {
            "metric": _baseline_metric() + ("" if ppr == 1 else f" [{ppr} poses batched per replay: NOT the headline configuration]"),
            "value": round(sum_steps / elapsed, 4), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps_done * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "sample.py 50-step sampling, SDXL UNet (random init), latent %d^2, CFG x3, %d ref views (synthetic ring cameras), "
                                   "%d target pose(s) over %d GPU(s); render on step 0 of each %d-step trajectory, cached afterwards.  Untimed, once per "
                                   "job in Sampler.prepare(): the FeatureNeRF reference tables Y = xref W1[:, :C]^T and lv (they depend on the "
                                   "`references` buffers and weights only; the reference recomputes the equivalent inside every render) and hipGraph "
                                   "capture.  Inside the timed render step, once per image: the text K / V projections (reused by the 49 cached steps)"
                                   % (args.latent, args.refs, n_poses, world, args.traj),
                       "render_step_ms": round(render_ms, 2), "steady_step_ms": round(steady_ms, 2), "prepare_ms": round(prepare_ms, 1),
                       **({"reference_tables_ms": round(tables_ms, 2), "render_step_ms_incl_tables": round(render_ms + tables_ms, 2),
                           "reference_tables_what": "the once-per-job FeatureNeRF reference tables (inside prepare_ms, outside the timed region): an eager "
                                                    "render step with every pose block's tables dropped minus the same step with them kept; a job of ONE "
                                                    "image pays render_step_ms_incl_tables for its first step"} if tables_ms is not None else {}),
                       **({"replay_note": f"one replay = one denoise step of {ppr} poses (CFG batch {3 * ppr}): value counts pose-steps, ms_per_step and "
                                          "render / steady_step_ms are per REPLAY"} if ppr > 1 else {}),
                       "prepare_ms_what": "Sampler.prepare(), once per job and outside every timed number: first eager render step (builds the reference "
                                          "tables Y / lv of the 51 distinct reference images for the 12 pose blocks, packs every block's weights) + capture "
                                          "of the steady-state and render hipGraphs; amortise it over the images of a job",
                       "attention_arith": "fp8 MFMA (e4m3 q / K / P / V, fp32 accumulate) in the fused cross-attentions" if args.fp8_attn else "bf16 MFMA",
                       "cfg_batch": 3 * ppr, "latent": args.latent,
                       "n_ref": args.refs, "poses": n_poses, "poses_per_gpu": len(mine), "poses_per_replay": ppr, "world_size": world,
                       "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                       **({"sanity_run": "CD360_BENCH_ONE_GPU: all ranks on GPU 0 over gloo -- not a measurement"} if one_gpu else {}),
                       "parallelism": "pose-dp%d" % world,
                       "hipgraph": not args.no_graph, "hipgraph_render_step": smp.rgraph is not None, "route": args.route,
                       "weight_prefetch": None if pf is None else {"lag": pf.lag, "wgs": pf.wgs, "what": "touch kernels of every GEMM-family launch's "
                                                                   "weights on a side branch of the captured graph (cd360/prefetch.py)"},
                       "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
                       "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                       "kernel_ms_per_step_source": "a separate EAGER replay of the same K steps after the timed region, HIP events around every launch "
                                                    "(events cannot bracket kernels inside a hipGraph replay): its sum exceeds ms_per_step by the "
                                                    "eager launch gaps the graph does not have"},
            "roofline": roof,
            "rooflines": roofs,
        }
        if args.fp8_attn:
            # BASELINE configs[4] tolerance report at workload level: the same render step (all 12 FeatureNeRF renders, pose-token attention
            # over 98 304 / 24 576 tokens per branch) and one cached step, launched eagerly, with the attention contractions in bf16 and in fp8
            from cd360 import sampling
            smp.use_graph = False

            def path_outputs(fp8):
                with routes.override(fp8_attn=fp8):
                    sampling.clear_rendered_feat(net)
                    e0 = smp.eps(x, 0)
                    rend = {n_: b_.rendered_feat.float().clone() for n_, b_ in sampling.pose_blocks(net)}
                    return e0, smp.eps(x, 1), rend

            relerr = lambda a_, b_: float((a_ - b_).abs().max() / b_.abs().max().clamp_min(1e-12))
            e0b, e1b, rb = path_outputs(False)
            e0f, e1f, rf = path_outputs(True)
            per_block = {n_: round(relerr(rf[n_], rb[n_]), 5) for n_ in rb}
            out["fp8_tolerance"] = {"rendered_feat_max_rel": max(per_block.values()), "rendered_feat_per_block": per_block,
                                    "eps_render_step_max_rel": round(relerr(e0f, e0b), 5), "eps_cached_step_max_rel": round(relerr(e1f, e1b), 5),
                                    "what": "max |fp8 - bf16| / max |bf16| of the 12 rendered feature maps (3 CFG branches each) and of the UNet output "
                                            "eps on the render step and on a cached step; same weights, inputs and kernels, only the arithmetic of the "
                                            "two attention contractions differs; against the fp32 oracle on a ray subset: "
                                            "tests/test_modules_gpu.py::test_cfgB_pose_block_render_fp8_attention_tolerance_report"}
            smp.use_graph = not args.no_graph
        if world == 1 and not args.no_train_step:
            # BASELINE configs[3] under the driver's eyes: one fine-tuning optimisation step at SDXL size (512^2 images, batch 4, 4 reference
            # views, trainkeys = pose: forward of both streams, four-term loss, backward, AdamW on fp32 masters) -- one warm + three timed
            # steps on the hand-written GEMM family (cd360_ms), then the same with every Linear on torch / hipBLASLt (library_ms)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_train
                del smp
                torch.cuda.empty_cache()
                graph_ = bench_train.measure(steps=5, warmup=1, batch=4, views=4, latent=64, profile=False, library=False, graph=True)
                libg_ = bench_train.measure(steps=5, warmup=1, batch=4, views=4, latent=64, profile=False, library=True, graph=True)
                mine_ = bench_train.measure(steps=3, warmup=1, batch=4, views=4, latent=64, profile=False, library=False)
                lib_ = bench_train.measure(steps=3, warmup=1, batch=4, views=4, latent=64, profile=False, library=True)
                out["train_step"] = {"ms": graph_["ms_per_step"], "bs": 4, "n_ref": 4, "latent": 64, "graph_ms": graph_["ms_per_step"],
                                     "library_graph_ms": libg_["ms_per_step"], "cd360_ms": mine_["ms_per_step"], "library_ms": lib_["ms_per_step"],
                                     "losses": graph_["losses"],
                                     "eager_losses": mine_["losses"], "peak_mem_gb": graph_["peak_mem_gb"],
                                     "what": "tools/bench_train.py shapes: main.py fine-tune step (train_co3d_concept.yaml), random-init SDXL UNet; "
                                             "graph_ms (= ms) = the whole step (forward, four-term loss, backward, AdamW on fp32 masters) captured once into "
                                             "a hipGraph and replayed (cd360.finetune.GraphedTrainStep), 1 warm + 5 timed replays; library_graph_ms = the same replay with "
                                             "every Linear on torch (hipBLASLt) -- the GPU-side comparison of the two GEMM families; cd360_ms = the same "
                                             "step launched eagerly (1 warm + 3 timed; host-bound: ~6400 launches), every Linear on cd360_gemm_bf16 / "
                                             "cd360_gemm_tn_bf16 (forward, dgrad, wgrad); library_ms = eager with the Linears on torch (hipBLASLt)"}
            except Exception as e:  # noqa: BLE001
                out["train_step"] = {"ms": None, "error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                # 32 threads: PyTorch's CPU kernels stop scaling (and regress badly) beyond one socket's worth of cores
                out["cpu_baseline"] = cpu_baseline(net, args.latent, min(os.cpu_count() or 1, 32))
            except Exception as e:  # noqa: BLE001  (e.g. not enough host RAM for the fp32 copy)
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()  # rank 0 may still be in its per-kernel event pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
