"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries tools/gpu_profile.sh wrote (pmc_FETCH_SIZE.csv,
pmc_WRITE_SIZE.csv: per-kernel averages in KB).  Kernel variants (template instantiations) are merged launch-weighted under the
name bench.py uses for its per-kernel timing.  Usage: make_traffic_json.py gpurun_out/prof_r1f profiles/r01f"""
import csv
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
NAMES = {"conv_igemm_kernel": "conv_igemm", "attn_fwd_kernel": "attn_self", "attn_self_kernel": "attn_self", "attn_smallk_kernel": "attn_smallk", "nerf_fused_kernel": "nerf_mlp_aggregate",
         "nerf_fused_line_kernel": "nerf_mlp_aggregate", "nerf_fused_rec_kernel": "nerf_mlp_aggregate", "nerf_geom_kernel": "nerf_mlp_aggregate",
         "gemm_mfma_kernel": "gemm8p", "row_stats_kernel": "row_stats",
         "geglu_kernel": "geglu", "volrender_kernel": "volrender", "gn_partial_kernel": "gn_silu", "gn_apply_kernel": "gn_silu", "gn_finalize_kernel": "gn_silu"}


def symbol_fragment(demangled: str) -> str:
    """Itanium-mangled fragment (name + integer / bool template arguments) of a kernel name as rocprofv3 prints it, e.g.
    `void (anonymous namespace)::gemm_mfma_kernel<2, 4, 1, 2, 4, 1, 4, 0>(...)` -> `16gemm_mfma_kernelILi2ELi4ELi1ELi2ELi4ELi1ELi4ELi0EE`:
    bench.py looks for it in the library it is running before it quotes this file's traffic for the kernel."""
    import re
    m = re.search(r"(\w+_kernel)(?:<([^>]*)>)?", demangled)
    name, targs = m.group(1), m.group(2)
    frag = f"{len(name)}{name}"
    if targs:
        parts = []
        for t in (x.strip() for x in targs.split(",")):
            parts.append({"true": "Lb1E", "false": "Lb0E"}.get(t, f"Li{t}E" if not t.startswith("-") else f"Lin{t[1:]}E"))
        frag += "I" + "".join(parts) + "E"
    return frag


def load(counter):
    out = {}
    for r in csv.DictReader(open(f"{src}/pmc_{counter}.csv")):
        key = next((v for k, v in NAMES.items() if k in r["kernel"]), None)
        if key is None:
            continue
        if key == "gemm8p":  # gemm_mfma_kernel<WM, WN, NCB, NMB, NBUF, KS, MV, EPI>: EPI 2-4 = fused q-projection + attention (256 x 256 tile: the
            import re        # pose tokens A3; 128 x 128 tile = <4, 2, 2, 1, ...>: the text cross-attention A2), 5 = 3x3 convolution, 6 = Linear + GN statistics
            m = re.search(r"gemm_mfma_kernel<([^>]*)>", r["kernel"])
            targs = [int(t) for t in m.group(1).split(",")] if m else []
            if targs and targs[-1] in (5, 11):  # (11: the halo form of the 3 x 3 convolution)
                key = "conv_igemm"
            elif targs and (2 <= targs[-1] <= 4 or 7 <= targs[-1] <= 10):  # attention epilogues (7-9: odd 16-key groups, 10: fp8)
                key = "qproj_attn" if targs[:4] == [2, 4, 2, 4] else "qproj_attn_text"  # only the pose tokens take the 256 x 256 tile
        d = out.setdefault(key, {"n": 0, "kb": 0.0, "symbols": set()})
        if "nerf_geom_kernel" not in r["kernel"]:  # (pass 1 of the two-pass render: its bytes count, its dispatch is the same launch as pass 2's)
            d["n"] += int(r["dispatches"])
        d["kb"] += float(r["total"])
        d["symbols"].add(symbol_fragment(r["kernel"]))
    return out


f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
kern = {}
for k in f:
    n = f[k]["n"]
    fetch, write = f[k]["kb"] / n, w.get(k, {"kb": 0.0})["kb"] / n
    kern[k] = {"fetch_size_kb_raw": round(fetch, 1), "write_size_kb": round(write, 1), "hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "dispatches": n,
               "symbols": sorted(f[k]["symbols"])}
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile "
                 "--no-graph` (tools/gpu_profile.sh), " + dst + "_pmc_*.csv",
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md HBM section; confirmed on geglu_kernel: "
                     "35141 KB raw vs 71.8 MB actually read; WRITE_SIZE exact) -> bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.  The counters sit on the "
                     "L2 <-> fabric side, so Infinity-Cache hits are included (upper bound on HBM traffic).  gn_silu / attn_fwd sum their sub-kernels "
                     "per launch of any of them.",
       "kernels": kern}
json.dump(doc, open(dst + "_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in kern.items()}))
