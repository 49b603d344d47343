"""The bench line's contract, checked on the newest committed line of `python bench.py --steps 20 --warmup 5` (profiles/*_bench_20steps_warmup5.json:
what the driver parses) and on bench.py's argument defaults -- no GPU needed.  Guards the key set, the metric / unit of BASELINE.json, the
roofline arithmetic (frac = achieved / peak; peak from MI355X_MICROARCH.md: 2.5 PFLOP/s dense bf16, 8 TB/s HBM) and the cpu_baseline object."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_20steps_warmup5.json")))
    assert files, "no committed driver-style bench line"
    with open(files[-1]) as f:
        return json.loads(f.read()), files[-1]


def test_bench_line_has_the_contract_keys_and_baseline_metric():
    line, _ = _latest_line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    norm = lambda m: m.replace("\u00b2", "^2").split(", 1/2/4/8")[0]  # lines committed before bench.py quoted BASELINE.json verbatim wrote 1024^2
    assert norm(line["metric"]) == norm(base["metric"]) and line["unit"] == "steps/s"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"metric": _baseline_metric()' in src and 'json.load(f)["metric"]' in src  # the line quotes BASELINE.json's metric verbatim
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "bf16" and "synthetic" in line["data"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-2 * line["value"]
    assert "workload" in line["config"] and "model" not in line["config"]


def test_bench_line_rooflines_are_consistent():
    line, _ = _latest_line()
    roofs = [line["roofline"]] + list(line.get("rooflines", {}).values())
    for r in roofs:
        assert r["bound"] in ("hbm", "mfma") and r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
        assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 2500.0)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 and 0.0 < r["frac"] < 1.0
        if r.get("traffic") is not None and r.get("alg_bytes_per_launch"):
            assert r["traffic"] >= 0.9 * r["alg_bytes_per_launch"] or r["bound"] == "mfma"  # counter bytes are not below the algorithmic bytes
    dominant = line["roofline"]
    assert dominant["kernel"] in line["config"]["kernel_ms_per_step"]
    assert dominant["kernel"] == max(line["config"]["kernel_ms_per_step"], key=line["config"]["kernel_ms_per_step"].get)
    src = dominant.get("traffic_source")
    assert src is None or os.path.exists(os.path.join(ROOT, src))


def test_bench_line_cpu_baseline_and_train_step():
    line, _ = _latest_line()
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["unit"] == line["unit"] and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    ts = line["train_step"]
    assert ts["bs"] == 4 and ts["n_ref"] == 4 and ts["latent"] == 64 and ts["ms"] == ts["graph_ms"] and ts["cd360_ms"] > 0 and ts["library_ms"] > 0
    assert all(b <= a * 1.02 for a, b in zip(ts["losses"][2:], ts["losses"][3:])) or ts["losses"][-1] < ts["losses"][0]


def test_bench_defaults_finish_quickly():
    src = open(os.path.join(ROOT, "bench.py")).read()
    gpus = re.search(r'add_argument\("--gpus", type=int, default=(\d+)', src)
    steps = re.search(r'add_argument\("--steps", type=int, default=(\d+)', src)
    warm = re.search(r'add_argument\("--warmup", type=int, default=(\d+)', src)
    assert gpus and int(gpus.group(1)) == 1 and steps and int(steps.group(1)) <= 100 and warm and int(warm.group(1)) <= 10


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 8` with no WORLD_SIZE in the environment must not die on an assert: it re-executes itself as 8 ranks under
    torch.distributed.run (127.0.0.1 rendezvous), the form the driver uses for N > 1; under a launcher (WORLD_SIZE set) it runs as a rank."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    argv = bench.self_launch_argv(8, ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"], port=29517)
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29517"
    script = [a for a in argv if a.endswith("bench.py")]
    assert len(script) == 1 and os.path.isabs(script[0])
    assert argv[argv.index(script[0]) + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    port = int(bench.self_launch_argv(2, ["bench.py"])[bench.self_launch_argv(2, ["bench.py"]).index("--master-port") + 1])
    assert 1024 < port < 65536
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.gpus > 1 and "WORLD_SIZE" not in os.environ' in src and "os.execv(sys.executable, self_launch_argv(" in src
