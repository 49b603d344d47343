#!/bin/bash
# scratch: run on the GPU box
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_r06i.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_r06i.log; grep -a "passed\|failed\|pytest exit" gpurun_out/pytest_r06i.log | tail -3
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_r06i_20.json; python -c "
import json; d=json.load(open('gpurun_out/bench_r06i_20.json')); print(d['value'], d['ms_per_step'], d['config']['steady_step_ms'], d['config']['render_step_ms'], d['roofline']['frac'])"
