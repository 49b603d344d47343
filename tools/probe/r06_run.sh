#!/bin/bash
# scratch: run on the GPU box
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final_bench.json
python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']); print(d['cpu_baseline']['value'], d['train_step']['ms'])"
