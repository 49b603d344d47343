#!/bin/bash
# end-of-iteration visit: full gpu tests, smoke, distributed-launch sanity (1 rank), headline bench
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -6 gpurun_out/smoke.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1; echo "dist1 exit $?"; tail -2 gpurun_out/bench_dist1.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_default.log
