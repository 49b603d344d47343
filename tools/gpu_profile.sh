#!/bin/bash
# rocprofv3 kernel trace of the bench workload + (optionally) the gpu tests.  Outputs under gpurun_out/.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
TAG=${1:-r1}
STEPS=${2:-6}
if [ "$3" == "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
fi
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$TAG
find /tmp/prof_$TAG -name "*stats*.csv" -exec cp {} gpurun_out/prof_$TAG/ \;
TRACE=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $TRACE 80 > gpurun_out/prof_$TAG/kernel_by_shape.csv
ls -la gpurun_out/prof_$TAG
head -45 gpurun_out/prof_$TAG/kernel_by_shape.csv
tail -3 gpurun_out/bench_prof_$TAG.log
if [ "$4" == "bench" ]; then timeout 900 python bench.py --steps 50 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1; echo "bench exit $?"; tail -2 gpurun_out/bench_$TAG.log; fi
