#!/bin/bash
# rocprofv3 kernel trace + stats of the fine-tune step (tools/bench_train.py, BASELINE config 4).  Outputs under gpurun_out/prof_$TAG.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-train}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -o train -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 3 --warmup 2 > $OUT/train_kt.log 2>&1; echo "kt exit $?"
find /tmp/prof_$TAG/kt -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
head -45 $OUT/*kernel_stats.csv | cut -c1-180
tail -1 $OUT/train_kt.log | cut -c1-400
