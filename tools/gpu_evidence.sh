#!/bin/bash
# The evidence pass of a round, one box, one call: rocprofv3 kernel trace + PMC passes of the bench workload (tools/gpu_profile.sh), one
# steady denoise step by kernel (tools/gpu_steady_diff.sh), the fine-tune step by kernel -- eager and as the hipGraph replay
# (tools/gpu_steady_diff_train.sh), then the driver-style and the default bench lines.  tools/gpu_evidence.sh TAG
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r}
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh $TAG 6 > gpurun_out/evidence_$TAG.log 2>&1
bash tools/gpu_steady_diff.sh $TAG >> gpurun_out/evidence_$TAG.log 2>&1
bash tools/gpu_steady_diff_train.sh ${TAG}_eager 1 4 >> gpurun_out/evidence_$TAG.log 2>&1
TRAIN_ARGS=--graph bash tools/gpu_steady_diff_train.sh ${TAG}_graph 1 5 >> gpurun_out/evidence_$TAG.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_20.log 2>&1; echo "bench20 exit $?"; tail -1 gpurun_out/bench_${TAG}_20.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_${TAG}_default.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_${TAG}_default.log | cut -c1-300
grep -a "exit\|steady step\|fine-tune step" gpurun_out/evidence_$TAG.log
