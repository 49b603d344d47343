#!/bin/bash
# round-4 evidence pass, one box: rocprofv3 kernel trace + PMC passes (tools/gpu_profile.sh), ONE steady step by kernel -- eager AND as the
# hipGraph replay the bench times --, the fine-tune step by kernel as the graph replay, SQ counters of the fused attention, bench lines
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r04a}
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh $TAG 6 > gpurun_out/evidence_$TAG.log 2>&1
bash tools/gpu_steady_diff.sh $TAG >> gpurun_out/evidence_$TAG.log 2>&1
STEADY_NOGRAPH=" " bash tools/gpu_steady_diff.sh ${TAG}_graph >> gpurun_out/evidence_$TAG.log 2>&1
TRAIN_ARGS=--graph bash tools/gpu_steady_diff_train.sh ${TAG}_graph 1 5 >> gpurun_out/evidence_$TAG.log 2>&1
bash tools/gpu_pmc_qattn.sh $TAG - > /dev/null 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_20.log 2>&1; echo "bench20 exit $?"; tail -1 gpurun_out/bench_${TAG}_20.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench_${TAG}_default.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_${TAG}_default.log | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 --fp8-attn --no-train-step --no-cpu-baseline > gpurun_out/bench_${TAG}_fp8.log 2>&1; echo "bench fp8 exit $?"
timeout 900 python bench.py --steps 20 --warmup 5 --poses 2 --poses-per-replay 2 --no-train-step --no-cpu-baseline > gpurun_out/bench_${TAG}_ppr2.log 2>&1; echo "bench ppr2 exit $?"
grep -a "exit\|steady step\|fine-tune step" gpurun_out/evidence_$TAG.log
