#!/bin/bash
# round-6 evidence pass, one box: rocprofv3 kernel trace + PMC passes (tools/gpu_profile.sh), ONE steady step by kernel -- eager AND as the
# hipGraph replay the bench times --, the fine-tune step by kernel as the graph replay, SQ counters of the fused attention, bench lines
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r06e}
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$TAG.log; grep -a "passed\|failed\|pytest exit" gpurun_out/pytest_$TAG.log | tail -5
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
timeout 900 python bench.py --steps 20 --warmup 5 --route sample_py --no-train-step --no-cpu-baseline > gpurun_out/bench_${TAG}_sample_py.log 2>&1; echo "bench sample_py exit $?"
timeout 900 python bench.py --steps 20 --warmup 5 --route module --no-train-step --no-cpu-baseline --no-profile > gpurun_out/bench_${TAG}_module.log 2>&1; echo "bench module exit $?"
for f in 20 default fp8 ppr2 sample_py module; do tail -1 gpurun_out/bench_${TAG}_$f.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d['config']; print('$f: %.2f steps/s steady %.2f render %.2f train %s' % (d['value'], c['steady_step_ms'], c['render_step_ms'], d.get('train_step',{}).get('ms')))
except Exception as e: print('$f: no line', e)"; done
# round 6 additions: SQ counters of the two-pass render's pass 2, the A/B switches of the round on this box
bash tools/gpu_pmc_kernel.sh ${TAG}_nerf nerf_fused "python tools/bench_kernels.py nerf1" "-" > /dev/null 2>&1
b() { python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['steady_step_ms'], d['config']['render_step_ms'])"; }
for i in 1 2; do b default; CD360_NO_STAGE=1 b no_stage; CD360_CONV_HALO=0 b no_halo; CD360_NERF_KERNEL=1 b nerf_one_pass; CD360_NO_OUT_CONV4=1 b no_out_conv4; CD360_GEMM_ASM4=1 b asm4_everywhere; [ -f $GRAFT_REPO_ROOT/custom-diffusion360_amd/lib/libcd360_old.so ] && CD360_LIB=$GRAFT_REPO_ROOT/custom-diffusion360_amd/lib/libcd360_old.so b epilogues_before; done > gpurun_out/ab_$TAG.txt 2>&1
cat gpurun_out/ab_$TAG.txt
python tools/probe/conv_halo_ab.py 2>&1 | grep "^conv" > gpurun_out/conv_halo_ab_$TAG.txt
