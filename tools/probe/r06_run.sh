for r in 1 2; do
python tools/probe/gemm_sched_ab.py 2>&1 | tail -1
for d in 1 2 3 4; do CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_sched$d.so python tools/probe/gemm_sched_ab.py 2>&1 | tail -1; done
done
