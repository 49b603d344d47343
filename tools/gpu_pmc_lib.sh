#!/bin/bash
# PMC passes over the LIBRARY GEMM (F.linear -> hipBLASLt) of one shape, to compare its counters with gemm8p's: tools/gpu_pmc_lib.sh TAG "M N K"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1; SHAPE=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lib_$TAG
mkdir -p $OUT
cd /tmp
cat > /tmp/libone.py <<'PY'
import sys, torch, torch.nn.functional as F
M, N, K = (int(v) for v in sys.argv[1:4])
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).cuda().bfloat16(); w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().bfloat16()
for _ in range(6): F.linear(a, w)
torch.cuda.synchronize()
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pl_$i -o m -- python /tmp/libone.py $SHAPE > $OUT/l_$i.log 2>&1
  CC=$(find /tmp/pl_$i -name "*counter_collection.csv" | head -1)
  python - "$CC" <<'PY' >> $OUT/summary.txt
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0])); meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "Cijk" not in n: continue
    a = agg[n[:110]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    meta[n[:110]] = {k: r.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
for k, d in agg.items():
    print("  ", k); print("      ", meta[k])
    for c, (n, v) in sorted(d.items()): print(f"      {c:34s} avg/dispatch {v / n:16.1f}  (n={n})")
PY
done
cat $OUT/summary.txt
