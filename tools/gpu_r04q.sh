#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 600 python tools/probe/op_census.py --top 70 > gpurun_out/r04q_census.log 2>&1; echo "census exit $?"; sed -n '/autograd> ops by/,$p' gpurun_out/r04q_census.log | head -80
