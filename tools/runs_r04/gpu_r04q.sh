#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 600 python tools/probe/op_census.py --top 70 > gpurun_out/r04q_census.log 2>&1; echo "census exit $?"; grep -v "nerf.py:1[67][0-9]" gpurun_out/r04q_census.log | head -150
