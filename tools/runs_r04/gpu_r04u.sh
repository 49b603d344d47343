#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
TRAIN_ARGS=--graph bash tools/gpu_steady_diff_train.sh r04u_graph 1 5 > gpurun_out/r04u.log 2>&1
head -3 gpurun_out/steady_train_r04u_graph/steady_train_step.csv
