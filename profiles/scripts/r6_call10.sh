#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call10.txt
: > $O
timeout 900 python tests/gpu_lu_sub.py >> $O 2>>gpurun_out/r6_call10.err
cat $O | cut -c1-300
