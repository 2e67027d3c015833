#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for cfg in 4 0 1 2; do
  echo "== TRX_GEMM_BIG=$cfg"
  TRX_GEMM_BIG=$cfg timeout 300 python -m pytest tests/test_blocks.py -q -m gpu -k "lu_row_split_panel or lu_solve" 2>&1 | grep -v "^$" | tail -25
done
