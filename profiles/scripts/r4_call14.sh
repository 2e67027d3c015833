#!/bin/bash
# Round 4, call 14: GPU suite at the final defaults; configs 3 and 4 with the engine's route memory (mixed-route fallback -> fp64 on the following calls).
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5
bash profiles/scripts/r4_configs.sh
