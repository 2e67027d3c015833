#!/bin/bash
# Round 5 final call on the round's last sources: full GPU suite, evidence (trace + counters + default line), the other configurations
R=$GRAFT_REPO_ROOT
cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05_gputests_tail.txt; cat gpurun_out/r05_gputests_tail.txt
bash profiles/scripts/r5_evidence.sh
bash profiles/scripts/r5_configs.sh
