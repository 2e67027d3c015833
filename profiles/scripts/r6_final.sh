#!/bin/bash
# Round 6 final call on the round's last sources: full GPU suite, evidence (trace + counters + default line with the CPU baseline), the other
# configurations and batch sizes.
R=$GRAFT_REPO_ROOT
cd $R
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r06_gputests_tail.txt; cat gpurun_out/r06_gputests_tail.txt
bash profiles/scripts/r6_evidence.sh
bash profiles/scripts/r6_configs.sh
timeout 300 python bench.py --precision native --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_native.json 2>/dev/null
timeout 300 python bench.py --eig-route fp64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_fp64route.json 2>/dev/null
python - <<'PY'
import json
for f in ("r06_bench_native", "r06_bench_fp64route"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1]); print(f, round(d["value"], 3), d["unit"], round(d["ms_per_step"], 1), "ms")
    except Exception as e:
        print(f, "FAILED", e)
PY
