#!/bin/bash
# Round 5, last GPU call: the committed tree once more (GPU suite, smoke, default bench line), the achieved error of precision="native",
# and -- for the record, defaults unchanged -- the AED window size after the super-step restructure (batch 16 and 128).
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
(timeout 200 python -m pytest tests/test_fullsize_properties.py -m gpu -q -s -k native_c64 2>&1 | grep -a "rel err\|passed\|failed") > $O/r5v_native_c64.txt; cat $O/r5v_native_c64.txt
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/r5v_gputests_tail.txt; cat $O/r5v_gputests_tail.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/r5v_smoke.txt; cat $O/r5v_smoke.txt
timeout 400 python bench.py > $O/r5v_bench_default.json 2> $O/r5v_bench_default.err; cat $O/r5v_bench_default.json | cut -c1-400
for b in 16 128; do
  for aed in 32 48 64; do
    echo "== batch $b TRX_QR_AED=$aed" >> $O/r5v_aed.txt
    TRX_QR_AED=$aed timeout 200 python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d.get('parity_sample'))" >> $O/r5v_aed.txt
  done
done
cat $O/r5v_aed.txt
