#!/bin/bash
# HIP runtime API trace of one bench step next to the kernel trace (no counters): which host calls take the time while the QR phase runs --
# does the one host thread that drives the iteration groups block in a memcpy / event call?   usage: profiles/scripts/api_trace.sh <tag>
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/api_$TAG
rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d /tmp/api_$TAG -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_api_bench.json 2> $R/gpurun_out/${TAG}_api_stderr.log
F=$(find /tmp/api_$TAG -name "*hip_api_stats.csv" | head -1)
[ -n "$F" ] && head -25 $F > $R/gpurun_out/${TAG}_hip_api_stats.csv
# the longest individual calls: name, duration
T=$(find /tmp/api_$TAG -name "*hip_api_trace.csv" | head -1)
[ -n "$T" ] && python - "$T" > $R/gpurun_out/${TAG}_hip_api_longest.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows.sort(key=dur, reverse=True)
print("# longest HIP runtime calls of the run (us)")
for r in rows[:40]:
    print("%10.1f  %s" % (dur(r) / 1e3, r["Function"]))
agg = collections.Counter()
for r in rows:
    if dur(r) > 1e6:
        agg[r["Function"]] += dur(r)
print("# calls longer than 1 ms, summed per function (ms)")
for k, v in agg.most_common(12):
    print("%10.1f  %s" % (v / 1e6, k))
PY
cat $R/gpurun_out/${TAG}_hip_api_stats.csv | cut -c1-160 | head -14
head -30 $R/gpurun_out/${TAG}_hip_api_longest.txt
