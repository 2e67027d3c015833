#!/bin/bash
# Kernel trace of the bench command (steady state: 1 warm-up + 3 timed steps) and the tables derived from it.
#   usage: profiles/scripts/trace_bench.sh <tag> [bench flags...]     -> gpurun_out/<tag>_{kernel_stats,phases,gaps}.txt, <tag>_stats.csv
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$TAG
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/tr_$TAG -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_stderr.log
DB=$(find /tmp/tr_$TAG -name "*.db" | head -1)
python $R/profiles/kernel_stats.py $DB 45 > $R/gpurun_out/${TAG}_kernel_stats.txt
python $R/profiles/kernel_stats.py $DB --json $R/gpurun_out/${TAG}_kernel_profile.json 128 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
python $R/profiles/kernel_stats.py $DB --phases > $R/gpurun_out/${TAG}_phases.txt
python $R/profiles/kernel_stats.py $DB --gaps-frac 0.55 > $R/gpurun_out/${TAG}_gaps.txt
python $R/profiles/kernel_stats.py $DB --concurrency > $R/gpurun_out/${TAG}_concurrency.txt 2>&1
python $R/profiles/kernel_stats.py $DB --excerpt 0.55 > $R/gpurun_out/${TAG}_excerpt.txt 2>&1
python $R/profiles/kernel_stats.py $DB --lanes 0.55 40 > $R/gpurun_out/${TAG}_lanes.txt 2>&1
python $R/profiles/kernel_stats.py $DB --lanes 0.3 40 >> $R/gpurun_out/${TAG}_lanes.txt 2>&1
python $R/profiles/kernel_stats.py $DB --by-grid gemm_mfma 40 > $R/gpurun_out/${TAG}_gemm_by_shape.txt
CSV=$(find /tmp/tr_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$CSV" ] && head -45 $CSV > $R/gpurun_out/${TAG}_stats.csv
tail -3 $R/gpurun_out/${TAG}_stderr.log
