#!/bin/bash
# Build libtrx.so at a list of commits (in throw-away worktrees, cross-compiled: no GPU needed) into profiles/_ab_libs/NN_<label>.so, so that
# ONE GPU call can measure the increments of a chain of changes (profiles/scripts/ab_prebuilt.sh).  The .so files are git-ignored and travel
# with gpurun.   usage: bash profiles/scripts/prebuild_libs.sh <label>=<commit> [<label>=<commit> ...]
set -e
ROOT=$(git rev-parse --show-toplevel)
OUT=$ROOT/profiles/_ab_libs
mkdir -p $OUT
i=0
for spec in "$@"; do
  label=${spec%%=*}; commit=${spec#*=}
  wt=/tmp/trx_wt_$$_$i
  git worktree add -q --detach $wt $commit
  (cd $wt && python torcwa_amd/csrc/build.py > /dev/null)
  cp $wt/torcwa_amd/libtrx.so $OUT/$(printf "%02d" $i)_$label.so
  git worktree remove --force $wt
  echo "$(printf "%02d" $i)_$label.so  <- $(git log -1 --format='%h %s' $commit | cut -c1-100)"
  i=$((i+1))
done
