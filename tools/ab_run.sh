#!/bin/bash
# Runs on the GPU box: same-call A/B of library builds (boxes of the pool differ by +-3 %, so only same-call numbers compare).
#   usage: tools/ab_run.sh OUT.log "lib1 lib2 .." "single qv .." [passes]
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; libs=$2; loads=$3; passes=${4:-2}
: > "$out"
for w in $loads; do
  for p in $(seq $passes); do
    for l in $libs; do
      EG_LIB=$l timeout -k 10 200 python3 tools/perf_probe.py 1000000 $w 3 >> "$out" 2>&1 || exit 1
    done
  done
done
cat "$out"
