#!/bin/bash
# Runs on the GPU box: same-call A/B of ENVIRONMENT settings of one library build (tools/ab_run.sh compares builds).
#   usage: tools/ab_env.sh OUT.log "ENV1=..,ENV2=.. | ENV3=.." "single qv .." [passes]      (settings separated by |, variables by ,)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; IFS='|' read -ra sets <<< "$2"; loads=$3; passes=${4:-2}
: > "$out"
for w in $loads; do
  for p in $(seq $passes); do
    for s in "${sets[@]}"; do
      s=$(echo $s | xargs)
      echo "== env: ${s:-default}" >> "$out"
      ( IFS=','; for kv in $s; do export "$(echo $kv | xargs)"; done; timeout -k 10 200 python3 tools/perf_probe.py 1000000 $w 3 >> "$out" 2>&1 ) || exit 1
    done
  done
done
grep -E "^== env|best" "$out"
