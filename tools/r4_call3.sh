#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
# larger batches: chunks of 2^20 become possible with the grouped workspace
for s in "EG_RING_GROUP=0" "EG_RING_GROUP=2,EG_CHUNK=1048576" "EG_RING_GROUP=3,EG_CHUNK=1048576" "EG_RING_GROUP=0" "EG_RING_GROUP=2,EG_CHUNK=1048576"; do
  echo "== 4M single: $s"; ( IFS=','; for kv in $s; do export "$kv"; done; timeout -k 10 200 python3 tools/perf_probe.py 4000000 single 3 2>&1 | grep best )
done
# larger elections: 40 options
for w in single multi; do for s in "EG_RING_GROUP=0" "EG_RING_GROUP=4" "EG_RING_GROUP=8" "EG_RING_GROUP=0" "EG_RING_GROUP=8"; do
  echo "== 40 options $w 400k: $s"; ( export $s EG_PROBE_OPTIONS=40; timeout -k 10 200 python3 tools/perf_probe.py 400000 $w 3 2>&1 | grep best )
done; done
