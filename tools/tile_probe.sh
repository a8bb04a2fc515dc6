#!/bin/bash
# Runs on the GPU box: the "tile pipeline" experiment of VERDICT r4 task 3 with the shipped engine.  A chunk IS a tile: its ballots go
# tables -> stage 1 -> encode + hash -> stage 2 -> encode + hash before the next chunk starts on the same work set, and the table
# addresses are reused chunk after chunk.  EG_CHUNK sets the tile size; two work sets, so the live tables are 2 x EG_CHUNK x 24 KiB
# (config A): 2560 -> 126 MB (under half of the 256 MB Infinity Cache), 5120 -> 252 MB, ... 524288 (default) -> 25.8 GB.
#   usage: tools/tile_probe.sh OUT.txt
# For every tile size: throughput (tools/perf_probe.py, 1 M single-choice ballots, best of 4), socket power and clocks sampled in the
# middle of 60 back-to-back steps, and - separate passes - the memory-side bytes (FETCH_SIZE x2 gfx950 + WRITE_SIZE, summed over every
# kernel of one step) per ballot.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; : > "$out"
export EG_COMB_BIG_MIN=1
for chunk in 2560 5120 10240 20480 65536 524288; do
  echo "=== EG_CHUNK=$chunk (live tables of two work sets: $((chunk * 2 * 24 / 1024)) MB)" >> "$out"
  EG_CHUNK=$chunk timeout -k 10 120 python3 tools/perf_probe.py 1000000 single 4 2>&1 | tail -n 1 >> "$out" || exit 1
  EG_CHUNK=$chunk timeout -k 10 200 python3 tools/perf_probe.py 1000000 single 60 > "$out.probe" 2>&1 &
  pid=$!
  for i in $(seq 60); do grep -q "iter 8:" "$out.probe" 2>/dev/null && break; sleep 0.5; done
  for i in 1 2 3; do
    kill -0 $pid 2>/dev/null || break
    amd-smi metric -g 0 --power --clock 2>&1 | grep -E "SOCKET_POWER|CLK:|GFX_0|GFX_4" | grep -v "MIN_CLK\|MAX_CLK\|LOCKED\|DEEP" | head -8 | tr -s ' ' | tr '\n' ';' >> "$out"; echo >> "$out"
    sleep 0.5
  done
  wait $pid
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/tile_${c}_$chunk
    EG_CHUNK=$chunk timeout -k 10 300 rocprofv3 --pmc $c -d gpurun_out/tile_${c}_$chunk -o pmc --output-format csv -- \
      python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --no-extra-configs --ballots 1000000 > gpurun_out/tile_${c}_$chunk.log 2>&1 || exit 1
  done
  python3 - "$chunk" >> "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
chunk = sys.argv[1]
tot = defaultdict(float); per = defaultdict(lambda: defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/tile_{c}_{chunk}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                v = float(r["Counter_Value"]); tot[c] += v; per[r["Kernel_Name"].split("(")[0][:60]][c] += v
b = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e9
print(f"  memory side, one step of 1 M ballots: FETCH_SIZE x2 = {2*tot['FETCH_SIZE']*1024/1e9:.1f} GB, WRITE_SIZE = {tot['WRITE_SIZE']*1024/1e9:.1f} GB -> {b:.1f} KB per ballot")
for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE']))[:4]:
    print(f"    {k}: {(2*v['FETCH_SIZE']+v['WRITE_SIZE'])*1024/1e9:.1f} KB per ballot (fetch x2 {2*v['FETCH_SIZE']*1024/1e9:.1f}, write {v['WRITE_SIZE']*1024/1e9:.1f})")
PY
done
cat "$out"
