#!/bin/bash
# Runs on the GPU box: samples power, clocks and temperature (rocm-smi) while the verifier runs back-to-back steps, to tell an issue-bound
# kernel from a power-capped clock.   usage: tools/clock_probe.sh OUT.txt [workload] [iters]
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; w=${2:-single}; iters=${3:-40}
{ echo "== idle"; rocm-smi --showpower --showclocks --showtemp --showmaxpower --showperflevel 2>&1 | grep -v "^$\|=====\|WARNING" ; } > "$out"
timeout -k 10 280 python3 tools/perf_probe.py 1000000 $w $iters > "$out.probe" 2>&1 &
pid=$!
sleep 12     # import, tables, generator
for i in $(seq 12); do
  kill -0 $pid 2>/dev/null || break
  { echo "== sample $i"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|junction\|edge\|hotspot" ; } >> "$out"
  sleep 0.4
done
wait $pid
tail -n 3 "$out.probe" >> "$out"
cat "$out"
