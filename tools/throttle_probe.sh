#!/bin/bash
# Runs on the GPU box: what holds the shader clock below its 2.4 GHz while the verifier runs?  Samples amd-smi's power, per-XCD clocks,
# temperatures and throttle status during back-to-back verification steps.   usage: tools/throttle_probe.sh OUT.txt [workload] [iters]
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; w=${2:-single}; iters=${3:-250}
: > "$out"
timeout -k 10 280 python3 tools/perf_probe.py 1000000 $w $iters > "$out.probe" 2>&1 &
pid=$!
for i in $(seq 40); do grep -q "iter 5:" "$out.probe" 2>/dev/null && break; sleep 0.5; done
for i in $(seq 6); do
  kill -0 $pid 2>/dev/null || break
  { echo "== busy sample $i"; amd-smi metric -g 0 --usage --power --clock --temperature 2>&1 | grep -v "JPEG\|VCN\|VCLK\|DCLK\|DEEP_SLEEP\|CLK_LOCKED\|MIN_CLK\|MAX_CLK" | head -60; } >> "$out"
  sleep 0.7
done
wait $pid
tail -n 2 "$out.probe" >> "$out"
