#!/bin/bash
# Runs on the GPU box (VERDICT r5 task 2): the FP64-limb field multiplication against the shipped integer one - bursts with the correctness
# check, then every case sustained for SECONDS with the package power sampled beside it (hwmon power1_input, every 0.2 s, unix time stamps;
# the bench prints the time span of every case's second half, and the mean power over each span is appended).
#   usage: tools/fp64_probe.sh OUT.txt [seconds]
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; secs=${2:-4}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I elastic_elgamal_amd/csrc -o tools/ubench/fp64_bench tools/ubench/fp64_bench.hip || exit 1
{ echo "# tools/ubench/fp64_bench (bursts)"; timeout -k 10 300 tools/ubench/fp64_bench; } > "$out" 2>&1 || exit 1
# the box shows the hwmon files of every card of the host: sample the one of the device under test (by its PCI address)
pci=$(tools/ubench/fp64_bench pci)
pw=$(ls /sys/bus/pci/devices/$pci/hwmon/hwmon*/power1_input 2>/dev/null | head -1)
echo "# power file: $pw" >> "$out"
( while true; do echo "$(date +%s.%N) $(cat $pw 2>/dev/null)"; sleep 0.2; done ) > "$out.power" &
sampler=$!
{ echo; echo "# tools/ubench/fp64_bench sustained $secs"; timeout -k 10 600 tools/ubench/fp64_bench sustained $secs; } >> "$out" 2>&1
rc=$?
kill $sampler 2>/dev/null
python3 - "$out" <<'PY'
import sys
out = sys.argv[1]
pw = [tuple(map(float, l.split())) for l in open(out + ".power") if len(l.split()) == 2]
lines = open(out).read().splitlines()
res = ["", "# mean package power over the second half of every sustained case (hwmon power1_input, W)"]
for l in lines:
    f = l.split("|")
    if len(f) == 6 and f[5].split() and f[5].split()[0].replace(".", "").isdigit():
        a, b = map(float, f[5].split())
        w = [p / 1e6 for t, p in pw if a <= t <= b]
        if w:
            res.append("%s | %s | w/SIMD %s | %7.1f W over %d samples" % (f[0].strip(), f[1].strip(), f[2].strip(), sum(w) / len(w), len(w)))
open(out, "a").write("\n".join(res) + "\n")
PY
rm -f "$out.power"
exit $rc
