#!/usr/bin/env python3
"""Compiles eg_hip.hip and eg_gen.hip with -Rpass-analysis=kernel-resource-usage (no GPU needed) and writes the per-kernel table kept
under profiles/.   usage: resource_usage.py <round-tag, e.g. r02>"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rows = []
for tu in ("eg_hip", "eg_gen"):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c",
                            "-o", f"{d}/{tu}.o", str(ROOT / "elastic_elgamal_amd" / "csrc" / f"{tu}.hip")], capture_output=True, text=True)
    for m in re.finditer(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                         r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", r.stderr, re.S):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        name = name[5:] if name.startswith("void ") else name
        rows.append((name, *m.groups()[1:]))
seen, out = set(), []
out.append(f"# hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage -c eg_hip.hip eg_gen.hip   (tools/resource_usage.py {tag})")
out.append("# VGPR/AGPR per lane, scratch bytes per lane, waves per SIMD the allocation admits.  The dominant kernel k_eq_table<false, T> and the")
out.append("# table builder k_base_tables<T> (T = teeth of the comb) run without scratch; round 1's single equation kernel (k_msm_jobs) had 159 VGPR spills / 480 B.")
out.append(f"{'kernel':72s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'sgpr_spill':>10s} {'vgpr_spill':>10s} {'LDS':>7s}")
for row in rows:
    if row[0] in seen:
        continue
    seen.add(row[0])
    out.append(f"{row[0][:72]:72s} {row[1]:>5s} {row[2]:>5s} {row[3]:>5s} {row[4]:>8s} {row[5]:>4s} {row[6]:>10s} {row[7]:>10s} {row[8]:>7s}")
(ROOT / "profiles" / f"{tag}_kernel_resource_usage.txt").write_text("\n".join(out) + "\n")
print("\n".join(out[:12]))
