#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only), priced with the issue costs
measured by tools/ubench/valu_rates.hip AT THE OCCUPANCY THE KERNEL RUNS AT (profiles/r03_ubench_valu_rates.txt; cycles per wave64
instruction and SIMD at 1 / 2 / 4 / 8 waves per SIMD: plain 32-bit VOP1/VOP2 4.55 / 2.30 / 2.26 / 2.20, v_mad_u64_u32 5.69 / 4.70 / 4.36 /
4.35, 64-bit shifts and adds 5.16 / 4.52 / 4.27 / 4.27, every other VALU 5.05 / 4.41 / 4.23 / 4.12; 3 waves = mean of 2 and 4).
A select (v_cndmask_b32 on vcc) is priced as a plain VOP2: that is what it costs between other instructions (r04_ubench_cndmask.txt).
   usage: isa_mix.py <listing.s> <kernel name substring> [min loop length] [waves per SIMD, default 3]"""
import collections
import re
import subprocess
import sys

listing, want = sys.argv[1], sys.argv[2]
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 200
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 3
PRICES = {1: (4.55, 5.69, 5.16, 5.05), 2: (2.30, 4.70, 4.52, 4.41), 4: (2.26, 4.36, 4.27, 4.23), 8: (2.20, 4.35, 4.27, 4.12)}
PRICES[3] = tuple((a + b) / 2 for a, b in zip(PRICES[2], PRICES[4]))
P_PLAIN, P_MAD, P_64, P_OTHER = PRICES[waves if waves in PRICES else (8 if waves > 4 else 4)]
WIDE = {"v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64", "v_ashrrev_i64", "v_mov_b64"}
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32", "v_ashrrev_i32", "v_not_b32",
         "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_cndmask_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}


def cost(op, text):
    if not op.startswith("v_"):
        return 0.0
    base = op.replace("_e32", "").replace("_e64", "")
    if base == "v_mad_u64_u32":
        return P_MAD
    if base in WIDE:
        return P_64
    if base in CHEAP and not op.endswith("_e64") and "v_cndmask_b32_e64" not in op:
        return P_PLAIN
    return P_OTHER


lines = open(listing).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.startswith("_Z") and l.rstrip().split(":")[0] and ":" in l:
        name = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout
        if want in name and start is None:
            start = i
        elif start is not None:
            end = i
            break
body = lines[start:end]
labels = {}
ins = []     # (op, text)
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if re.match(r"^[a-z_0-9]+$", op):
        ins.append((op, t))


def summarize(a, b, title):
    h = collections.Counter()
    cyc = collections.Counter()
    for op, t in ins[a:b]:
        base = op.replace("_e32", "").replace("_e64", "")
        h[base] += 1
        cyc[base] += cost(op, t)
    total = sum(cyc.values())
    n_valu = sum(v for k, v in h.items() if k.startswith("v_"))
    print(f"== {title}: {b - a} instructions ({n_valu} VALU), VALU issue estimate {total:.0f} cycles per wave at {waves} waves per SIMD "
          f"= {total / max(n_valu, 1):.3f} per VALU instruction")
    for k, v in sorted(cyc.items(), key=lambda kv: -kv[1])[:18]:
        print(f"   {k:24s} {h[k]:6d}  {v:9.0f} cyc  {100 * v / max(total, 1):5.1f} %")
    other = [(k, h[k]) for k in h if cyc[k] == 0]
    print("   non-VALU:", ", ".join(f"{k}×{n}" for k, n in sorted(other, key=lambda kv: -kv[1])[:12]))


summarize(0, len(ins), "whole kernel (static)")
for i, (op, t) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] <= i and i - labels[tgt] >= min_len:
            summarize(labels[tgt], i + 1, f"loop {tgt}")
