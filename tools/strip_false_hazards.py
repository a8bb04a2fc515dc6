#!/usr/bin/env python3
"""Removes from a gfx950 device listing the `s_nop 0` that directly follows an inline-asm block (`;;#ASMEND`) and precedes a
v_mad_u64_u32: the compiler's hazard recogniser treats every asm block as a possible partial-register (dst_sel) writer and puts one wait
state before a consumer of its outputs.  Our asm blocks are empty value fences and plain v_add_u32, which have no such hazard.
   usage: strip_false_hazards.py in.s out.s"""
import sys
lines = open(sys.argv[1]).read().split("\n")
out, removed = [], 0
for i, l in enumerate(lines):
    if l.strip() == "s_nop 0" and out and out[-1].strip() == ";;#ASMEND" and i + 1 < len(lines) and lines[i + 1].strip().startswith("v_mad_u64_u32"):
        removed += 1
        continue
    out.append(l)
open(sys.argv[2], "w").write("\n".join(out))
print(f"{sys.argv[1]}: removed {removed} wait states", file=sys.stderr)
