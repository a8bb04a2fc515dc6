#!/usr/bin/env python3
"""Turns the rocprofv3 outputs written by tools/profile_round.sh (under gpurun_out/) into the text summaries kept
under profiles/.   usage: profile_summary.py <round-tag, e.g. r01>"""
import csv, sys, collections
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    return name if len(name) <= 64 else name[:64]


# ---- kernel stats ------------------------------------------------------------------------------------------------
rows = list(csv.DictReader(open(OUT / "prof_stats" / "stats_kernel_stats.csv")))
lines = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline   (MI355X)",
         "# 1 warm-up + 3 timed steps of 1M single-choice 5-option ballots (one chunk: one k_base_tables and two k_msm_jobs launches per step), plus the untimed generator launch",
         f"{'kernel':64s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>11s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>8s}"]
for r in rows:
    lines.append(f"{short(r['Name']):64s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:12.3f} {float(r['AverageNs'])/1e6:11.4f} "
                 f"{int(r['MinNs'])/1e6:9.3f} {int(r['MaxNs'])/1e6:9.3f} {float(r['Percentage']):8.4g}")
(ROOT / "profiles" / f"{tag}_bench_kernel_stats.txt").write_text("\n".join(lines) + "\n")

# ---- PMC passes ----------------------------------------------------------------------------------------------------
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ1", "pmc_SQ2"):
    f = OUT / d / "pmc_counter_collection.csv"
    if not f.exists():
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE":
            launches[k].add(r["Dispatch_Id"])
want = ["eg::k_msm_jobs", "eg::k_base_tables", "eg::k_encode_batch", "eg::k_decode_points", "eg::k_hash"]
lines = ["# rocprofv3 --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --ballots 262144   (MI355X)",
         "# separate passes for FETCH_SIZE, WRITE_SIZE and two groups of SQ counters; values summed over the launches of one step",
         "# (1 chunk of 262144 ballots).  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports 16-B/lane reads by 2x",
         "# (MI355X_MICROARCH.md, HBM section), hence the x2.", ""]
for k in want:
    c = acc.get(k)
    if not c:
        continue
    n = max(len(launches[k]), 1)
    per_launch = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / n / 1e9
    lines.append(f"{k}: launches={n}  FETCH_SIZE={c['FETCH_SIZE']:.0f} KiB  WRITE_SIZE={c['WRITE_SIZE']:.0f} KiB  "
                 f"-> per launch (FETCH x2 corrected + WRITE) = {per_launch:.2f} GB")
    sq = "  ".join(f"{n_}={c[n_]:.3e}" for n_ in sorted(c) if n_.startswith("SQ_"))
    lines.append("    " + sq)
    if c.get("SQ_WAVE_CYCLES"):
        lines.append(f"    VALU-active share of wave cycles = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.3f}   "
                     f"issue-stall share = {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
(ROOT / "profiles" / f"{tag}_bench_pmc_counters.txt").write_text("\n".join(lines) + "\n")
print((ROOT / "profiles" / f"{tag}_bench_pmc_counters.txt").read_text())
