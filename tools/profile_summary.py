#!/usr/bin/env python3
"""Turns the outputs of tools/profile_round.sh (under gpurun_out/) into the summaries kept under profiles/:
   <round>_bench_line*.json, <round>_kernel_stats_<workload>.txt, <round>_pmc_counters.txt and traffic.json (the memory-side
   bytes per ballot and launch of every profiled kernel, which bench.py reads for `roofline.traffic`).
   usage: profile_summary.py [--traffic-only] <round-tag, e.g. r03>
   --traffic-only (what tools/profile_round.sh runs on the GPU box between the counter passes and the bench lines): only the PMC summary and
   traffic.json, so that the bench lines printed afterwards carry the traffic of the build they measure."""
import collections
import csv
import hashlib
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"
PROF = ROOT / "profiles"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
TRAFFIC_ONLY = "--traffic-only" in sys.argv[1:]
tag = args[0] if args else "r03"
PMC_BALLOTS = 1000000      # the size the bench line quotes (tools/profile_round.sh); a step is cut into chunks, one launch per chunk and stage
WORKLOADS = {"single": "single-5", "multi": "multi-16", "qv": "qv-5-20"}
DESCR = {"single": "1M single-choice 5-option ballots (BASELINE configs[1])", "multi": "1M multi-choice 3-of-16 ballots (configs[3])",
         "qv": "1M quadratic-voting ballots, 5 options / 20 credits (configs[2])"}


def source_hash() -> str:
    """Hash of the device and host sources of the library without comments and white space (tools/srchash.py): ties profiles/traffic.json
    to the build it was measured on (bench.py prints "traffic_stale": true when the tree it runs from hashes differently)."""
    sys.path.insert(0, str(ROOT / "tools"))
    from srchash import code_hash
    return code_hash(ROOT)


def kname(name):
    name = name.split("(")[0].strip()
    return name[5:] if name.startswith("void ") else name


# ---- bench lines ---------------------------------------------------------------------------------------------------
for src, dst in () if TRAFFIC_ONLY else (("bench_single.json", "bench_line.json"), ("bench_multi.json", "bench_line_multi16.json"), ("bench_qv.json", "bench_line_qv.json"),
                 ("bench_10M.json", "bench_line_10M.json"), ("bench_tampered1pct.json", "bench_line_tampered1pct.json"),
                 ("bench_msm.json", "bench_line_msm.json"), ("bench_in_process2.json", "bench_line_in_process2_one_gpu.json"),
                 ("bench_bare2.json", "bench_line_bare_gpus2_one_gpu.json"), ("bench_in_process2_host.json", "bench_line_in_process2_from_host_one_gpu.json"),
                 ("bench_spawn1.json", "bench_line_spawn_gpus1_rccl.json"), ("bench_driver_cmd.json", "bench_line_driver_cmd.json")):
    f = OUT / src
    if f.exists() and f.read_text().strip():
        line = f.read_text().strip().splitlines()[-1]
        json.loads(line)
        (PROF / f"{tag}_{dst}").write_text(line + "\n")

# ---- round 5 probes: copied as they are (the command that made each is its first line) ----------------------------------------
for src, dst, head in () if TRAFFIC_ONLY else (
        ("hbm_power_probe.txt", "hbm_power_probe.txt", "# python3 tools/hbm_power_probe.py   (MI355X; package power from hwmon while device-to-device copies stream)"),
        ("json_stream_probe.txt", "json_stream_probe.txt", "# python3 tools/json_stream_probe.py   (MI355X, 16 host threads; 1 M single-choice ballots as 1.39 GB of JSON)"),
        ("json_trace.txt", "json_stream_trace.txt", "# EG_JSON_TRACE=1 python3 tools/json_trace_probe.py   (MI355X; timeline of eg_verify_choice_json on 1 M ballots, three calls)"),
        ("ubench_fp64.txt", "ubench_fp64.txt", "# tools/fp64_probe.sh   (MI355X; FP64-limb field multiplication against the shipped integer one: bursts with the correctness check, then sustained with the package power)"),
        ("msm_by_size.txt", "msm_by_size.txt", "# python3 tools/msm_probe.py   (MI355X; one vartime_multi_mul, operands in HBM, Straus and bucket paths forced; encodings vs prepared points)")):
    f = OUT / src
    if f.exists() and f.read_text().strip():
        body = "\n".join(l for l in f.read_text().splitlines() if "amdgpu.ids" not in l)
        (PROF / f"{tag}_{dst}").write_text(head + "\n" + body + "\n")

# ---- kernel stats --------------------------------------------------------------------------------------------------
for w in () if TRAFFIC_ONLY else WORKLOADS:
    f = OUT / f"prof_stats_{w}" / "stats_kernel_stats.csv"
    if not f.exists():
        continue
    rows = [r for r in csv.DictReader(open(f)) if "k_selfbench_fmul" not in r["Name"]]      # the box calibration (valu_roofline.box) is not part of a step
    all_ns = sum(int(r["TotalDurationNs"]) for r in rows) or 1
    for r in rows:
        r["Percentage"] = 100.0 * int(r["TotalDurationNs"]) / all_ns
    lines = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --workload {w} --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated   (MI355X)",
             f"# 1 warm-up + 3 timed steps of {DESCR[w]}, plus the untimed generator launch (the launches of the box calibration, k_selfbench_fmul, left out)",
             f"{'kernel':64s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>11s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>8s}"]
    for r in rows:
        lines.append(f"{kname(r['Name'])[:64]:64s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:12.3f} {float(r['AverageNs'])/1e6:11.4f} "
                     f"{int(r['MinNs'])/1e6:9.3f} {int(r['MaxNs'])/1e6:9.3f} {float(r['Percentage']):8.4g}")
    (PROF / f"{tag}_kernel_stats_{w}.txt").write_text("\n".join(lines) + "\n")
    # the same command with EG_STREAMS=1 (one work set, one stream): nothing overlaps, so a kernel's total is its share of the step
    f = OUT / f"prof_serial_{w}" / "stats_kernel_stats.csv"
    if f.exists():
        rows = list(csv.DictReader(open(f)))
        skip = ("encrypt", "k_build_fixed_table", "k_comb_window_bases", "k_setup_points", "k_const_points", "at::", "k_selfbench_fmul")
        step = sum(int(r["TotalDurationNs"]) for r in rows if not any(x in r["Name"] for x in skip))
        lines = [f"# EG_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --workload {w} --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --no-extra-configs   (MI355X)",
                 f"# ONE work set on one stream: kernels run one after the other, so total_ms / 4 steps is a kernel's cost per step of {DESCR[w]}",
                 f"# (share = of the verification kernels; generator and table set-up excluded: {step / 4e6:.1f} ms per step)",
                 f"{'kernel':64s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>11s} {'share':>8s}"]
        for r in rows:
            if any(x in r["Name"] for x in skip):
                continue
            lines.append(f"{kname(r['Name'])[:64]:64s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:12.3f} {float(r['AverageNs'])/1e6:11.4f} "
                         f"{100 * int(r['TotalDurationNs']) / step:7.2f}%")
        (PROF / f"{tag}_kernel_stats_serial_{w}.txt").write_text("\n".join(lines) + "\n")

# ---- PMC passes ----------------------------------------------------------------------------------------------------
want = [f"eg::k_eq_table<{m}, {t}>" for m in ("false", "true") for t in (5, 6)] + ["eg::k_eq_direct"] + \
       [f"eg::{k}<{t}>" for k in ("k_eq_generic", "k_base_tables", "k_sum_tables") for t in (5, 6)] + ["eg::k_encode_batch", "eg::k_decode_points", "eg::k_hash"]
text = [f"# rocprofv3 --pmc <counter> -- python3 bench.py --steps 1 --warmup 0 --workload W --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --ballots {PMC_BALLOTS}   (MI355X)",
        "# separate passes per counter (FETCH_SIZE, WRITE_SIZE; for the single-choice workload also two groups of SQ counters); values are",
        "# summed over the launches of the one step (its chunks x stages).  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the",
        "# bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section; calibrated for this access shape in <round>_fetch_calibration.txt),",
        "# hence the x2.", ""]
traffic = {"round": tag, "ballots_per_step_measured": PMC_BALLOTS, "source_hash": source_hash(), "workloads": {}}
try:
    traffic["commit"] = subprocess.check_output(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    traffic["commit"] = None
for w, key in WORKLOADS.items():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    dirs = [f"pmc_FETCH_SIZE_{w}", f"pmc_WRITE_SIZE_{w}"] + ([f"pmc_SQ1_{w}", f"pmc_SQ2_{w}"] if w == "single" else [])
    found = False
    for d in dirs:
        f = OUT / d / "pmc_counter_collection.csv"
        if not f.exists():
            continue
        found = True
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "FETCH_SIZE":
                launches[k].add(r["Dispatch_Id"])
    if not found:
        continue
    text.append(f"== workload {w}: {DESCR[w].replace('1M', str(PMC_BALLOTS))}")
    n_chunks = max([len(v) for k, v in launches.items() if k.startswith("eg::k_base_tables")] + [1])          # the table builder runs once per chunk
    per_launch_ballots = PMC_BALLOTS / n_chunks
    ent = {"ballots_per_launch": per_launch_ballots, "chunks_per_step": n_chunks, "kernels": {}}
    for k in want:
        c = acc.get(k)
        if not c:
            continue
        n = max(len(launches[k]), 1)
        per_launch = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / n
        ent["kernels"][k] = {"launches": n, "fetch_kib": c["FETCH_SIZE"], "write_kib": c["WRITE_SIZE"],
                             "bytes_per_launch": per_launch, "bytes_per_ballot_launch": per_launch / per_launch_ballots}
        text.append(f"{k}: launches={n}  FETCH_SIZE={c['FETCH_SIZE']:.0f} KiB  WRITE_SIZE={c['WRITE_SIZE']:.0f} KiB  "
                    f"-> per launch (FETCH x2 + WRITE) = {per_launch/1e9:.2f} GB = {per_launch/per_launch_ballots/1e3:.1f} KB per ballot")
        sq = "  ".join(f"{n_}={c[n_]:.3e}" for n_ in sorted(c) if n_.startswith("SQ_"))
        if sq:
            text.append("    " + sq)
        if c.get("SQ_WAVE_CYCLES"):
            text.append(f"    VALU-active share of wave cycles = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.3f}   "
                        f"issue-stall share = {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
    traffic["workloads"][key] = ent
    text.append("")
if traffic["workloads"]:
    (PROF / f"{tag}_pmc_counters.txt").write_text("\n".join(text) + "\n")
    (PROF / "traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
    print("\n".join(text))
