#!/usr/bin/env python3
"""bench.py -- EncryptedChoice ballot verifications/sec on MI355X (BASELINE.json metric).

A step = one pass of the hot path (EncryptedChoice::verify for every ballot + homomorphic tally + the one
tally exchange) over one batch of synthetic ballots that is already resident in HBM.  Workload at N = 1:
BASELINE.json configs[1], 1M single-choice 5-option ballots.  With N ranks each rank holds its own 1M-ballot
shard (weak scaling; ballots are independent, no data-path collective) and the per-rank tallies are
all-gathered over RCCL once per step.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N --steps K --warmup W          # bare: starts the N ranks itself, as a child torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Inputs are generated on the GPU by the library's own prover kernel
(eg_choice_encrypt_batch_device); the CPU oracle is used only for the cpu_baseline leg and as a checker.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# before anything can initialise the HIP/HSA runtime (importing torch does not, torch.cuda.* does): the host driver of this
# pool only supports dmabuf IPC, which RCCL needs for its intra-node transport
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PUBLIC_KEY_HEX = "a6adb6e9c0ae8d54c26e6e56b5ccd7a16bb0e1951abe4d7ee7028e3d4eca8531"  # seed-12345 key of the snapshots
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU model: field operations per building block counted by the host-check build
# (tests/hostcheck: hc_op_counts) as (fe_mul, fe_sq); one fe_mul issues 98 and one fe_sq 62 v_mad_u64_u32 (fe25519.cuh: 81 / 45 limb
# products + 16 to fold the high columns + 1 for the top carry).
OPS = {5: {"decode": (27, 257), "direct_table": (64, 0), "direct_mul": (1269, 1008), "comb": (91, 0), "encode": (32, 255),
           "base_table": (805, 816), "base_mul": (551, 200), "enc_batch_each": (23, 10), "enc_batch_inversion": (11, 254),
           "multi_first": (558, 200), "multi_extra": (408, 0), "sum_table_first": (149, 0), "sum_table_extra": (40, 0),
           "comb_wide": (77, 0)},
       6: {"decode": (27, 257), "direct_table": (64, 0), "direct_mul": (1269, 1008), "comb": (91, 0), "encode": (32, 255),
           "base_table": (994, 860), "base_mul": (463, 168), "enc_batch_each": (23, 10), "enc_batch_inversion": (11, 254),
           "multi_first": (470, 168), "multi_extra": (344, 0), "sum_table_first": (296, 0), "sum_table_extra": (48, 0),
           "comb_wide": (77, 0)}}      # by comb shape of the plan's per-ballot tables (5 x 51 or 6 x 43: eg_plan_describe "teeth")
# memory-side traffic per ballot and launch of the profiled kernels comes from profiles/traffic.json, which
# tools/profile_summary.py writes from the separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of tools/profile_round.sh
TRAFFIC_JSON = ROOT / "profiles" / "traffic.json"
MAD_PER_MUL, MAD_PER_SQ = 98, 62
MAD_PEAK_T = 36.8              # profiles/r03_ubench_valu_rates.txt: v_mad_u64_u32 alone, 8 waves/SIMD, T lane-ops/s chip-wide from the wall
                               # clock (4.2 issue cycles per wave64 instruction; 32.8 T at the two or three waves per SIMD the kernels hold)
FMUL_PEAK_G = 275.0            # profiles/r03_ubench_field_bench.txt: the shipped field multiplication in a bare chain, G/s chip-wide
SQ_WEIGHT = 0.74               # a squaring in the same bench: 372 G/s
FMUL_SUSTAINED_G = 252.0       # profiles/r03_ubench_field_sustained.txt: the same chain run back to back for seconds (3 waves per SIMD): the power
                               # management holds 2.25 GHz at 1.24 kW under it, not the burst clock
MAD_PEAK_SCLK_MHZ = 2420.0     # shader clock the chip held in that micro-benchmark (same file, s_memtime / s_memrealtime) ...
FMUL_PEAK_SCLK_MHZ = 2270.0    # ... and in the field-multiplication chain: both are bursts of tens of milliseconds.  Back-to-back verification
                               # steps run at the clock the power management settles on (about 2.08 GHz at 1.31 kW of the 1.4 kW cap on the
                               # boxes measured, `clock` in the JSON line), so the fractions are also given against the peaks scaled to it
DOMINANT_KERNEL = "eg::k_eq_table<false, {teeth}>"   # one table-backed base + fixed-base combs: every ring equation (kernels.cuh); the
                                                     # second template argument is the comb shape of the plan's tables


def plan_field_ops(desc: dict, wide_combs: bool = False):
    """(fe_mul, fe_sq) per ballot of the shipped pipeline, from the flattened verification plan (eg_plan_describe) and the
    per-building-block counts above.  Not counted: the few point additions of the derived points, scalar arithmetic, hashing."""
    def add(*xs):
        return (sum(x[0] for x in xs), sum(x[1] for x in xs))

    def mul(x, k):
        return (x[0] * k, x[1] * k)

    OPS = globals()["OPS"][desc.get("teeth", 6)]
    return add(mul(OPS["decode"], desc["wire_points"]),
               mul(OPS["base_table"], desc["bases"]),
               mul(OPS["base_mul"], desc["single_table_jobs"] + desc["loose_table_terms"]),
               mul(OPS["multi_first"], desc["chains"]),                   # several table-backed bases on one doubling chain
               mul(OPS["multi_extra"], desc["chain_extra_terms"]),
               mul(add(OPS["direct_table"], OPS["direct_mul"]), desc["direct_terms"]),
               mul(OPS["sum_table_first"], desc["sum_tables"]),           # tables of sums of ring bases, made from their tables
               mul(OPS["sum_table_extra"], desc["sum_table_members"] - desc["sum_tables"]),
               # ring-group walk: every further group re-opens and re-stores the T accumulator entries of a sum (one multiplication each way)
               (2 * desc.get("teeth", 6) * desc["sum_tables"] * max(desc.get("table_groups", 1) - 1, 0) if desc.get("ring_group") else 0, 0),
               mul(OPS["comb_wide" if wide_combs else "comb"], desc["combs"]),   # wide tables: 11 instead of 13 additions per comb
               mul(OPS["enc_batch_each"], desc["deferred"]),
               mul(OPS["enc_batch_inversion"], desc["inversion_groups"]),
               mul(OPS["encode"], desc["plain_encodes"]))
# algorithmic bytes per ballot = packed ballot + 4-byte status word: 740 (single 5), 2084 (multi 16), 2148 (qv 5/20)


class ClockSampler:
    """Shader clock and package power of one GPU while the timed region runs, read from the hwmon files of its PCI function
    (freq1_input in Hz, power1_input in microwatts: what rocm-smi prints).  Measurement only; absent files give an empty result."""

    def __init__(self, torch, device_index: int, period_s: float = 0.01):
        import glob
        import threading
        self.files, self.samples, self.cap_w = None, [], None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for dev in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(dev)) == addr:
                    for h in glob.glob(dev + "/hwmon/hwmon*"):
                        if os.path.exists(h + "/freq1_input") and os.path.exists(h + "/power1_input"):
                            self.files = (h + "/freq1_input", h + "/power1_input")
                            try:
                                self.cap_w = int(open(h + "/power1_cap").read()) / 1e6
                            except (OSError, ValueError):
                                pass
        except (AttributeError, OSError):
            pass
        self.period, self._stop, self._thread = period_s, threading.Event(), None
        self._threading = threading

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append((int(open(self.files[0]).read()) / 1e6, int(open(self.files[1]).read()) / 1e6))
            except (OSError, ValueError):
                pass
            self._stop.wait(self.period)

    def start(self):
        if self.files:
            self._thread = self._threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread:
            self._stop.set()
            self._thread.join()
        if not self.samples:
            return None
        f = sorted(x[0] for x in self.samples)
        w = [x[1] for x in self.samples]
        return {"sclk_mhz": sum(f) / len(f), "sclk_mhz_median": f[len(f) // 2], "sclk_mhz_min": f[0], "sclk_mhz_max": f[-1],
                "power_w": sum(w) / len(w), "power_cap_w": self.cap_w, "samples": len(f),
                "source": "hwmon freq1_input / power1_input of the device, sampled every 10 ms over the timed region"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ballots", type=int, default=1_000_000, help="ballots per GPU per step (weak scaling: the default mode)")
    ap.add_argument("--total-ballots", type=int, default=0,
                    help="fixed-total mode (BASELINE.json configs[4]: 10M ballots sharded across the GPUs of a node): the batch of "
                         "this many ballots is split into contiguous shards, one per rank; the line says \"scaling\": \"strong\"")
    ap.add_argument("--options", type=int, default=None)
    ap.add_argument("--workload", choices=["single", "multi", "qv", "msm"], default="single",
                    help="single = BASELINE configs[1] (the bench line); multi = 3-of-16 (configs[3]); qv = 5 options / 20 credits (configs[2]); "
                         "msm = the primitive tier (Group::vartime_double_mul_generator x --ballots and one 2^16-term vartime_multi_mul; "
                         "benches/basics.rs:284-319 of the reference times these helpers)")
    ap.add_argument("--credits", type=int, default=20)
    ap.add_argument("--seed", type=int, default=20260612)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--tampered-percent", type=float, default=0.0,
                    help="flip one response bit in this share of the ballots before timing (SURVEY 8d: verdict parity and tally "
                         "exclusion with invalid ballots in the batch); the headline line uses 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wire-ingest", action="store_true",
                    help="skip the wire-ingest leg (JSON text of the same ballots through the native packer on the host cores)")
    ap.add_argument("--no-host-inclusive", action="store_true",
                    help="skip the PCIe-inclusive leg (host buffers through eg_verify_*_batch) that follows the timed loop at N = 1")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the extra untimed step that measures the dominant kernel alone on one work set (profile runs: the kernel "
                         "statistics then hold the launches of the warm-up and timed steps only)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks that all use device 0 and exchange through gloo: runs the whole multi-rank path of this file on a box "
                         "with one GPU (the ranks time-share it: the value means nothing, the JSON line and the tally check do)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short runs of the other BASELINE configs (multi-choice 3-of-16, quadratic voting 5 / 20, the primitive "
                         "tier) that follow the headline measurement at N = 1 and are printed as extra.configs")
    ap.add_argument("--in-process-devices", type=int, default=0, metavar="N",
                    help="the same step over N GPUs through ONE process (eg_verify_*_batch_multi_device: one context, one resident slab and "
                         "one stream per GPU, the slabs' tallies merged inside the library - no torch.distributed, no collective); with "
                         "--rehearse-one-gpu the N contexts all sit on device 0.  A second scaling measurement beside the ranked path")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--spawn", action="store_true",
                    help="started bare, launch the rank(s) as a child torch.distributed.run even for --gpus 1: the whole bare-launch path "
                         "(count the GPUs without touching HIP, child launcher, rendezvous, RCCL init with --force-dist) on a 1-GPU box")
    ap.add_argument("--from-host", action="store_true",
                    help="with --in-process-devices: the whole batch sits in ONE pinned host buffer and goes through eg_verify_*_batch_multi "
                         "(uploads inside the library, one host thread per GPU) - what a single-process host with its ballots in host memory "
                         "gets, PCIe included; the line says so (config.input = host) and is never the headline")
    ap.add_argument("--selfbench-seconds", type=float, default=2.0,
                    help="length of the field-multiplication calibration run before the timed region (valu_roofline.box); 0 = skip")
    args = ap.parse_args()
    args.box_fmul_g = None           # set by the calibration run (valu_roofline.box); the extra configs quote their fractions against it
    return args


def cpu_model() -> str:
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cores() -> int:
    """Hardware threads this process may actually use (affinity mask and cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return n


def measure_msm(args, ctx, eg, torch, dev, steps, warmup):
    """n x vartime_double_mul_generator and one 2^16-term vartime_multi_mul on operands resident in HBM (eg_vartime_multi_mul_batch_device);
    one step = both.  Returns the timings and the buffers (for the checker leg)."""
    grp = eg.Ristretto(ctx)
    n, big = args.ballots, 1 << 16
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(args.seed)

    def scalars(count):          # uniform 252-bit scalars (canonical)
        t = torch.randint(0, 256, (count, 32), dtype=torch.uint8, generator=g)
        t[:, 31] &= 0x0F
        return t.reshape(-1).to(dev)

    def points(count):           # valid encodings: [x]G made on the GPU
        src = scalars(count)
        out = torch.empty(32 * count, dtype=torch.uint8, device=dev)
        grp.vartime_multi_mul_device(count, 0, 0, 0, out.data_ptr(), d_r=src.data_ptr(), stream=stream)
        return out

    k, r, p = scalars(n), scalars(n), points(n)
    bk, bp = scalars(big), points(big)
    out1 = torch.empty(32 * n, dtype=torch.uint8, device=dev)
    ok1 = torch.empty(n, dtype=torch.uint8, device=dev)
    out2 = torch.empty(32, dtype=torch.uint8, device=dev)
    scratch = torch.empty(max(grp.msm_scratch_bytes(1, big), 16), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_double, t_big = 0.0, 0.0

    def step(timed):
        nonlocal t_double, t_big
        ev[0].record()
        grp.vartime_multi_mul_device(n, 1, k.data_ptr(), p.data_ptr(), out1.data_ptr(), d_r=r.data_ptr(), d_ok=ok1.data_ptr(), stream=stream)
        ev[1].record()
        grp.vartime_multi_mul_device(1, big, bk.data_ptr(), bp.data_ptr(), out2.data_ptr(), d_scratch=scratch.data_ptr(), stream=stream)
        ev[2].record()
        if timed:
            torch.cuda.synchronize()
            t_double += ev[0].elapsed_time(ev[1])
            t_big += ev[1].elapsed_time(ev[2])

    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return {"n": n, "big": big, "elapsed": elapsed, "double_ms": t_double / steps, "big_ms": t_big / steps,
            "bufs": (k, p, r, bk, bp, out1, ok1, out2)}


def bench_msm(args, ctx, eg, torch, dev, world):
    """Primitive tier as its own bench line.  The oracle is timed beside it."""
    if world != 1:
        raise SystemExit("--workload msm is a one-GPU bench")
    m = measure_msm(args, ctx, eg, torch, dev, args.steps, args.warmup)
    n, big, elapsed, double_ms, big_ms = m["n"], m["big"], m["elapsed"], m["double_ms"], m["big_ms"]
    k, p, r, bk, bp, out1, ok1, out2 = m["bufs"]
    value = n / (double_ms * 1e-3)
    line = {
        "metric": "Group::vartime_double_mul_generator operations/sec (Ristretto backend, primitive tier)",
        "value": value, "unit": "ops/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{n} x [k]P + [r]G with k, r uniform and P = [x]G, resident in HBM (eg_vartime_multi_mul_batch_device, terms = 1); "
                               f"then one {big}-term vartime_multi_mul; NOT the BASELINE metric (that is the default workload)",
                   "ops": n, "big_terms": big, "seed": args.seed, "all_points_decoded": bool(int(ok1.min().item()) == 1)},
        "multi_mul_65536": {"ms": big_ms, "terms_per_s": big / (big_ms * 1e-3),
                            "note": "one product, cut so that it covers the chip (65 536 lanes, one term each here; problems in batches share "
                                    "doubling chains in chunks of 8 terms); the 252 sequential doublings of a ladder are ~0.4 ms for a lane "
                                    "whatever the algorithm; wave-shuffle reduction, one encoding"},
        "roofline": {"bound": "hbm", "kernel": "eg::k_prim_msm", "achieved": 128.0 * n / (double_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": 128.0 * n / (double_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": double_ms,
                     "note": "algorithmic bytes per operation: two scalars, a point, an encoding = 128 B against ~2600 field "
                             "multiplications: VALU-bound like the batch tier"},
    }
    if not args.no_cpu_baseline:
        from oracle import oracle as o

        hk, hp, hr = bytes(k.cpu().numpy()), bytes(p.cpu().numpy()), bytes(r.cpu().numpy())
        got = bytes(out1.cpu().numpy())
        t0, m = time.perf_counter(), 0
        same = True
        while time.perf_counter() - t0 < min(args.cpu_seconds, 6.0) and m < n:
            want = o.point_double_mul_generator(hk[32 * m: 32 * m + 32], hp[32 * m: 32 * m + 32], hr[32 * m: 32 * m + 32])
            same = same and want == got[32 * m: 32 * m + 32]
            m += 1
        cpu_s = time.perf_counter() - t0
        t1 = time.perf_counter()
        want_big = o.point_multi_mul(bytes(bk.cpu().numpy()), bytes(bp.cpu().numpy()))
        big_cpu_s = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": m / cpu_s, "unit": "ops/s", "cores": 1, "kind": "port",
                                "sample": f"first {m} of the same operations through oracle/ (one thread, C restatement, not dalek)",
                                "results_match_gpu": bool(same), "multi_mul_65536_s": big_cpu_s,
                                "multi_mul_65536_matches_gpu": bool(want_big == bytes(out2.cpu().numpy())), "cpu_model": cpu_model()}
    print(json.dumps(line))


def oracle_sample_check(args, torch, p, b, st, m, bad, n_random=5000, n_bad=1000):
    """Verdicts of a sample of the batch - up to n_bad of the tampered ballots and n_random drawn at random - through the CPU oracle, all
    hardware threads; returns (sample size, tampered ones in it, verdicts identical).  After the timed steps; the checker, never the product."""
    from oracle import oracle as o

    pk = bytes.fromhex(PUBLIC_KEY_HEX)
    op = o.QvParams(pk, p.n_options, args.credits) if p.kind_name == "qv" else o.ChoiceParams(pk, p.n_options, p.kind_name == "single")
    g = torch.Generator(device="cpu").manual_seed(args.seed + 1)
    idx = torch.randperm(m, generator=g)[: min(n_random, m)]
    n_in = 0
    if bad is not None and len(bad):
        idx = torch.cat([bad[:n_bad].cpu(), idx])
        n_in = min(n_bad, len(bad))
    idx_d = idx.to(b.device)
    sample = bytes(b.view(m, p.ballot_size)[idx_d].cpu().numpy())
    want = op.verify_batch(sample, threads=effective_cores())
    got = st[idx_d].cpu().tolist()
    return len(idx), n_in, want == [x & 0xFFFFFFFF for x in got]


def extra_configs(args, ctx, eg, torch, dev, pk, stream, steps: int = 10):
    """Short runs of the BASELINE configs that are not the headline - configs[3] multi-choice 3-of-16, configs[2] quadratic voting
    5 options / 20 credits, configs[4]'s 10 M single-choice batch on this one GPU, configs[1] with 1 % tampered ballots -
    and of the primitive tier, on the same GPU, ballots resident in HBM; a step is what a step of the headline is (tally reset, verify,
    tally encode).  10 timed steps each (3 for the 10 M batch: 1.6 s a step), one warm-up.  The headline's params object has been closed
    by the caller."""
    res = {}
    n = args.ballots
    runs = (("multi16", lambda: eg.ChoiceParams.multi_choice(ctx, pk, 16), {"n_selected": 3}, n, 0.0, steps),
            ("qv", lambda: eg.QuadraticVotingParams(ctx, pk, 5, args.credits), {}, n, 0.0, steps),
            ("single10M", lambda: eg.ChoiceParams.single_choice(ctx, pk, 5), {}, 10 * n, 0.0, 3),
            ("tampered1pct", lambda: eg.ChoiceParams.single_choice(ctx, pk, 5), {}, n, 1.0, steps))
    for name, make, gen_kw, m, tampered, k in runs:
        try:
            p = make()
            b = torch.empty(m * p.ballot_size, dtype=torch.uint8, device=dev)
            st = torch.empty(m, dtype=torch.int32, device=dev)
            tl = torch.empty(64 * p.n_options, dtype=torch.uint8, device=dev)
            p.encrypt_batch_device(args.seed, 0, m, b.data_ptr(), stream=stream, **gen_kw)
            n_t = int(m * tampered / 100.0)
            bad = None
            if n_t:
                g = torch.Generator(device="cpu").manual_seed(args.seed)
                bad = torch.randperm(m, generator=g)[:n_t].to(dev)
                b.view(m, p.ballot_size)[bad, p.ballot_size - 32] ^= 1       # last response scalar stays canonical

            def one():
                p.tally_reset(stream)
                p.verify_batch_device(m, b.data_ptr(), st.data_ptr(), stream)
                p.tally_encode_device(tl.data_ptr(), stream)

            one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                one()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            accepted = int((st == 0).sum().item())
            res[name] = {"value": m * k / dt, "ms_per_step": dt / k * 1e3, "accepted": accepted, "ballots": m, "tampered": n_t,
                         "ballot_bytes": p.ballot_size, "steps": k, "accepted_is_all_but_tampered": accepted == m - n_t,
                         "tally_matches_engine": bytes(tl.cpu().numpy()) == p.tally_encode()}
            if p.kind_name != "single":
                desc = eg.plan_describe(p.kind_name, p.n_options, args.credits if p.kind_name == "qv" else 0)
                fm, fs = plan_field_ops(desc, wide_combs=ctx.comb_table_bits()[1] != 0)
                res[name]["fmul_equiv_frac_sustained"] = m * k / dt * (fm + SQ_WEIGHT * fs) / 1e9 / FMUL_SUSTAINED_G
                if args.box_fmul_g:
                    res[name]["fmul_equiv_frac_sustained_box"] = m * k / dt * (fm + SQ_WEIGHT * fs) / 1e9 / args.box_fmul_g
            if not args.no_cpu_baseline:      # >= 5000 ballots of every leg through the oracle (the tampered leg: 1000 tampered ones among them)
                cs, cb, same = oracle_sample_check(args, torch, p, b, st, m, bad)
                res[name].update({"cpu_sample": cs, "cpu_sample_tampered": cb, "cpu_sample_verdicts_match": bool(same)})
            del b, st, tl
            p.close()
            torch.cuda.empty_cache()
        except Exception as e:      # an extra must never cost the headline line
            res[name] = {"error": repr(e)}
    try:
        m = measure_msm(args, ctx, eg, torch, dev, steps, 1)
        res["msm"] = {"value": m["n"] / (m["double_ms"] * 1e-3), "unit": "ops/s", "ms_per_step": m["elapsed"] / steps * 1e3,
                      "double_mul_ms": m["double_ms"], "multi_mul_65536_ms": m["big_ms"], "ops": m["n"],
                      "all_points_decoded": bool(int(m["bufs"][6].min().item()) == 1)}
    except Exception as e:
        res["msm"] = {"error": repr(e)}
    return res


def host_inclusive_leg(torch, params, ballots, status, B, resident_value, before_each=None, iters=3):
    """PCIe-inclusive rate (first H2D byte to last status byte D2H): the rank's own batch from a pinned host buffer through the
    host-pointer entry point (pipelined uploads, eg_verify_*_batch).  Reported beside `value`, never as it.  before_each: a barrier, so that
    with several ranks every iteration starts on all of them at the same time (they share one host's memory and PCIe root).  A rank on which
    the leg fails (no pinned memory left, say) still takes part in every barrier - the other ranks must not be left waiting in one - and
    reports {"error": ...}."""
    err, host, host_status = None, None, None
    try:
        host = torch.empty(ballots.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(ballots)
        host_status = torch.empty(B, dtype=torch.int32, pin_memory=True)
        torch.cuda.synchronize()
    except Exception as e:
        err = repr(e)
    times = []
    for it in range(iters + 1):                      # the first call sizes the staging buffers
        if before_each:
            before_each()
        if err:
            continue
        try:
            t0 = time.perf_counter()
            params.verify_batch_host_ptr(B, host.data_ptr(), host_status.data_ptr())
            times.append(time.perf_counter() - t0)
        except Exception as e:
            err = repr(e)
    if err:
        return {"error": err, "value": 0.0, "verdicts_match_device_path": False}
    hs = sum(times[1:]) / iters
    return {"value": B / hs, "unit": "ballots/s", "ms": hs * 1e3, "iterations": iters, "pinned": True,
            "bytes_h2d": B * params.ballot_size, "bytes_d2h": 4 * B,
            "verdicts_match_device_path": bool(torch.equal(host_status, status.cpu())),
            "vs_value": B / hs / resident_value}


def ballots_as_json(args, eg, torch, params, ballots, B, n_opt, reps_for):
    """The first min(B, 1000) ballots of the batch as JSON objects in serde's layout (src/serde.rs:19-80), and a text builder: text(reps) =
    one JSON array holding those objects `reps` times."""
    from elastic_elgamal_amd import ingest as eging, serde as egserde

    distinct = min(B, 1000)
    raw = bytes(ballots[: distinct * params.ballot_size].cpu().numpy())
    if args.workload == "qv":
        objs = [eging.unpack_qv_ballot(raw[i * params.ballot_size : (i + 1) * params.ballot_size], n_opt, args.credits) for i in range(distinct)]
    else:
        objs = [egserde.unpack_encrypted_choice(raw[i * params.ballot_size : (i + 1) * params.ballot_size], n_opt, args.workload == "single")
                for i in range(distinct)]
    one = [json.dumps(o) for o in objs]
    return distinct, raw, one, (lambda reps: ("[" + ",".join(one * reps) + "]").encode())


def json_inclusive_leg(args, eg, torch, params, ballots, status, B, n_opt, resident_value, threads, before_each=None, text=None, distinct=None):
    """The whole wire path inside the library (eg_verify_*_json): JSON text in host memory -> status words, on as many objects as the
    step has ballots (the first 1000 ballots repeated); host threads pack piece k+1 while the GPU verifies piece k.  Like
    host_inclusive_leg, a rank on which the leg fails still takes part in every barrier."""
    import ctypes
    import numpy as np

    err, jn, jstatus, jreps = None, 0, None, 1
    try:
        if text is None:
            distinct, _, _, build = ballots_as_json(args, eg, torch, params, ballots, B, n_opt, None)
            text = build(max(1, B // distinct))
        jreps = max(1, B // distinct)
        jn = distinct * jreps
        jstatus = (ctypes.c_uint32 * jn)()
    except Exception as e:
        err = repr(e)
    json_s, jgot = None, 0
    for _ in range(3):
        if before_each:
            before_each()
        if err:
            continue
        try:
            t0 = time.perf_counter()
            jgot = params.verify_json_into(text, jstatus, threads)
            dt = time.perf_counter() - t0
            json_s = dt if json_s is None else min(json_s, dt)
        except Exception as e:
            err = repr(e)
    if err:
        return {"error": err, "value": 0.0, "verdicts_match_device_path": False, "threads": threads}, None
    jarr = np.frombuffer(jstatus, dtype=np.uint32)
    first_status = status[:distinct].cpu().numpy().astype(np.uint32)
    return {"value": jn / json_s, "unit": "ballots/s", "objects": jn, "json_bytes": len(text), "ms": json_s * 1e3,
            "threads": threads, "vs_value": jn / json_s / resident_value,
            "verdicts_match_device_path": bool(jgot == jn and np.array_equal(jarr.reshape(jreps, distinct), np.tile(first_status, (jreps, 1)))),
            "note": "eg_verify_*_json: JSON text in host memory -> status words; parse, upload and verify pipelined"}, jarr


def in_process_json_leg(args, eg, torch, params, ballots, status, counts, total, n_opt, resident_value):
    """--in-process-devices: as many objects as the step has ballots (the first 1000 ballots of slab 0 repeated) as ONE JSON array through
    eg_verify_*_json_multi: one splitter and pool of host threads, the packed windows dealt to the N params objects, verdicts in text order."""
    import ctypes
    import numpy as np

    B0 = counts[0]
    distinct, _, _, build = ballots_as_json(args, eg, torch, params[0], ballots[0], B0, n_opt, None)
    reps = max(1, total // distinct)
    text = build(reps)
    jn = distinct * reps
    jstatus = (ctypes.c_uint32 * jn)()
    cores = effective_cores()
    for p in params:
        p.tally_reset()
    best, got = None, 0
    for _ in range(3):
        t0 = time.perf_counter()
        got = eg.verify_json_multi_into(params, text, jstatus, cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    first_status = status[0][:distinct].cpu().numpy().astype(np.uint32) if not args.from_host else None
    jarr = np.frombuffer(jstatus, dtype=np.uint32)
    same = bool(got == jn and (first_status is None or np.array_equal(jarr.reshape(reps, distinct), np.tile(first_status, (reps, 1)))))
    return {"value": jn / best, "unit": "ballots/s", "objects": jn, "json_bytes": len(text), "ms": best * 1e3, "threads": cores,
            "vs_value": jn / best / resident_value, "verdicts_match_device_path": same, "devices": len(params),
            "note": "eg_verify_*_json_multi: ONE parser (splitter + pool of host threads) for the node, its packed windows dealt to the params "
                    "objects by load; PCIe-inclusive; against `value` of this line (the resident rate of the same GPUs)"}


def die(code: int, msg: str):
    """Loud, early end of this rank: message on stderr, non-zero exit (torch.distributed.run then ends the other ranks)."""
    print(f"bench.py rank {os.environ.get('RANK', '0')}: FATAL: {msg}", file=sys.stderr, flush=True)
    raise SystemExit(code)


def count_gpus_without_hip(topology: str = "/sys/class/kfd/kfd/topology/nodes", env=None):
    """GPUs a HIP process started from this environment would see, counted WITHOUT touching HIP: the KFD topology nodes with
    simd_count > 0 (CPU nodes have 0), then ROCR_VISIBLE_DEVICES (indices into those, or GPU-<uuid> names) and HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES (indices into what ROCr left visible) applied the way the runtimes apply them: entries in order, the list ends at
    the first one that is not a visible device.  None when the topology is not readable (no KFD in this container)."""
    env = os.environ if env is None else env
    gpus = []
    try:
        nodes = sorted(os.listdir(topology), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    for node in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(topology, node, "properties")).read().splitlines() if " " in line)
        except OSError:
            continue                      # a node this user may not read (another tenant's partition): not ours
        if int(props.get("simd_count", "0")) > 0:
            gpus.append(props.get("unique_id", "").strip())
    n = len(gpus)
    rocr = env.get("ROCR_VISIBLE_DEVICES")
    if rocr is not None:
        seen = 0
        for tok in rocr.split(","):
            tok = tok.strip()
            if tok.upper().startswith("GPU-"):
                try:
                    want = int(tok[4:], 16)
                except ValueError:
                    want = None
                ok = want is not None and any(int(u) == want for u in gpus if u.isdigit())
            else:
                ok = tok.isdigit() and int(tok) < n
            if not ok:
                break
            seen += 1
        n = seen
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(name)
        if v is None:
            continue
        seen = 0
        for tok in v.split(","):
            tok = tok.strip()
            if not (tok.isdigit() and int(tok) < n):
                break
            seen += 1
        n = seen
        break                             # HIP reads HIP_VISIBLE_DEVICES first and CUDA_VISIBLE_DEVICES only in its absence
    return n


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment): start the N ranks as a CHILD
    `python -m torch.distributed.run ... bench.py <same arguments>` and hand back its exit code.  Runs before this process has imported
    torch.cuda or loaded the library, so nothing here has touched the GPU; it is a child process, never an exec (a process that has
    initialised the GPU must not be replaced, and under `rocprofv3 -- python3 bench.py` the profiler's preload has).  The child's stdout
    is this process's stdout: rank 0's one JSON line stays the last line.  examples/voting.rs:199-203 is the loop the N slabs stand in for."""
    import socket
    import subprocess

    # NOTHING before the spawn initialises HIP: this process imports neither torch nor the library, and the GPUs are counted from the
    # KFD topology in sysfs (round 5 asked torch.cuda.device_count(), which on this ROCm build goes through amdsmi and falls back to
    # hipGetDeviceCount - a runtime initialisation in the parent - when amdsmi fails).
    if not args.rehearse_one_gpu:
        have = count_gpus_without_hip()
        if have is not None and have < args.gpus:
            print(f"bench.py: FATAL: --gpus {args.gpus} but this node shows {have} GPU(s) (use --rehearse-one-gpu to time-share device 0 "
                  "over gloo)", file=sys.stderr, flush=True)
            return 2
        if have is None:
            print("bench.py: the KFD topology is not readable here: cannot count the GPUs before launching (the ranks will say)", file=sys.stderr, flush=True)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print("bench.py: --gpus %d without a launcher: starting the ranks as a child: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.stdout.flush()
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def bench_in_process(args):
    """--in-process-devices N: what a single-process host (examples/voting.rs:179-213 is one) does on a node with N GPUs.  One context, one
    params object, one resident slab and one stream per GPU; a step = tally reset on every GPU, eg_verify_*_batch_multi_device (one host
    thread per GPU inside the library enqueues the slab's verification and waits for its stream), eg_*_tally_encode_multi (the running
    tallies merged in the library).  Same JSON line as the ranked path; `parallelism` says which one ran."""
    import torch

    import elastic_elgamal_amd as eg
    from elastic_elgamal_amd import distributed as egd

    N = args.in_process_devices
    have = torch.cuda.device_count()
    if not args.rehearse_one_gpu and have < N:
        die(2, f"--in-process-devices {N} but this node shows {have} GPU(s) (use --rehearse-one-gpu to put the N contexts on device 0)")
    if args.workload == "msm":
        die(2, "--in-process-devices times the ballot workloads")
    devs = [0] * N if args.rehearse_one_gpu else list(range(N))
    pk = bytes.fromhex(PUBLIC_KEY_HEX)
    n_opt = args.options or {"single": 5, "multi": 16, "qv": 5}[args.workload]
    strong = args.total_ballots > 0
    total = args.total_ballots if strong else args.ballots * N
    ctxs, params, ballots, status, streams, counts = [], [], [], [], [], []
    t0 = time.time()
    for d, dev_i in enumerate(devs):
        torch.cuda.set_device(dev_i)
        dev = torch.device("cuda", dev_i)
        c = eg.Context(dev_i)
        if args.workload == "single":
            p = eg.ChoiceParams.single_choice(c, pk, n_opt)
        elif args.workload == "multi":
            p = eg.ChoiceParams.multi_choice(c, pk, n_opt)
        else:
            p = eg.QuadraticVotingParams(c, pk, n_opt, args.credits)
        first, last = egd.shard_range(total, d, N)           # GPU d owns the contiguous slab of voters [first, last)
        B = last - first
        b = torch.empty(max(B, 1) * p.ballot_size, dtype=torch.uint8, device=dev)
        s = torch.cuda.Stream(device=dev)
        kw = {"n_selected": 3} if args.workload == "multi" else {}
        p.encrypt_batch_device(args.seed, first, B, b.data_ptr(), stream=s.cuda_stream, **kw)
        n_t = int(B * args.tampered_percent / 100.0)
        if n_t:
            with torch.cuda.stream(s):
                g = torch.Generator(device="cpu").manual_seed(args.seed + d)
                bad = torch.randperm(B, generator=g)[:n_t].to(dev)
                b.view(-1, p.ballot_size)[bad, p.ballot_size - 32] ^= 1
        ctxs.append(c); params.append(p); ballots.append(b); streams.append(s); counts.append(B)
        status.append(torch.empty(max(B, 1), dtype=torch.int32, device=dev))
    for dev_i in set(devs):
        torch.cuda.synchronize(dev_i)
    gen_s = time.time() - t0
    ptr_b, ptr_s, ptr_q = [x.data_ptr() for x in ballots], [x.data_ptr() for x in status], [x.cuda_stream for x in streams]
    merged = [None]
    host = host_status = None
    if args.from_host:
        # the WHOLE batch in ONE pinned host buffer, slab after slab in ballot order (what examples/voting.rs:179-213 holds after reading its
        # ballots): eg_verify_*_batch_multi cuts it into the same contiguous slabs and uploads each to its GPU inside the library
        sz = params[0].ballot_size
        host = torch.empty(max(total, 1) * sz, dtype=torch.uint8, pin_memory=True)
        host_status = torch.empty(max(total, 1), dtype=torch.int32, pin_memory=True)
        at = 0
        for b, c in zip(ballots, counts):
            host[at * sz : (at + c) * sz].copy_(b[: c * sz])
            at += c
        for dev_i in set(devs):
            torch.cuda.synchronize(dev_i)

    def step():
        for p in params:
            p.tally_reset()
        if args.from_host:
            eg.verify_batch_multi_host_ptr(params, total, host.data_ptr(), host_status.data_ptr())
        else:
            eg.verify_batch_multi_device(params, counts, ptr_b, ptr_s, ptr_q)
        merged[0] = eg.tally_encode_multi(params)

    for _ in range(args.warmup):
        step()
    sampler = ClockSampler(torch, devs[0])
    sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for dev_i in set(devs):
        torch.cuda.synchronize(dev_i)
    elapsed = time.perf_counter() - t0
    clock = sampler.stop()
    if args.from_host:
        accepted = int((host_status[:total] == 0).sum().item())
    else:
        accepted = sum(int((st[:c] == 0).sum().item()) for st, c in zip(status, counts))
    n_tampered = sum(int(c * args.tampered_percent / 100.0) for c in counts)
    # the merged tally is the sum of the per-GPU running tallies, each of which is the tally of its own slab: re-merge them with the
    # primitive tier one by one, and (one GPU's worth of work, untimed) let ONE engine verify every slab and compare its running tally
    grp = eg.Ristretto(ctxs[0])
    again = bytes(64 * n_opt)
    for p in params:
        again = grp.element_add(again, p.tally_encode())[0]
    tally_ok = merged[0] == again and accepted == total - n_tampered
    check_one_engine = total <= 4_000_000
    if check_one_engine:
        one = params[0]
        one.tally_reset()
        scratch = torch.empty(max(counts) if counts else 1, dtype=torch.int32, device=torch.device("cuda", devs[0]))
        for d, (b, c) in enumerate(zip(ballots, counts)):
            src = b if devs[d] == devs[0] else b.to(torch.device("cuda", devs[0]))
            torch.cuda.set_device(devs[0])
            one.verify_batch_device(c, src.data_ptr(), scratch.data_ptr(), 0)
        tally_ok = tally_ok and one.tally_encode() == merged[0]
    out = {
        "metric": "EncryptedChoice ballot verifications/sec" if args.workload != "qv" else "QuadraticVotingBallot verifications/sec",
        "value": total * args.steps / elapsed, "unit": "ballots/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"{counts[0]} {args.workload} {n_opt}-option ballots per GPU per step, resident in HBM; verify + homomorphic tally "
                               f"+ in-library tally merge over {N} GPUs of ONE process (eg_verify_*_batch_multi_device)",
                   "ballots_per_gpu": counts[0], "total_ballots": total, "options": n_opt, "ballot_bytes": params[0].ballot_size,
                   "seed": args.seed, "accepted": accepted, "tampered": n_tampered, "tally_exchange_ok": bool(tally_ok),
                   "tally_checked_against_one_engine": check_one_engine, "generator_s": round(gen_s, 3),
                   "parallelism": f"in-process{N}" + ("-on-one-gpu" if args.rehearse_one_gpu else ""), "devices": devs,
                   "input": "host" if args.from_host else "hbm"},
        "clock": clock,
    }
    if args.from_host:
        out["config"]["workload"] = (f"{total} {args.workload} {n_opt}-option ballots in ONE pinned host buffer per step; eg_verify_*_batch_multi cuts it "
                                     f"into {N} contiguous slabs, uploads and verifies each on its GPU (one host thread per GPU inside the library), "
                                     "merges the tallies: PCIe-INCLUSIVE, not the headline metric")
        out["host_inclusive"] = {"value": out["value"], "unit": "ballots/s", "bytes_h2d": total * params[0].ballot_size, "bytes_d2h": 4 * total,
                                 "pinned": True, "h2d_gb_per_s": total * params[0].ballot_size * args.steps / elapsed / 1e9,
                                 "note": "`value` of this line IS the PCIe-inclusive rate (first H2D byte to last status byte D2H, every step)"}
    # the JSON text of the whole batch through the multi-GPU JSON entry (one parser, its packed windows dealt to the N params objects)
    if not args.no_wire_ingest:
        try:
            out["json_inclusive"] = in_process_json_leg(args, eg, torch, params, ballots, status, counts, total, n_opt, out["value"])
        except Exception as e:            # an extra leg must never cost the line
            out["json_inclusive"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    if not tally_ok:
        raise SystemExit(5)


def main():
    args = parse()
    if args.in_process_devices:
        return bench_in_process(args)
    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))           # nothing above this line has imported torch or loaded the library: HIP is untouched
    import torch
    import torch.distributed as dist

    import elastic_elgamal_amd as eg
    from elastic_elgamal_amd import distributed as egd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: under torch.distributed.run --gpus must equal the number of ranks "
                         "(started bare, `python bench.py --gpus N` launches the N ranks itself)")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one explicit stream for everything: the library's launches, torch's copies and RCCL's collective are ordered on it
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    use_dist = world > 1 or args.force_dist
    if args.force_dist:
        egd.ALWAYS_COLLECTIVE = True           # the helpers then run their collectives even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        try:
            if args.rehearse_one_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        except Exception as e:      # a rank that cannot join must end the job now, with a message, not hang the others in a collective
            die(2, f"torch.distributed init failed ({'gloo' if args.rehearse_one_gpu else 'nccl = RCCL'}): {e!r}")
        # Preflight of the one collective of the path, before anything is generated or timed: an all-gather of `world` x 320 uint8 with
        # known contents through the very helper the steps use (all_gather_into_tensor on the current stream under RCCL).  A wrong
        # dtype / layout / transport (e.g. no dmabuf IPC) shows here as an exception or as wrong bytes, in a process that has done
        # nothing else yet.
        try:
            probe = torch.full((320,), rank + 1, dtype=torch.uint8, device=dev)
            got = egd.gather_tallies(probe)
            torch.cuda.synchronize()
            want = torch.arange(1, world + 1, dtype=torch.uint8, device=dev).view(world, 1).expand(world, 320)
            wrong = got.dtype != torch.uint8 or tuple(got.shape) != (world, 320) or not torch.equal(got, want)
            if egd.sum_over_ranks(1 if wrong else 0, dev):        # every rank learns of it and ends here, none is left waiting in a collective
                die(3, "tally all-gather preflight returned wrong data" + (f" on this rank: dtype {got.dtype}, shape {tuple(got.shape)}" if wrong else " on another rank"))
        except SystemExit:
            raise
        except Exception as e:
            die(3, f"tally all-gather preflight failed: {e!r}")

    ctx = eg.Context(local_rank)
    pk = bytes.fromhex(PUBLIC_KEY_HEX)
    if args.workload == "msm":
        return bench_msm(args, ctx, eg, torch, dev, world)
    n_opt = args.options or {"single": 5, "multi": 16, "qv": 5}[args.workload]
    if args.workload == "single":
        params = eg.ChoiceParams.single_choice(ctx, pk, n_opt)
    elif args.workload == "multi":
        params = eg.ChoiceParams.multi_choice(ctx, pk, n_opt)
    else:
        params = eg.QuadraticVotingParams(ctx, pk, n_opt, args.credits)
    strong = args.total_ballots > 0
    total = args.total_ballots if strong else args.ballots * world
    first, last = egd.shard_range(total, rank, world)     # rank r owns the contiguous slab of voters [first, last)
    B = last - first
    stream = torch.cuda.current_stream().cuda_stream

    # ---- synthetic input, generated on the GPU and left resident in HBM ------------------------------------
    ballots = torch.empty(B * params.ballot_size, dtype=torch.uint8, device=dev)
    t0 = time.time()
    if args.workload == "multi":
        params.encrypt_batch_device(args.seed, first, B, ballots.data_ptr(), n_selected=3, stream=stream)
    else:
        params.encrypt_batch_device(args.seed, first, B, ballots.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    n_tampered = int(B * args.tampered_percent / 100.0)
    if n_tampered:
        g = torch.Generator(device="cpu").manual_seed(args.seed + rank)
        bad = torch.randperm(B, generator=g)[:n_tampered].to(dev)
        ballots.view(B, params.ballot_size)[bad, params.ballot_size - 32] ^= 1   # last response scalar stays canonical
    status = torch.empty(B, dtype=torch.int32, device=dev)
    local_tally = torch.empty(64 * n_opt, dtype=torch.uint8, device=dev)
    final_tally = torch.empty(64 * n_opt, dtype=torch.uint8, device=dev)
    bad_terms = torch.zeros(1, dtype=torch.int32, device=dev)   # gathered encodings that failed to decode (must stay 0)

    # HIP events around the ONE exchange of a step (tally encode -> all-gather -> k_points_sum), on the stream all of it is enqueued on:
    # a scaling point that comes out sub-linear can then be told apart (verification, exchange, or a slow rank: `per_rank`, `exchange`)
    ex_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 1))]

    def step(k=None):
        params.tally_reset(stream)
        params.verify_batch_device(B, ballots.data_ptr(), status.data_ptr(), stream)
        if k is not None:
            ex_ev[k][0].record()
        params.tally_encode_device(local_tally.data_ptr(), stream)
        gathered = egd.gather_tallies(local_tally)            # the ONE collective (RCCL all-gather, 64*n bytes/rank)
        ctx.points_sum_device(gathered.shape[0], 2 * n_opt, gathered.data_ptr(), final_tally.data_ptr(), stream,
                              d_bad=bad_terms.data_ptr())
        if k is not None:
            ex_ev[k][1].record()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def exchange_consistent() -> bool:
        """The tally that went through encode -> all-gather -> k_points_sum is the engine's own tally (one rank) / identical on
        every rank (several ranks) and no gathered encoding failed to decode: proves the stream ordering of the exchange."""
        torch.cuda.synchronize()
        exchanged = bytes(final_tally.cpu().numpy())
        if world == 1 and not use_dist:
            ok = exchanged == params.tally_encode()
        else:
            digest = float(int.from_bytes(__import__("hashlib").sha256(exchanged).digest()[:6], "big"))   # exact in a double
            # every rank runs every collective, whatever it has seen so far (a rank that skipped one would hang the others)
            hi = egd.max_over_ranks(digest, dev)
            lo = -egd.max_over_ranks(-digest, dev)
            ok = hi == digest == lo
            if world == 1:
                ok = ok and exchanged == params.tally_encode()
        bad = egd.sum_over_ranks(int(bad_terms.item()), dev)
        verdicts = egd.sum_over_ranks(0 if ok else 1, dev)       # ... and every rank learns whether ANY rank disagreed
        return bool(ok and bad == 0 and verdicts == 0)

    for _ in range(args.warmup):
        step()
    if use_dist:
        # One untimed step whose exchanged tally is checked on every rank BEFORE the timed loop: a job whose ranks disagree (or that
        # gathered an encoding that does not decode) ends here with a non-zero exit code instead of printing a number.
        if not args.warmup:
            step()
        if not exchange_consistent():
            die(4, "the exchanged tally differs between ranks or holds an undecodable encoding (checked before the timed steps)")
    if use_dist:
        # RCCL's version banner (NCCL_DEBUG=VERSION) sits in the C stdio buffer of every rank: push it out now so that the
        # JSON line below stays the last line of the job's stdout
        barrier()
        import ctypes
        ctypes.CDLL(None).fflush(None)
    # ---- the VALU roof of THIS box (every rank, at the same time: the GPUs of a node share its power budget and cooling), after the
    # warm-up and outside the timed region: the shipped fe_mul in a bare chain for --selfbench-seconds (eg_selfbench_fmul)
    dev_index = dev.index if dev.index is not None else 0
    box = None
    if args.selfbench_seconds > 0:
        barrier()
        bs = ClockSampler(torch, dev_index)
        bs.start()
        try:
            box_g, box_mhz = ctx.selfbench_fmul(args.selfbench_seconds)
            bc = bs.stop()
            box = {"fmul_sustained_g": box_g, "sclk_mhz": box_mhz, "power_w": bc["power_w"] if bc else None,
                   "sclk_mhz_hwmon": bc["sclk_mhz"] if bc else None, "seconds": args.selfbench_seconds, "waves_per_simd": 3,
                   "source": "eg_selfbench_fmul: the shipped fe_mul in a bare dependent chain, launched back to back; rate and clock "
                             "(s_memtime / s_memrealtime) over the second half of the run, on the GPU and in the process that runs the bench"}
        except eg.EgError as e:             # a calibration must never cost the line
            bs.stop()
            box = {"error": repr(e)}
    ctx.profile_enable(True)
    ctx.profile_read()
    ctx.profile_read_tables()
    sampler = ClockSampler(torch, dev_index)       # every rank samples its own GPU (`per_rank`); rank 0's is the line's `clock`
    barrier()
    sampler.start()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0            # this rank's own steps, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    clock = sampler.stop()
    exchange_ms = sum(a.elapsed_time(b) for a, b in ex_ev[: args.steps]) / max(args.steps, 1)
    msm_ms, msm_launches, all_ms = ctx.profile_read()
    tables_ms, tables_launches = ctx.profile_read_tables()
    # one more, untimed step with the chunks run one after the other on one work set: in the timed steps a launch shares the chip with the
    # other work set's kernels (that is what the two streams are for), so its duration is not its cost
    iso_ms, iso_launches, iso_tables_ms, iso_tables_launches = 0.0, 0, 0.0, 0
    if not args.no_isolated:
        ctx.profile_enable(2)
        step()
        barrier()
        iso_ms, iso_launches, _ = ctx.profile_read()
        iso_tables_ms, iso_tables_launches = ctx.profile_read_tables()
    ctx.profile_enable(False)
    elapsed = egd.max_over_ranks(elapsed, dev)

    accepted = int((status == 0).sum().item())
    tally_ok = exchange_consistent()          # again after the timed steps (and the only check of a one-rank run)
    accepted_all = egd.sum_over_ranks(accepted, dev)
    value = total * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- legs that EVERY rank runs at the same time when there are several (the scaling risk is host-side staging - N x 736 MB
    # through one host's memory and PCIe root - not the 320-byte collective): the PCIe-inclusive rate of each rank's own slab from a pinned
    # host buffer, and the JSON text of it through the native parser with the host's cores divided among the ranks
    multi = world > 1
    hi_all = js_all = None
    if multi and not args.no_host_inclusive:
        hi_all = host_inclusive_leg(torch, params, ballots, status, B, value / world, before_each=barrier)
    if multi and not args.no_wire_ingest:
        js_all, _ = json_inclusive_leg(args, eg, torch, params, ballots, status, B, n_opt, value / world,
                                       threads=max(1, effective_cores() // world), before_each=barrier)
    if multi or use_dist:
        rows = egd.gather_rows([rank, dev_index, own_elapsed / args.steps * 1e3, clock["sclk_mhz"] if clock else 0.0,
                                clock["power_w"] if clock else 0.0, accepted, exchange_ms * 1e3,
                                (box or {}).get("fmul_sustained_g") or 0.0, (box or {}).get("sclk_mhz") or 0.0,
                                hi_all["value"] if hi_all else -1.0, 1.0 if hi_all and hi_all["verdicts_match_device_path"] else 0.0,
                                js_all["value"] if js_all else -1.0, 1.0 if js_all and js_all["verdicts_match_device_path"] else 0.0], dev)
    else:
        rows = None

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        if not tally_ok:
            raise SystemExit(5)
        return

    # ---- roofline of the dominant kernel (k_eq_table<false>), HIP events on the launch stream ------------------------
    kind = {"single": "single", "multi": "multi", "qv": "qv"}[args.workload]
    desc = eg.plan_describe(kind, n_opt, args.credits if args.workload == "qv" else 0)
    n_stages = desc["stages"]   # one launch of the dominant kernel per stage and chunk
    dominant = DOMINANT_KERNEL.format(teeth=desc.get("teeth", 6))
    launches_per_step = max(1, msm_launches // max(args.steps, 1))
    avg_launch_ms = msm_ms / max(msm_launches, 1)
    n_chunks = max(1, launches_per_step // n_stages)
    units_per_launch = -(-B // n_chunks)
    chunk = units_per_launch
    alg_bytes = params.ballot_size + 4
    achieved_gbs = alg_bytes * units_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    # memory-side bytes per ballot and launch of the dominant kernel, measured by the PMC passes of the last profile round
    traffic_src = None
    try:
        sys.path.insert(0, str(ROOT / "tools"))
        from srchash import code_hash                   # comments and white space do not count
        tree_hash = code_hash(ROOT)
    except (OSError, ImportError):
        tree_hash = None
    try:
        tj = json.loads(TRAFFIC_JSON.read_text())
        key = f"{args.workload}-{n_opt}" + (f"-{args.credits}" if args.workload == "qv" else "")
        ent = tj["workloads"][key]["kernels"][dominant]
        traffic_bpbl = float(ent["bytes_per_ballot_launch"])
        traffic_src = {"file": "profiles/traffic.json", "round": tj.get("round"), "commit": tj.get("commit"),
                       "ballots_per_launch": tj["workloads"][key].get("ballots_per_launch"), "source_hash": tj.get("source_hash"),
                       "tree_hash": tree_hash}
    except (OSError, KeyError, ValueError):
        traffic_bpbl = None
    iso_units = -(-B // max(1, iso_launches // n_stages)) if iso_launches else 0
    iso_achieved = alg_bytes * iso_units / (iso_ms / iso_launches * 1e-3) / 1e9 if iso_launches else 0.0
    out = {
        "metric": "EncryptedChoice ballot verifications/sec" if args.workload != "qv" else "QuadraticVotingBallot verifications/sec",
        "value": value,
        "unit": "ballots/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": {
                "single": f"{B} single-choice {n_opt}-option EncryptedChoice ballots per GPU per step, resident in HBM "
                          "(BASELINE.json configs[1]); verify + homomorphic tally + tally all-gather",
                "multi": f"{B} multi-choice 3-of-{n_opt} EncryptedChoice ballots per GPU per step (BASELINE.json configs[3])",
                "qv": f"{B} QuadraticVotingBallot ballots, {n_opt} options / {args.credits} credits (BASELINE.json configs[2])",
            }[args.workload],
            "ballots_per_gpu": B,
            "total_ballots": total,
            "options": n_opt,
            "ballot_bytes": params.ballot_size,
            "chunk_ballots": chunk,
            "seed": args.seed,
            "accepted": accepted_all,
            "tally_exchange_ok": bool(tally_ok),
            "tampered": n_tampered * world,
            "generator_s": round(gen_s, 3),
            "parallelism": f"shard{world}" if world > 1 else "single",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dominant,
            # headline figures = the kernel ALONE on the chip (`isolated` below: one extra untimed step, chunks one after the other on
            # one work set); in the timed steps a launch shares the CUs with the other work set's kernels, so its duration there
            # (`*_concurrent`) is not its cost.  Without the extra step (--no-isolated) the concurrent figures stand in.
            "achieved": iso_achieved if iso_launches else achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (iso_achieved if iso_launches else achieved_gbs) / HBM_PEAK_GBS,
            "mode": "isolated" if iso_launches else "concurrent",
            "achieved_concurrent": achieved_gbs,
            "frac_concurrent": achieved_gbs / HBM_PEAK_GBS,
            "traffic": (traffic_bpbl * units_per_launch / (avg_launch_ms * 1e-3) / 1e9
                        if traffic_bpbl is not None and avg_launch_ms > 0 else None),
            "traffic_bytes_per_launch": traffic_bpbl * units_per_launch if traffic_bpbl is not None else None,
            "traffic_source": traffic_src,
            "traffic_stale": bool(traffic_src is None or traffic_src.get("source_hash") != tree_hash),   # counters measured on another build of csrc/
            "traffic_note": "GB/s like `achieved`: memory-side bytes per ballot and launch of this kernel from the separate rocprofv3 "
                            "PMC passes of the last profile round (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; "
                            "tools/profile_round.sh -> profiles/traffic.json) x this run's ballots per launch / this run's "
                            "launch time; it is the per-ballot table lookups, not the ballots",
            "avg_launch_ms": iso_ms / iso_launches if iso_launches else avg_launch_ms,
            "avg_launch_ms_concurrent": avg_launch_ms,
            "isolated": ({"avg_launch_ms": iso_ms / iso_launches, "launches": iso_launches,
                          "units_per_launch": -(-B // max(1, iso_launches // n_stages)),
                          "achieved": alg_bytes * -(-B // max(1, iso_launches // n_stages)) / (iso_ms / iso_launches * 1e-3) / 1e9,
                          "frac": alg_bytes * -(-B // max(1, iso_launches // n_stages)) / (iso_ms / iso_launches * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "second_kernel_avg_launch_ms": iso_tables_ms / max(iso_tables_launches, 1),
                          "note": "one extra untimed step with the chunks run one after the other on ONE work set (eg_profile_enable 2): "
                                  "the same kernel without the other stream's kernels beside it"} if iso_launches else None),
            "launches_per_step": launches_per_step,
            "units_per_launch": units_per_launch,
            "kernel_share_of_step": msm_ms / max(all_ms, 1e-9),
            "second_kernel": {"kernel": f"eg::k_base_tables<{desc.get('teeth', 6)}>", "avg_launch_ms": tables_ms / max(tables_launches, 1),
                              "launches_per_step": tables_launches // max(args.steps, 1),
                              "share_of_step": tables_ms / max(all_ms, 1e-9)},
            "note": "modular-integer VALU work: ~740 algorithmic bytes but ~4.4e4 field multiplications per ballot, so the "
                    "HBM fraction is ~1e-3 by construction (SURVEY 0.6); the binding roof is VALU integer multiply-add "
                    "throughput, see DESIGN.md",
        },
    }

    if True:
        narrow_bits, wide_bits = ctx.comb_table_bits()        # wide comb tables exist once an engine has seen 2^19 items
        fm, fs = plan_field_ops(desc, wide_combs=wide_bits != 0)
        mads = fm * MAD_PER_MUL + fs * MAD_PER_SQ
        out["config"]["comb_bits"] = wide_bits or narrow_bits
        out["valu_roofline"] = {
            "bound": "valu-int-mad",
            "fe_mul_per_ballot": fm,
            "fe_sq_per_ballot": fs,
            "mad_per_ballot": mads,
            "achieved": value / world * mads / 1e12,
            "peak": MAD_PEAK_T,
            "unit": "T v_mad_u64_u32 lane-ops/s per GPU",
            "frac": value / world * mads / 1e12 / MAD_PEAK_T,
            "fmul_equiv_frac": value / world * (fm + SQ_WEIGHT * fs) / 1e9 / FMUL_PEAK_G,
            "fmul_equiv_frac_sustained": value / world * (fm + SQ_WEIGHT * fs) / 1e9 / FMUL_SUSTAINED_G,
            "note": "the multiply-adds of the field operations only (no carries, adds, selects, hashing); peaks measured on this chip: the "
                    "instruction alone at 8 waves per SIMD, and the field multiplication in a bare chain, in a burst (fmul_equiv_frac) and "
                    "sustained for seconds (fmul_equiv_frac_sustained); the *_at_sclk fractions scale the burst peaks to the clock this "
                    "run held",
        }
        if clock:
            vr = out["valu_roofline"]
            vr["frac_at_sclk"] = vr["frac"] * MAD_PEAK_SCLK_MHZ / clock["sclk_mhz"]
            vr["fmul_equiv_frac_at_sclk"] = vr["fmul_equiv_frac"] * FMUL_PEAK_SCLK_MHZ / clock["sclk_mhz"]
    if clock:
        out["clock"] = clock

    # ---- per-rank figures and the exchange of a step on its own (N > 1, or one rank with --force-dist) -----------------------------------
    if rows is not None:
        keys = ("rank", "device", "ms_per_step", "sclk_mhz", "power_w", "accepted", "exchange_us_per_step", "fmul_box_g", "fmul_box_sclk_mhz",
                "host_inclusive_value", "host_inclusive_ok", "json_inclusive_value", "json_inclusive_ok")
        out["per_rank"] = [{k: (int(v) if k in ("rank", "device", "accepted") else (bool(v) if k.endswith("_ok") else v))
                            for k, v in zip(keys, r)} for r in rows]
        for pr in out["per_rank"]:          # a leg that did not run on every rank at once (one rank: it runs below, on its own) is null, not zero
            for leg in ("host_inclusive", "json_inclusive"):
                if pr[leg + "_value"] < 0:
                    pr[leg + "_value"] = pr[leg + "_ok"] = None
        out["exchange"] = {"us_per_step": max(r[6] for r in rows), "us_per_step_rank0": exchange_ms * 1e3,
                           "bytes_per_rank": 64 * n_opt, "bytes_gathered_per_rank": 64 * n_opt * world,
                           "share_of_step": max(r[6] for r in rows) / 1e3 / ms_per_step,
                           "note": "HIP events on the step's stream around eg_*_tally_encode_device -> all-gather of the 64 n-byte tallies "
                                   "(RCCL; gloo through host memory in a one-GPU rehearsal) -> eg_points_sum_device; max over ranks: it "
                                   "includes the wait for the slowest rank to reach the collective"}
        out["slowest_rank"] = max(out["per_rank"], key=lambda r: r["ms_per_step"])["rank"]
    if box:
        out["valu_roofline"]["box"] = box
        if box.get("fmul_sustained_g"):
            vr = out["valu_roofline"]
            vr["fmul_equiv_frac_sustained_box"] = value / world * (vr["fe_mul_per_ballot"] + SQ_WEIGHT * vr["fe_sq_per_ballot"]) / 1e9 / box["fmul_sustained_g"]

    # ---- PCIe-inclusive rate (first H2D byte to last status byte D2H): the same batch from a pinned host buffer
    # through the host-pointer entry point (pipelined uploads, eg_verify_*_batch).  Reported beside `value`, never as it.
    if hi_all is not None:             # several ranks: every rank ran it on its own slab, all at the same time
        out["host_inclusive"] = dict(hi_all)
        vals = [r[9] for r in rows]
        out["host_inclusive"]["all_ranks"] = {"sum_value": sum(vals), "min_value": min(vals), "max_value": max(vals), "ranks": world,
                                              "verdicts_match_device_path": all(r[10] == 1.0 for r in rows), "vs_value": sum(vals) / value,
                                              "note": "every rank verifies its own slab from its own pinned host buffer through eg_verify_*_batch, "
                                                      "all ranks at the same time (a barrier before each iteration): the sum is the "
                                                      "PCIe-inclusive rate of the node through ONE host"}
    elif not args.no_host_inclusive and world == 1:
        out["host_inclusive"] = host_inclusive_leg(torch, params, ballots, status, B, value)
    if js_all is not None:
        out["json_inclusive"] = dict(js_all)
        vals = [r[11] for r in rows]
        out["json_inclusive"]["all_ranks"] = {"sum_value": sum(vals), "min_value": min(vals), "max_value": max(vals), "ranks": world,
                                              "threads_per_rank": js_all["threads"], "verdicts_match_device_path": all(r[12] == 1.0 for r in rows),
                                              "vs_value": sum(vals) / value,
                                              "note": "every rank parses and verifies the JSON text of its own slab (eg_verify_*_json) at the same "
                                                      "time, the host's cores divided among the ranks"}

    # ---- wire ingest: the same ballots as JSON text in serde's layout through the native packer -------------
    def wire_legs():
        """wire_ingest, json_inclusive and json_stream of a one-GPU line; a closure over the batch, so that a failure in an extra leg costs
        only that leg (the line is printed with {"error": ...} in its place)."""
        from elastic_elgamal_amd import serde as egserde
        import ctypes
        import numpy as np

        distinct, raw, one, build = ballots_as_json(args, eg, torch, params, ballots, B, n_opt, None)
        reps = max(1, min(100, B // distinct))
        text = build(reps)
        n_obj = distinct * reps
        cores = effective_cores()
        kw = {"credits": args.credits} if args.workload == "qv" else {"single": args.workload == "single"}
        by_threads = {}
        best, got, native = None, 0, None
        for th in sorted({t for t in (4, 8, 16, 32, 64) if t < cores} | {cores}):      # the packer by thread count, up to what the box grants
            native = eg.JsonPacker(n_opt, n_obj, threads=th, **kw)   # caller-owned output buffers, reused: the C call is what is timed
            b_th = None
            for _ in range(4 if th == cores else 2):
                t0 = time.perf_counter()
                got = native.pack(text)
                dt = time.perf_counter() - t0
                b_th = dt if b_th is None else min(b_th, dt)
            by_threads[str(th)] = n_obj / b_th
            best = b_th                      # the last one is `cores`
        packed = native.packed.raw[: distinct * params.ballot_size]
        st = list(native.status[:got])
        t0 = time.perf_counter()
        packer = egserde.pack_qv_ballot if args.workload == "qv" else egserde.pack_encrypted_choice
        ref = b"".join(packer(o) for o in json.loads(text[: 1 + sum(len(x) + 1 for x in one) - 1].decode() + "]"))
        py_s = time.perf_counter() - t0
        jreps = max(1, B // distinct)
        jtext = text if jreps == reps else build(jreps)
        jn = distinct * jreps
        out["json_inclusive"], jarr = json_inclusive_leg(args, eg, torch, params, ballots, status, B, n_opt, value, threads=cores, text=jtext,
                                                         distinct=distinct)
        # the same text through the STREAMING entry (eg_verify_json_begin / _feed / _end): pieces of 64 MB (read in place), of 1 MB copied
        # by feed on the caller's thread, and of 1 MB handed over without a copy (eg_verify_json_feed_owned)
        if jarr is None:
            raise RuntimeError("json_inclusive failed: " + str(out["json_inclusive"].get("error")))
        piece_rates = {}
        jbase = ctypes.cast(ctypes.c_char_p(jtext), ctypes.c_void_p).value
        for label, piece, owned in (("64MB", 64 << 20, False), ("1MB", 1 << 20, False), ("1MB_owned", 1 << 20, True)):
            best_s, ok = None, True
            for _ in range(3):
                sstatus = (ctypes.c_uint32 * jn)()
                t0 = time.perf_counter()
                js = params.json_stream(threads=cores)
                for at in range(0, len(jtext), piece):
                    (js.feed_owned_ptr if owned else js.feed_ptr)(jbase + at, min(piece, len(jtext) - at))
                taken, _ = js.end_into(sstatus)
                dt = time.perf_counter() - t0
                best_s = dt if best_s is None else min(best_s, dt)
                ok = ok and taken == jn and np.array_equal(np.frombuffer(sstatus, dtype=np.uint32), jarr)
            piece_rates[label] = {"value": jn / best_s, "ms": best_s * 1e3, "vs_value": jn / best_s / value, "verdicts_match_one_shot": bool(ok)}
        out["json_stream"] = {"unit": "ballots/s", "objects": jn, "threads": cores, "pieces": piece_rates,
                              "note": "eg_verify_json_begin / _feed / _end on the same text as json_inclusive, fed in pieces of the given size; "
                                      "1MB_owned: eg_verify_json_feed_owned (the library reads the caller's block in place and calls its "
                                      "release function when the worker is through with it: no copy on the caller's thread)"}
        del jtext
        out["wire_ingest"] = {"value": n_obj / best, "unit": "ballots/s", "threads": cores, "json_bytes": len(text), "objects": n_obj,
                              "json_mb_per_s": len(text) / best / 1e6, "all_packed": st.count(0) == n_obj,
                              "equals_device_ballots": got == n_obj and packed == raw == ref,
                              "by_threads": by_threads,
                              "python_mirror_value": distinct / py_s,
                              "note": "JSON text (serde layout, base64url) -> packed bytes on the host, before the PCIe-inclusive path above; "
                                      "by_threads: the same call with fewer parser threads (up to the hardware threads this process may use)"}


    if not args.no_wire_ingest and world == 1:
        try:
            wire_legs()
        except Exception as e:            # an extra leg must never cost the line
            out.setdefault("wire_ingest", {"error": repr(e)})

    # ---- CPU baseline: the oracle ("port": CPU restatement, not curve25519-dalek) on a bounded sample --------------
    if not args.no_cpu_baseline:          # at every N, on rank 0 (the other ranks have nothing left to do: the host's cores are rank 0's)
        from oracle import oracle as o

        cores = effective_cores()
        op = (o.QvParams(pk, n_opt, args.credits) if args.workload == "qv"
              else o.ChoiceParams(pk, n_opt, args.workload == "single"))
        # bounded sample: verify slices of the same batch until ~cpu_seconds of wall time have been spent
        slice_n = max(64, 32 * cores)
        cpu_status, cpu_s, sample = [], 0.0, 0
        while cpu_s < args.cpu_seconds and sample < B:
            m = min(slice_n, B - sample)
            host = bytes(ballots[sample * params.ballot_size : (sample + m) * params.ballot_size].cpu().numpy())
            t0 = time.perf_counter()
            cpu_status += op.verify_batch(host, threads=cores)
            cpu_s += time.perf_counter() - t0
            sample += m
        one = min(sample, 256)
        host = bytes(ballots[: one * params.ballot_size].cpu().numpy())
        t0 = time.perf_counter()
        op.verify_batch(host, threads=1)
        one_s = time.perf_counter() - t0
        gpu_status = status[:sample].cpu().tolist()
        out["cpu_baseline"] = {
            "value": sample / cpu_s,
            "unit": "ballots/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {sample} ballots of the same batch, oracle/ C restatement (not curve25519-dalek), "
                      f"{cores} threads, {cpu_s:.1f} s",
            "single_thread_value": one / one_s,
            "cpu_model": cpu_model(),
            "verdicts_match_gpu": cpu_status == gpu_status,
        }
    # ---- the other BASELINE configs, after the headline measurement and outside its timed region (VERDICT r3 task 6) ------------
    if world == 1 and not use_dist and not args.no_extra_configs and args.workload == "single" and not strong:
        del ballots, status
        params.close()                         # frees the chunk workspace and the key's comb tables before the next election's are made
        torch.cuda.empty_cache()
        args.box_fmul_g = (box or {}).get("fmul_sustained_g")
        out["extra"] = {"configs": extra_configs(args, ctx, eg, torch, dev, pk, stream),
                        "note": "10 timed steps each (3 for the 10 M batch; 1 warm-up), same process and GPU, after the headline measurement; value "
                                "in ballots/s (msm: vartime_double_mul_generator operations/s); NOT part of `value`"}
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if not tally_ok:
        raise SystemExit(5)          # the line is printed (tally_exchange_ok: false), the job still fails


if __name__ == "__main__":
    main()
