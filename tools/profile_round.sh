#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines and rocprofv3 passes of one round, written under gpurun_out/.
#   usage: tools/profile_round.sh [pmc [only]]      (without "pmc": bench lines + kernel-trace stats only; "pmc only": the counter passes alone)
# Counters are collected in their own passes (--pmc never combined with tracing), the program after `--` is python3 itself.
# tools/profile_summary.py <round> then turns the outputs into profiles/<round>_*.txt and profiles/traffic.json.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PMC_BALLOTS=1000000     # the size the bench line quotes: launches of the same 2^18-ballot chunks
if [ "$1" = "pmc" ]; then
# counters FIRST, then profiles/traffic.json is rewritten on the box from them (with the hash of this tree), so that the bench lines below
# print a roofline.traffic that belongs to the build they measure ("traffic_stale": false)
(
export EG_COMB_BIG_MIN=1     # one step without warm-up: give it the wide comb tables that the timed steps of the bench line use
for w in single multi qv; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${c}_$w
    timeout -k 10 300 rocprofv3 --pmc $c -d gpurun_out/pmc_${c}_$w -o pmc --output-format csv -- \
      python3 bench.py --steps 1 --warmup 0 --workload $w --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --no-extra-configs --selfbench-seconds 0 --ballots $PMC_BALLOTS > gpurun_out/pmc_${c}_$w.log 2>&1 || exit 1
  done
done
rm -rf gpurun_out/pmc_SQ1_single gpurun_out/pmc_SQ2_single
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES -d gpurun_out/pmc_SQ1_single -o pmc --output-format csv -- \
  python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --no-extra-configs --selfbench-seconds 0 --ballots $PMC_BALLOTS > gpurun_out/pmc_SQ1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVES -d gpurun_out/pmc_SQ2_single -o pmc --output-format csv -- \
  python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --no-extra-configs --selfbench-seconds 0 --ballots $PMC_BALLOTS > gpurun_out/pmc_SQ2.log 2>&1 || exit 1
) || exit 1
python3 tools/profile_summary.py --traffic-only > gpurun_out/traffic_summary.log 2>&1 || exit 1
fi
# "pmc-only": the counter passes alone (one gpurun call has 20 minutes); run `python3 tools/profile_summary.py --traffic-only rNN` at home
# afterwards - the CSVs come back under gpurun_out/ - and the next call's bench lines find the traffic of their own build in profiles/traffic.json
if [ "$2" = "only" ]; then echo "counter passes done"; exit 0; fi
for w in single multi qv; do
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || exit 1
  tail -c 300 gpurun_out/bench_$w.json; echo
done
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --total-ballots 10000000 > gpurun_out/bench_10M.json 2> gpurun_out/bench_10M.err || exit 1
timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --tampered-percent 1 > gpurun_out/bench_tampered1pct.json 2> gpurun_out/bench_tampered.err || exit 1
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --workload msm > gpurun_out/bench_msm.json 2> gpurun_out/bench_msm.err || exit 1
for w in single multi qv; do
  rm -rf gpurun_out/prof_stats_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_$w -o stats --output-format csv -- \
    python3 bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --no-extra-configs --selfbench-seconds 0 > gpurun_out/prof_stats_$w.log 2>&1 || exit 1
done
# the same with ONE work set (EG_STREAMS=1): chunks one after the other on one stream, so that a kernel's duration is its cost
for w in single multi qv; do
  rm -rf gpurun_out/prof_serial_$w
  EG_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serial_$w -o stats --output-format csv -- \
    python3 bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline --no-host-inclusive --no-wire-ingest --no-isolated --no-extra-configs --selfbench-seconds 0 > gpurun_out/prof_serial_$w.log 2>&1 || exit 1
done
# round 5: the probes behind DESIGN's paragraphs on memory-side watts, the JSON stream, the multi-scalar multiplication by size, and the
# in-process multi-GPU leg rehearsed with two contexts on this one GPU (its value means nothing: two engines time-share the chip)
timeout -k 10 120 python3 tools/hbm_power_probe.py > gpurun_out/hbm_power_probe.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/json_stream_probe.py > gpurun_out/json_stream_probe.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/json_trace_probe.py > gpurun_out/json_trace.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/msm_probe.py > gpurun_out/msm_by_size.txt 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --in-process-devices 2 --rehearse-one-gpu --steps 3 --warmup 1 --ballots 500000 > gpurun_out/bench_in_process2.json 2> gpurun_out/bench_in_process2.err || exit 1
timeout -k 10 400 python3 bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --ballots 500000 --no-isolated > gpurun_out/bench_bare2.json 2> gpurun_out/bench_bare2.err || exit 1
# round 6: the in-process leg from ONE pinned host buffer (eg_verify_*_batch_multi), the non-rehearsed bare launch at N = 1 over RCCL
# (--spawn --force-dist: count the GPUs from sysfs -> child torch.distributed.run -> RCCL init -> preflight -> line), the driver's own
# command, and the FP64-limb multiplier micro-benchmark with the package power beside it
timeout -k 10 300 python3 bench.py --in-process-devices 2 --rehearse-one-gpu --from-host --steps 3 --warmup 1 --ballots 500000 > gpurun_out/bench_in_process2_host.json 2> gpurun_out/bench_in_process2_host.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --force-dist --spawn --steps 5 --warmup 1 --no-extra-configs > gpurun_out/bench_spawn1.json 2> gpurun_out/bench_spawn1.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.json 2> gpurun_out/bench_driver_cmd.err || exit 1
timeout -k 10 300 tools/fp64_probe.sh gpurun_out/ubench_fp64.txt 4 || exit 1
echo "profile round done"
