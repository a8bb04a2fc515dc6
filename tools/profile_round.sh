cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -q -x -m gpu 2>&1 | tail -2
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_single.json 2> gpurun_out/bench_single.err; tail -c 600 gpurun_out/bench_single.json
python bench.py --steps 3 --warmup 1 --workload multi --ballots 250000 > gpurun_out/bench_multi.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --workload qv --ballots 250000 > gpurun_out/bench_qv.json 2>/dev/null
rm -rf gpurun_out/prof_stats gpurun_out/pmc_*
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d gpurun_out/pmc_$c -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --ballots 262144 > gpurun_out/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES -d gpurun_out/pmc_SQ1 -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --ballots 262144 > gpurun_out/pmc_SQ1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVES -d gpurun_out/pmc_SQ2 -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --ballots 262144 > gpurun_out/pmc_SQ2.log 2>&1
find gpurun_out -name "*.csv" | head -30
