cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in elastic_elgamal_amd/libeg_hip.so build_variants/libeg_tabnostore.so; do
  n=$(basename $v .so)
  rm -rf gpurun_out/prof_$n
  EG_LIB=$v rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$n -o stats --output-format csv -- python3 tools/perf_probe.py 1000000 single 2 > gpurun_out/prof_$n.log 2>&1
  echo $n $(grep -E "k_sum_tables" gpurun_out/prof_$n/stats_kernel_stats.csv | cut -d, -f4)
done
