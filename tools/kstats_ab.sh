#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of tools/perf_probe.py for several library builds.   usage: kstats_ab.sh WORKLOAD lib1 lib2 ..
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
w=$1; shift
for v in "$@"; do
  n=$(basename $v .so)
  rm -rf gpurun_out/kst_${n}_$w
  EG_LIB=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kst_${n}_$w -o stats --output-format csv -- python3 tools/perf_probe.py 1000000 $w 2 > gpurun_out/kst_${n}_$w.log 2>&1 || exit 1
  tail -n 1 gpurun_out/kst_${n}_$w.log
done
