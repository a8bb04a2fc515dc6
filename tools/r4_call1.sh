#!/bin/bash
# round 4, GPU call 1: cndmask micro-benchmark rows + same-call A/B of the select-free builds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout -k 10 120 tools/ubench/valu_rates 25 cndmask; timeout -k 10 60 tools/ubench/valu_rates 25 "v_xor +"; timeout -k 10 60 tools/ubench/valu_rates 25 v_bfi; timeout -k 10 60 tools/ubench/valu_rates 25 "v_add_u32" ) > gpurun_out/r04_ubench_cndmask.txt 2>&1
cat gpurun_out/r04_ubench_cndmask.txt
tools/ab_run.sh gpurun_out/ab_r4_mask.log "build_variants/libeg_base.so build_variants/libeg_mask1.so build_variants/libeg_mask2.so" "single qv" 2 | grep best
