#!/bin/bash
# round 4, GPU call: ring-group walk - parity test, workspace per ballot, same-call A/B of group sizes
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ring_group or options_2_3 or unusual_option or choice_batch_vs or multi_choice_3 or golden" > gpurun_out/t3.txt 2>&1 || { tail -n 40 gpurun_out/t3.txt; exit 1; }
tail -n 3 gpurun_out/t3.txt
for g in 0 2; do echo "== EG_RING_GROUP=$g"; EG_RING_GROUP=$g timeout -k 10 300 python3 tools/workspace_probe.py 2>&1 | grep "bytes of chunk"; done > gpurun_out/ws_r4.txt 2>&1
cat gpurun_out/ws_r4.txt
tools/ab_env.sh gpurun_out/ab_r4_groups.log "EG_RING_GROUP=0 | EG_RING_GROUP=2 | EG_RING_GROUP=3 | EG_RING_GROUP=2,EG_CHUNK=1048576" "single" 2
tools/ab_env.sh gpurun_out/ab_r4_groups_multi.log "EG_RING_GROUP=0 | EG_RING_GROUP=4 | EG_RING_GROUP=8 | EG_RING_GROUP=4,EG_CHUNK=262144" "multi" 2
