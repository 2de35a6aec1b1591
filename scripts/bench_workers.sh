#!/bin/bash
# usage (through gpurun): scripts/bench_workers.sh  -> step / feature-stage time for several SIFT worker counts and queue counts
cd $GRAFT_REPO_ROOT
for cfg in "10 8" "8 8" "12 8" "16 8" "12 16" "16 16" "10 8"; do
  set -- $cfg
  APS_SIFT_WORKERS=$1 GPU_MAX_HW_QUEUES=$2 python3 bench.py --steps 6 --warmup 2 --cpu-baseline off --end-to-end off --global-probe off 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read())
print('workers $1 queues $2: step', b['ms_per_step'], 'resident', b['ms_per_step_resident'], 'features', b['stages_ms_per_step']['features'], 'matching', b['stages_ms_per_step']['matching'], 'render', b['stages_ms_per_step']['render'])"
done
