#!/bin/bash
# usage (GPU box, repo root): tools/r04_fusion_ab.sh ROUNDS [bench.py arguments ...]
# Round 4, item "first tile-sort pass inside the emit pass": alternating bench runs with gs_set_list_fusion 0 / 1 on ONE box.
# GSPLAT_HIP_LIBRARY (a variant build) is honoured.  FUSION_VALUES="0 1" by default.
ROUNDS=${1:-2}; shift
export GPU_MAX_HW_QUEUES=8
for r in $(seq $ROUNDS); do
  for fusion in ${FUSION_VALUES:-0 1}; do
    python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra --list-fusion $fusion "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages']
print('${AB_LABEL:-tree} fusion $fusion', 'fps', d['value'], 'one-frame-then-sync ms', d['extra']['latency']['one_frame_then_sync']['ms'], 'stages ms', {k: round(st[k]['ms'], 4) for k in st})"
  done
done
