#!/bin/bash
# usage (GPU box, repo root): tools/ab_storage.sh ROUNDS
# What spatially ordered splat storage buys (README.md:138 of the reference): the bench scene stored in the file's order
# (gs_renderer_set_storage_order CALLER) against the same splats stored along a 3-D Morton curve (MORTON), alternating runs on one box -- frames/s and serial stage times on one GPU,
# then the bands of an 8-way split (4 frames per chain, 4 chains in flight), one GPU standing in for each rank.
ROUNDS=${1:-2}
export GPU_MAX_HW_QUEUES=8
for r in $(seq $ROUNDS); do
  for order in caller morton; do
    python bench.py --steps 512 --warmup 32 --cpu-frames 0 --storage-order $order 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']['low_opacity_scene']; st=d['stages']
print('$order', 'fps', d['value'], 'one-frame-then-sync ms', d['extra']['latency']['one_frame_then_sync']['ms'], 'stages ms', {k: st[k]['ms'] for k in st}, '| low-opacity fps', e['value'], 'raster_ms', e['raster_ms'])"
  done
done
for order in caller morton; do
  echo "--- 8-way split, storage order: $order"
  python tools/band_sim.py --worlds 1,8 --frames 1024 --inflight 4 --views 4 --storage-order $order 2>/dev/null | grep -E "^world"
done
