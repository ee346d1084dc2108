#!/bin/bash
# usage (GPU box, repo root): tools/ab_colour_bound.sh ROUNDS
# Upper bound of what a per-splat colour cache in the raster kernel could save ("colour evaluated once per distinct
# consumed splat"): the raster kernel with colours on demand (position + 192-byte coefficient row fetched and evaluated per staged entry)
# against the SAME kernel reading finished colours from the colour buffer (GS_NO_LAZY_SH=1: the SH pass runs first) -- what a cache with a
# 100 % hit rate and no cost of its own would make of it.  Serial stage times of the default scene and of the low-opacity scene.
ROUNDS=${1:-2}
for r in $(seq $ROUNDS); do
  for mode in on_demand colour_buffer; do
    if [ $mode = colour_buffer ]; then export GS_NO_LAZY_SH=1; else unset GS_NO_LAZY_SH; fi
    python bench.py --steps 512 --warmup 32 --cpu-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']['low_opacity_scene']; st=d['stages']
print('$mode', 'fps', d['value'], 'raster ms', st['raster']['ms'], 'sh ms', st['sh']['ms'], '| low-opacity: fps', e['value'], 'raster ms', e['raster_ms'], 'sh ms', e['stages_ms']['sh'])"
  done
done
