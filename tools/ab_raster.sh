#!/bin/bash
# usage (GPU box, repo root): tools/ab_raster.sh ROUNDS lib1.so lib2.so ...   ("tree" = the in-tree library)
# Raster-kernel variants on ONE box: lone-tile latency (tools/raster_lone_tile.py), then alternating bench runs -- frames/s, the serial
# raster time of the default scene and of the low-opacity scene (extra).
ROUNDS=$1; shift
pick() { if [ "$1" = "tree" ]; then unset GSPLAT_HIP_LIBRARY; else export GSPLAT_HIP_LIBRARY=$PWD/$1; fi; }
for lib in "$@"; do pick $lib; echo "== $lib"; python tools/raster_lone_tile.py --sizes 1024,4096 --reps 10 2>/dev/null | tail -1; python tools/raster_lone_tile.py --sizes 512,2048 --tiles 64 --reps 10 2>/dev/null | tail -1; done
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    pick $lib
    python bench.py --steps 512 --warmup 32 --cpu-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']['low_opacity_scene']
print('$lib', 'fps', d['value'], 'raster_ms', d['stages']['raster']['ms'], 'sum_ms', d['gpu_ms_sum_of_stages'], '| low-opacity fps', e['value'], 'raster_ms', e['raster_ms'])"
  done
done
