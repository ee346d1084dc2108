#!/bin/bash
# usage (GPU box): tools/r05_layers_small.sh -- where two depth layers start to pay (splat count, resolution) and which cut suits the headline workload
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('%9.2f frames/s | layers %d | pairs %9d | listed %9d (%.3f)' % (d['value'], c['depth_layers']['in_use'], c['pairs_per_frame'], c['depth_layers']['listed_pairs_last_frame'], c['depth_layers']['listed_pairs_last_frame']/max(1,c['pairs_per_frame'])))"; }
for cfg in "--splats 200000" "--splats 500000" "--splats 1000000 --width 1280 --height 720" "--splats 1000000 --epanechnikov"; do
for a in 1 2; do
  echo "[$cfg] [--depth-layers $a] $(python bench.py $cfg --steps 512 --warmup 32 --cpu-frames 0 --no-extra --depth-layers $a 2>/dev/null | tail -1 | line)"
done; done
for f in 0.05 0.1 0.15 0.2 0.3 0.45; do
  echo "[C3] [--depth-layers 2 --layer-fractions $f] $(python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra --depth-layers 2 --layer-fractions $f 2>/dev/null | tail -1 | line)"
done
for f in 0.05,0.3 0.1,0.4 0.15,0.5; do
  echo "[C3] [--depth-layers 3 --layer-fractions $f] $(python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra --depth-layers 3 --layer-fractions $f 2>/dev/null | tail -1 | line)"
done
C5="--splats 5000000 --width 3840 --height 2160"
for f in 0.03,0.15 0.06,0.25 0.1,0.35 0.15,0.5; do
  echo "[C5] [--depth-layers 3 --layer-fractions $f] $(python bench.py $C5 --steps 128 --warmup 16 --cpu-frames 0 --no-extra --depth-layers 3 --layer-fractions $f 2>/dev/null | tail -1 | line)"
done
for f in 0.05 0.1 0.15 0.25; do
  echo "[5M 1080p] [--depth-layers 2 --layer-fractions $f] $(python bench.py --splats 5000000 --steps 128 --warmup 16 --cpu-frames 0 --no-extra --depth-layers 2 --layer-fractions $f 2>/dev/null | tail -1 | line)"
done
