#!/bin/bash
# frames/s with 1, 2 and 3 depth layers (gs_renderer_set_depth_layers, default cuts) over scene sizes and resolutions, one box
for cfg in "--splats 1000000" "--splats 2000000" "--splats 5000000" "--splats 1000000 --width 3840 --height 2160" "--splats 5000000 --width 3840 --height 2160"; do
for a in "--depth-layers 1" "--depth-layers 2" "--depth-layers 3"; do
  v=$(python bench.py $cfg --steps 256 --warmup 32 --cpu-frames 0 --no-extra $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['pairs_per_frame'])")
  echo "[$cfg] [$a] -> $v"
done; done
