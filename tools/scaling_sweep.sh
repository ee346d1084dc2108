#!/bin/bash
# frames/s of the one-call renderer (four frames in flight, captured frames) over the splat count at 1920x1080 and over the resolution at
# 1 M splats; same synthetic scene generator and orbit as the default bench line.  usage (GPU box): tools/scaling_sweep.sh
cd "$(dirname "$0")/.."
line() { python bench.py --steps 256 --warmup 32 --cpu-frames 0 --no-extra "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; s=d['stages']
print(f\"{c['splats']:>8d} splats {c['width']}x{c['height']}: {d['value']:8.1f} frames/s  pairs {c['pairs_per_frame']:>9d} consumed {c['consumed_pairs_per_frame']:>8d}  stages ms clip {s['clip']['ms']:.3f} list {s['tile_list']['ms']:.3f} sort {s['sort']['ms']:.3f} raster {s['raster']['ms']:.3f}\")"; }
echo "== splat count at 1920x1080"
for n in 100000 300000 1000000 2000000 5000000; do line --splats $n; done
echo "== resolution at 1 M splats"
for wh in "1280 720" "1920 1080" "2560 1440" "3840 2160"; do set -- $wh; line --width $1 --height $2; done
