#!/bin/bash
# usage (GPU box, repo root): tools/r05_layers_report.sh > gpurun_out/r05_depth_layers_raw.txt
# Depth layers (gs_renderer_set_depth_layers) measured on one box: frames/s and pairs listed with 1, 2, 3, 4 layers and with the renderer's
# own choice (0) over scene sizes and resolutions; then the per-kernel statistics of BASELINE.json configs[4] on one GPU with 1 and 3 layers
# (captured frames, one in flight).
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('%9.2f frames/s | layers in use %d | pairs %9d | listed %9d (%.3f)' % (d['value'], c['depth_layers']['in_use'], c['pairs_per_frame'], c['depth_layers']['listed_pairs_last_frame'], c['depth_layers']['listed_pairs_last_frame']/max(1,c['pairs_per_frame'])))"; }
echo "== frames/s (4 frames in flight, captured), pairs per frame and pairs listed per frame (last timed frame)"
for cfg in "--splats 1000000" "--splats 2000000" "--splats 5000000" "--splats 1000000 --width 3840 --height 2160" "--splats 5000000 --width 3840 --height 2160" "--splats 1000000 --opacity-mean -3"; do
for a in 1 2 3 4 0; do
  echo "[$cfg] [--depth-layers $a] $(python bench.py $cfg --steps 256 --warmup 32 --cpu-frames 0 --no-extra --depth-layers $a 2>/dev/null | tail -1 | line)"
done; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in 1 3; do
echo "== kernel statistics, 5 M splats 3840x2160, --depth-layers $L (captured frames, one in flight; the trace holds every frame of the run)"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp$L -o s -- python3 $ROOT/bench.py --splats 5000000 --width 3840 --height 2160 --steps 32 --warmup 4 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --depth-layers $L > /dev/null 2>&1
f=$(find /tmp/lp$L -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    name = re.sub(r"\(anonymous namespace\)::|gs::|void ", "", r["Name"]); name = name[:name.index("(")] if "(" in name else name
    print("  %-62s calls %6s  avg %10.1f us  total %9.1f ms  %5.1f %%" % (name[:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
done
