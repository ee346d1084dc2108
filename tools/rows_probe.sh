#!/bin/bash
# usage (GPU box, repo root): tools/rows_probe.sh  -- frame time of tile-row bands of growing height (native frame loop, 1 and 4 frames in flight)
export GPU_MAX_HW_QUEUES=8 PUMP_MASKS=0 PUMP_DEPTHS=1 PUMP_KEEP=1
python3 tools/frame_pump.py > /dev/null 2>&1   # writes the scene + cameras, builds tools/frame_pump (depth 9 runs nothing useful: skipped below)
for rows in "29 34" "29 39" "24 44" "14 54" "0 68"; do
  for d in 1 4; do
    timeout -k 5 60 tools/frame_pump /tmp/gs_pump/scene.ply /tmp/gs_pump/cams.bin 1920 1080 $rows $d 512 1 0 || exit 1
  done
done
rm -f /tmp/gs_pump/scene.ply
