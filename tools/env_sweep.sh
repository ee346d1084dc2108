#!/bin/bash
# usage (GPU box, repo root): tools/env_sweep.sh "K=V" ...  -- frame_pump (native frame loop, 4 frames in flight) under HIP runtime
# switches, one run per argument; stops at the first run that hangs (frame_pump.py gives every run 60 s).
export PUMP_MASKS=0 PUMP_DEPTHS=4
for e in "$@"; do
  echo "== $e"
  python3 tools/frame_pump.py GPU_MAX_HW_QUEUES=8 $e 2>/dev/null | grep -E "inflight|timeout" | sed -e 's/"graph.*"ms_per_frame"/"ms"/' || exit 1
done
