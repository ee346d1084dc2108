#!/bin/bash
# usage (GPU box, repo root): tools/r04_ab.sh ROUNDS [lib ...]   (lib: "tree" = the in-tree library; "old" = the checkout under ab_old/
# (git archive of the previous round + its built library); or a library path under ab_variants/ for the in-tree bench.py)
# Round 4 A/B: alternating bench runs of library builds on ONE box (boxes differ by +-5 %), both storage orders; prints frames/s, the
# latency loop and the serial stage times of each run.  Then one band of an 8-way split (band_sim) per library and storage order.
ROUNDS=${1:-2}; shift
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=(old tree)
export GPU_MAX_HW_QUEUES=8
for r in $(seq $ROUNDS); do
  for lib in "${LIBS[@]}"; do
    for order in caller morton; do
      if [ "$lib" = "tree" ] || [ "$lib" = "old" ]; then unset GSPLAT_HIP_LIBRARY; else export GSPLAT_HIP_LIBRARY=$PWD/$lib; fi
      BENCH=bench.py; [ "$lib" = "old" ] && BENCH=ab_old/bench.py
      python $BENCH --steps 512 --warmup 32 --cpu-frames 0 --no-extra --storage-order $order 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages']
print('$lib', '$order', 'fps', d['value'], 'one-frame-then-sync ms', d['extra']['latency']['one_frame_then_sync']['ms'], 'stages ms', {k: round(st[k]['ms'], 4) for k in st})"
    done
  done
done
if [ -z "$NO_BANDS" ]; then
for lib in "${LIBS[@]}"; do
  for order in caller morton; do
    if [ "$lib" = "tree" ] || [ "$lib" = "old" ]; then unset GSPLAT_HIP_LIBRARY; else export GSPLAT_HIP_LIBRARY=$PWD/$lib; fi
    SIM=tools/band_sim.py; [ "$lib" = "old" ] && SIM=ab_old/tools/band_sim.py
    echo "--- 8-way split, $lib, storage order $order"
    python $SIM --worlds 1,8 --frames 1024 --inflight 4 --views 4 --storage-order $order 2>/dev/null | grep -E "^world"
  done
done
fi
