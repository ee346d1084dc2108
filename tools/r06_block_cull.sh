#!/bin/bash
# Round 6: block-level band cull, A/B on one box (tools/band_sim.py renders each rank's band in turn on the one GPU: a projection, gather excluded).
set -e
export GPU_MAX_HW_QUEUES=8   # (four chains in flight need their own hardware queues: DESIGN.md section 5; rounds 2-5 measured band_sim with it)
out=gpurun_out/r06_cull; mkdir -p $out
for bc in 0 1 0 1; do
  echo "--- C3 (1 M splats, 1920x1080), 8 bands, 4 frames per chain, block cull $bc" >> $out/band_sim_c3.txt
  python tools/band_sim.py --worlds 8 --views 4 --frames 1024 --inflight 4 --block-cull $bc 2>/dev/null | grep world >> $out/band_sim_c3.txt
done
for bc in 0 1 0 1; do
  echo "--- C5 (5 M splats, 3840x2160), 8 bands, 1 frame per chain, depth layers -1, block cull $bc" >> $out/band_sim_c5.txt
  python tools/band_sim.py --splats 5000000 --width 3840 --height 2160 --worlds 8 --views 1 --frames 256 --inflight 4 --depth-layers -1 --block-cull $bc 2>/dev/null | grep world >> $out/band_sim_c5.txt
done
for bc in 0 1; do
  echo "--- C5 (5 M splats, 3840x2160), 8 bands, 4 frames per chain, depth layers -1, block cull $bc" >> $out/band_sim_c5.txt
  python tools/band_sim.py --splats 5000000 --width 3840 --height 2160 --worlds 8 --views 4 --frames 256 --inflight 4 --depth-layers -1 --block-cull $bc 2>/dev/null | grep world >> $out/band_sim_c5.txt
done
