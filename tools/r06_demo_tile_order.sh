#!/bin/bash
# usage (GPU box, repo root): tools/r06_demo_tile_order.sh -- the C++ viewer loop (tools/gsplat_demo --one-call --events: gs::CameraControls moved by a
# recorded orbit drag, one frame, then a synchronise) on BASELINE.json configs[2] with and without gs_renderer_set_tile_start_order (--tile-order)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
from cudagaussianrenderer_amd import scenes
os.makedirs("/tmp/gs_demo", exist_ok=True)
scenes.write_synthetic_ply("/tmp/gs_demo/scene.ply", 1_000_000, sh_degree=3, seed=1234)
with open("/tmp/gs_demo/orbit.txt", "w") as f:
    for i in range(450):  # middle button held (orbit about the anchor), the pointer moving 2 pixels per frame
        f.write("0.016 2 0 %d 540\n" % (700 + 2 * i))
PY
for rep in 1 2; do
  for flags in "--one-call" "--one-call --tile-order" "" "--tile-order"; do
    tools/gsplat_demo /tmp/gs_demo/scene.ply 1920 1080 400 --events /tmp/gs_demo/orbit.txt --warm-up 50 $flags | grep "Frame loop\|Rasterize\|Total" | cut -c1-200
  done
done
rm -rf /tmp/gs_demo
