#!/bin/bash
# The reference's own frame loop (Demo.cpp:394-479: eight staged calls, blocking per-stage timers) through the C++ facade, on
#  (a) BASELINE.json configs[2]: 1 M splats, SH degree 3, 1920x1080, and
#  (b) the shape of the reference's published table (README.md:144-153): 816 k splats, SH degree 0, 1024x1024.
# Prints the reference's Stats table (Demo.cpp:541-562) for each.  usage: tools/staged_path.sh [frames]
set -e
cd "$(dirname "$0")/.."
FRAMES=${1:-200}
make -s -C cudagaussianrenderer_amd/csrc demo >/dev/null
TMP=$(mktemp -d)
python3 - "$TMP" <<'PY'
import sys
sys.path.insert(0, ".")
from cudagaussianrenderer_amd import scenes
scenes.write_synthetic_ply(sys.argv[1] + "/c3.ply", 1_000_000, sh_degree=3, seed=1234)
scenes.write_synthetic_ply(sys.argv[1] + "/lilly.ply", 815_957, sh_degree=0, seed=1234)
PY
echo "== (a) 1 M splats, SH 3, 1920x1080, $FRAMES frames"
./tools/gsplat_demo "$TMP/c3.ply" 1920 1080 "$FRAMES"
echo "== (a') the same scene and camera, the loop body as ONE call per frame from the same C++ host (gs_renderer_render, synchronised after each)"
./tools/gsplat_demo "$TMP/c3.ply" 1920 1080 "$FRAMES" --one-call --warm-up 16
echo "== (b) 815 957 splats, SH 0, 1024x1024, $FRAMES frames (README.md:144-153 shape; reference: 12.50 ms on an RTX 3050 laptop)"
./tools/gsplat_demo "$TMP/lilly.ply" 1024 1024 "$FRAMES"
rm -rf "$TMP"
