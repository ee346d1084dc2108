#!/bin/bash
# rocprofv3 kernel statistics of the reference's frame loop over the eight staged calls (tools/gsplat_demo), C3 and C5.
# usage: tools/staged_path_profile.sh OUTDIR   (writes OUTDIR/staged_c3_kernel_stats.csv, OUTDIR/staged_c5_kernel_stats.csv)
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/staged_prof}
mkdir -p "$OUT"
TMP=$(mktemp -d)
export TMPDIR=/tmp
python3 - "$TMP" <<'PY'
import sys
sys.path.insert(0, ".")
from cudagaussianrenderer_amd import scenes
scenes.write_synthetic_ply(sys.argv[1] + "/c3.ply", 1_000_000, sh_degree=3, seed=1234)
scenes.write_synthetic_ply(sys.argv[1] + "/c5.ply", 5_000_000, sh_degree=3, seed=1234)
PY
ROOT=$(pwd)
for cfg in "c3 1920 1080 100" "c5 3840 2160 20"; do
  set -- $cfg
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$TMP/prof_$1" -o run -- "$ROOT/tools/gsplat_demo" "$TMP/$1.ply" $2 $3 $4 --warm-up 4 > "$ROOT/$OUT/staged_$1_stdout.txt" 2>&1) || true
  f=$(find "$TMP/prof_$1" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/staged_$1_kernel_stats.csv"
done
rm -rf "$TMP"
