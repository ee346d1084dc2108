set -e
TMP=$(mktemp -d)
python3 - "$TMP" <<'PY'
import sys
sys.path.insert(0, ".")
from cudagaussianrenderer_amd import scenes
scenes.write_synthetic_ply(sys.argv[1] + "/c5.ply", 5_000_000, sh_degree=3, seed=1234)
PY
for b in sorted reference-order; do
  echo "== C5: 5 M splats, SH 3, 3840x2160, builder $b"
  timeout -k 10 240 ./tools/gsplat_demo "$TMP/c5.ply" 3840 2160 20 "$TMP/out_$b.ppm" --warm-up 4 --builder $b
done
cmp "$TMP/out_sorted.ppm" "$TMP/out_reference-order.ppm" && echo "images identical"
rm -rf "$TMP"
