#!/bin/bash
# usage (GPU box): tools/bench_other.sh TAG -- bench lines of the other BASELINE.json configurations on one GPU, one JSON line each
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_bench_other.jsonl
: > $OUT
# configs[4] on one GPU: 5 M splats, SH 3, 3840x2160 (full frame; the 8-GPU form needs a node)
python bench.py --splats 5000000 --width 3840 --height 2160 --steps 128 --warmup 16 --cpu-frames 1 --no-extra | tail -1 >> $OUT
# the reference's EPANECHNIKOV_KERNEL option on the headline workload
python bench.py --epanechnikov --steps 256 --warmup 32 --cpu-frames 1 --no-extra | tail -1 >> $OUT
# configs[1]-like: 100 k splats, SH 0 is the procedural generator's case (tests); here the PLY path at 100 k / 1024x1024 / SH 3
python bench.py --splats 100000 --width 1024 --height 1024 --log-scale-mean -3.0 --steps 256 --warmup 32 --cpu-frames 2 --no-extra | tail -1 >> $OUT
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(d["config"]["workload"][:70], "|", d["value"], "frames/s |", d["config"]["pairs_per_frame"], "pairs | parity", d.get("parity", {}).get("max_lsb"), d.get("parity", {}).get("ok"),
          "| cpu", d.get("cpu_baseline", {}).get("value"))
PY
