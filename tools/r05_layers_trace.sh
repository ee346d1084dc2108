#!/bin/bash
# usage (GPU box): tools/r05_layers_trace.sh [bench args] -- per-LAYER kernel durations of a layered frame (rocprofv3 kernel trace of captured frames,
# one in flight): the raster / count / emit / second-scatter launches of the last frames, by their position in the frame
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=${@:-"--splats 5000000 --width 3840 --height 2160 --depth-layers 3"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o t -- python3 $ROOT/bench.py $ARGS --steps 24 --warmup 4 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 > /dev/null 2>&1
f=$(find /tmp/lt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|gs::|void ", "", n)
    return n[:n.index("(")] if "(" in n else n
# frames = runs of kernels between clip kernels; keep the last 16 frames
frames, cur = [], []
for r in rows:
    n = short(r["Kernel_Name"])
    if n.startswith("splatClipDataKernel") or n.startswith("radixHistogramKernel<unsigned int, 9, 8, DepthKeySource"):
        if cur and any(short(x["Kernel_Name"]).startswith("rasterizeTilesKernel") for x in cur): frames.append(cur); cur = []
    cur.append(r)
import os
want = int(os.environ.get("LAYERS", "3"))
frames = [f for f in frames if sum(short(x["Kernel_Name"]).startswith("rasterizeTilesKernel") for x in f) == want][-16:]
acc = collections.OrderedDict()
for fr in frames:
    seen = collections.Counter()
    for r in fr:
        n = short(r["Kernel_Name"]); seen[n] += 1
        acc.setdefault((n, seen[n]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for (n, i), v in acc.items():
    if len(v) < len(frames) // 2: continue
    m = sum(v) / len(v); tot += m
    if m >= 6: print("  %-62s #%d  %8.1f us" % (n[:62], i, m))
print("  kernel time per frame %.1f us over %d frames; span %.1f us" % (tot, len(frames), sum((int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in frames) / len(frames)))
PY
