#!/bin/bash
# usage (GPU box, repo root): tools/fetch_calibration.sh   -> gpurun_out/fetch_calibration.txt
# FETCH_SIZE of known access patterns (tools/micro/fetch_calibration.hip) under rocprofv3 --pmc: bytes tallied per byte the lanes asked for.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $ROOT/tools/micro/fetch_calibration.hip -o /tmp/fetch_calibration || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fetch_cal
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_cal -o cal -- /tmp/fetch_calibration > $OUT/fetch_calibration.log 2>&1
f=$(find $OUT/fetch_cal -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/fetch_calibration.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
count = 4 << 20
asked = {"streamKernel": 16, "gatherKernel<1, true>": 16, "gatherKernel<1, false>": 16, "gatherKernel<2, true>": 32, "gatherKernel<2, false>": 32,
         "gatherKernel<12, true>": 192, "gatherKernel<12, false>": 192, "gatherKernel<4, false>": 64, "gatherKernel<8, false>": 128, "gatherWordKernel": 4}
what = {"streamKernel": "streaming, 16 B per lane", "gatherKernel<1, true>": "16-B gather, records side by side", "gatherKernel<1, false>": "16-B gather, one per 256-B slot",
        "gatherKernel<2, true>": "32-B gather, side by side (the raster record)", "gatherKernel<2, false>": "32-B gather, one per 256-B slot",
        "gatherKernel<12, true>": "192-B row gather, side by side (the coefficient row)", "gatherKernel<12, false>": "192-B row gather, one per 256-B slot",
        "gatherKernel<4, false>": "64-B gather, one per 256-B slot", "gatherKernel<8, false>": "128-B gather, one per 256-B slot", "gatherWordKernel": "4-B gather, one per 256-B slot"}
print("FETCH_SIZE of %d accesses per pattern to lines nobody touched before (4 GiB table); factor = bytes tallied / bytes the lanes asked for" % count)
for r in rows:
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if name not in asked:
        continue
    kb = float(r["Counter_Value"])
    per = kb * 1024.0 / count
    print("%-52s asked %4d B  tallied %7.1f B per access  factor %.3f" % (what[name], asked[name], per, per / asked[name]))
PY
rm -rf $OUT/fetch_cal
