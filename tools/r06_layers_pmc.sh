#!/bin/bash
# usage (GPU box, repo root): tools/r06_layers_pmc.sh TAG -- the LAYERED raster kernel's SQ counters on the headline workload with two depth layers
# (captured frames, one frame in flight; separate --pmc pass, kernel trace only) -> gpurun_out/TAG_pmc_raster_layered.json with the source hash
# bench.py checks before it attaches roofline.timed_loop_kernel.valu
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lpmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/lpmc -o pmc -- python3 $ROOT/bench.py --steps 16 --warmup 8 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --depth-layers 2 > /dev/null 2> $OUT/${TAG}_layers_pmc.err
f=$(find /tmp/lpmc -name "*counter_collection.csv" | head -1)
PMC_KEEP_TEMPLATES=1 python3 $ROOT/tools/pmc_summary.py $f | grep -v "at::\|rocclr\|unpack\|sphericalHarm" > $OUT/${TAG}_layers_pmc.txt
cd $ROOT
python3 - $TAG <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
tag = sys.argv[1]
row = None
for line in open(f"gpurun_out/{tag}_layers_pmc.txt"):
    if line.startswith("void rasterizeTilesKernel<2, false, 3, true>"):
        row = (eval(line[line.index("{"):line.index("}") + 1]), int(line.rsplit(" ", 1)[1]))
if row is None:
    raise SystemExit("no LAYERED raster line")
sq, n = row
busy = sq["SQ_BUSY_CYCLES"] / 32  # per shader engine
out = {"kernel": "rasterizeTilesKernel<2, false, 3, true>", "source_hash": bench.source_hash(),
       "workload": "bench.py default scene, --steps 16 --warmup 8 --inflight 1 --depth-layers 2 (captured frames, one in flight)", "dispatches": n,
       "SQ_INSTS_VALU_per_dispatch": sq["SQ_INSTS_VALU"], "sq": sq,
       "valu_issue_frac": round(4 * sq["SQ_ACTIVE_INST_VALU"] / (1024 * busy), 3), "waves_per_simd": round(4 * sq["SQ_WAVE_CYCLES"] / (1024 * busy), 3),
       "units": "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles, SQ_BUSY_CYCLES per shader engine (MI355X_MICROARCH.md)"}
json.dump(out, open(f"gpurun_out/{tag}_pmc_raster_layered.json", "w"), indent=1)
print(json.dumps(out))
PY
