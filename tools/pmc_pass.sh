#!/bin/bash
# usage (GPU box, repo root): tools/pmc_pass.sh TAG "COUNTER1 COUNTER2 ..." [bench flags]
# One rocprofv3 --pmc pass (kernel trace only, as MI355X_MICROARCH.md prescribes) over a short serial bench run; per-kernel means
# to gpurun_out/TAG_pmc_<first counter>.txt
TAG=$1; COUNTERS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
first=$(echo $COUNTERS | cut -d' ' -f1)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$first -o pmc -- python3 $ROOT/bench.py --steps 16 --warmup 4 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --no-graph "$@" > $OUT/${TAG}_pmc_$first.json 2> $OUT/${TAG}_pmc_$first.err
f=$(find $OUT/${TAG}_pmc_$first -name "*counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $f > $OUT/${TAG}_pmc_$first.txt
rm -rf $OUT/${TAG}_pmc_$first   # the raw per-dispatch table (gpurun_out/ travels back only below 64 MiB); the per-kernel means stay
grep -i raster $OUT/${TAG}_pmc_$first.txt
