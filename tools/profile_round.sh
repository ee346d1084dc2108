#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh r01
# Writes under gpurun_out/<tag>_*: the default bench line, rocprofv3 kernel stats of the same command, and HBM traffic
# counters (separate --pmc passes, kernel trace only, as MI355X_MICROARCH.md prescribes).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/${TAG}_bench_c3.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o stats -- python3 $ROOT/bench.py --steps 256 --warmup 32 --cpu-frames 0 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
echo "stats done"
# the same frames one at a time (one frame in flight, plain launches): per-kernel averages without overlap from other frames,
# comparable with the per-stage HIP-event times bench.py reports in "stages" / "roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -o stats -- python3 $ROOT/bench.py --steps 256 --warmup 32 --cpu-frames 0 --inflight 1 --no-graph > $OUT/${TAG}_bench_under_rocprof_serial.json 2> $OUT/${TAG}_stats_serial.err
echo "serial stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$c -o pmc -- python3 $ROOT/bench.py --steps 16 --warmup 4 --cpu-frames 0 --inflight 1 --no-graph > $OUT/${TAG}_pmc_$c.json 2> $OUT/${TAG}_pmc_$c.err
  echo "$c done"
done
find $OUT -name "*counter_collection.csv" | while read f; do python3 $ROOT/tools/pmc_summary.py $f > ${f%.csv}_summary.txt; done
ls $OUT
