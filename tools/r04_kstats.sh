#!/bin/bash
# usage (GPU box, repo root): tools/r04_kstats.sh TAG [bench.py arguments ...]
# rocprofv3 kernel statistics of the bench frames one at a time (one frame in flight, plain launches): per-kernel average durations
# without overlap from other frames -> gpurun_out/TAG_kernel_stats.csv, and a short table on stdout.  GSPLAT_HIP_LIBRARY is honoured.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_stats_serial
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -o stats -- python3 $ROOT/bench.py --steps 256 --warmup 32 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --no-graph "$@" > $OUT/${TAG}_bench_under_rocprof_serial.json 2> $OUT/${TAG}_stats_serial.err
echo "rocprof rc=$?"
cd $ROOT
f=$(find $OUT/${TAG}_stats_serial -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
find $OUT/${TAG}_stats_serial -name "*kernel_trace.csv" -delete; find $OUT/${TAG}_stats_serial -name "*agent_info.csv" -delete
python3 - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    name = re.sub(r"gs::\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(gs::Screen.*|\(int, .*|\(unsigned.*|\(float.*|\(HIP_vector.*|\(long.*", "", name)[:90]
    print(f'{name:90s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"]) / 1e3:8.2f} us  total% {float(r["Percentage"]):5.1f}')
PY
