#!/bin/bash
# usage (GPU box, repo root): tools/band_profile.sh [rowBegin rowEnd [views [morton]]]
# rocprofv3 kernel statistics of ONE band of the 8-way split (default: tile rows 29..34 of the C3 scene), frames one at a time
# and four in flight, driven by the native loop tools/frame_pump: which kernels make up a band frame's chain.
RB=${1:-29}; RE=${2:-34}; VIEWS=${3:-1}; MORTON=${4:-1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export GPU_MAX_HW_QUEUES=8
cd $ROOT
PUMP_MASKS=0 PUMP_DEPTHS=1 PUMP_KEEP=1 python3 tools/frame_pump.py > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for D in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/band_stats_d$D -o stats -- $ROOT/tools/frame_pump /tmp/gs_pump/scene.ply /tmp/gs_pump/cams.bin 1920 1080 $RB $RE $D 512 1 0 $VIEWS $MORTON > $OUT/band_stats_d$D.json 2> $OUT/band_stats_d$D.err
  echo "inflight $D rc=$?"; cat $OUT/band_stats_d$D.json
  f=$(find $OUT/band_stats_d$D -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $((512 + 8 * D)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:7.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
frames = int(sys.argv[2])  # frame_pump: 8 * D warm-up frames + 512 timed ones
print(f"kernel time per frame ({frames} frames):", round(tot / frames / 1e3, 1), "us")
PY
  find $OUT/band_stats_d$D -name "*kernel_trace.csv" -delete
done
rm -f /tmp/gs_pump/scene.ply
