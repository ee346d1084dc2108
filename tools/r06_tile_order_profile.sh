#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_exp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2; do
  rm -rf /tmp/top$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/top$mode -o stats -- python3 $ROOT/tools/tile_order_profile.py $((mode > 1 ? 1 : mode)) plain $([ $mode = 2 ] && echo cull) > $OUT/tile_order_profile_$mode.txt 2>/dev/null
  f=$(find /tmp/top$mode -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $OUT/tile_order_profile_$mode.txt <<'PY'
import csv, sys
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{r["Name"].replace("gs::(anonymous namespace)::", "")[:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
done
cat $OUT/tile_order_profile_0.txt $OUT/tile_order_profile_1.txt $OUT/tile_order_profile_2.txt
