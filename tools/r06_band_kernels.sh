#!/bin/bash
# usage (GPU box, repo root): tools/r06_band_kernels.sh -- rocprofv3 kernel statistics of one centre band of the 8-way split, chains one at a time,
# with and without the block-level band cull: BASELINE.json configs[3] (1 M splats, 1080p, rows 29..34, 4 views per chain) and configs[4]
# (5 M splats, 2160p, rows 59..76, 1 view per chain, the renderer's choice of depth layers)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_cull
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
summ() {
python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
chains = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].replace("gs::(anonymous namespace)::", "").replace("void ", "")
    print(f'{name[:78]:78s} calls/chain {int(r["Calls"]) / chains:6.2f} avg {float(r["AverageNs"])/1e3:8.2f} us  per chain {float(r["TotalDurationNs"])/chains/1e3:8.2f} us {100*float(r["TotalDurationNs"])/tot:5.1f} %')
print(f"kernel time per chain: {tot / chains / 1e3:.1f} us")
PY
}
for bc in 0 1; do
  for cfg in "c3 --splats 1000000 --width 1920 --height 1080 --rows 29 34 --views 4 --depth-layers 1" "c5 --splats 5000000 --width 3840 --height 2160 --rows 59 76 --views 1 --depth-layers -1"; do
    set -- $cfg; tag=$1; shift
    rm -rf /tmp/bk_${tag}_$bc
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk_${tag}_$bc -o stats -- python3 $ROOT/tools/band_kernels.py "$@" --block-cull $bc --chains 64 > $OUT/band_kernels_${tag}_bc$bc.txt 2> /tmp/bk_${tag}_$bc.err
    f=$(find /tmp/bk_${tag}_$bc -name "*kernel_stats.csv" | head -1)
    summ "$f" 64 >> $OUT/band_kernels_${tag}_bc$bc.txt
  done
done
ls $OUT
