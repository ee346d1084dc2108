cd /tmp && export TMPDIR=/tmp
for L in 1 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp$L -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 256 --warmup 32 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --depth-layers $L > /dev/null 2>&1
f=$(find /tmp/lp$L -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/layers_kstats_$L.csv
done
