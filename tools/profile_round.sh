#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh r03 [a|b|c]
# Writes under gpurun_out/<tag>_*; tools/collect_profiles.sh copies the summaries into profiles/.  Three parts, each within one gpurun call:
#   a  the default bench line, rocprofv3 kernel stats of the same command (pipelined) and of the frames one at a time (serial), HBM traffic
#      and SQ counters of the kernels for the default scene AND the low-opacity scene (separate --pmc passes, kernel trace only, as
#      MI355X_MICROARCH.md prescribes), the stage ablation;
#   b  BASELINE.json configs[4] (5 M splats, 3840x2160) on one GPU: kernel stats and FETCH/WRITE passes; the other configurations' bench lines;
#      the raster kernel's per-tile timelines; the micro benchmarks;
#   c  the multi-GPU projections on one GPU: bands of 1/2/4/8-way splits (1, 2, 4 frames per chain; caller and Morton storage), the kernel
#      statistics of one band chain, the native frame pump.
TAG=${1:-r03}; PART=${2:-a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ $PART = a ]; then
python3 bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/${TAG}_bench_c3.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
# (--spin-up 0: the untimed second of rendering would put ~80 000 more kernel records into every trace)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o stats -- python3 $ROOT/bench.py --steps 256 --warmup 32 --cpu-frames 0 --no-extra --spin-up 0 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err; echo "stats rc=$?"
# the same frames one at a time (one frame in flight, plain launches): per-kernel averages without overlap from other frames,
# comparable with the per-stage HIP-event times bench.py reports in "stages" / "roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -o stats -- python3 $ROOT/bench.py --steps 256 --warmup 32 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --no-graph > $OUT/${TAG}_bench_under_rocprof_serial.json 2> $OUT/${TAG}_stats_serial.err; echo "serial stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}lo_stats_serial -o stats -- python3 $ROOT/bench.py --steps 128 --warmup 16 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --no-graph --opacity-mean -3 > $OUT/${TAG}lo_bench_under_rocprof_serial.json 2> $OUT/${TAG}lo_stats_serial.err; echo "low-opacity serial stats rc=$?"
cd $ROOT
tools/pmc_pass.sh $TAG "FETCH_SIZE"
tools/pmc_pass.sh $TAG "WRITE_SIZE"
tools/pmc_pass.sh $TAG "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
tools/pmc_pass.sh $TAG "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
tools/pmc_pass.sh $TAG "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INST_CYCLES_VMEM"
# the low-opacity scene (raw opacity mean -3: the raster kernel is the dominant stage)
tools/pmc_pass.sh ${TAG}lo "FETCH_SIZE" --opacity-mean -3
tools/pmc_pass.sh ${TAG}lo "WRITE_SIZE" --opacity-mean -3
tools/pmc_pass.sh ${TAG}lo "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" --opacity-mean -3
tools/pmc_pass.sh ${TAG}lo "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" --opacity-mean -3
echo "pmc done"
tools/ablation.sh 2>/dev/null | grep ablate > $OUT/${TAG}_ablation.txt
fi
if [ $PART = b ]; then
C5="--splats 5000000 --width 3840 --height 2160"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}c5_stats_serial -o stats -- python3 $ROOT/bench.py $C5 --steps 32 --warmup 4 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --no-graph > $OUT/${TAG}c5_bench_under_rocprof_serial.json 2> $OUT/${TAG}c5_stats_serial.err; echo "C5 serial stats rc=$?"
cd $ROOT
tools/pmc_pass.sh ${TAG}c5 "FETCH_SIZE" $C5
tools/pmc_pass.sh ${TAG}c5 "WRITE_SIZE" $C5
tools/bench_other.sh $TAG
# the micro benchmarks are built here from their sources (never a binary that happens to lie around)
for m in valu_lds_rate graph_streams lds_atomic_order; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/$m.hip -o /tmp/micro_$m -lpthread 2>/dev/null && timeout -k 5 200 /tmp/micro_$m > $OUT/${TAG}_micro_$m.txt 2>/dev/null
done
python3 tools/raster_trace.py --frames 3 2>/dev/null > $OUT/${TAG}_raster_trace.txt
python3 tools/raster_trace.py --frames 3 --opacity-mean -3 2>/dev/null > $OUT/${TAG}_raster_trace_low_opacity.txt
fi
if [ $PART = c ]; then
for v in 1 2 4; do echo "--- frames per chain: $v (Morton storage)"; GPU_MAX_HW_QUEUES=8 python3 tools/band_sim.py --worlds 1,2,4,8 --frames 1024 --inflight 4 --views $v 2>/dev/null | grep -E "world|bands"; done > $OUT/${TAG}_band_sim.txt; echo "band_sim rc=$?"
echo "--- frames per chain: 4, splats in the caller's order" >> $OUT/${TAG}_band_sim.txt
GPU_MAX_HW_QUEUES=8 python3 tools/band_sim.py --worlds 1,8 --frames 1024 --inflight 4 --views 4 --storage-order caller 2>/dev/null | grep -E "world|bands" >> $OUT/${TAG}_band_sim.txt
echo "--- frames per chain: 4, 8 GPUs, bands re-cut twice from their measured times (bench.py --rebalance 2)" >> $OUT/${TAG}_band_sim.txt
GPU_MAX_HW_QUEUES=8 python3 tools/band_sim.py --worlds 8 --frames 1024 --inflight 4 --views 4 --rebalance 2 2>/dev/null | grep -E "world|bands" >> $OUT/${TAG}_band_sim.txt
{ for v in 1 4; do echo "=== tile rows 29..34 (a centre band of the 8-way split), $v frame(s) per chain, Morton storage"; tools/band_profile.sh 29 34 $v 1; done
  echo "=== tile rows 29..34, 4 frames per chain, splats in the caller's order"; tools/band_profile.sh 29 34 4 0; } > $OUT/${TAG}_band_chain_kernels.txt 2>&1; echo "band chain rc=$?"
PUMP_MASKS=0 PUMP_DEPTHS=1,2,3,4 python3 tools/frame_pump.py GPU_MAX_HW_QUEUES=8 2>/dev/null | grep inflight > $OUT/${TAG}_frame_pump.txt
fi
# the per-dispatch kernel traces are large (gpurun_out/ travels back only below 64 MiB): keep the per-kernel statistics
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT | grep $TAG
