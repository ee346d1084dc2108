#!/bin/bash
# usage: tools/collect_profiles.sh r03   -- copies the summaries tools/profile_round.sh left in gpurun_out/ into profiles/
TAG=${1:-r03}
G=gpurun_out
[ -s $G/${TAG}_bench_c3.json ] && [ -s $G/${TAG}_pmc_FETCH_SIZE.txt ] || { echo "no $TAG outputs under $G (was gpurun_out/ copied back?)"; exit 1; }
# (a bench line taken AFTER the counters were collected carries them -- roofline.traffic; it is kept as gpurun_out/TAG_bench_final.json)
if [ -s $G/${TAG}_bench_final.json ]; then tail -1 $G/${TAG}_bench_final.json > profiles/${TAG}_bench_c3.json; else tail -1 $G/${TAG}_bench_c3.json > profiles/${TAG}_bench_c3.json; fi
tail -1 $G/${TAG}_bench_under_rocprof.json > profiles/${TAG}_bench_under_rocprof_pipelined.json
tail -1 $G/${TAG}_bench_under_rocprof_serial.json > profiles/${TAG}_bench_under_rocprof_serial.json
cp $G/${TAG}_stats/stats_kernel_stats.csv profiles/${TAG}_kernel_stats_pipelined.csv
cp $G/${TAG}_stats_serial/stats_kernel_stats.csv profiles/${TAG}_kernel_stats_serial.csv
cp $G/${TAG}_pmc_FETCH_SIZE.txt profiles/${TAG}_pmc_fetch_size_kb.txt
cp $G/${TAG}_pmc_WRITE_SIZE.txt profiles/${TAG}_pmc_write_size_kb.txt
cat $G/${TAG}_pmc_SQ_BUSY_CYCLES.txt $G/${TAG}_pmc_SQ_ACTIVE_INST_LDS.txt $G/${TAG}_pmc_SQ_WAIT_ANY.txt > profiles/${TAG}_pmc_sq_counters.txt
for f in band_sim band_chain_kernels ablation frame_pump raster_trace raster_trace_low_opacity micro_valu_lds_rate micro_graph_streams micro_atomic_rate micro_graph_memset; do
  [ -s $G/${TAG}_$f.txt ] && cp $G/${TAG}_$f.txt profiles/${TAG}_$f.txt
done
[ -s $G/${TAG}_bench_other.jsonl ] && cp $G/${TAG}_bench_other.jsonl profiles/${TAG}_bench_other.jsonl
# the low-opacity scene (raster-dominated) and BASELINE.json configs[4] on one GPU: serial kernel statistics + counter passes
[ -s $G/${TAG}lo_stats_serial/stats_kernel_stats.csv ] && cp $G/${TAG}lo_stats_serial/stats_kernel_stats.csv profiles/${TAG}_kernel_stats_low_opacity_serial.csv
[ -s $G/${TAG}lo_bench_under_rocprof_serial.json ] && tail -1 $G/${TAG}lo_bench_under_rocprof_serial.json > profiles/${TAG}_bench_under_rocprof_low_opacity_serial.json
ls $G/${TAG}lo_pmc_*.txt > /dev/null 2>&1 && cat $G/${TAG}lo_pmc_FETCH_SIZE.txt $G/${TAG}lo_pmc_WRITE_SIZE.txt $G/${TAG}lo_pmc_SQ_BUSY_CYCLES.txt $G/${TAG}lo_pmc_SQ_ACTIVE_INST_LDS.txt > profiles/${TAG}_pmc_low_opacity.txt
[ -s $G/${TAG}c5_stats_serial/stats_kernel_stats.csv ] && cp $G/${TAG}c5_stats_serial/stats_kernel_stats.csv profiles/${TAG}_kernel_stats_c5_serial.csv
[ -s $G/${TAG}c5_bench_under_rocprof_serial.json ] && tail -1 $G/${TAG}c5_bench_under_rocprof_serial.json > profiles/${TAG}_bench_under_rocprof_c5_serial.json
ls $G/${TAG}c5_pmc_*.txt > /dev/null 2>&1 && cat $G/${TAG}c5_pmc_FETCH_SIZE.txt $G/${TAG}c5_pmc_WRITE_SIZE.txt > profiles/${TAG}_pmc_c5.txt
python3 - "$TAG" <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
tag = sys.argv[1]
def raster(path):
    for line in open(path):
        if line.startswith("void rasterizeTilesKernel"):
            return eval(line[line.index("{"):line.index("}") + 1]), int(line.rsplit(" ", 1)[1])
    raise SystemExit("no raster line in " + path)
f, n = raster(f"profiles/{tag}_pmc_fetch_size_kb.txt")
w, _ = raster(f"profiles/{tag}_pmc_write_size_kb.txt")
sq = {}
for line in open(f"profiles/{tag}_pmc_sq_counters.txt"):
    if line.startswith("void rasterizeTilesKernel"):
        sq.update(eval(line[line.index("{"):line.index("}") + 1]))
simds, ses = 256 * 4, 32
busy = sq["SQ_BUSY_CYCLES"] / ses          # SQ_BUSY_CYCLES counts per shader engine (8 XCDs x 4): cycles the kernel kept an SE busy
out = {
    "kernel": "rasterizeTilesKernel<2, gaussian, SH 3>", "source_hash": bench.source_hash(),
    "workload": "bench.py default scene (1 M splats, SH 3, 1920x1080, seed 1234), --steps 16 --warmup 4 --inflight 1 --no-graph",
    "fetch_size_kb_raw": f["FETCH_SIZE"], "write_size_kb": w["WRITE_SIZE"], "dispatches": n,
    # FETCH_SIZE tallies 64 bytes per memory-side read request whatever the request's size (profiles/r04_fetch_calibration.txt: 4-, 16-, 32- and
    # 64-byte gathers 64 B each, a 128-byte line 64 B, a 192-byte coefficient row 128 B, a 16-byte-per-lane stream half its bytes).
    "fetch_requests_per_launch": round(f["FETCH_SIZE"] * 1024 / 64),
    "traffic_bytes_per_launch_low": (f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024,
    "traffic_bytes_per_launch_high": (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024,
    "traffic_bytes_per_launch": (1.25 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024,
    "traffic_source": f"profiles/{tag}_pmc_fetch_size_kb.txt + {tag}_pmc_write_size_kb.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                      "kernel trace only, mean per dispatch; FETCH_SIZE corrected by the calibration MI355X_MICROARCH.md prescribes for non-streaming reads "
                      "(tools/fetch_calibration.sh -> profiles/r04_fetch_calibration.txt): the counter tallies 64 B per read request, a request moves 64 B (a "
                      "gather of up to 64 B) or 128 B (a whole line); the kernel's misses are one 32-byte record, one 16-byte position (64 B each) and one "
                      "192-byte coefficient row (192 B, tallied 128) per entry = 320 B moved per 256 B tallied: x 1.25.  _low is the raw tally (every request "
                      "64 B), _high counts every request as a full 128-byte line",
    "sq": sq,
    "sq_source": f"profiles/{tag}_pmc_sq_counters.txt; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md:488), SQ_BUSY_CYCLES per shader engine",
    "busy_cycles_per_se": busy,
    "waves_per_simd": round(4 * sq["SQ_WAVE_CYCLES"] / (simds * busy), 3),
    "valu_issue_frac": round(4 * sq["SQ_ACTIVE_INST_VALU"] / (simds * busy), 3),
    "cycles_per_valu_instruction": round(4 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"], 2),
    "waves_per_simd_if_counters_were_cycles": round(sq["SQ_WAVE_CYCLES"] / (simds * busy), 3),
}
json.dump(out, open(f"profiles/{tag}_pmc_raster.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("traffic_bytes_per_launch", "waves_per_simd", "valu_issue_frac", "cycles_per_valu_instruction", "source_hash")}))
PY
ls profiles | grep $TAG
