#!/bin/bash
# usage: tools/collect_profiles.sh r01   -- copies the summaries tools/profile_round.sh left in gpurun_out/ into profiles/
TAG=${1:-r01}
G=gpurun_out
tail -1 $G/${TAG}_bench_c3.json > profiles/${TAG}_bench_c3.json
tail -1 $G/${TAG}_bench_under_rocprof.json > profiles/${TAG}_bench_under_rocprof_pipelined.json
tail -1 $G/${TAG}_bench_under_rocprof_serial.json > profiles/${TAG}_bench_under_rocprof_serial.json
cp $G/${TAG}_stats/stats_kernel_stats.csv profiles/${TAG}_kernel_stats_pipelined.csv
cp $G/${TAG}_stats_serial/stats_kernel_stats.csv profiles/${TAG}_kernel_stats_serial.csv
cp $G/${TAG}_pmc_FETCH_SIZE/pmc_counter_collection_summary.txt profiles/${TAG}_pmc_fetch_size_kb.txt
cp $G/${TAG}_pmc_WRITE_SIZE/pmc_counter_collection_summary.txt profiles/${TAG}_pmc_write_size_kb.txt
python3 - "$TAG" <<'PY'
import json, re, sys
tag = sys.argv[1]
def raster(path, key):
    for line in open(path):
        if line.startswith("void rasterizeTilesKernel"):
            d = eval(line[line.index("{"):line.index("}") + 1])
            return d[key], int(line.rsplit(" ", 1)[1])
f, n = raster(f"profiles/{tag}_pmc_fetch_size_kb.txt", "FETCH_SIZE")
w, _ = raster(f"profiles/{tag}_pmc_write_size_kb.txt", "WRITE_SIZE")
json.dump({"kernel": "rasterizeTilesKernel<2>",
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 16 --warmup 4 --inflight 1 --no-graph, mean per dispatch over {n} dispatches; profiles/{tag}_pmc_fetch_size_kb.txt, {tag}_pmc_write_size_kb.txt",
           "fetch_size_kb_raw": f, "write_size_kb": w,
           "correction": "MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of WIDE coalesced streaming reads; the raster's reads are 4-B list entries plus 8/16-B gathers (uncalibrated width), so the raw value is reported as the lower bound and 2x as the upper bound",
           "traffic_bytes_per_launch_low": (f + w) * 1024, "traffic_bytes_per_launch_high": (2 * f + w) * 1024},
          open(f"profiles/{tag}_pmc_raster.json", "w"), indent=1)
# the bench line of this very run read the PREVIOUS counter file: attach the counters collected with it
bench = json.load(open(f"profiles/{tag}_bench_c3.json"))
bench["roofline"]["traffic"] = round((f + w) * 1024)
bench["roofline"]["traffic_source"] = (f"profiles/{tag}_pmc_raster.json (FETCH_SIZE raw + WRITE_SIZE; gather widths are uncalibrated, "
                                       f"upper bound {round((2 * f + w) * 1024)} with the wide-read 2x correction)")
json.dump(bench, open(f"profiles/{tag}_bench_c3.json", "w"))
PY
ls profiles
