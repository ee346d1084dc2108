#!/usr/bin/env python3
"""What runs beside what in the pipelined frame loop: from a rocprofv3 kernel trace (CSV) of `bench.py` with four frames in flight,
the distribution of the number of kernels in flight, the share of the time in which NO heavy kernel (count / emit / raster / a
scatter / clip) is running, and per kernel the average number of OTHER kernels that ran beside it.
usage: tools/pipeline_timeline.py <kernel_trace.csv>"""
import csv, re, sys, collections

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gs::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("gs::(anonymous namespace)::", "").replace("void ", ""))
    name = re.sub(r"<unsigned (short|int), (\d), (\d)(, gs::DepthKeySource|, NoKeySource)?>", lambda m: "<%s%s%s>" % ("u16" if m.group(1) == "short" else "u32", "," + m.group(2), ",keygen" if m.group(4) and "Depth" in m.group(4) else ""), name)
    ev.append((s, e, name, r.get("Queue_Id", "")))
ev.sort()
# the densest window: the pipelined loop is where starts are closest together -- take the middle half of the trace by dispatch index
lo, hi = ev[len(ev) // 4][0], ev[3 * len(ev) // 4][0]
win = [x for x in ev if x[0] >= lo and x[1] <= hi]
heavy = ("tileListFastKernel", "rasterizeTilesKernel", "radixScatterKernel", "splatClipDataKernel")
points = []
for s, e, name, q in win:
    points.append((s, 1, name))
    points.append((e, -1, name))
points.sort()
running = collections.Counter()
hist = collections.Counter()
no_heavy = 0
t_prev = points[0][0]
for t, d, name in points:
    dt = t - t_prev
    n = sum(running.values())
    hist[n] += dt
    if not any(running[k] for k in running if k.startswith(heavy)):
        no_heavy += dt
    running[name] += d
    t_prev = t
total = sum(hist.values())
frames = sum(1 for x in win if x[2].startswith("rasterizeTilesKernel"))
print("window %.1f ms, %d frames -> %.4f ms per frame; queues: %s" % (total / 1e6, frames, total / 1e6 / max(frames, 1), sorted(set(x[3] for x in win))))
print("kernels in flight: " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / total) for k, v in sorted(hist.items())))
print("mean kernels in flight %.2f; time with no heavy kernel (count, emit, raster, scatter, clip) running: %.1f %%" % (sum(k * v for k, v in hist.items()) / total, 100.0 * no_heavy / total))
# per kernel: duration and mean concurrency of others during it
by = collections.defaultdict(list)
starts = sorted(win)
for s, e, name, q in win:
    by[name].append(e - s)
print("%-44s %8s %10s %8s" % ("kernel", "calls", "mean us", "share %"))
allt = sum(sum(v) for v in by.values())
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-44s %8d %10.1f %8.1f" % (name[:44], len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / allt))
