#!/bin/bash
# usage (GPU box, repo root): tools/r05_ab_flags.sh "FLAGS_T" "FLAGS_A" [common bench flags] -- two settings of bench.py, order-balanced
# (T A A T T A A T T A after a warm-up run) on ONE box: frames/s of the timed loop
T=$1; A=$2; shift 2
export GPU_MAX_HW_QUEUES=8
python bench.py --steps 512 --warmup 64 --cpu-frames 0 --no-extra "$@" > /dev/null 2>&1
for which in T A A T T A A T T A; do
  if [ $which = T ]; then F=$T; else F=$A; fi
  echo "$which $(python bench.py --steps 2048 --warmup 64 --cpu-frames 0 --no-extra $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['depth_layers']['in_use'])")"
done | tee /tmp/abf.txt
python - "$T" "$A" <<'PY'
import collections, sys
v = collections.defaultdict(list)
for line in open("/tmp/abf.txt"):
    w, x, _ = line.split()
    v[w].append(float(x))
t, a = sum(v["T"]) / len(v["T"]), sum(v["A"]) / len(v["A"])
print("T [%s] mean %.1f, A [%s] mean %.1f: T %+.2f %%" % (sys.argv[1], t, sys.argv[2], a, 100 * (t / a - 1)))
PY
