#!/bin/bash
# usage (GPU box, repo root): tools/r05_ab_lib.sh ALT.so [bench flags] -- the in-tree library against another build of it, order-balanced
# (T A A T T A A T T A after a warm-up run) on ONE box: frames/s of bench.py's timed loop
ALT=$1; shift
export GPU_MAX_HW_QUEUES=8
python bench.py --steps 512 --warmup 64 --cpu-frames 0 --no-extra "$@" > /dev/null 2>&1
for which in T A A T T A A T T A; do
  if [ $which = T ]; then unset GSPLAT_HIP_LIBRARY; else export GSPLAT_HIP_LIBRARY=$PWD/$ALT; fi
  echo "$which $(python bench.py --steps 2048 --warmup 64 --cpu-frames 0 --no-extra "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['depth_layers']['in_use'])")"
done | tee /tmp/ab.txt
python - <<'PY'
import collections
v = collections.defaultdict(list)
for line in open("/tmp/ab.txt"):
    w, x, _ = line.split()
    v[w].append(float(x))
t, a = sum(v["T"]) / len(v["T"]), sum(v["A"]) / len(v["A"])
print("tree mean %.1f, alternative mean %.1f: tree %+.2f %%" % (t, a, 100 * (t / a - 1)))
PY
