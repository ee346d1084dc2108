#!/bin/bash
# usage: tools/bench_sweep.sh  -> runs a few bench variants, prints the per-stage ms
for args in "--inflight 1" "--inflight 2" "--inflight 3" "--inflight 3 --no-graph" "--inflight 1 --no-graph" "--inflight 3 --opacity-mean -3"; do
  timeout -k 10 300 python bench.py --steps 512 --warmup 32 --cpu-frames 0 $args 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('$args', 'fps', r['value'], 'P', r['config']['pairs_per_frame'], 'Pc', r['config']['consumed_pairs_per_frame'], {k:v['ms'] for k,v in r['stages'].items()})"
done
