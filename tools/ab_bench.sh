#!/bin/bash
# usage (GPU box, repo root): tools/ab_bench.sh ROUNDS lib1.so lib2.so ...   ("" = the in-tree library; "old" = ab_old/bench.py)
# Alternates short bench runs of library builds on the SAME box and prints frames/s per run: box-to-box variation is ~10 %,
# run-to-run on one box ~0.5 %, so kernel variants are compared this way.
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = "old" ]; then v=$(python ab_old/bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra | tail -1 | grep -o '"value": [0-9.]*');
    elif [ "$lib" = "tree" ]; then v=$(python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra | tail -1 | grep -o '"value": [0-9.]*');
    else v=$(GSPLAT_HIP_LIBRARY=$PWD/$lib python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra | tail -1 | grep -o '"value": [0-9.]*'); fi
    echo "$lib $v"
  done
done
