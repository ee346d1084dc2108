#!/bin/bash
# usage (GPU box, repo root): tools/r04_fusion_variants.sh ROUNDS name ...   -- alternating bench runs (fusion on) of the in-tree library
# ("tree"; also run with fusion off as the reference point) and of the variant builds ab_variants/lib_<name>.so, on ONE box.
ROUNDS=${1:-2}; shift
for r in $(seq $ROUNDS); do
  FUSION_VALUES="0 1" AB_LABEL=tree tools/r04_fusion_ab.sh 1 $BENCH_ARGS
  for name in "$@"; do
    FUSION_VALUES=1 AB_LABEL=$name GSPLAT_HIP_LIBRARY=$PWD/ab_variants/lib_$name.so tools/r04_fusion_ab.sh 1 $BENCH_ARGS
  done
done
