#!/bin/bash
# usage (GPU box): tools/ablation.sh  -- frames/s of the pipelined bench with stages left out, cumulatively from the back of the
# frame (so that no kernel ever reads data a skipped stage should have produced): marginal cost of every stage in the pipelined loop
# (--depth-layers 1: the stages of the ONE-LIST frame; a layered frame has no stage that could be left out alone)
for m in 0 64 96 112 120 124 126 1 2; do
  python bench.py --steps 512 --warmup 32 --cpu-frames 0 --no-extra --depth-layers 1 --ablate $m | tail -1 | cut -c1-100
done
