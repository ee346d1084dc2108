#!/bin/bash
# world-8 band timing matrix
run() { echo "$@"; timeout -k 10 120 python tools/band_sim.py --worlds 8 "$@" 2>&1 | grep "^world" || exit 1; }
run --inflight 6
run --inflight 3 --sort-backend 2
run --inflight 6 --sort-backend 2
run --inflight 3 --ppl 1
run --inflight 1 --ppl 1
run --inflight 1 --sort-backend 2
