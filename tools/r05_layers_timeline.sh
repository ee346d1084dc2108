#!/bin/bash
# usage (GPU box): tools/r05_layers_timeline.sh [bench flags] -- tools/pipeline_timeline.py on a rocprofv3 kernel trace of the pipelined loop
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o t -- python3 $ROOT/bench.py --steps 1024 --warmup 64 --cpu-frames 0 --no-extra --spin-up 0 "$@" > /dev/null 2>&1
python3 $ROOT/tools/pipeline_timeline.py $(find /tmp/ptl -name "*kernel_trace.csv" | head -1)
