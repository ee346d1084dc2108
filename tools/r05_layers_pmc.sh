#!/bin/bash
# usage (GPU box): tools/r05_layers_pmc.sh -- VALU instructions per kernel of captured frames with one and with two depth layers (headline workload,
# one frame in flight; separate --pmc pass, kernel trace only), per-dispatch means by template instantiation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in 1 2; do
  rm -rf /tmp/lpmc$L
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/lpmc$L -o pmc -- python3 $ROOT/bench.py --steps 16 --warmup 8 --cpu-frames 0 --no-extra --spin-up 0 --inflight 1 --depth-layers $L "$@" > /dev/null 2>&1
  f=$(find /tmp/lpmc$L -name "*counter_collection.csv" | head -1)
  echo "== --depth-layers $L"
  PMC_KEEP_TEMPLATES=1 python3 $ROOT/tools/pmc_summary.py $f | grep -v "at::\|rocclr\|unpack\|sphericalHarm"
done
