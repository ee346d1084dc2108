#!/bin/bash
# usage (on the GPU box): tools/gpu_steps.sh TAG "cmd1" "cmd2" ...
# Runs the commands one after another, each under its own `timeout -k 10 LIMIT` (LIMIT = $STEP_LIMIT, default 420 s), stdout+stderr
# of step i to gpurun_out/TAG_step<i>.log.  A failing step does not stop the sequence; a step that had to be KILLED at its limit
# does (no further GPU work after a hang).
TAG=$1; shift
LIMIT=${STEP_LIMIT:-420}
mkdir -p gpurun_out
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "== step $i: $cmd" | tee gpurun_out/${TAG}_step${i}.log
  timeout -k 10 $LIMIT bash -c "$cmd" >> gpurun_out/${TAG}_step${i}.log 2>&1
  rc=$?
  echo "== step $i rc=$rc"
  tail -3 gpurun_out/${TAG}_step${i}.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $i was killed at its limit: stopping"; exit 1; fi
done
exit 0
