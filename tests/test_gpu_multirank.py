"""The N > 1 control flow of bench.py on ONE GPU: two ranks share cuda:0 and gather over gloo (`--rehearse`; RCCL needs one
GPU per rank, which the test box does not have).  Covers the band split, culled band renders with frames in flight, captured
frames driven from worker threads, the per-frame gather ordering, and the byte-for-byte check of the gathered frame against a
single-GPU render."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_multirank_rehearsal(ranks):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + ranks), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "24", "--warmup", "6",
           "--splats", "200000", "--cpu-frames", "0", "--rehearse"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    result = json.loads(line)
    assert result["n_gpus"] == ranks and result["steps"] == 24 and result["value"] > 0
    assert result["scaling"] == "strong"
    assert len(result["config"]["tile_row_bands"]) == ranks
    assert result["config"]["gathered_frame_equals_single_gpu_frame"] is True
