"""bench.py's N > 1 branches on a machine with one GPU: two ranks launched by torch.distributed.run, both on device
0, collectives over gloo (ORBX_BENCH_REHEARSE=1).  Checks that the ranks get through every barrier, reduction and
the gather + merge of the sharded search, and that rank 0 prints one well-formed line; the numbers mean nothing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_reach_the_json_line():
    env = dict(os.environ, ORBX_BENCH_REHEARSE="1", ORBX_BENCH_NT_MAP="300001", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--batches-per-step", "2", "--batch-1080p", "2", "--no-host-path", "--cpu-pairs", "0"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f"the two-rank rehearsal did not finish within its limit: {(e.stderr or b'')[-2000:]}")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d
    assert d["config"]["frames_per_step"] == 2 * 4 * 2 * 2            # 2 images x 4 pairs x 2 ranks x 2 batches
    assert d["hamming"]["result_check"]["ok"] and d["hamming"]["workload"].startswith("5000 queries x 300001")
    assert d["extract_1080p"]["frames_per_s"] > 0
    assert "cpu_baseline" not in d                                   # rank 0 at N = 1 only
