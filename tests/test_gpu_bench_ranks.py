"""bench.py's N > 1 path — one process per rank, tile columns dealt to the ranks, ONE all_gather per frame — run with eight ranks that
share the box's single GPU (MPR_BENCH_SHARE_GPU=1, the gather through gloo): not a measurement, a check that the path the driver
launches on an 8-GPU node (python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...) works end to end and says what it
has to say: the gathered frame equals a single-GPU frame on every rank, the world size the collective saw, the gather's own time,
rank 0's kernel times."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("ranks", [8])
def test_bench_line_of_eight_ranks_on_one_device(ranks):
    env = dict(os.environ, MPR_BENCH_SHARE_GPU="1", MPR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "4", "--warmup", "2", "--size", "512"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["steps"] == 4 and out["warmup"] == 2
    assert out["verified_against_single_gpu"] is True
    c = out["collective"]
    assert c["world_size_seen"] == ranks and c["shared_one_gpu"] is True and c["all_gather_ms_mean_rank0"] > 0 and c["bytes_per_rank"] > 0
    assert out["config"]["parallelism"] == "tile-columns x%d" % ranks and out["scaling"] == "strong"
    assert "eval_voxels_f" in out["kernel_ms_rank0"] and "eval_tiles_i" in out["kernel_ms_rank0"]
    assert out["roofline"]["frac"] > 0 and out["value"] > 0
