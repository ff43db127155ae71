"""bench.py contract: the one-line JSON, and the sharded multi-rank path (2 ranks sharing the one GPU over
gloo: functional check of view sharding + all-reduce + max-over-ranks timing; the real runs use RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly ONE JSON line (rank 0 only)
    return json.loads(lines[0])


def test_bench_line_and_sharded_ranks_reproduce_the_single_rank_trajectory():
    args = ["--steps", "3", "--warmup", "1", "--grid", "64", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py"] + args)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels", "parity"):
        assert k in one, k
    assert one["n_gpus"] == 1 and one["dtype"] == "f32" and one["scaling"] == "strong" and one["vs_baseline"] is None
    assert one["roofline"]["bound"] == "mfma" and 0 < one["roofline"]["frac"] < 1.2
    assert one["parity"]["grad_rel_l2"] < one["parity"]["tolerance"]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--no-kernel-profile"] + args,
               env={"NFS_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["views_per_rank"] == 4
    assert abs(two["final_loss"] - one["final_loss"]) <= 1e-5 * abs(one["final_loss"])
