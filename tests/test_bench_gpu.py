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
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels", "parity", "sustained", "other_configs",
              "scaling_by"):
        assert k in one, k
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["vs_baseline"] is None
    # float32 everywhere; the deep layers' GEMMs in float32-equivalent split-limb arithmetic on the bf16 MFMA by default,
    # the same step on the f32-input MFMA beside it
    assert one["dtype"] == "f32 (3xbf16 split-limb MFMA, f32 accumulate)" and one["f32_mfma_gemm"]["dtype"] == "f32"
    assert 0 < one["f32_mfma_gemm"]["value"] and "split_limb_gemm" not in one
    assert one["roofline"]["bound"] == "mfma" and 0 < one["roofline"]["frac"] < 1.0 and one["roofline"]["peak"] == 2500.0
    assert 0 < one["roofline"]["f32_input_launches"]["frac"] < 1.0 and one["roofline"]["f32_input_launches"]["peak"] == 157.3
    assert "frac_net" not in one["roofline"] and 0 < one["roofline"]["conv_family"]["frac"] < 1.0
    assert all(r.get("frac", 0.0) < 1.0 for r in one["kernels"])          # executed flops / bytes: never above the peak
    assert 0 < one["render_advect_family"]["survey_fused"]["frac_hbm"] < one["render_advect_family"]["as_built"]["frac_hbm"]
    assert one["parity"]["grad_rel_l2"] < one["parity"]["tolerance"]
    assert "64^3" in one["metric"] and one["sustained"]["windows"] >= 3
    assert len(one["other_configs"]) == 11 and not any("error" in c for c in one["other_configs"]), one["other_configs"]
    # configs[0] end to end on SURVEY 8(d)'s particle count, the sparse lattice of earlier rounds beside it
    assert sorted(c["particles"] for c in one["other_configs"] if "particles" in c) == [1344, 16384]
    # the data-dependent part of the headline is visible: the same step without skipping, and a density with nothing to skip
    skip = one["dead_region_skipping"]
    assert 0 < skip["live_voxel_fraction"] < 1 and 0 < skip["accumulated_box_fraction"] <= 1
    assert skip["skipping_off"]["value"] > 0 and skip["dense_density"]["live_voxel_fraction"] > 0.99
    widened = one["other_configs"][-1]["ops"]                         # SURVEY 8(f) operators: one timed call each
    assert len(widened) == 10 and all(o["ms"] > 0 and 0 < o["frac_hbm"] < 1 for o in widened), widened
    fast = args + ["--no-kernel-profile", "--no-other-configs", "--no-sustained"]
    # (a) the launcher form the driver uses, views sharded (strong scaling): 2 ranks share the GPU over gloo
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--scaling-by", "views"]
               + fast, env={"NFS_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["views_per_rank"] == 4 and two["scaling"] == "strong"
    assert abs(two["final_loss"] - one["final_loss"]) <= 1e-5 * abs(one["final_loss"])
    # (b) ``python bench.py --gpus 2`` WITHOUT a launcher must become two ranks by itself; the default for N > 1 is the
    # BASELINE metric itself -- the same metric, workload and unit as the N = 1 line, views sharded (strong scaling) --
    # with the frame-sharded sequence (weak scaling) of the same box beside it
    env = {"NFS_DIST_BACKEND": "gloo"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    v2 = _run([sys.executable, "bench.py", "--gpus", "2"] + args + ["--no-kernel-profile", "--no-sustained"], env=env)
    assert v2["n_gpus"] == 2 and v2["world_size_seen"] == 2 and v2["scaling"] == "strong" and v2["scaling_by"] == "views"
    assert v2["metric"] == one["metric"] and v2["unit"] == one["unit"]
    assert v2["config"]["workload"] == one["config"]["workload"] and v2["config"]["views_per_rank"] == 4
    assert abs(v2["final_loss"] - one["final_loss"]) <= 1e-5 * abs(one["final_loss"])
    assert v2["frames_weak"]["scaling"] == "weak" and v2["frames_weak"]["value"] > 0
    # frames as the headline on request; the same two-frame sequence on one rank: identical trajectory (frame sharding +
    # halo exchange are exact)
    seq2 = _run([sys.executable, "bench.py", "--gpus", "2", "--scaling-by", "frames"] + fast, env=env)
    assert seq2["scaling"] == "weak" and seq2["scaling_by"] == "frames" and seq2["config"]["frames"] == 2
    assert abs(seq2["views_strong"]["final_loss"] - one["final_loss"]) <= 1e-5 * abs(one["final_loss"])
    assert abs(seq2["final_loss"] - v2["frames_weak"]["final_loss"]) <= 1e-5 * abs(seq2["final_loss"])
    seq1 = _run([sys.executable, "bench.py", "--scaling-by", "frames", "--frames-per-rank", "2"] + fast)
    assert seq1["n_gpus"] == 1 and seq1["config"]["frames"] == 2
    assert abs(seq2["final_loss"] - seq1["final_loss"]) <= 1e-5 * abs(seq1["final_loss"])


def test_bench_gpus_4_and_8_over_gloo():
    """the rank counts of a SCALE run, functionally: ``bench.py --gpus 8`` (one view per rank, slabs of 64 / 8 planes)
    and ``--gpus 4`` (two views per rank) as the driver launches them, the ranks sharing the one GPU over gloo.  Same
    metric / workload as the one-rank line, the world size the ranks saw, and the trajectory of the one-rank run."""
    args = ["--steps", "3", "--warmup", "1", "--grid", "64", "--no-cpu-baseline", "--no-kernel-profile",
            "--no-other-configs", "--no-sustained", "--no-split-limb"]
    one = _run([sys.executable, "bench.py"] + args)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    for n in (8, 4):
        r = _run([sys.executable, "bench.py", "--gpus", str(n)] + args, env={"NFS_DIST_BACKEND": "gloo"})
        assert r["n_gpus"] == n and r["world_size_seen"] == n and r["collective_backend"] == "gloo"
        assert r["config"]["views_per_rank"] == 8 // n and r["scaling"] == "strong"
        assert r["metric"] == one["metric"] and r["config"]["workload"] == one["config"]["workload"]
        assert abs(r["final_loss"] - one["final_loss"]) <= 1e-5 * abs(one["final_loss"]), (n, r["final_loss"])
        assert r["config"]["field_work"].startswith("D-slab sharded"), r["config"]
        # every rank's own record: backend, group size, device, the step's collectives timed on the device
        assert len(r["ranks"]) == n and sorted(x["rank"] for x in r["ranks"]) == list(range(n))
        for x in r["ranks"]:
            assert x["backend"] == "gloo" and x["world_size"] == n and x["views_local"] == 8 // n
            assert set(x["collectives"]) == {"reduce_scatter", "all_gather"}, x["collectives"]
            assert all(c["calls_per_step"] == 1.0 and c["device_ms_per_step"] > 0 for c in x["collectives"].values())
        assert r["collective_ms_per_step_max_over_ranks"] > 0 and r["distinct_devices"] == 1      # (one GPU shared here)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "1"], cwd=ROOT, capture_output=True,
                       text=True, env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
