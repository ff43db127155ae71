"""world_size-2 gloo test (CPU) of the N>1 logic: view sharding covers every view exactly once, the
all-reduced gradient equals the single-process gradient of the summed view losses, and the replicas
stay identical after the (TF-semantics) Adam step.  The per-view loss here is a small pure-torch
stand-in -- the collective logic under test is independent of the HIP kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _view_loss(field, rot):
    # toy differentiable "render + loss" of one view
    return ((field * rot.sum()).sin() ** 2).sum() + (field ** 2).sum() * rot[0, 0]


def _adam_tf(x, m, v, g, t, lr=0.1, b1=0.9, b2=0.999, eps=1e-8):
    m = b1 * m + (1 - b1) * g; v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return x - lr_t * m / (v.sqrt() + eps), m, v


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_flow_style_amd import parallel
    from neural_flow_style_amd import synthetic as S
    mats = [torch.tensor(m, dtype=torch.float32) for m in S.uniform_views(8)]
    mine = parallel.shard(list(range(8)))
    torch.manual_seed(0)
    x = torch.randn(6, 5, 4); m = torch.zeros_like(x); v = torch.zeros_like(x)
    for t in range(1, 4):
        xr = x.clone().requires_grad_()
        loss = sum(_view_loss(xr, mats[i]) for i in mine)
        (g,) = torch.autograd.grad(loss, xr)
        tot = loss.detach().clone()
        parallel.all_reduce_sum_([g, tot])
        x, m, v = _adam_tf(x, m, v, g, t)
        assert parallel.replicas_identical(x)
    q.put((rank, mine, x.numpy(), float(tot)))
    dist.destroy_process_group()


def test_view_sharding_allreduce_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs.sort(key=lambda o: o[0])
    assert sorted(outs[0][1] + outs[1][1]) == list(range(8))          # every view exactly once
    assert np.array_equal(outs[0][2], outs[1][2])                      # replicas identical
    # single-process reference: gradient of the sum over all 8 views
    from neural_flow_style_amd import synthetic as S
    mats = [torch.tensor(m, dtype=torch.float32) for m in S.uniform_views(8)]
    torch.manual_seed(0)
    x = torch.randn(6, 5, 4); m = torch.zeros_like(x); v = torch.zeros_like(x)
    for t in range(1, 4):
        xr = x.clone().requires_grad_()
        loss = sum(_view_loss(xr, mats[i]) for i in range(8))
        (g,) = torch.autograd.grad(loss, xr)
        x, m, v = _adam_tf(x, m, v, g, t)
    np.testing.assert_allclose(outs[0][2], x.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(outs[0][3], float(loss), rtol=1e-5)


def test_shard_is_a_partition_for_any_world():
    from neural_flow_style_amd import parallel
    for world in (1, 2, 4, 8):
        got = sum((parallel.shard(list(range(8)), r, world) for r in range(world)), [])
        assert sorted(got) == list(range(8))
        assert len({len(parallel.shard(list(range(8)), r, world)) for r in range(world)}) == 1
