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


# ---- frame sharding of a sequence (configs[3]) ----------------------------------------------------------------------

def test_plan_frames_is_a_partition_and_keeps_optimiser_groups_whole():
    from neural_flow_style_amd import parallel
    for F, interp, fpo in ((60, 1, 1), (60, 1, 10), (61, 2, 4), (7, 1, 3), (5, 2, 1), (3, 1, 1)):
        keys = list(range(0, F, interp))
        for world in (1, 2, 3, 8):
            plan = parallel.plan_frames(F, interp, fpo, world)
            assert len(plan) == world
            assert sum(plan, []) == keys                               # contiguous blocks in frame order
            owner = {t: r for r, ts in enumerate(plan) for t in ts}
            for t in keys:                                             # one Adam state never straddles ranks
                assert all(owner[s] == owner[t] for s in keys if s // fpo == t // fpo)
            n_groups = [len({t // fpo for t in ts}) for ts in plan]
            assert max(n_groups) - min(n_groups) <= 1


def test_frames_needed_covers_the_filter_window_and_tolerates_idle_ranks():
    """styler_grid.Styler.frames_needed(rank, world): a rank's density frames are its block of the plan, its velocity
    frames span every crossing its temporal filter makes; a world larger than the number of optimiser groups leaves
    ranks with two empty sets instead of an error (host logic only: no device is touched)"""
    from neural_flow_style_amd import parallel
    from neural_flow_style_amd.styler_grid import Styler
    from neural_flow_style_amd.util import temporal_weights
    st = Styler.__new__(Styler)
    for F, interp, fpo, sigma, world in ((12, 1, 1, 0.9, 3), (12, 1, 4, 2.0, 2), (3, 1, 1, 1.0, 8), (9, 2, 1, 0.0, 2)):
        st.num_frames, st.interp, st.frames_per_opt, st.window_sigma, st.pg = F, interp, fpo, sigma, None
        plan = parallel.plan_frames(F, interp, fpo, world)
        keys = list(range(0, F, interp))
        seen = []
        for r in range(world):
            dens, vels = st.frames_needed(r, world)
            assert dens == set(plan[r])
            seen += sorted(dens)
            if not plan[r]:
                assert vels == set()
                continue
            if sigma > 0:
                W = temporal_weights(len(keys), sigma)
                for t in plan[r]:
                    for jj in np.nonzero(W[keys.index(t)])[0]:
                        s_ = keys[jj]
                        # every frame crossed between s_ and t (forwards: s_..t-1, backwards: t..s_-1) has a velocity
                        assert set(range(min(s_, t), max(s_, t))) <= vels
            else:
                assert vels == set()
        assert seen == keys


def test_temporal_weights_is_the_matrix_of_denoise():
    """W @ x == util.denoise(x, (sigma,0,..)) -- the function pinned to the reference's own util.denoise by
    tests/golden/util_reference.npz; rows sum to one; non-zero band <= 4 sigma"""
    from neural_flow_style_amd.util import denoise, temporal_weights
    rng = np.random.RandomState(0)
    for n, sg in ((1, 2.0), (2, 2.0), (6, 0.8), (9, 3.0), (60, 2.0)):
        x = rng.randn(n, 4, 3).astype(np.float32)
        W = temporal_weights(n, sg)
        np.testing.assert_allclose(np.einsum("ts,sij->tij", W, x.astype(np.float64)), denoise(x, (sg, 0, 0)), atol=2e-7)
        np.testing.assert_allclose(W.sum(1), 1.0, atol=1e-12)
        r = int(4 * sg + 0.5)
        assert all(W[t, s] == 0 for t in range(n) for s in range(n) if abs(t - s) > r)


def _exchange_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_flow_style_amd import parallel
    from neural_flow_style_amd.util import temporal_weights
    F, sigma = 11, 0.9
    plan = parallel.plan_frames(F, 1, 2, world)
    owner = {t: r for r, ts in enumerate(plan) for t in ts}
    W = temporal_weights(F, sigma)
    have = {t: torch.full((3, 2), float(t)) + torch.arange(6.).view(3, 2) * 0.01 for t in plan[rank]}
    need = set(plan[rank])
    for t in plan[rank]:
        need |= set(int(s) for s in np.nonzero(W[t])[0])
    got = parallel.exchange_frames(have, need, owner, torch.zeros(3, 2))
    # the filtered value of my frames, computed from what arrived
    mine = {t: sum(float(W[t, s]) * got[s] for s in sorted(need) if W[t, s] != 0).numpy() for t in plan[rank]}
    q.put((rank, sorted(need), {t: v.numpy() for t, v in got.items()}, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_of_frame_updates_world3():
    """three gloo ranks: every rank receives exactly the frames its temporal filter reaches, bit-identical to the
    owner's tensors, and the filtered updates equal the single-process denoise"""
    from neural_flow_style_amd.util import denoise
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    F = 11
    full = np.stack([(torch.full((3, 2), float(t)) + torch.arange(6.).view(3, 2) * 0.01).numpy() for t in range(F)])
    want = denoise(full, (0.9, 0, 0))
    seen = set()
    for rank, need, got, mine in outs:
        assert sorted(got) == need
        for t, v in got.items():
            assert np.array_equal(v, full[t])
        for t, v in mine.items():
            np.testing.assert_allclose(v, want[t], atol=1e-6)
            seen.add(t)
    assert seen == set(range(F))


def _slab_worker(rank, world, port, q, D=11):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_flow_style_amd import parallel
    H, W = 3, 4
    cs, plan = parallel.slab_plan(D, world)
    z0, z1 = plan[rank]
    # every rank's local "gradient" (a function of rank and voxel), packed as engine.GridStylizer packs it: chunk k =
    # planes [k cs - 2, k cs + cs + 2) of the zero-padded volume + one plane whose first element is the local loss
    g = torch.arange(D * H * W, dtype=torch.float32).view(D, H, W) * (rank + 1)
    gpad = torch.zeros(D + 5, H, W)
    gpad[2:D + 2] = g
    gpad[D + 4].view(-1)[0] = 10.0 + rank
    idx = parallel.slab_pack_index(D, world)
    pack = gpad.index_select(0, torch.tensor(idx)).view(world, cs + 5, H, W).contiguous()
    recv = torch.empty(cs + 5, H, W)
    parallel.reduce_scatter_sum(recv, pack)
    # all-gather of a per-slab result (here: the received interior, i.e. the summed gradient of the slab)
    mine = torch.zeros(cs, H, W)
    mine[:z1 - z0] = recv[2:2 + (z1 - z0)]
    full = torch.empty(world, cs, H, W)
    parallel.all_gather_into(full, mine)
    q.put((rank, recv.numpy(), full.view(world * cs, H, W)[:D].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _slab_exchange_case(world, D):
    from neural_flow_style_amd import parallel
    H, W = 3, 4
    cs, plan = parallel.slab_plan(D, world)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_worker, args=(r, world, port, q, D)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = np.arange(D * H * W, dtype=np.float32).reshape(D, H, W) * (world * (world + 1) / 2.0)   # ranks 1 + ... + world
    padded = np.zeros((D + 4, H, W), np.float32)
    padded[2:D + 2] = total
    for rank, recv, full in got:
        z0 = rank * cs                       # (chunk origin; an idle rank's plan entry is clipped to (D, D))
        want = np.zeros((cs + 4, H, W), np.float32)
        hi = min(z0 + cs + 4, D + 4)
        if hi > z0:
            want[:hi - z0] = padded[z0:hi]
        np.testing.assert_array_equal(recv[:cs + 4], want)
        assert recv[cs + 4].reshape(-1)[0] == 10.0 * world + world * (world - 1) / 2.0      # sum of (10 + rank)
        np.testing.assert_array_equal(full, total)


def test_slab_reduce_scatter_with_overlapping_chunks_and_all_gather_world3():
    """the exchange of the D-slab sharded step (engine.GridStylizer._step_slab) on a ragged split (11 planes over 3
    ranks: 4 + 4 + 3): every rank receives the SUM over ranks of its slab with a two-plane halo (zero beyond the volume)
    and the summed loss; the all-gather of the slabs rebuilds the whole volume"""
    from neural_flow_style_amd import parallel
    cs, plan = parallel.slab_plan(11, 3)
    assert cs == 4 and plan == [(0, 4), (4, 8), (8, 11)]
    assert parallel.slab_plan(200, 8) == (25, [(25 * r, 25 * r + 25) for r in range(8)])
    assert parallel.slab_plan(5, 8)[1][5:] == [(5, 5)] * 3                      # idle ranks: empty slabs
    _slab_exchange_case(3, 11)


def test_slab_exchange_world8_ragged_split_with_an_idle_rank():
    """the world size a SCALE run uses, on a depth that does not divide: 27 planes over 8 ranks = 6 slabs of 4, one of 3
    and an EMPTY one (rank 7 owns no plane: it still sends its chunks and receives zeros)"""
    from neural_flow_style_amd import parallel
    cs, plan = parallel.slab_plan(27, 8)
    assert cs == 4 and plan[6] == (24, 27) and plan[7] == (27, 27)
    _slab_exchange_case(8, 27)


def test_slab_exchange_world4_and_world8_at_the_headline_split():
    """200 planes over 4 and 8 ranks divide evenly (50 / 25 planes): the plan, and the index table of the send buffer --
    every plane of the gradient appears in its owner's chunk at offset 2, the halo planes in the neighbours' chunks, the
    loss plane closes every chunk"""
    from neural_flow_style_amd import parallel
    for world in (4, 8):
        D = 200
        cs, plan = parallel.slab_plan(D, world)
        assert cs * world == D and plan == [(cs * r, cs * r + cs) for r in range(world)]
        idx = np.asarray(parallel.slab_pack_index(D, world)).reshape(world, cs + 5)
        for r in range(world):
            assert idx[r, cs + 4] == D + 4
            np.testing.assert_array_equal(idx[r, 2:cs + 2], np.arange(r * cs, r * cs + cs) + 2)     # the slab itself
            lo = [z + 2 for z in (r * cs - 2, r * cs - 1)]              # (planes 0, 1 of gpad are zero planes)
            hi = [z + 2 if z < D + 2 else 0 for z in (r * cs + cs, r * cs + cs + 1)]
            assert list(idx[r, :2]) == lo and list(idx[r, cs + 2:cs + 4]) == hi
    _slab_exchange_case(4, 10)
