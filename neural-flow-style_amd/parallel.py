"""Multi-GPU plumbing of the one exchange step the path has: views (or
frames) shard over ranks, every rank holds a full replica of the field, and ONE all-reduce(sum) of
the field gradient per iteration makes the replicas take the identical Adam step.

The functions are backend-agnostic (`nccl` = RCCL over xGMI on the GPU box, `gloo` in the CPU
tests): they only touch `torch.distributed`, never the HIP kernels.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def rank_world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard(items, rank=None, world=None, group=None):
    """round-robin shard of a list of work units (views / frames): unit i goes to rank i % world.
    With world | len(items) every rank gets the same count (the bench asserts it)."""
    if rank is None or world is None:
        rank, world = rank_world(group)
    return list(items[rank::world])


# Optional timing of the collectives (bench.py --gpus N): when TIMER is a dict every collective below is bracketed by an
# event pair on the current stream (the stream the collective's result is waited on) and a host clock; entries are
# name -> [(start_event, end_event, payload_bytes, host_seconds)].  Off by default: two event records per call.
TIMER = None


def _timed(name, t, fn):
    if TIMER is None or not t.is_cuda:
        return fn()
    import time
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    h0 = time.perf_counter()
    r = fn()
    h1 = time.perf_counter()
    e1.record()
    TIMER.setdefault(name, []).append((e0, e1, int(t.numel() * t.element_size()), h1 - h0))
    return r


def timer_summary(steps):
    """per collective: calls and milliseconds per step (device: between the event pairs; host: inside the call), payload
    bytes per call.  Call after a device synchronisation; empties the timer."""
    global TIMER
    out = {}
    for name, recs in (TIMER or {}).items():
        out[name] = {"calls_per_step": len(recs) / float(steps),
                     "device_ms_per_step": sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs) / float(steps),
                     "host_ms_per_step": 1e3 * sum(h for _, _, _, h in recs) / float(steps),
                     "payload_bytes_per_call": recs[0][2]}
    TIMER = None
    return out


def all_reduce_sum_(tensors, group=None):
    """in-place sum over ranks of a list of tensors (field gradient, loss); no-op for world 1"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    for t in tensors:
        _timed("all_reduce", t, lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group))
    return tensors


def reduce_scatter_sum(out, inp, group=None):
    """out [chunk...] = sum over ranks of inp[rank] (inp [world, chunk...], contiguous): the first half of an all-reduce,
    every rank keeps only its own chunk"""
    assert out.is_contiguous() and inp.is_contiguous() and inp.numel() == out.numel() * dist.get_world_size(group)
    # (flat views: gloo wants input.shape[0] == world * output.shape[0])
    _timed("reduce_scatter", inp,
           lambda: dist.reduce_scatter_tensor(out.view(-1), inp.view(-1), op=dist.ReduceOp.SUM, group=group))
    return out


def all_gather_into(out, inp, group=None):
    """out [world, chunk...] <- every rank's inp [chunk...]: the second half of an all-reduce"""
    assert out.is_contiguous() and inp.is_contiguous() and out.numel() == inp.numel() * dist.get_world_size(group)
    _timed("all_gather", out, lambda: dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=group))
    return out


def slab_plan(D, world):
    """D planes in ``world`` contiguous slabs of ``cs = ceil(D / world)`` planes (the last ranks may be short or empty):
    -> (cs, [(z0, z1)] per rank)"""
    cs = -(-int(D) // int(world))
    return cs, [(min(r * cs, D), min((r + 1) * cs, D)) for r in range(world)]


def slab_pack_index(D, world):
    """plane indices that build the send buffer of the D-slab reduce-scatter from the padded gradient volume
    gpad [D + 5] (planes [2, D + 2) = the gradient, 0, 1, D + 2, D + 3 zero, D + 4 the loss plane): chunk k = planes
    [k cs - 2, k cs + cs + 2) of the volume (zero outside it) + the loss plane.  The definition ``nfs_slab_pack``
    implements as one copy kernel (tests hold the kernel to ``gpad.index_select(0, this)``)."""
    cs, _ = slab_plan(D, world)
    idx = []
    for k in range(world):
        for j in range(cs + 4):
            z = k * cs - 2 + j
            idx.append(z + 2 if -2 <= z < D + 2 else 0)                   # (plane 0 is a zero plane)
        idx.append(D + 4)
    return idx


def replicas_identical(t, group=None, atol=0.0):
    """debug check: max |t - t_rank0| over ranks is <= atol"""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    ref = t.clone()
    dist.broadcast(ref, src=0, group=group)
    diff = (t - ref).abs().max()
    dist.all_reduce(diff, op=dist.ReduceOp.MAX, group=group)
    return float(diff) <= atol


# ---- frame sharding of a sequence ("Frames") ------------------------------------------------

def plan_frames(num_frames, interp=1, frames_per_opt=1, world=1):
    """Contiguous blocks of key frames per rank.  Key frames are 0, interp, 2*interp, ... (styler_3p.py:304); frames
    of one optimiser group ``t // frames_per_opt`` (styler_3p.py:315) stay on one rank because their Adam state is
    updated sequentially.  Returns a list (per rank) of lists of key-frame indices; blocks differ by at most one
    group."""
    keys = list(range(0, num_frames, max(int(interp), 1)))
    groups = {}
    for t in keys:
        groups.setdefault(t // max(int(frames_per_opt), 1), []).append(t)
    gids = sorted(groups)
    out = []
    for rnk in range(world):
        lo = len(gids) * rnk // world
        hi = len(gids) * (rnk + 1) // world
        out.append([t for g in gids[lo:hi] for t in groups[g]])
    return out


def _is_gloo(group=None):
    return dist.get_backend(group) == "gloo"


def exchange_frames(have, need, owner, like, group=None, need_by_rank=None):
    """Point-to-point exchange of per-frame tensors (the temporal filter's halo): ``have`` {frame: tensor} are this
    rank's frames, ``need`` the frames this rank wants, ``owner`` {frame: rank} (identical on all ranks).
    ``need_by_rank`` (list over ranks of frame collections, identical on all ranks) says what every rank wants; it is
    a pure function of the frame plan and the filter matrix, so callers compute it locally -- when it is not given
    the need lists are all-gathered (a few integers, but a host round trip per call).  Returns {frame: tensor} for
    every frame of ``need``.  gloo has no GPU point-to-point: tensors are staged through the host there (CPU tests /
    one-GPU functional checks); under nccl (RCCL) the device buffers go over xGMI directly, all transfers of a call
    in one group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {t: have[t] for t in need}
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if need_by_rank is None:
        needs = [None] * world
        dist.all_gather_object(needs, sorted(int(t) for t in need), group=group)
    else:
        needs = [sorted(int(t) for t in n_) for n_ in need_by_rank]
        assert needs[rank] == sorted(int(t) for t in need), "need_by_rank[rank] must equal this rank's need"
    stage = _is_gloo(group) and like.is_cuda
    out, ops_, keep = {}, [], []
    for t in sorted(int(t) for t in need):
        if owner[t] == rank:
            out[t] = have[t]
    # a fixed global order of (src, dst, frame) triples keeps the send/recv pairing identical on both ends
    for src in range(world):
        for dst in range(world):
            if src == dst:
                continue
            for t in needs[dst]:
                if owner[t] != src:
                    continue
                if rank == src:
                    buf = have[t].detach().contiguous()
                    buf = buf.cpu() if stage else buf
                    keep.append(buf)
                    ops_.append(dist.P2POp(dist.isend, buf, dst, group=group))
                elif rank == dst:
                    buf = torch.empty(like.shape, dtype=like.dtype, device="cpu" if stage else like.device)
                    out[t] = buf
                    ops_.append(dist.P2POp(dist.irecv, buf, src, group=group))
    if ops_:
        for req in dist.batch_isend_irecv(ops_):
            req.wait()
    if stage:
        out = {t: (v.to(like.device) if not v.is_cuda else v) for t, v in out.items()}
    return out


def all_reduce_sum_coalesced_(tensors, group=None):
    """one collective for several tensors of one dtype/device (field gradient + loss): flattened into one buffer,
    reduced, copied back"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n
    return tensors
