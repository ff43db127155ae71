#!/usr/bin/env python
"""Headline benchmark: stylisation iterations/sec on the 200^3 smoke grid with 8 views
(BASELINE.json metric; workload = configs[2], the configuration the metric is quoted on).

  python bench.py --gpus N --steps K --warmup W        (N > 1 without WORLD_SIZE: re-executes itself under
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      torch.distributed.run)

One step at N = 1 = one stylisation iteration: advect -> smooth/clamp -> for all 8 views: rotate+render -> VGG-19
conv1_1..conv5_1 -> Gram style loss -> full adjoint chain -> TF-Adam update of the 200^3 x 3 velocity field.  (The
advect of iteration i + 1 is formed inside the Adam kernel of iteration i -- the updated velocity is in registers
there --, so every timed step still contains exactly one advect, one update and everything between them.)  Inputs are
synthetic (seed 123) and resident in HBM before the timed region.

N > 1 (``--scaling-by``):
  views (default)   the BASELINE metric: the 8 views of ONE frame sharded over the ranks (strong scaling); the exchange is
                    one all-reduce's worth of link traffic per iteration (reduce-scatter of the 32 MB density-field
                    gradient over D-slabs, slab-local field work, all-gather of the smoothed density).  Same metric,
                    workload and unit as the N = 1 line.  The frame-sharded sequence is measured in the same run and
                    reported under "frames_weak".
  frames            BASELINE configs[3] in weak scaling: N x ``--frames-per-rank`` frames of 200^3, every rank keeps
                    full 8-view batches for its own frames; one step = one iteration of the sequence loop (a
                    stylisation step per frame, halo exchange of the per-frame updates over RCCL point-to-point,
                    their temporal alignment by transport + Gaussian).  value = frame-iterations/s of the whole job;
                    the view-sharded number rides along under "views_strong".

Rank 0 prints ONE JSON line with the contract keys plus
  "roofline"      dominant kernel (the f32-MFMA Winograd GEMM) timed live with HIP events on its launch stream, in the
                  headline configuration,
  "kernels"       every C-ABI family the same way (HBM GB/s or TFLOP/s + fraction of peak; single-stream pass),
  "sustained"     >= 2 s of stepping in 20-step windows (median / min / max),
  "cpu_baseline"  the CPU oracle timed on this box's host cores on a bounded sample (N=1 only),
  "parity"        gradient relative-L2 of the HIP path vs the oracle on a small case (N=1 only),
  "other_configs" short side measurements of BASELINE configs[0], [1], [3], [4] (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TF = 2500.0   # v_mfma_f32_16x16x32_bf16 / 32x32x16 dense peak (MI355X_MICROARCH.md; never the sparse figure)
DTYPE_SPLIT = "f32 (3xbf16 split-limb MFMA, f32 accumulate)"
STYLE_LAYERS = ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--scaling-by", choices=["views", "frames"], default=None,
                    help="N>1: what is sharded (default views = the BASELINE metric, strong scaling; frames = the "
                         "sequence of configs[3] in weak scaling; N=1 is the single-frame 8-view workload either way)")
    ap.add_argument("--frames-per-rank", type=int, default=1)
    ap.add_argument("--window-sigma", type=float, default=2.0, help="temporal filter of the sequence (config default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the small HIP-vs-oracle gradient check")
    ap.add_argument("--no-skip-control", action="store_true",
                    help="skip the dead-region-skipping controls (the same step with skipping off / on a dense density)")
    ap.add_argument("--no-split-limb", action="store_true", help="skip the secondary split-limb GEMM measurement")
    ap.add_argument("--cpu-views", type=int, default=1,
                    help="views 0..n-1 in the bounded CPU sample, besides view V-1 (so the default times two views)")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="BASELINE.md's protocol for the CPU baseline: 1 warm-up + median of 3 whole iterations (~20 min)")
    return ap.parse_args()


def relaunch(args):
    """``python bench.py --gpus N`` outside a launcher: become N ranks (one per GPU) under torch.distributed.run"""
    import socket
    backend = os.environ.get("NFS_DIST_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU over RCCL); "
                         "NFS_DIST_BACKEND=gloo lets ranks share a GPU for a functional check"
                         % (args.gpus, torch.cuda.device_count()))
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


LIVE_BOX_FRAC = 1.0       # share of the volume inside the boxes the rotate adjoint accumulates (set from the mask in main())
EVER_WAVE_FRAC = 1.0      # share of the 256-voxel waves of the advect + Adam kernel that hold an ever-live voxel (likewise)


def ever_wave_fraction(gs):
    """(ever-live voxels / all, 256-voxel waves with an ever-live voxel / all) from the Adam state's mask"""
    ev = gs.adam.ever_mask()
    if ev is None:
        return 1.0, 1.0
    n = gs.d0.numel()
    words = ev.view(torch.int64).cpu().numpy().view(np.uint64)
    nz = words != 0
    pad = (-len(nz)) % 4
    waves = np.concatenate([nz, np.zeros(pad, bool)]).reshape(-1, 4).any(1)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
    # the kernel predicates its streamed accesses per lane (unless NFS_EVER_LANES=0): memory then moves whole cache lines of
    # ever-live voxels -- counted here in runs of 16 voxels (192 bytes of a 12-byte-per-voxel stream: a line and a half)
    if os.environ.get("NFS_EVER_LANES", "1") != "0":
        pad16 = (-n) % 16
        runs = np.concatenate([bits, np.zeros(pad16, np.uint8)]).reshape(-1, 16).any(1)
        return float(bits.mean()), float(runs.mean())
    return float(bits.mean()), float(waves[:(n + 255) // 256].mean())


def live_box_fraction(gs):
    """what nfs_rotate_bwd_coef_live works on, from the stylizer's current mask: (live voxels / all, volume of the per-tile
    boxes / all, tiles skipped / all) -- the host-side restatement of the kernel's per-tile box (14 x 14 x 34 tiles)"""
    D, H, W = gs.d0.shape
    dil = 1 if gs.k > 0 else 0
    words = gs._live_buf.view(torch.int64).cpu().numpy().view(np.uint64)
    m = np.unpackbits(words.view(np.uint8), bitorder="little")[:D * H * W].reshape(D, H, W).astype(bool)
    TZ, TY, TX = 14, 14, 34
    vol = skipped = tiles = 0
    for z0 in range(0, D, TZ):
        for y0 in range(0, H, TY):
            for x0 in range(0, W, TX):
                tiles += 1
                z1, y1, x1 = min(z0 + TZ, D), min(y0 + TY, H), min(x0 + TX, W)
                sub = m[max(z0 - dil, 0):min(z1 + dil, D), max(y0 - dil, 0):min(y1 + dil, H), max(x0 - dil, 0):min(x1 + dil, W)]
                if not sub.any():
                    skipped += 1
                    continue
                ext = []
                for ax, (lo, hi, n) in enumerate(((z0, z1, D), (y0, y1, H), (x0, x1, W))):
                    idx = np.nonzero(sub.any(axis=tuple(a for a in range(3) if a != ax)))[0] + max(lo - dil, 0)
                    ext.append(min(hi - 1, idx.max() + dil) - max(lo, idx.min() - dil) + 1)
                vol += ext[0] * ext[1] * ext[2]
    return float(m.mean()), vol / float(D * H * W), skipped / float(tiles)


def build_problem(G, V, device, rank, world, dense=False):
    from neural_flow_style_amd import engine, vgg
    from neural_flow_style_amd import synthetic as S
    from neural_flow_style_amd import transform as T
    rng = np.random.RandomState(123)
    d0 = S.blob_density(G, rng)
    if dense:
        # the control for the data-dependent part of the step: a density that varies everywhere (no empty space, no
        # plateau), so that no voxel's velocity gradient vanishes and nothing is skipped
        d0 = (0.05 + 0.1 * np.random.RandomState(321).rand(G, G, G)).astype(np.float32)
    vel = S.curl_velocity(G, rng, max_cells=2.0)
    simg = S.style_image(G, G, rng)
    mats = S.uniform_views(V)
    net = vgg.VGG(vgg.synthetic_weights(123, upto="conv5_1"), device)
    loss = engine.RenderStyleLoss(net, STYLE_LAYERS, [1.0] * 5, 1.0, transmit=0.01)
    loss.set_style_image(simg)
    pg = None
    if world > 1:
        import torch.distributed as dist
        pg = dist.group.WORLD
    gs = engine.GridStylizer(loss, torch.tensor(d0, device=device), k=3, target="v", lr=1e-3, process_group=pg)
    gs.var.copy_(torch.tensor(vel))
    rot_local = T.rot_to_device(mats[rank::world], device)
    return gs, rot_local, dict(d0=d0, vel=vel, simg=simg, mats=mats)


# ---- algorithmic work per call ---------------------------------------
def work_of(name, a):
    """-> (kind, amount): kind 'B' bytes or 'F' flops, from the C-ABI call arguments.  Convolutions count the MFMA flops
    their kernels EXECUTE (Winograd F(4x4) / F(5x5) products incl. tile padding, nfs_conv3x3_executed_flops), so that a
    fraction of the MFMA peak stays a fraction; the direct-conv flops of the same calls are reported beside them as
    `algorithmic`."""
    from neural_flow_style_amd import _lib
    ex = _lib.lib().nfs_conv3x3_executed_flops
    if name == "nfs_conv3x3_fwd":
        B, H, W, Ci, Co = a[4:9]
        return "F", ex(B, H, W, Ci, Co, 0)
    if name == "nfs_conv3x3_dgrad":
        B, H, W, Ci, Co = a[5:10]
        return "F", ex(B, H, W, Co, Ci, 0)
    if name == "nfs_conv3x3_fwd_pool":
        B, H, W, Ci, Co = a[5:10]
        return "F", ex(B, H, W, Ci, Co, 1)
    if name == "nfs_conv3x3_dgrad_pool":
        B, H, W, Ci, Co = a[6:11]
        return "F", ex(B, H, W, Co, Ci, 1)
    if name in ("nfs_gram_style_group_fwd", "nfs_gram_group_bwd"):
        import ctypes
        n = a[1]
        L_ = ctypes.cast(a[0], ctypes.POINTER(_lib.GramLayer * n)).contents
        # forward: the symmetric tile pairs t1 <= t2 of 64-channel tiles are executed; gradient: the full product
        if name == "nfs_gram_group_bwd":
            return "F", sum(2.0 * y.B * y.HW * y.C * y.C for y in L_)
        return "F", sum(2.0 * y.B * y.HW * 64 * 64 * ((y.C // 64) * (y.C // 64 + 1) // 2) for y in L_)
    if name == "nfs_gram_fwd":
        B, HW, C = a[2:5]
        return "F", 2.0 * B * HW * C * C
    if name == "nfs_gram_bwd":
        B, HW, C = a[3:6]
        return "F", 2.0 * B * HW * C * C
    if name == "nfs_rotate_render_fwd":
        V, D, H, W = a[5:9]
        # read d once per view + write img; + write of the kept rotated volume when requested
        return "B", (8.0 if a[4] else 4.0) * V * D * H * W + 4.0 * V * H * W
    if name == "nfs_render_bwd":
        V, D, H, W = a[4:8]
        return "B", 8.0 * V * D * H * W + 4.0 * V * H * W
    if name == "nfs_rotate_render_fwd_coef":
        V, D, H, W = a[6:10]
        # read d once per view + write the kept u volume + img / ray sums / the three per-segment planes
        return "B", 8.0 * V * D * H * W + (8.0 + 48.0) * V * H * W
    if name == "nfs_render_ray_coef":
        V, H, W = a[4:7]
        return "B", (4.0 + 48.0 + 32.0) * V * H * W
    if name == "nfs_rotate_bwd_coef":
        V, D, H, W = a[4:8]
        # read u once + the (A, B) planes + write g_d (the adjoint reads no render-adjoint volume any more)
        return "B", 4.0 * V * D * H * W + 32.0 * V * H * W + 4.0 * D * H * W
    if name == "nfs_rotate_bwd_coef_live":
        V, D, H, W = a[4:8]
        # only the boxes the live mask leaves are accumulated: u and the (A, B) planes are read for that fraction of the
        # volume (LIVE_BOX_FRAC, measured from the mask in main()); g_d is written everywhere
        return "B", LIVE_BOX_FRAC * (4.0 * V * D * H * W + 32.0 * V * H * W) + 4.0 * D * H * W
    if name == "nfs_rotate_bwd":
        V, D, H, W, C = a[3:8]
        return "B", 4.0 * V * D * H * W * C + 8.0 * D * H * W * C
    if name == "nfs_rotate_render_bwd":
        V, D, H, W = a[5:9]
        return "B", 8.0 * V * D * H * W + 4.0 * V * H * W
    if name == "nfs_advect_fwd":
        D, H, W, C = a[3:7]
        return "B", (8.0 * C + 12.0) * D * H * W
    if name == "nfs_advect_fwd_live":
        D, H, W = a[4:7]
        return "B", 20.125 * D * H * W
    if name == "nfs_advect_bwd":
        D, H, W, C = a[5:9]
        return "B", ((12.0 if a[3] else 8.0) * C + 24.0) * D * H * W
    if name == "nfs_advect_bwd_adam":
        D, H, W = a[5:8]
        # g_out 4 + gathered d 4 + vel/m/v read 36 + vel/m/v write 36 bytes per voxel
        return "B", 80.0 * D * H * W
    if name == "nfs_advect_bwd_adam_fwd":
        D, H, W = a[6:9]
        # the same + the next iteration's forward sample written (4 bytes; its gathers hit the lines the adjoint just read)
        return "B", 84.0 * D * H * W
    if name == "nfs_advect_bwd_adam_fwd_live":
        D, H, W = a[7:10]
        return "B", 84.125 * D * H * W             # + one mask bit per voxel
    if name == "nfs_advect_bwd_adam_fwd_live_ever":
        D, H, W = a[8:11]
        # only the ever-live part of the volume moves its 84 bytes per voxel (EVER_WAVE_FRAC, measured from the Adam state's
        # mask in main(): 16-voxel runs with an ever-live voxel, or whole 256-voxel waves with NFS_EVER_LANES=0); every
        # wave reads its four mask words twice
        return "B", (84.125 * EVER_WAVE_FRAC + 0.25) * D * H * W
    if name in ("nfs_smooth3d_relu_fwd",):
        D, H, W = a[2:5]
        return "B", 8.0 * D * H * W
    if name in ("nfs_smooth3d_relu_bwd",):
        D, H, W = a[3:6]
        return "B", 8.0 * D * H * W
    if name == "nfs_adam_tf_step":
        return "B", 28.0 * a[4]
    if name == "nfs_transport_step":
        D, H, W, C = a[7:11]
        # read u 12 + gather g 4C + write out 4C (+ read addend 4C) bytes per voxel
        return "B", (12.0 + (12.0 if a[4] else 8.0) * C) * D * H * W
    return None, 0.0


def conv_direct_flops(name, a):
    """direct-convolution flops 2*B*H*W*9*Ci*Co of a conv C-ABI call (what the layer computes), else 0"""
    off = {"nfs_conv3x3_fwd": 4, "nfs_conv3x3_dgrad": 5, "nfs_conv3x3_fwd_pool": 5, "nfs_conv3x3_dgrad_pool": 6}.get(name)
    if off is None:
        return 0.0
    B, H, W, Ci, Co = a[off:off + 5]
    return 2.0 * B * H * W * 9 * Ci * Co


def pmc_traffic(kernel_substr):
    """Mean HBM bytes per launch (read + write) of a kernel (all template instances whose name contains
    ``kernel_substr``) from the committed rocprofv3 PMC passes (profiles/r*_traffic.json, produced by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command; counters cannot be
    read from inside the run).  None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        tab = json.load(open(files[-1]))["kernels"]
        rows = [v for k, v in tab.items() if kernel_substr in k]
        if not rows:
            return None
        if all("read_bytes_mean" in v for v in rows):
            n = sum(v["launches_seen"] for v in rows)
            return sum((v["read_bytes_mean"] + v["write_bytes_mean"]) * v["launches_seen"] for v in rows) / n
        return sum(v["read_bytes_per_launch"] + v["write_bytes_per_launch"] for v in rows) / len(rows)
    except Exception:
        return None


def event_pair_overhead_us(device):
    """What an event pair adds around one call on a busy stream: median elapsed time of the pair around a
    1-element fill kernel queued behind a long kernel, minus nothing (the ~2 us the empty kernel itself takes stay
    in, so the correction below is slightly generous to the overhead, i.e. conservative for the kernels)."""
    from neural_flow_style_amd import _lib, ops
    big = torch.empty(1 << 24, dtype=torch.float32, device=device)
    one = torch.empty(64, dtype=torch.float32, device=device)
    ts = []
    for _ in range(40):
        ops.fill(big, 0.0)
        _lib.PROFILE = {}
        ops.fill(one[:1], 0.0)
        rec, _lib.PROFILE = _lib.PROFILE, None
        ts.append(rec["nfs_fill"][0][:2])
    torch.cuda.synchronize()
    us = sorted(1e3 * a.elapsed_time(b) for a, b in ts)
    return us[len(us) // 2]


def kernel_table(profile, steps):
    """per C-ABI family: time from HIP event pairs around every call (single-stream pass; the pair's own dispatch
    latency, `event_pair_overhead_us`, is INSIDE these figures: they are the conservative side)"""
    rows = []
    for name, recs in sorted(profile.items()):
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
        kind, _ = work_of(name, recs[0][2])
        amount = sum(work_of(name, r[2])[1] for r in recs)
        row = {"kernel": name, "launches_per_step": len(recs) / steps, "ms_per_step": ms / steps,
               "avg_launch_us": 1e3 * ms / len(recs)}
        if kind == "B":
            ach = amount / (ms * 1e-3) / 1e9
            row.update(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS)
        elif kind == "F":
            ach = amount / (ms * 1e-3) / 1e12
            row.update(bound="mfma", achieved=ach, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=ach / MFMA_F32_PEAK_TF)
            direct = sum(conv_direct_flops(name, r[2]) for r in recs)
            if direct:
                row.update(flops="executed (Winograd products incl. tile padding)",
                           algorithmic_tflops=direct / (ms * 1e-3) / 1e12, executed_over_algorithmic=amount / direct)
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def cpu_baseline(data, G, V, n_views, gs=None, device=None, full=False):
    """The oracle (CPU restatement of the reference graph; the TF-1.15 reference cannot run here) timed on the host
    cores on a bounded sample.  The iteration is taken apart into what runs once (prologue: advect + smooth/clamp;
    epilogue: their adjoints + ApplyAdam) and what runs per view (rotate -> render -> VGG -> Gram losses and the adjoint
    down to the smoothed density), each timed by itself, so that the extrapolation to V views is
    prologue + V * per-view + epilogue with the per-view figure measured on >= 2 DIFFERENT views (their spread is
    reported).  The one-view gradients the oracle produces on the way are not thrown away: with ``gs`` (the HIP
    stylizer of the same problem) they are compared with the HIP gradients of the same views -- oracle parity AT THE
    HEADLINE SIZE (returned under "full_size_parity").  ``full``: BASELINE.md's own protocol (1 warm-up + median of 3
    whole 8-view iterations; ~20 min of CPU at 200^3)."""
    from oracle import nfs_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    O.FAST_WARP = True   # multi-threaded grid_sample for the 8-tap warps (identical numerics, tested)
    w = O.synthetic_vgg19_weights(123, upto="conv5_1")
    sfe = O.style_target_features(torch.tensor(data["simg"])[None], w, STYLE_LAYERS, upto="conv5_1")
    cfg = dict(k=3, transmit=0.01, style_layer=STYLE_LAYERS, w_style_layer=[1.0] * 5, w_style=1.0, upto="conv5_1")
    d0 = torch.tensor(data["d0"])[None, ..., None]
    clock = time.perf_counter

    def iteration(view_ids, data=data, d0=d0, sfe=sfe, want_grads=False):
        """one oracle iteration over the views ``view_ids`` -> (seconds by phase, per-view losses, per-view dL/dvel)"""
        vel = torch.tensor(data["vel"])[None].requires_grad_()
        t0 = clock()
        d_s = O.smooth3d_relu(O.advect(d0, vel), cfg["k"])
        t_pro = clock() - t0
        leaf = d_s.detach().requires_grad_()
        g_ds, t_views, losses, g_views = torch.zeros_like(leaf), [], [], []
        for v in view_ids:
            t0 = clock()
            rot = torch.tensor(np.asarray(data["mats"][v:v + 1], np.float32))
            l = O.grid_view_loss(leaf, rot, cfg, w, sfe)
            (gv,) = torch.autograd.grad(l, leaf)
            t_views.append(clock() - t0)
            losses.append(float(l.detach()))
            g_views.append(gv)
            g_ds += gv
        t0 = clock()
        (g,) = torch.autograd.grad(d_s, vel, g_ds, retain_graph=want_grads)
        opt = O.TFAdam(); opt.step(vel.detach().clone(), g, 1e-3)
        t_epi = clock() - t0
        g_vel = []
        if want_grads:                      # (untimed) the velocity gradient of every view by itself, for the parity check
            for gv in g_views:
                g_vel.append(torch.autograd.grad(d_s, vel, gv, retain_graph=True)[0][0])
        return dict(prologue=t_pro, views=t_views, epilogue=t_epi), losses, g_vel

    if full:
        iteration(list(range(V)))
        ts = []
        for _ in range(3):
            t0 = clock(); iteration(list(range(V))); ts.append(clock() - t0)
        ts.sort()
        return {"value": 1.0 / ts[1], "unit": "iters/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle (PyTorch-CPU restatement, f32): BASELINE.md's protocol -- 1 warm-up + median of 3 whole "
                          "%d-view iterations at %d^3 (%.1f / %.1f / %.1f s)" % (V, G, ts[0], ts[1], ts[2]),
                "seconds_measured": sum(ts)}, None

    # the sample: views V-1 and 0..n_views-1.  No separate warm-up pass: the first timed view doubles as one -- its time
    # against the later views' is reported (per_view_spread; measured 31.94 / 31.93 s on a 256-core host, and a 32^3
    # warm-up pass cost 29 s of its own in thread-pool and primitive set-up without changing either)
    sample = [V - 1] + list(range(n_views))
    warm = 0.0
    t0 = clock()
    sec, losses, g_vel = iteration(sample, want_grads=gs is not None)
    dt = clock() - t0
    per_view = float(np.mean(sec["views"]))
    est_iter = sec["prologue"] + V * per_view + sec["epilogue"]
    out = {"value": 1.0 / est_iter, "unit": "iters/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "oracle (PyTorch-CPU restatement, f32) at %d^3: ONE "
                     "iteration over %d of the %d views (views %s) timed by phase: prologue advect+smooth %.2f s, per view "
                     "(rotate, render, VGG conv1_1..conv5_1, Gram losses, adjoint) %s s, epilogue (adjoints of smooth and "
                     "advect, ApplyAdam) %.2f s; one %d-view iteration = prologue + %d x mean view + epilogue = %.1f s.  "
                     "BASELINE.md asks for 1 warm-up + median of 3 whole iterations (~20 min of CPU here): "
                     "--cpu-baseline-full runs that" % (G, len(sample), V, sample, sec["prologue"],
                                                        "/".join("%.1f" % t for t in sec["views"]), sec["epilogue"], V, V,
                                                        est_iter),
           "seconds_measured": dt, "seconds_warmup": warm, "seconds_prologue": sec["prologue"],
           "seconds_per_view": sec["views"], "seconds_epilogue": sec["epilogue"],
           "per_view_spread": (max(sec["views"]) - min(sec["views"])) / per_view}
    parity = None
    if gs is not None:
        from neural_flow_style_amd import ops
        from neural_flow_style_amd import transform as T

        def against_oracle():
            rows = []
            gs.var.copy_(torch.tensor(data["vel"]))    # the timed steps moved the variable: back to the oracle's input
            for v, lo, go in zip(sample, losses, g_vel):
                lh, gh = gs.gradient(T.rot_to_device(data["mats"][v:v + 1], device))
                gh = gh.double().cpu()
                rows.append({"view": v, "grad_rel_l2": float((gh - go.double()).norm() / go.double().norm()),
                             "loss_rel": abs(float(lh.double().sum()) - lo) / abs(lo)})
            return {"grad_rel_l2": max(r["grad_rel_l2"] for r in rows), "loss_rel": max(r["loss_rel"] for r in rows),
                    "case": "%d^3, 1 view, conv1_1..conv5_1: dL/d velocity field [%d,%d,%d,3] of the HIP path vs the CPU "
                            "oracle, views %s of the benchmark's lattice (worst view reported)" % (G, G, G, G, sample),
                    "per_view": rows, "tolerance": 1e-3}
        mode0 = ops.gemm_mode(None)
        parity = against_oracle()
        parity["gemm_mode"] = mode0
        if mode0 == 1:
            # the same comparison with the Winograd GEMMs on the f32-input MFMA: the split-limb arithmetic is the default
            # BECAUSE it is no further from the float64-checked oracle than this
            ops.gemm_mode(0)
            try:
                parity["f32_mfma"] = {k: v for k, v in against_oracle().items() if k in ("grad_rel_l2", "loss_rel", "per_view")}
            finally:
                ops.gemm_mode(mode0)
    return out, parity


def small_parity(device):
    """gradient relative L2 of the HIP path vs the oracle on a 24^3, 3-view, 5-layer case"""
    from oracle import nfs_oracle as O
    from neural_flow_style_amd import engine, vgg
    from neural_flow_style_amd import synthetic as S
    from neural_flow_style_amd import transform as T
    G, V = 24, 3
    rng = np.random.RandomState(123)
    d0 = S.blob_density(G, rng)
    vel0 = (rng.randn(G, G, G, 3) * 0.3 / (G - 1)).astype(np.float32)
    simg = S.style_image(G, G, rng)
    mats = S.uniform_views(V)
    w = O.synthetic_vgg19_weights(123, upto="conv5_1")
    sfe = O.style_target_features(torch.tensor(simg)[None], w, STYLE_LAYERS, upto="conv5_1")
    cfg = dict(k=3, transmit=0.05, style_layer=STYLE_LAYERS, w_style_layer=[1.0] * 5, w_style=1.0, upto="conv5_1")
    v = torch.tensor(vel0)[None].requires_grad_()
    total, _, _ = O.grid_forward(torch.tensor(d0)[None, ..., None], v, torch.tensor(np.asarray(mats, np.float32)),
                                 cfg, w, sfe)
    (go,) = torch.autograd.grad(total, v)
    net = vgg.VGG(vgg.synthetic_weights(123, upto="conv5_1"), device)
    loss = engine.RenderStyleLoss(net, STYLE_LAYERS, [1.0] * 5, 1.0, transmit=0.05)
    loss.set_style_image(simg)
    gs = engine.GridStylizer(loss, torch.tensor(d0, device=device), k=3, target="v")
    gs.var.copy_(torch.tensor(vel0))
    _, gh = gs.gradient(T.rot_to_device(mats, device))
    r = float((gh.double().cpu() - go[0].double()).norm() / go[0].double().norm())
    return {"grad_rel_l2": r, "case": "24^3 grid, 3 views, conv1_1..conv5_1, vs CPU oracle", "tolerance": 1e-3}


def frame_data(G, t, base):
    """frame t of the synthetic sequence: the blob density and the curl-noise
    simulation velocity of the single-frame problem, shifted by 3 cells per frame along W (periodic roll)"""
    return np.roll(base["d0"], 3 * t, axis=2), np.roll(base["vel"], 3 * t, axis=2)


def build_sequence(args, device, rank, world, n_frames, base, pg):
    """BASELINE configs[3]: ``styler_grid.Styler`` on ``n_frames`` frames of G^3, frames sharded over the ranks"""
    from neural_flow_style_amd.config import get_config
    from neural_flow_style_amd.styler_grid import Styler
    from neural_flow_style_amd import parallel
    G, V = args.grid, args.views
    cfg, _ = get_config([])
    for k, v in dict(network="vgg_19.ckpt", data_dir="/nonexistent", synthetic_weights=True, resolution=[G, G, G], k=3,
                     num_frames=n_frames, batch_size=1, frames_per_opt=1, window_sigma=args.window_sigma, interp=1,
                     lr=1e-3, iter=1, octave_n=1, style_layer=STYLE_LAYERS, w_style_layer=[1.0] * 5, w_style=1.0,
                     w_content=0, transmit=0.01, rotate=True, n_views=V, v_batch=1, sample_type="uniform",
                     resize_scale=1.0, style_target=base["simg"], grid_variable="v").items():
        setattr(cfg, k, v)
    cfg.rng = np.random.RandomState(123)
    st = Styler(cfg)
    st.pg = pg
    st.rot_mat_ = [np.asarray(m, np.float32) for m in base["mats"]]     # the benchmark's 8-view lattice
    st.load_img([G, G])
    # only the frames this rank touches go to its HBM: its own densities + the velocities its filter window crosses
    # (a rank beyond the number of frames owns nothing and only takes part in the collectives)
    mine, vel_frames = st.frames_needed(rank, world)
    dd, uu, vi = {}, {}, {}
    for t in sorted(mine | vel_frames):
        d_t, u_t = frame_data(G, t, base)
        if t in vel_frames:
            uu[t] = u_t
        if t in mine:
            dd[t] = d_t
            vi[t] = base["vel"]            # non-zero initial stylisation velocity (as the single-frame bench)
    mine = sorted(mine)
    st.prepare({"d": dd, "v": uu, "v_init": vi}, frames_on_device=mine)
    return st


def time_steps(step, barrier, warmup, steps, device, world):
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt, last


def sustained(step, barrier, units, device, world, seconds=2.0, window=20):
    """>= ``seconds`` of back-to-back stepping in ``window``-step windows (each bracketed by barrier + sync, max over
    ranks): the clock the chip settles at under load, beside the short contract timing"""
    rates = []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < seconds or len(rates) < 3:
        dt, _ = time_steps(step, barrier, 0, window, device, world)
        rates.append(units * window / dt)
        if len(rates) >= 200:
            break
    rates.sort()
    return {"median": rates[len(rates) // 2], "min": rates[0], "max": rates[-1], "windows": len(rates),
            "steps_per_window": window, "seconds": time.perf_counter() - t_all}


def other_configs(device, base):
    """short, driver-visible side numbers for the other BASELINE configurations (each a few seconds)"""
    from neural_flow_style_amd import _lib, engine, ops, vgg
    from neural_flow_style_amd import synthetic as S
    from neural_flow_style_amd import transform as T
    out = []

    def ev_time(f, reps):
        f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # configs[1]: smokegun 100^3, one view
    try:
        G = 100
        rng = np.random.RandomState(123)
        d0 = S.blob_density(G, rng); vel = S.curl_velocity(G, rng, max_cells=2.0); simg = S.style_image(G, G, rng)
        net = vgg.VGG(vgg.synthetic_weights(123, upto="conv5_1"), device)
        loss = engine.RenderStyleLoss(net, STYLE_LAYERS, [1.0] * 5, 1.0, transmit=0.01)
        loss.set_style_image(simg)
        gs = engine.GridStylizer(loss, torch.tensor(d0, device=device), k=3, target="v", lr=1e-3)
        gs.var.copy_(torch.tensor(vel))
        rot = T.rot_to_device(S.uniform_views(1), device)
        for _ in range(5):
            gs.step(rot, loss_view=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            gs.step(rot, loss_view=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        out.append({"config": "configs[1] smokegun 100^3 single-frame, 1 view, conv1_1..conv5_1", "value": 1.0 / dt,
                    "unit": "iters/s", "ms_per_step": 1e3 * dt, "hipgraph": bool(gs.use_graph)})
    except Exception as e:  # pragma: no cover
        out.append({"config": "configs[1]", "error": repr(e)})

    # configs[0]: dambreak2d 128x128 colour field, conv3_1, style mask + TV: one step of the loss chain
    try:
        H = W = 128
        rng = np.random.RandomState(7)
        net = vgg.VGG(vgg.synthetic_weights(123, upto="conv3_1"), device)
        il = engine.ImageStyleLoss(net, ["conv3_1"], [1.0], 1.0, w_tv=0.01, style_mask=True)
        il.set_style_image(S.style_image(H, W, rng))
        d = torch.rand(1, H, W, 3, device=device)
        dg = (torch.rand(1, H, W, 1, device=device) > 0.4).float()
        ms = ev_time(lambda: il.loss_and_grad(d, dg), 50)
        out.append({"config": "configs[0] dambreak2d 128x128 colour field, conv3_1 (style mask + TV): loss + gradient of "
                              "the image (VGG fwd/dgrad + masked Gram), per Adam iteration", "value": 1e3 / ms,
                    "unit": "iters/s", "ms_per_step": ms, "algorithmic_gflop": 2 * 3.68,
                    "frac_mfma_f32": 2 * 3.68e9 / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF})
    except Exception as e:  # pragma: no cover
        out.append({"config": "configs[0]", "error": repr(e)})

    # configs[0] end to end: Styler(config).run of the 2-D colour stylizer (particle colours -> splat -> masked style + TV ->
    # adjoint -> TF-Adam), 50 iterations at 128 x 128 -- on SURVEY 8(d)'s workload (16 384 particles: two per cell and
    # dimension on the dam-break lattice, scene/dambreak2d.py:96-103) and, beside it, on the 1 344 particles rounds 3-5
    # quoted (tools/dambreak_bench.py)
    for n_side, n_keep in ((280, 16384), (80, None)):
        try:
            from neural_flow_style_amd.config import get_config as _gc
            from neural_flow_style_amd.styler_2p import Styler as Styler2
            rng = np.random.RandomState(7)
            pp2 = S.dambreak_particles(n_side, rng, n_keep)
            rr2 = rng.uniform(900, 1100, (pp2.shape[0], 1)).astype(np.float32)
            c2, _ = _gc([])
            for k_, v_ in dict(network="vgg_19.ckpt", data_dir="/nonexistent", synthetic_weights=True, resolution=[128, 128],
                               domain=[3.2, 3.2], radius=0.0125, nsize=2, support=4, rest_density=1000, clip=False,
                               target_field="c", num_frames=1, batch_size=1, frames_per_opt=200, window_sigma=3, lr=0.01,
                               iter=50, octave_n=1, octave_scale=1.7, style_layer=["conv3_1"], w_style_layer=[1.0], w_style=1.0,
                               w_content=0, style_mask=True, w_tv=0.01, style_target=S.style_image(128, 128, rng),
                               resize_scale=1.0).items():
                setattr(c2, k_, v_)
            c2.rng = np.random.RandomState(c2.seed)
            import contextlib, io as _io
            with contextlib.redirect_stdout(_io.StringIO()):           # (the stylizer prints its octave sizes)
                st2 = Styler2(c2)
                st2.load_img([128, 128])
                st2.run({"p": [pp2], "r": [rr2]})                       # warm-up (weight packing, workspaces)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res2 = st2.run({"p": [pp2], "r": [rr2]})
                torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / c2.iter
            out.append({"config": "configs[0] dambreak2d 128x128 END TO END: Styler(config).run of the 2-D colour stylizer, %d "
                                  "particles%s, conv3_1, style mask + TV, 50 Adam iterations"
                                  % (pp2.shape[0], " (SURVEY 8(d)'s workload)" if n_keep else " (the sparse lattice of rounds 3-5)"),
                        "particles": int(pp2.shape[0]),
                        "value": 1.0 / dt2, "unit": "iters/s", "ms_per_step": 1e3 * dt2,
                        "loss_first_last": [float(res2["l"][0][0]), float(res2["l"][0][-1])],
                        "loss_chain_mode": getattr(st2._graph_loss, "mode", None) if st2._graph_loss else "eager"})
            del st2
        except Exception as e:  # pragma: no cover
            out.append({"config": "configs[0] end to end (%d-lattice)" % n_side, "error": repr(e)})

    # configs[4]: chocolate-scale splat, 5e5 particles -> 200^3 (grid-cell order as Styler.run processes them)
    try:
        N, G = 500000, 200
        rng = np.random.RandomState(0)
        p = torch.tensor(S.blob_particles(N, rng), device=device)
        p = p[T.grid_order(p, [G, G, G])].contiguous()
        scfg = ops.make_splat_cfg(3, [G, G, G], [G, G, G], 0.5, 4, 1000.0, 1, False, 0)
        g = torch.randn(G, G, G, 1, device=device)
        tf_ = ev_time(lambda: ops.p2g_fwd(p, scfg), 20)
        tb_ = ev_time(lambda: ops.p2g_bwd(p, scfg, g, need_p=True), 20)
        bf = N * 12.0 + 8.0 * G ** 3          # positions + zero fill + the grid written once
        bb = N * 24.0 + 4.0 * G ** 3
        out.append({"config": "configs[4] SPH splat p2g, 5e5 particles -> 200^3 (27 cells each), in grid order (8-cell bricks, as Styler.run sorts them)",
                    "fwd_ms": tf_, "bwd_ms": tb_, "fwd_frac_hbm": bf / (tf_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "bwd_frac_hbm": bb / (tb_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "fwd_cell_updates_per_s": N * 27 / (tf_ * 1e-3)})
    except Exception as e:  # pragma: no cover
        out.append({"config": "configs[4]", "error": repr(e)})

    # configs[4] end to end: one iteration of the particle stylizer at the chocolate scale -- 5e5 particles -> 200^3,
    # 'p' field (the variable is a displacement per particle), liquid render (transmit 0.2), rotate False, k 3, VGG-19
    # conv1_1..conv4_1, TF-Adam (test_chocolate.py:148-252 with the VGG network) -- and where its time goes
    try:
        from neural_flow_style_amd.config import get_config
        from neural_flow_style_amd.styler_3p import Styler
        N, G = 500000, 200
        rng = np.random.RandomState(0)
        cfg, _ = get_config([])
        for k_, v_ in dict(network="vgg_19.ckpt", data_dir="/nonexistent", synthetic_weights=True, resolution=[G, G, G],
                           domain=[12.8] * 3, radius=0.025, support=4, nsize=1, rest_density=1000, k=3, clip=False,
                           target_field="p", num_frames=1, batch_size=1, frames_per_opt=120, window_sigma=9, interp=1,
                           lr=0.002, iter=1, octave_n=1, style_layer=STYLE_LAYERS[:4], w_style_layer=[1.0] * 4,
                           w_style=1.0, w_content=0, transmit=0.2, render_liquid=True, rotate=False, resize_scale=1.0,
                           num_kernels=1, kernel_scale=2, style_target=S.style_image(G, G, rng)).items():
            setattr(cfg, k_, v_)
        cfg.rng = np.random.RandomState(123)
        stp = Styler(cfg)
        stp.load_img([G, G])
        stp.loss.set_style_image(stp._style_feature(stp.style_img, [G, G]))
        pp = torch.tensor(S.blob_particles(N, rng), device=device)
        pp = pp[T.grid_order(pp, [G, G, G])].contiguous()                    # as Styler.run orders them
        var = torch.zeros(N, 3, device=device)
        adam = engine.TFAdamState()

        def p_iter():
            losses, g = stp._value_and_grad(pp, None, var, [G, G, G], stp._identity)
            adam.step(var, g.contiguous(), cfg.lr)
            return losses
        for _ in range(4):            # (lazy state, then the stylizer's measured eager-vs-hipGraph choice for the loss chain)
            p_iter()
        ms_it = ev_time(p_iter, 10)
        # stage split (each stage alone, same operands)
        v_ = var.detach().clone().requires_grad_(True)
        ms_f = ev_time(lambda: stp._field(pp, None, v_, [G, G, G]), 10)
        _, d_out, _ = stp._field(pp, None, v_, [G, G, G])
        d3 = d_out.detach().reshape(G, G, G).contiguous()
        g_d = torch.zeros_like(d3)
        ms_l = ev_time(lambda: stp.loss.loss_and_grad(d3, stp._identity, g_d), 10)
        gd5 = g_d.reshape(d_out.shape)

        def back():
            _, d_o, _ = stp._field(pp, None, v_, [G, G, G])
            v_.grad = None
            d_o.backward(gd5)
        ms_fb = ev_time(back, 10)
        out.append({"config": "configs[4] chocolate-scale particle stylizer END TO END: 5e5 particles -> 200^3, 'p' field, "
                              "liquid render (transmit 0.2), one view (rotate False), VGG-19 conv1_1..conv4_1, TF-Adam "
                              "on the displacements: one iteration",
                    "value": 1e3 / ms_it, "unit": "iters/s", "ms_per_step": ms_it,
                    "stages_ms": {"field forward (splat p2g + smooth/clamp, through autograd)": ms_f,
                                  "render + VGG + Gram losses + adjoint down to the grid": ms_l,
                                  "field forward + backward (smooth adjoint + splat gather adjoint)": ms_fb},
                    "loss_chain_mode": getattr(stp._graph_loss, "mode", None) if stp._graph_loss else "eager",
                    "note": "stages timed alone on the same operands (their sum exceeds the iteration by the forward "
                            "pass counted twice); loss_chain_mode: what the stylizer measured to be faster on this "
                            "box for its one-view loss chain, eager submission or hipGraph replay"})
        del stp
    except Exception as e:  # pragma: no cover
        out.append({"config": "configs[4] end to end", "error": repr(e)})

    # configs[2] at fewer local views: the per-rank step of the view-sharded (strong-scaling) run at 4 / 2 / 1 views per
    # rank, measured on this one GPU -- what a rank of an 8-view job on 2 / 4 / 8 GPUs executes per iteration before the
    # all-reduce (DESIGN.md section 7 derives the strong-scaling bound from these)
    try:
        G2 = int(base["d0"].shape[0])
        net2 = vgg.VGG(vgg.synthetic_weights(123, upto="conv5_1"), device)
        rows = {}
        for nv in (8, 4, 2, 1):
            loss2 = engine.RenderStyleLoss(net2, STYLE_LAYERS, [1.0] * 5, 1.0, transmit=0.01)
            loss2.set_style_image(base["simg"])
            gs2 = engine.GridStylizer(loss2, torch.tensor(base["d0"], device=device), k=3, target="v", lr=1e-3)
            gs2.var.copy_(torch.tensor(base["vel"]))
            rot2 = T.rot_to_device(base["mats"][:nv], device)
            for _ in range(4):
                gs2.step(rot2, loss_view=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                gs2.step(rot2, loss_view=True)
            torch.cuda.synchronize()
            rows[str(nv)] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / 30, "hipgraph": bool(gs2.use_graph)}
            del gs2, loss2
        t8 = rows["8"]["ms_per_step"]
        # the same with the field work sharded over D-slabs (engine.GridStylizer._slab_setup): a rank then runs the loss
        # chain of its views + the field ops on 1/world of the planes (+ halo) + the packing gather; measured piecewise on
        # this GPU (rank 1's slab), the two collectives (reduce-scatter + all-gather = one all-reduce's traffic) excluded
        loss2 = engine.RenderStyleLoss(net2, STYLE_LAYERS, [1.0] * 5, 1.0, transmit=0.01)
        loss2.set_style_image(base["simg"])
        gs2 = engine.GridStylizer(loss2, torch.tensor(base["d0"], device=device), k=3, target="v", lr=1e-3)
        gs2.var.copy_(torch.tensor(base["vel"]))
        d_s2 = gs2.forward_field()
        slab_rows = {}
        for world_ in (2, 4, 8):
            nv = len(base["mats"]) // world_
            rot2 = T.rot_to_device(base["mats"][:nv], device)
            t_loss = ev_time(lambda: gs2._loss_gradient(d_s2, rot2), 20)
            cs = G2 // world_
            z0 = cs
            lo, hi = z0 - 1, z0 + cs + 1
            vel_s = gs2.var[lo:hi].clone(); m_s = torch.zeros_like(vel_s); v_s = torch.zeros_like(vel_s)
            gpad = torch.zeros(G2 + 5, G2, G2, device=device)
            idx = torch.arange(world_ * (cs + 5), device=device) % (G2 + 5)
            pack = torch.empty(world_ * (cs + 5), G2, G2, device=device)
            g_chunk = torch.randn(cs + 4, G2, G2, device=device)

            d_adv = ops.advect_fwd_slab(gs2.d0, vel_s, lo)

            def field_slab():
                # (the forward advect of the next iteration is written by the Adam kernel: engine.GridStylizer._adv_target)
                ops.smooth3d_relu_fwd(d_adv, 3.0)
                torch.index_select(gpad, 0, idx, out=pack)
                g_adv = ops.smooth3d_relu_bwd(d_s2[z0 - 2:z0 + cs + 2], g_chunk, 3.0)
                ops.advect_bwd_adam_slab(gs2.d0, vel_s, g_adv[1:1 + (hi - lo)], m_s, v_s, lo, 1e-3, adv_next=d_adv)
            t_field = ev_time(field_slab, 20)
            slab_rows[str(world_)] = {"local_views": nv, "loss_chain_ms": t_loss, "field_slab_ms": t_field,
                                      "ms_per_step": t_loss + t_field}
        out.append({"config": "configs[2] per-rank step at 8 / 4 / 2 / 1 local views (one GPU; the compute side of view-"
                              "sharded strong scaling on 1 / 2 / 4 / 8 GPUs, collective not included)",
                    "local_views": rows,
                    "compute_bound_speedup": {"2_gpus": t8 / rows["4"]["ms_per_step"],
                                              "4_gpus": t8 / rows["2"]["ms_per_step"],
                                              "8_gpus": t8 / rows["1"]["ms_per_step"]},
                    "slab_sharded_field_work": slab_rows,
                    "compute_bound_speedup_slab": {"%s_gpus" % w_: t8 / r_["ms_per_step"] for w_, r_ in slab_rows.items()},
                    "note": "local_views: the whole step of one rank with the field work replicated (NFS_SLAB_SHARD=0); "
                            "slab_sharded_field_work: loss chain of the rank's views + the field ops on its D-slab, measured "
                            "piecewise (what a rank executes with the default sharding); the 32 MB reduce-scatter + "
                            "all-gather over xGMI come on top (SURVEY 8(e): 0.05-0.37 ms)"})
        del gs2, loss2
    except Exception as e:  # pragma: no cover
        out.append({"config": "configs[2] per-rank step", "error": repr(e)})

    # configs[3]: the sequence loop on 4 frames of 200^3 (one GPU): frame-iterations/s incl. temporal alignment, and
    # the transport step kernel against its HBM roofline
    try:
        G3 = int(base["d0"].shape[0])
        ns = argparse.Namespace(grid=G3, views=len(base["mats"]), window_sigma=2.0)
        st = build_sequence(ns, device, 0, 1, 4, base, None)
        for _ in range(2):
            st.iterate()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            st.iterate()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        gq = torch.randn(G3, G3, G3, 3, device=device); uq = torch.tensor(base["vel"], device=device)
        oq = torch.empty_like(gq)
        ms = ev_time(lambda: ops.transport_step(gq, uq, 1.0, 0.5, gq, 0.5, out=oq), 20)
        bytes_ = (12.0 + 36.0) * G3 ** 3
        out.append({"config": "configs[3] smokegun %d^3 sequence, 4 frames x %d views, window_sigma 2 (one GPU)"
                              % (G3, ns.views),
                    "value": 4 / dt, "unit": "frame-iters/s", "ms_per_iteration": 1e3 * dt,
                    "transport_step_ms": ms, "transport_step_frac_hbm": bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        del st
    except Exception as e:  # pragma: no cover
        out.append({"config": "configs[3]", "error": repr(e)})

    # the reference driver's OWN configuration (test_smokegun.py:128-148): 200 x 300 x 200 density field, one unrotated
    # view, render resized by 1.5 to 300 x 450, Inception-v1 style layers conv2d2 / mixed3b / mixed4b
    try:
        from neural_flow_style_amd import inception
        rng = np.random.RandomState(3)
        net = inception.InceptionV1(inception.synthetic_weights(123, upto="mixed4b"), device)
        layers = ["conv2d2", "mixed3b", "mixed4b"]
        il = engine.RenderStyleLoss(net, layers, [1.0] * 3, 1.0, transmit=0.01, resize_scale=1.5, rotate=False)
        il.set_style_image(S.style_image(300, 450, rng))
        dd = torch.tensor(S.blob_density(200, rng), device=device)
        dd = torch.nn.functional.pad(dd, (0, 0, 50, 50)).contiguous()                    # [200,300,200]
        g_d = torch.zeros_like(dd)

        def inc_step():
            g_d.zero_()
            return il.loss_and_grad(dd, None, g_d)
        ms = ev_time(inc_step, 50)
        _lib.PROFILE = {}
        for _ in range(5):
            inc_step()
        torch.cuda.synchronize()
        prof, _lib.PROFILE = _lib.PROFILE, None
        shapes = prof.pop("shapes:nfs_conv2d_group", [])

        def fl(B, H, W, Cin, Cout, kh, kw, stride):
            Ho, Wo = ops.same_out(H, kh, stride)[0], ops.same_out(W, kw, stride)[0]
            K = (kh * kw + 3) // 4 * 16 if Cin <= 4 else kh * kw * ((Cin + 15) // 16 * 16)
            return 2.0 * B * Ho * Wo * ((Cout + 63) // 64 * 64) * K
        ex = (sum(fl(*r[2][10:18]) for r in prof.get("nfs_conv2d_fwd", []))
              + sum(fl(*(sh + (1,))) for call in shapes for sh in call)) / 5
        tc = sum(r[0].elapsed_time(r[1]) for k in ("nfs_conv2d_fwd", "nfs_conv2d_group") for r in prof.get(k, [])) / 5
        out.append({"config": "the reference driver's own test_smokegun.py configuration: 200x300x200 density field, one "
                              "unrotated view, render resized x1.5 to 300x450, Inception-v1 (tensorflow_inception_graph) "
                              "style layers conv2d2 / mixed3b / mixed4b: loss + gradient of the density field",
                    "value": 1e3 / ms, "unit": "iters/s", "ms_per_step": ms,
                    "inception_conv": {"ms_per_step": tc, "executed_tflops": ex / tc / 1e9,
                                       "frac_mfma_f32": ex / tc / 1e9 / MFMA_F32_PEAK_TF,
                                       "launches_per_step": (len(prof.get("nfs_conv2d_fwd", []))
                                                             + 2 * len(prof.get("nfs_conv2d_group", []))) / 5,
                                       "note": "nfs_conv2d_fwd + nfs_conv2d_group calls (implicit-GEMM SAME convolutions on "
                                               "the f32 MFMA, the branches of a module as grouped launches), executed "
                                               "flops incl. the padding of K to 16 and N to 64, HIP event pairs"}})
        del il, net
    except Exception as e:  # pragma: no cover
        out.append({"config": "reference test_smokegun.py configuration (Inception-v1)", "error": repr(e)})

    # BASELINE configs[2] through the reference's OWN loop: the driver's main() with its override block and the command line
    # that selects VGG-19 and 8 rotated views -- particles ('d' field: a density offset per particle, two splat kernels),
    # 200 x 300 x 200 grid, render resized x1.5 to 300 x 450, conv1_1..conv5_1, and the reference's SEQUENTIAL view mode:
    # one TF-Adam step per view batch on the same variable, iterates averaged (styler_3p.py:326-352) -- inherently serial
    # over the views, so one iteration = 8 loss evaluations + 8 Adam steps.  Wall time of Styler.run per iteration as a user
    # of the driver sees it: two run lengths differenced (set-up and the final inference cancel).
    try:
        import contextlib, importlib, io as _io, tempfile
        from config import get_config as _gcfg
        drv = importlib.import_module("test_smokegun")
        argv0, runs = sys.argv, []
        # (the first run builds the lazy state: discarded; each length twice, the faster kept: a run is ~0.7 s of set-up --
        # weights, demo data, style target -- whose jitter, differenced against only 8 iterations, moved this figure between
        # 12 and 18 ms from box to box)
        for it in (4, 4, 28, 4, 28):
            sys.argv = ["test_smokegun.py", "--num_frames", "1", "--target_frame", "70", "--network", "vgg_19.ckpt",
                        "--rotate", "true", "--n_views", "8", "--w_style", "1", "--synthetic_weights", "true",
                        "--iter", str(it)]                  # (the driver's main() looks at sys.argv for the flags given)
            try:
                c3, _ = _gcfg()
                tmp = tempfile.mkdtemp()
                c3.log_dir, c3.data_dir = os.path.join(tmp, "log"), os.path.join(tmp, "nodata")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(_io.StringIO()):
                    res = drv.main(c3)
                torch.cuda.synchronize()
                runs.append((it, time.perf_counter() - t0, sum(len(l) for l in res["l"])))
            finally:
                sys.argv = argv0
        i0, i1 = 4, 28
        ta = min(t_ for it_, t_, _ in runs[1:] if it_ == i0)
        tb = min(t_ for it_, t_, _ in runs[1:] if it_ == i1)
        ms_iter = 1e3 * (tb - ta) / (i1 - i0)
        out.append({"config": "BASELINE configs[2] through the reference's own driver and loop: test_smokegun.py --network "
                              "vgg_19.ckpt --rotate true --n_views 8 (demo data: 50k particles, 'd' field, 200x300x200 grid, "
                              "300x450 render, conv1_1..conv5_1), views_mode sequential (styler_3p.py:326-352: one Adam "
                              "step per view, iterates averaged): Styler.run wall time per iteration",
                    "value": 1e3 / ms_iter, "unit": "iters/s", "ms_per_step": ms_iter, "loss_evaluations_per_iteration": 8,
                    "ms_per_loss_evaluation": ms_iter / 8.0,
                    "note": "two run lengths (%d and %d iterations: %.2f s, %.2f s) differenced; the sequential mode is "
                            "serial over the views by construction (each step starts from the previous view's update) -- "
                            "replicas only on several GPUs (SURVEY 8(e)); the headline's views=sum batch of 8 is the form "
                            "that shards" % (i0, i1, ta, tb)})
    except Exception as e:  # pragma: no cover
        out.append({"config": "configs[2] through the reference driver (sequential views)", "error": repr(e)})

    # the operators either side of the path, each at the size its caller uses -- time per call and the
    # fraction of 8 TB/s its ALGORITHMIC bytes make (compulsory reads + writes; gathers counted once per element read)
    try:
        from neural_flow_style_amd import util as U
        G, N = 200, 1000000
        rng = np.random.RandomState(5)
        gen = torch.Generator(device=device).manual_seed(5)
        ops_out = []

        def add(name, ref, f, nbytes, reps=20):
            ms = ev_time(f, reps)
            ops_out.append({"op": name, "reference": ref, "ms": ms, "algorithmic_mb": nbytes / 1e6,
                            "gb_s": nbytes / ms / 1e6, "frac_hbm": nbytes / ms / 1e6 / HBM_PEAK_GBS})

        grid1 = torch.rand(G, G, G, 1, device=device, generator=gen)
        grid3 = torch.rand(G, G, G, 3, device=device, generator=gen)
        # grid -> particle sampling of the SimG2P resampler (test_smokegun_resim.py:36-47, 96): 1e6 particles in 200^3,
        # seeded the way the resampler seeds them -- cell by cell in grid order, jittered inside the cell (here the
        # central 100^3 cells) -- and, as the worst case, the same number at uniformly random positions in random order
        ax = torch.arange(50, 150, device=device, dtype=torch.float32)
        cells = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
        pts = ((cells + torch.rand(N, 3, device=device, generator=gen)) / G).contiguous()
        pts_rand = (torch.rand(N, 3, device=device, generator=gen) * 0.9 + 0.05).contiguous()
        bytes_d = 4.0 * 104 ** 3 + N * (12 + 4)          # the cells the particles touch + positions in + values out
        bytes_v = 12.0 * 102 ** 3 + N * (12 + 12)
        add("g2p cubic, density (64 taps), particles in cell order", "transform.py:778-1108",
            lambda: ops.g2p_fwd(grid1, pts, cubic=True), bytes_d)
        add("g2p linear, velocity (8 taps x 3), particles in cell order", "transform.py:1110-1231",
            lambda: ops.g2p_fwd(grid3, pts, cubic=False), bytes_v)
        add("g2p cubic, density, random positions in random order", "transform.py:778-1108",
            lambda: ops.g2p_fwd(grid1, pts_rand, cubic=True), 4.0 * G ** 3 + N * (12 + 4))
        vel = torch.tensor(S.curl_velocity(G, rng, max_cells=2.0), device=device)
        dens = grid1.contiguous()
        add("advect order 2 (MacCormack: forward pass + corrected pass)", "transform.py:570-582",
            lambda: ops.advect_maccormack(dens, vel), 2 * (4.0 + 12.0 + 4.0) * G ** 3 + 4.0 * G ** 3)
        add("curl of a 3-component stream function", "transform.py:517-555", lambda: ops.curl_fwd(grid3), 24.0 * G ** 3)
        add("curl adjoint", "adjoint of transform.py:517-555", lambda: ops.curl_bwd(grid3), 24.0 * G ** 3)
        gfield = torch.randn(G, G, G, 3, device=device, generator=gen)
        add("Laplacian-pyramid normalisation of a 200^3 x 3 gradient (3 levels)", "util.py:57-110",
            lambda: U.lap_normalize(gfield, scale_n=3, is_3d=True, c=3), 2 * 12.0 * G ** 3, reps=5)
        # histogram loss of one 300 x 450 x 256-channel feature map against a style feature of the same size
        F_ = torch.rand(1, 75, 112, 256, device=device, generator=gen)
        Ft = torch.rand(1, 75, 112, 256, device=device, generator=gen)
        lacc = torch.zeros(1, device=device)
        gacc = torch.zeros_like(F_)
        add("histogram loss + gradient, 75 x 112 x 256 feature map", "styler_base.py:187-209, util.py:317-399",
            lambda: ops.hist_loss(F_, Ft, 1.0, lacc, gacc), 4.0 * F_.numel() * 4)
        Fi = torch.rand(1, 300, 450, 3, device=device, generator=gen) * 255
        Fit = torch.rand(1, 300, 450, 3, device=device, generator=gen) * 255
        gacc_i = torch.zeros_like(Fi)
        add("histogram loss + gradient, the 300 x 450 x 3 loss-net input (the default hist layer)",
            "styler_base.py:187-209, config.py:97", lambda: ops.hist_loss(Fi, Fit, 1.0, lacc, gacc_i), 4.0 * Fi.numel() * 4)
        img = torch.rand(4, 512, 1024, 3, device=device, generator=gen)
        coords = (torch.rand(4, 2, 512, 1024, device=device, generator=gen) * 2 - 1).contiguous()
        add("batch_warp2d of four 512 x 1024 x 3 images", "transform.py:206-236, 280-341", lambda: ops.warp2d_fwd(img, coords),
            4.0 * img.numel() * 2 + 4.0 * coords.numel())
        out.append({"config": "SURVEY 8(f) operators outside the iteration (resampler, order-2 advection, stream function, "
                              "gradient normalisation, histogram loss, 2-D warp): one call each",
                    "ops": ops_out})
        del grid1, grid3, pts, pts_rand, cells, vel, dens, gfield, F_, Ft, img, coords
    except Exception as e:  # pragma: no cover
        out.append({"config": "SURVEY 8(f) operators", "error": repr(e)})
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch(args)                      # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)"
                         % (args.gpus, world, args.gpus, args.gpus))
    # one rank per GPU; NFS_DIST_BACKEND=gloo lets several ranks share one GPU to exercise the sharded path on a
    # single-GPU box (functional check only: the ranks then time-share the device)
    backend = os.environ.get("NFS_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD
        seen = dist.get_world_size()
        assert seen == world, (seen, world)
    # the BASELINE metric at every N is the 8-view single-frame problem with the VIEWS sharded (strong scaling);
    # --scaling-by frames makes the frame-sharded sequence (configs[3], weak scaling) the headline instead
    mode = args.scaling_by or "views"

    from neural_flow_style_amd import _lib
    G, V = args.grid, args.views

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the single-frame problem (N=1 headline; view-sharded strong scaling for N>1) ---------------------------------
    views_ok = V % world == 0
    gs = rot_local = None
    if mode == "views" or views_ok:
        assert mode != "views" or views_ok, "views must divide over ranks"
        gs, rot_local, base = build_problem(G, V, device, rank, world)
    else:
        _, _, base = build_problem(G, V, device, 0, 1)

    from neural_flow_style_amd import ops as _ops
    gemm_mode0 = _ops.gemm_mode(None)          # the library's arithmetic for the Winograd GEMMs (1: split-limb, the default)
    out = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
           "dtype": DTYPE_SPLIT if gemm_mode0 == 1 else "f32", "data": "synthetic", "unit": "iters/s", "scaling_by": mode,
           "gemm_arithmetic": (
               "inputs, outputs, accumulators and every stored tensor are float32.  The batched Winograd GEMMs of the deep "
               "VGG layers write each float32 operand exactly as three bf16 limbs (round to nearest at each level: 3 x 8 "
               "significand bits = float32's 24) and run the six leading limb products on v_mfma_f32_16x16x32_bf16 with "
               "float32 accumulation: each product is carried to 2^-26, below float32's own rounding unit.  Measured "
               "against the float64 oracle the step's gradient is no worse than with the f32-input MFMA (parity.full_size "
               "vs parity.full_size_f32_mfma; tests/test_ops_gpu.py::test_split_limb_gemm_is_float32_accurate).  "
               "NFS_GEMM_MODE=0 runs the same GEMMs on v_mfma_f32_16x16x4_f32: `f32_mfma_gemm` below."
               if gemm_mode0 == 1 else "float32-input MFMA (NFS_GEMM_MODE=0)"),
           "metric": "stylization iters/sec on %d^3 smoke grid, %d views" % (G, V)}
    cfg_common = {"grid": G, "views": V, "image": [G, G], "style_layers": STYLE_LAYERS,
                  "vgg_weights": "synthetic He-normal seed 123 (no checkpoint offline)"}

    def views_step():
        return gs.step(rot_local, loss_view=True)

    def settle(step, n=5):
        """untimed set-up steps before the W warm-up steps: lazy state (packed filters, tile tuner), and -- at one or
        two local views -- the stylizer's measured eager-vs-hipGraph choice with its capture"""
        for _ in range(n):
            step()

    def frames_run():
        """BASELINE configs[3] in weak scaling: world x frames_per_rank frames, sharded by frames"""
        F_ = world * args.frames_per_rank
        st = build_sequence(args, device, rank, world, F_, base, pg)

        def frames_step():
            return st.iterate()

        settle(frames_step, 2)
        dt, last = time_steps(frames_step, barrier, args.warmup, args.steps, device, world)
        res = dict(value=F_ * args.steps / dt, ms_per_step=1e3 * dt / args.steps, scaling="weak",
                   final_loss=float(last.sum()))
        cfg_ = dict(cfg_common, workload="smokegun %d^3 sequence of %d frames (%d per rank), %d rotated views per "
                    "frame, VGG-19 conv1_1..conv5_1 Gram style loss, per-frame velocity variable through "
                    "advect + TF-Adam, updates aligned across frames by transport + Gaussian sigma %g "
                    "(BASELINE configs[3]; one step = one iteration over all frames; value = frame-"
                    "iterations/s; at one frame this is BASELINE configs[2])"
                    % (G, F_, args.frames_per_rank, V, args.window_sigma),
                    frames=F_, frames_per_rank=args.frames_per_rank, window_sigma=args.window_sigma,
                    parallelism="frames sharded over %d rank(s) in contiguous blocks, full %d-view batches per "
                                "rank, point-to-point halo exchange of the %d MB per-frame updates the temporal "
                                "filter reaches" % (world, V, 12 * G ** 3 // 2 ** 20))
        return res, cfg_, frames_step, F_

    if mode == "views":
        settle(views_step)
        dt, last = time_steps(views_step, barrier, args.warmup, args.steps, device, world)
        out.update(value=args.steps / dt, ms_per_step=1e3 * dt / args.steps, scaling="strong", final_loss=float(last))
        # how the step was submitted: eagerly, or as a hipGraph replay -- the stylizer measures at its second step whether
        # the host can issue the ~80 launches faster than the GPU runs them (hosts of one pool differ by 2x)
        out["submission"] = {"hipgraph": bool(gs.use_graph),
                             "trial_issue_ms": None if not getattr(gs, "graph_trial", None) else 1e3 * gs.graph_trial[0],
                             "trial_wall_ms": None if not getattr(gs, "graph_trial", None) else 1e3 * gs.graph_trial[1]}
        out["config"] = dict(cfg_common, workload="smokegun %d^3 single-frame, %d rotated views, VGG-19 conv1_1..conv5_1 "
                             "Gram style loss, grid velocity variable through advect + TF-Adam (BASELINE configs[2])"
                             % (G, V), views_per_rank=V // world,
                             parallelism="views sharded over %d rank(s), ONE all-reduce's worth of link traffic for the "
                                         "%d MB density-field gradient + loss per iteration" % (world, 4 * G ** 3 // 2 ** 20),
                             field_work=("D-slab sharded: reduce-scatter of the gradient over slabs of %d planes (two-plane "
                                         "halos in the chunks) -> slab-local smooth adjoint, advect adjoint + Adam, advect, "
                                         "smooth -> all-gather of the smoothed density" % gs.slab.cs)
                             if gs.slab is not None else
                             ("replicated on every rank behind one all-reduce(sum)" if world > 1 else "one rank"))
        step_fn, units = views_step, 1
        if gs._live_kw() and not args.no_skip_control:
            # data dependence of the headline, made visible: the rotate adjoint skips what only feeds voxels whose
            # velocity gradient is an exact zero (empty space, plateaus).  (a) the same problem with skipping off,
            # (b) a dense density on which nothing can be skipped -- both NOT the headline
            global LIVE_BOX_FRAC, EVER_WAVE_FRAC
            lf, bf, sk = live_box_fraction(gs)
            LIVE_BOX_FRAC = bf
            ef, EVER_WAVE_FRAC = ever_wave_fraction(gs)
            ctl = {"live_voxel_fraction": lf, "accumulated_box_fraction": bf, "tiles_skipped_fraction": sk,
                   "ever_live_voxel_fraction": ef, "adam_bytes_moved_fraction": EVER_WAVE_FRAC,
                   "note": "velocity variable: dL/dv(x) = g(x) * grad d0(x - v) is an exact zero where the eight "
                           "back-traced density corners are equal, whatever g(x) is; the rotate adjoint therefore sums "
                           "only the per-tile bounding boxes of the voxels within the smoothing stencil of a live voxel "
                           "(nfs_rotate_bwd_coef_live), and the fused advect-adjoint + ApplyAdam kernel leaves out the 256-voxel "
                           "waves none of whose voxels has ever been live (m = v = +0 there and the gradient is +-0: an exact "
                           "no-op; nfs_advect_bwd_adam_fwd_live_ever).  The Adam update is BIT-identical with and without it "
                           "(tests/test_dead_skip_gpu.py); the synthetic smoke of SURVEY 8(d) is %.0f %% live" % (100 * lf)}
            # (a stylizer of its own: a step without the masks ends the headline stylizer's ever-live bookkeeping for good)
            gso, roto, _ = build_problem(G, V, device, rank, world)
            gso.dead_skip = False
            ostep = lambda: gso.step(roto, loss_view=True)
            settle(ostep)
            dto, _ = time_steps(ostep, barrier, args.warmup, args.steps, device, world)
            ctl["skipping_off"] = {"value": args.steps / dto, "unit": "iters/s", "ms_per_step": 1e3 * dto / args.steps}
            del gso, roto
            torch.cuda.empty_cache()
            if world == 1 and not args.no_other_configs:
                gsd, rotd, _ = build_problem(G, V, device, rank, world, dense=True)
                dstep = lambda: gsd.step(rotd, loss_view=True)
                settle(dstep)
                dtd, _ = time_steps(dstep, barrier, args.warmup, args.steps, device, world)
                ctl["dense_density"] = {"value": args.steps / dtd, "unit": "iters/s", "ms_per_step": 1e3 * dtd / args.steps,
                                        "live_voxel_fraction": live_box_fraction(gsd)[0],
                                        "density": "0.05 + 0.1 U[0,1) per voxel: every stencil has differing corners"}
                del gsd, rotd
                torch.cuda.empty_cache()
            out["dead_region_skipping"] = ctl
        if world > 1 and not args.no_other_configs:
            # the same box, the other sharding: a sequence with one frame per rank (weak scaling)
            res, cfg_, _, _ = frames_run()
            out["frames_weak"] = dict(res, unit="frame-iters/s", workload=cfg_["workload"], parallelism=cfg_["parallelism"])
    else:
        res, cfg_, step_fn, units = frames_run()
        out.update(res)
        out["config"] = cfg_
        if gs is not None and world > 1:
            # the same box, the other sharding: the 8 views of ONE frame over the ranks (strong scaling)
            settle(views_step)
            dtv, lv = time_steps(views_step, barrier, args.warmup, args.steps, device, world)
            out["views_strong"] = {"value": args.steps / dtv, "unit": "iters/s", "ms_per_step": 1e3 * dtv / args.steps,
                                   "scaling": "strong", "views_per_rank": V // world, "final_loss": float(lv),
                                   "collective": "one all-reduce(sum) of %d MB + loss per iteration"
                                                 % (4 * G ** 3 // 2 ** 20)}
    if world > 1:
        import torch.distributed as dist
        from neural_flow_style_amd import parallel as _par
        out["collective_backend"] = dist.get_backend()
        out["world_size_seen"] = dist.get_world_size()
        # what every rank saw: the backend (nccl = RCCL), the group size, its device, and the collectives of a step timed
        # on the device (event pair on the stream that waits for the collective) -- a SCALE run then shows that RCCL had N
        # ranks on N devices and how far the exchange is from the 0.05-0.37 ms DESIGN section 7 estimates for it
        tsteps = max(2, min(args.steps, 5))
        barrier()
        _par.TIMER = {}
        for _ in range(tsteps):
            step_fn()
        barrier()
        coll = _par.timer_summary(tsteps)
        props = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "backend": dist.get_backend(),
                "world_size": dist.get_world_size(), "device_index": int(device.index), "device_name": props.name,
                "device_uuid": str(getattr(props, "uuid", "")), "views_local": None if rot_local is None else int(rot_local.shape[0]),
                "hipgraph": None if gs is None else bool(gs.use_graph), "collectives": coll,
                "collective_device_ms_per_step": sum(c["device_ms_per_step"] for c in coll.values())}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        out["ranks"] = allr
        out["collective_ms_per_step_max_over_ranks"] = max(r["collective_device_ms_per_step"] for r in allr)
        out["distinct_devices"] = len(set((r["device_uuid"] or r["device_index"]) for r in allr))

    if not args.no_sustained:
        out["sustained"] = dict(sustained(step_fn, barrier, units, device, world), unit=out["unit"])

    # ---- the same workload with the Winograd GEMMs in the OTHER arithmetic (see nfs_gemm_mode) ---------------------------
    if gs is not None and mode == "views" and not args.no_split_limb:
        from neural_flow_style_amd import ops
        other = 1 - gemm_mode0
        prev = ops.gemm_mode(other)
        try:
            gs.use_graph = False
            dts, lasts = time_steps(views_step, barrier, max(args.warmup, 2), args.steps, device, world)
            out["f32_mfma_gemm" if other == 0 else "split_limb_gemm"] = {
                "value": args.steps / dts, "unit": "iters/s", "ms_per_step": 1e3 * dts / args.steps,
                "final_loss": float(lasts), "dtype": "f32" if other == 0 else DTYPE_SPLIT,
                "note": "NOT the headline: the same step with nfs_gemm_mode(%d) -- " % other +
                        ("every Winograd GEMM on the float32-input MFMA (v_mfma_f32_16x16x4_f32, dense peak %.1f TFLOP/s): "
                         "the arithmetic of rounds 1-4's headline" % MFMA_F32_PEAK_TF if other == 0 else
                         "every float32 GEMM operand written exactly as three bf16 limbs, the six leading limb products "
                         "on v_mfma_f32_16x16x32_bf16, float32 accumulation")}
        finally:
            ops.gemm_mode(prev)

    # ---- per-kernel live measurement (extra profiled steps, after the headline timing) -----------
    if not args.no_kernel_profile and gs is not None:
        import ctypes
        psteps = max(2, min(args.steps, 5))
        gs.use_graph = False          # the per-call timers hook the C-ABI calls: a graph replay would bypass them
        L = _lib.lib()
        # pass 1 -- the HEADLINE configuration: HIP event pair around every launch of the GEMM kernel, recorded inside
        # the library on the stream the kernel is launched on
        L.nfs_gemm_timer(1)
        for _ in range(psteps):
            gs.step(rot_local, loss_view=True)
        torch.cuda.synchronize()
        s_ms, s_fl, s_n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
        s_by = ctypes.c_double()
        L.nfs_gemm_timer_read_kind(1, ctypes.byref(s_ms), ctypes.byref(s_fl), ctypes.byref(s_n), ctypes.byref(s_by))
        g_ms, g_fl, g_n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
        L.nfs_gemm_timer_read(ctypes.byref(g_ms), ctypes.byref(g_fl), ctypes.byref(g_n))             # the f32-input rest
        L.nfs_gemm_timer(0)
        # pass 2 -- event pairs around every C-ABI call; clean per-family timings need a single stream (an event pair
        # on one stream also counts the other stream's kernels sharing the chip)
        side = gs.loss.gram_side_stream
        gs.loss.vgg_streams = 1
        gs.loss.view_groups = 1
        gs.loss.gram_side_stream = False
        _lib.PROFILE = {}
        for _ in range(psteps):
            gs.step(rot_local, loss_view=True)
        torch.cuda.synchronize()
        prof, _lib.PROFILE = _lib.PROFILE, None
        gs.loss.gram_side_stream = side
        ov_us = event_pair_overhead_us(device)
        rows = kernel_table(prof, psteps)
        conv = [r for r in rows if r["kernel"] in ("nfs_conv3x3_fwd", "nfs_conv3x3_dgrad", "nfs_conv3x3_fwd_pool",
                                                   "nfs_conv3x3_dgrad_pool")]
        ms = sum(r["ms_per_step"] for r in conv)
        fl = sum(r["achieved"] * r["ms_per_step"] for r in conv)              # executed TF/s * ms
        fa = sum(r.get("algorithmic_tflops", 0.0) * r["ms_per_step"] for r in conv)   # direct-conv TF/s * ms
        n_launch = sum(r["launches_per_step"] for r in conv)
        tf = g_fl.value / (g_ms.value * 1e-3) / 1e12 if g_ms.value > 0 else 0.0
        f32_roof = {
            "kernel": "nfs::winograd_gemm_rb16_kernel / winograd_gemm_rb16_group_kernel on v_mfma_f32_16x16x4_f32"
                      + (": the Gram gradients of the five style layers (one grouped launch; mask / scale / symmetric "
                         "operand: f32-input MFMA in both arithmetic modes)" if s_n.value else ""),
            "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
            "launches_per_step": g_n.value / psteps, "ms_per_step": g_ms.value / psteps,
            "avg_launch_us": 1e3 * g_ms.value / max(g_n.value, 1)}
        if s_n.value:
            # the dominant kernel in the default arithmetic: executed bf16-MFMA flops = 6 limb products per float32 product
            tfe = s_fl.value / (s_ms.value * 1e-3) / 1e12
            out["roofline"] = {
                "kernel": "nfs::winograd_gemm_rb16s_kernel: the 49 Winograd F(5x5,3x3) / 36 F(4x4,3x3) products of every conv "
                          "layer from conv3_1 on (forward and data gradient) as batched GEMMs in split-limb arithmetic on "
                          "v_mfma_f32_16x16x32_bf16 (A split into three bf16 limb planes while staged into LDS, B = the "
                          "filters' limb planes made at pack time -- launches of fewer than 128 rows split the float32 "
                          "pack in registers instead; six limb products per 16 x 16 x 32 block): %d "
                          "launches/step, %.2f ms/step = the largest share of the step"
                          % (s_n.value // psteps, s_ms.value / psteps),
                "bound": "mfma", "achieved": 6.0 * tfe, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "frac": 6.0 * tfe / MFMA_BF16_PEAK_TF, "f32_equivalent_tflops": tfe,
                "traffic": pmc_traffic("winograd_gemm_rb16s"),
                "flops_per_launch": 6.0 * s_fl.value / max(s_n.value, 1),
                "f32_equivalent_flops_per_launch": s_fl.value / max(s_n.value, 1),
                "avg_launch_us": 1e3 * s_ms.value / max(s_n.value, 1), "event_pair_overhead_us": None,
                "ms_per_step": s_ms.value / psteps,
                "configuration": "headline, %d local views, one stream" % rot_local.shape[0],
                "note": "achieved = EXECUTED bf16 MFMA flops (6 limb products x 2*Z*T*K*N per launch) / summed launch "
                        "durations, HIP events on the launch stream around every launch (nfs_gemm_timer) in the headline "
                        "configuration; peak = the dense bf16 MFMA rate.  f32_equivalent_tflops = 2*Z*T*K*N / time, the "
                        "figure comparable with rounds 1-4 (f32-input MFMA, peak %.1f)" % MFMA_F32_PEAK_TF,
                # the same launches against HBM: each reads V and the packed filters and writes M exactly once when
                # nothing is re-fetched (tools/pmc_rb16s_traffic.sh: 98 MB fetched for 71 MB of operands at conv4_2)
                "hbm_side": {"algorithmic_bytes_per_launch": s_by.value / max(s_n.value, 1),
                             "achieved_gbs": s_by.value / (s_ms.value * 1e-3) / 1e9, "peak_gbs": HBM_PEAK_GBS,
                             "frac_hbm": s_by.value / (s_ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "Z (4 T K + 6 K N + 4 T N) bytes per launch (V read, the filters' bf16 limb planes read, M written): at the bf16 rate these products are bound as "
                                     "much by their compulsory traffic as by the matrix pipe (DESIGN.md section 3, K7s16)"},
                "f32_input_launches": f32_roof}
        else:
            out["roofline"] = {}
        _roof_f32 = {
            "kernel": "nfs::winograd_gemm_rb16_kernel (+ its grouped form winograd_gemm_rb16_group_kernel: the five Gram "
                      "gradients in one launch): batched f32-MFMA GEMM on v_mfma_f32_16x16x4_f32, filters from L2 "
                      "straight into registers -- the 49 Winograd F(5x5,3x3) or 36 F(4x4,3x3) products of every conv layer "
                      "from conv3_1 on, forward and data gradient, and the Gram gradient; the narrower layers run in the "
                      "single-kernel form winograd_fused_kernel: %d launches/step, %.2f ms/step = the largest share of "
                      "the step" % (g_n.value // psteps, g_ms.value / psteps),
            "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
            "traffic": pmc_traffic("winograd_gemm_"),
            "flops_per_launch": g_fl.value / max(g_n.value, 1), "avg_launch_us": 1e3 * g_ms.value / max(g_n.value, 1),
            "event_pair_overhead_us": ov_us,
            "ms_per_step": g_ms.value / psteps,
            "configuration": "headline, %d local views, one stream" % rot_local.shape[0],
            "note": "achieved = executed MFMA flops (2*Z*T*K*N per launch) / summed launch durations, HIP events on "
                    "the launch stream around every launch (nfs_gemm_timer) in the headline configuration.  An event "
                    "pair adds dispatch latency that rocprofv3's kernel-only durations do not contain "
                    "(event_pair_overhead_us, measured around a 1-element fill kernel, for information: it is NOT "
                    "subtracted anywhere); profiles/ holds the rocprofv3 in-step average of the same launches",
            # the whole conv family seen from the operator boundary (input transform + GEMM + output transform, the
            # single-kernel form, or the direct kernel): EXECUTED MFMA flops over the ABI-call time
            "conv_family": {"launches_per_step": n_launch, "ms_per_step": ms, "executed_tflops": fl / ms,
                            "frac": fl / ms / MFMA_F32_PEAK_TF, "algorithmic_tflops": fa / ms,
                            "note": "frac = executed MFMA flops (Winograd products incl. tile padding, "
                                    "nfs_conv3x3_executed_flops) / ABI-call time / f32 MFMA peak; algorithmic_tflops = "
                                    "the direct-conv flops 2*B*H*W*9*Ci*Co of the same calls over the same time, for "
                                    "reference (Winograd executes 2.25-4.6x fewer multiplies: not a roofline figure; in the "
                                    "split-limb arithmetic the deep layers' products run on the bf16 pipe, so this fraction "
                                    "of the f32-input peak is an f32-EQUIVALENT rate, not a pipe utilisation)"}}
        if s_n.value:
            out["roofline"]["event_pair_overhead_us"] = ov_us
            out["roofline"]["conv_family"] = _roof_f32["conv_family"]
        else:
            out["roofline"] = _roof_f32
        # render + advect family against the HBM roofline (north_star's >= 40 % target): measured kernels, two byte
        # accountings -- the kernels as built (the rotated volume is KEPT for the adjoint: written once, read once more)
        # and SURVEY 8(d)'s fully fused counts (rotate+render fwd 4VG^3 + 4VG^2, adjoint 8VG^3 + 4VG^2)
        fam = [r for r in rows if r["kernel"] in ("nfs_rotate_render_fwd", "nfs_render_bwd", "nfs_rotate_bwd",
                                                  "nfs_rotate_render_fwd_coef", "nfs_render_ray_coef", "nfs_rotate_bwd_coef",
                                                  "nfs_rotate_bwd_coef_live", "nfs_advect_fwd", "nfs_advect_fwd_live",
                                                  "nfs_advect_bwd_adam", "nfs_advect_bwd", "nfs_advect_bwd_adam_fwd",
                                                  "nfs_advect_bwd_adam_fwd_live", "nfs_advect_bwd_adam_fwd_live_ever")]
        if fam:
            fms = sum(r["ms_per_step"] for r in fam)
            built = sum(r["achieved"] * r["ms_per_step"] for r in fam)        # GB/s * ms = MB
            Vl, G3 = int(rot_local.shape[0]), float(G) ** 3
            # (the adjoint's share scaled by what the live mask leaves of it: bytes not moved are not counted)
            live_adj = LIVE_BOX_FRAC if any(r["kernel"] == "nfs_rotate_bwd_coef_live" for r in fam) else 1.0
            fused = (4.0 * Vl * G3 + 4.0 * Vl * G * G) + live_adj * (8.0 * Vl * G3 + 4.0 * Vl * G * G)
            fused += sum(r["achieved"] * r["ms_per_step"] * 1e6 for r in fam if "advect" in r["kernel"])
            out["render_advect_family"] = {
                "kernels": [r["kernel"] for r in fam], "ms_per_step": fms,
                "as_built": {"bytes_per_step": built * 1e6, "frac_hbm": built / fms / HBM_PEAK_GBS,
                             "note": "as the kernels move them: with the u / coefficient form of the adjoint (round 4) the "
                                     "forward writes a kept volume (8VG^3) and the rotate adjoint reads it once (4VG^3 + "
                                     "4G^3) -- the render-adjoint pass (8VG^3) is gone; with NFS_RENDER_COEF=0: fwd 8VG^3, "
                                     "render adjoint 8VG^3, rotate adjoint 4VG^3 + 8G^3"},
                "live_box_fraction": live_adj,
                "survey_fused": {"bytes_per_step": fused, "frac_hbm": fused / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "SURVEY 8(d) fully fused rotate+render (fwd 4VG^3 + 4VG^2, adjoint 8VG^3 + "
                                         "4VG^2) + the advect kernels as built, over the same measured time"}}
        out["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]

    if rank == 0 and world == 1:
        if not args.no_parity:
            try:
                out["parity"] = small_parity(device)
            except Exception as e:  # pragma: no cover
                out["parity"] = {"error": repr(e)}
        if not args.no_other_configs:
            out["other_configs"] = other_configs(device, base)
        if not args.no_cpu_baseline:
            cb, fsp = cpu_baseline(base, G, V, args.cpu_views, gs=gs, device=device, full=args.cpu_baseline_full)
            out["cpu_baseline"] = cb
            if fsp is not None:
                f32p = fsp.pop("f32_mfma", None)
                out.setdefault("parity", {})["full_size"] = fsp
                if f32p is not None:
                    out["parity"]["full_size_f32_mfma"] = f32p
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
