#!/usr/bin/env python
"""Headline benchmark: stylisation iterations/sec on the 200^3 smoke grid with 8 views
(BASELINE.json metric; workload = configs[2], the configuration the metric is quoted on).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one stylisation iteration: advect -> smooth/clamp -> for all 8 views: rotate+render
-> VGG-19 conv1_1..conv5_1 -> Gram style loss -> full adjoint chain -> (all-reduce of the field
gradient over ranks) -> TF-Adam update of the 200^3 x 3 velocity field.  The 8 views are sharded
over the N ranks (strong scaling: total work is fixed).  Inputs are synthetic (seed 123) and
resident in HBM before the timed region.

Rank 0 prints ONE JSON line with the contract keys plus
  "roofline"     dominant kernel (the f32-MFMA 3x3 conv) measured live with events on the launch stream,
  "kernels"      the same measurement for every kernel family (HBM GB/s or TFLOP/s + fraction of peak),
  "cpu_baseline" the CPU oracle timed on this box's host cores on a bounded sample (N=1 only),
  "parity"       gradient relative-L2 of the HIP path vs the oracle on a small case (N=1 only).
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
STYLE_LAYERS = ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=1, help="views in the bounded CPU sample")
    return ap.parse_args()


def build_problem(G, V, device, rank, world):
    from neural_flow_style_amd import engine, vgg
    from neural_flow_style_amd import synthetic as S
    from neural_flow_style_amd import transform as T
    rng = np.random.RandomState(123)
    d0 = S.blob_density(G, rng)
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
    """-> (kind, amount): kind 'B' bytes or 'F' flops, from the C-ABI call arguments"""
    if name == "nfs_conv3x3_fwd":
        B, H, W, Ci, Co = a[4:9]
        return "F", 2.0 * B * H * W * 9 * Ci * Co
    if name == "nfs_conv3x3_dgrad":
        B, H, W, Ci, Co = a[5:10]
        return "F", 2.0 * B * H * W * 9 * Ci * Co
    if name == "nfs_conv3x3_fwd_pool":
        B, H, W, Ci, Co = a[5:10]
        return "F", 2.0 * B * H * W * 9 * Ci * Co
    if name == "nfs_conv3x3_dgrad_pool":
        B, H, W, Ci, Co = a[6:11]
        return "F", 2.0 * B * H * W * 9 * Ci * Co
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
    if name == "nfs_rotate_bwd":
        V, D, H, W, C = a[3:8]
        return "B", 4.0 * V * D * H * W * C + 8.0 * D * H * W * C
    if name == "nfs_rotate_render_bwd":
        V, D, H, W = a[5:9]
        return "B", 8.0 * V * D * H * W + 4.0 * V * H * W
    if name == "nfs_advect_fwd":
        D, H, W, C = a[3:7]
        return "B", (8.0 * C + 12.0) * D * H * W
    if name == "nfs_advect_bwd":
        D, H, W, C = a[5:9]
        return "B", ((12.0 if a[3] else 8.0) * C + 24.0) * D * H * W
    if name == "nfs_advect_bwd_adam":
        D, H, W = a[5:8]
        # g_out 4 + gathered d 4 + vel/m/v read 36 + vel/m/v write 36 bytes per voxel
        return "B", 80.0 * D * H * W
    if name in ("nfs_smooth3d_relu_fwd",):
        D, H, W = a[2:5]
        return "B", 8.0 * D * H * W
    if name in ("nfs_smooth3d_relu_bwd",):
        D, H, W = a[3:6]
        return "B", 8.0 * D * H * W
    if name == "nfs_adam_tf_step":
        return "B", 28.0 * a[4]
    return None, 0.0


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


def kernel_table(profile, steps, overhead_us=0.0):
    rows = []
    for name, recs in sorted(profile.items()):
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
        ms_net = max(ms - 1e-3 * overhead_us * len(recs), 0.25 * ms)
        kind, _ = work_of(name, recs[0][2])
        amount = sum(work_of(name, r[2])[1] for r in recs)
        row = {"kernel": name, "launches_per_step": len(recs) / steps, "ms_per_step": ms / steps,
               "avg_launch_us": 1e3 * ms / len(recs), "ms_per_step_net": ms_net / steps}
        if kind == "B":
            ach = amount / (ms * 1e-3) / 1e9
            row.update(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                       frac_net=amount / (ms_net * 1e-3) / 1e9 / HBM_PEAK_GBS)
        elif kind == "F":
            ach = amount / (ms * 1e-3) / 1e12
            row.update(bound="mfma", achieved=ach, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=ach / MFMA_F32_PEAK_TF,
                       frac_net=amount / (ms_net * 1e-3) / 1e12 / MFMA_F32_PEAK_TF)
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def cpu_baseline(data, G, V, n_views):
    """The oracle (CPU restatement of the reference graph; the TF-1.15 reference cannot run here)
    timed on the host cores on a bounded sample: forward+backward of ``n_views`` of the V views at
    the full grid size, extrapolated to one full iteration."""
    from oracle import nfs_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    O.FAST_WARP = True   # multi-threaded grid_sample for the 8-tap warps (identical numerics, tested)
    w = O.synthetic_vgg19_weights(123, upto="conv5_1")
    sfe = O.style_target_features(torch.tensor(data["simg"])[None], w, STYLE_LAYERS, upto="conv5_1")
    cfg = dict(k=3, transmit=0.01, style_layer=STYLE_LAYERS, w_style_layer=[1.0] * 5, w_style=1.0, upto="conv5_1")
    d0 = torch.tensor(data["d0"])[None, ..., None]
    vel = torch.tensor(data["vel"])[None].requires_grad_()
    rot = torch.tensor(np.asarray(data["mats"][:n_views], np.float32))
    t0 = time.perf_counter()
    total, _, _ = O.grid_forward(d0, vel, rot, cfg, w, sfe)
    (g,) = torch.autograd.grad(total, vel)
    opt = O.TFAdam(); opt.step(vel.detach(), g, 1e-3)
    dt = time.perf_counter() - t0
    # prologue/epilogue (advect, smooth, their adjoints, Adam) are inside dt once; views dominate
    est_iter = dt * V / n_views
    return {"value": 1.0 / est_iter, "unit": "iters/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle (PyTorch-CPU restatement, f32) fwd+bwd+Adam of %d of %d views at %d^3, %.1f s "
                      "measured, scaled by %d/%d to one iteration" % (n_views, V, G, dt, V, n_views),
            "seconds_measured": dt}


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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # one rank per GPU; NFS_DIST_BACKEND=gloo lets several ranks share one GPU to exercise the sharded path on a
    # single-GPU box (functional check only: the ranks then time-share the device)
    backend = os.environ.get("NFS_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert args.views % world == 0, "views must divide over ranks"

    from neural_flow_style_amd import _lib
    G, V = args.grid, args.views
    gs, rot_local, data = build_problem(G, V, device, rank, world)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        gs.step(rot_local)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = gs.step(rot_local)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = 1e3 * dt / args.steps

    out = {
        "metric": "stylization iters/sec on 200^3 smoke grid, 8 views",
        "value": args.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "smokegun %d^3 single-frame, %d rotated views, VGG-19 conv1_1..conv5_1 Gram style "
                               "loss, grid velocity variable through advect + TF-Adam (BASELINE configs[2])" % (G, V),
                   "grid": G, "views": V, "views_per_rank": V // world, "image": [G, G],
                   "style_layers": STYLE_LAYERS, "vgg_weights": "synthetic He-normal seed 123 (no checkpoint offline)",
                   "parallelism": "views sharded over %d rank(s), all-reduce(sum) of the %d MB density-field "
                                  "gradient" % (world, 4 * G ** 3 // 2 ** 20)},
        "final_loss": float(last),
    }

    # ---- per-kernel live measurement (extra profiled steps, after the headline timing) -----------
    if not args.no_kernel_profile:
        psteps = max(2, min(args.steps, 5))
        gs.use_graph = False          # the per-call timers hook the C-ABI calls: a graph replay would bypass them
        # clean per-kernel timings need a single stream (with two concurrent view groups the event pairs of one
        # stream also count the other stream's kernels sharing the chip); the headline above uses the default
        gs.loss.vgg_streams = 1
        gs.loss.view_groups = 1
        gs.loss.gram_side_stream = False
        import ctypes
        L = _lib.lib()
        _lib.PROFILE = {}
        L.nfs_gemm_timer(1)                 # HIP event pair around every launch of the GEMM kernel, on its stream
        for _ in range(psteps):
            gs.step(rot_local)
        torch.cuda.synchronize()
        prof, _lib.PROFILE = _lib.PROFILE, None
        g_ms, g_fl, g_n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
        L.nfs_gemm_timer_read(ctypes.byref(g_ms), ctypes.byref(g_fl), ctypes.byref(g_n))
        L.nfs_gemm_timer(0)
        ov_us = event_pair_overhead_us(device)
        rows = kernel_table(prof, psteps, ov_us)
        conv = [r for r in rows if r["kernel"] in ("nfs_conv3x3_fwd", "nfs_conv3x3_dgrad", "nfs_conv3x3_fwd_pool",
                                                   "nfs_conv3x3_dgrad_pool")]
        ms = sum(r["ms_per_step"] for r in conv)
        fl = sum(r["achieved"] * r["ms_per_step"] for r in conv)  # TF/s * ms
        n_launch = sum(r["launches_per_step"] for r in conv)
        tf = g_fl.value / (g_ms.value * 1e-3) / 1e12
        out["roofline"] = {
            "kernel": "nfs::winograd_gemm_kernel (batched f32-MFMA GEMM: the 36 Winograd F(4x4,3x3) products of every "
                      ">=64-channel conv layer, forward and data gradient, and the Gram gradient): %d launches/step, "
                      "%.2f ms/step = the largest share of the step" % (g_n.value // psteps, g_ms.value / psteps),
            "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
            "traffic": pmc_traffic("winograd_gemm_kernel"),
            "flops_per_launch": g_fl.value / max(g_n.value, 1), "avg_launch_us": 1e3 * g_ms.value / max(g_n.value, 1),
            "event_pair_overhead_us": ov_us,
            "frac_net": g_fl.value / (max(g_ms.value - 1e-3 * ov_us * g_n.value, 0.25 * g_ms.value) * 1e-3) / 1e12
                        / MFMA_F32_PEAK_TF,
            "ms_per_step": g_ms.value / psteps,
            "note": "achieved = executed MFMA flops (2*Z*T*K*N per launch) / summed launch durations, HIP events on "
                    "the launch stream (nfs_gemm_timer); frac_net subtracts event_pair_overhead_us per launch (the "
                    "dispatch latency an event pair adds on a busy stream, measured around a 1-element fill kernel) "
                    "and is what rocprofv3's kernel-only durations correspond to",
            # the whole conv family seen from the operator boundary: what the layer computes (direct-conv flops)
            # over the time of the ABI call (input transform + GEMM + output transform, or the direct kernel)
            "conv_family": {"launches_per_step": n_launch, "ms_per_step": ms, "algorithmic_tflops": fl / ms,
                            "algorithmic_over_mfma_peak": fl / ms / MFMA_F32_PEAK_TF,
                            "note": "ALGORITHMIC direct-conv flops (2*B*H*W*9*Ci*Co) / ABI-call time; Winograd "
                                    "executes 4x fewer multiplies, so this may exceed the MFMA peak"}}
        out["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]

    if rank == 0 and world == 1:
        try:
            out["parity"] = small_parity(device)
        except Exception as e:  # pragma: no cover
            out["parity"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, G, V, args.cpu_views)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
