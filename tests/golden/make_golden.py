#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference (TensorFlow 1.15) cannot be imported in this container, so these vectors are
outputs of ``oracle/nfs_oracle.py`` (the cited-line restatement), NOT of the reference itself;
the only vector that comes from the reference tree is the 5x5 warp docstring
(transform.py:1859-1885), stored verbatim in ``warp2d_kat.npz``.  Every file records its seed
and shapes.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nfs_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), **kw)
    print(name, {k: np.asarray(v).shape for k, v in kw.items()})


def make_g2p():
    """(6) SURVEY 8(f)-1: grid -> particle sampling and the SimG2P resampler, seed 11"""
    rng = np.random.RandomState(11)
    g = torch.tensor(rng.randn(1, 9, 7, 11, 3).astype(np.float32))
    p = torch.tensor(rng.uniform(-0.1, 1.1, (1, 300, 3)).astype(np.float32))
    G = 12
    zz, yy, xx = np.meshgrid(*[np.linspace(0, 1, G)] * 3, indexing="ij")
    d = np.exp(-((zz - 0.5) ** 2 + (yy - 0.45) ** 2 + (xx - 0.55) ** 2) / 0.04).astype(np.float32)
    d[d < 0.05] = 0
    u = (rng.randn(G, G, G, 3) * 0.01).astype(np.float32)
    cfg = dict(domain=[G, G, G], radius=0.5, rest_density=1000.0, nsize=1, support=4, octave_n=2, octave_scale=2.0,
               lr=0.002, iter=3)
    pid = np.array(np.where(d > 0.3)).T.astype(np.float64) + 0.5
    x = torch.tensor((pid / G).astype(np.float32))
    r = O.simg2p_optimize(x, torch.tensor(d), torch.tensor(u), cfg, [G, G, G])
    save("ops_g2p.npz", seed=11, g=g, p=p, cubic=O.g2p(g, p, is_2d=False), linear=O.g2p(g, p, is_2d=False, is_linear=True),
         d=d, u=u, x=x, p_adv=r["p_adv"], p_new=r["p_new"], losses=np.array(r["l"], np.float64), r_smp=r["r_smp"],
         d_diff=r["d_diff"], cfg_keys=np.array(sorted(cfg)), cfg_vals=np.array([str(cfg[k]) for k in sorted(cfg)]))


def make_inception():
    """(7) SURVEY 8(f)-3: the Inception-v1 restatement down to mixed3b on a 32 x 40 image -- four tensor kinds (a ReLU
    output, two module outputs incl. the 480-channel one, a '*_pre_relu' tensor) and the gradient of a fixed
    functional of them with respect to the image.  Weights: the seeded synthetic set (seed 17) of
    neural-flow-style_amd/inception.py, regenerated from the seed by the tests (the arrays would be 2.7 MB)."""
    from neural_flow_style_amd import inception
    w = inception.synthetic_weights(17, upto="mixed3b")
    rng = np.random.RandomState(17)
    img = (rng.rand(1, 32, 40, 3) * 255).astype(np.float32)
    names = ["conv2d2", "mixed3a", "mixed3b", "mixed3b_3x3_bottleneck_pre_relu"]
    x = torch.tensor(img, requires_grad=True)
    feats = O.inception_v1_features(x, w, "mixed3b")
    proj = {n: rng.randn(*feats[n].shape).astype(np.float32) for n in names}
    total = sum((feats[n] * torch.tensor(proj[n])).sum() / float(feats[n].detach().abs().mean()) for n in names)
    (gx,) = torch.autograd.grad(total, x)
    save("inception_mixed3b.npz", seed=17, img=img, grad_img=gx,
         scale=np.array([float(feats[n].detach().abs().mean()) for n in names], np.float64),
         **{"feat_" + n: feats[n].detach() for n in names}, **{"proj_" + n: proj[n] for n in names})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "g2p":      # regenerate only the 8(f)-1 fixture
        make_g2p()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "inception":   # ... only the 8(f)-3 fixture
        make_inception()
        return
    # (1) the reference's own known-answer vector
    save("warp2d_kat.npz", img=np.arange(25, dtype=np.float32).reshape(5, 5),
         zoom_in=np.array([[6, 6.5, 7, 7.5, 8], [8.5, 9, 9.5, 10, 10.5], [11, 11.5, 12, 12.5, 13],
                           [13.5, 14, 14.5, 15, 15.5], [16, 16.5, 17, 17.5, 18]], np.float32))

    # (2) per-operator forward + gradient on a 12^3 grid / 200 particles, seed 7
    rng = np.random.RandomState(7)
    G = 12
    d = torch.tensor(rng.rand(1, G, G, G, 1).astype(np.float32), requires_grad=True)
    vel = torch.tensor((rng.randn(1, G, G, G, 3) * 0.15).astype(np.float32), requires_grad=True)
    th, ph = np.deg2rad(8.0), np.deg2rad(-4.0)
    ry = np.array([[np.cos(th), 0, -np.sin(th)], [0, 1, 0], [np.sin(th), 0, np.cos(th)]])
    rz = np.array([[np.cos(ph), -np.sin(ph), 0], [np.sin(ph), np.cos(ph), 0], [0, 0, 1]])
    R = torch.tensor(np.stack([np.eye(3), ry @ rz]).astype(np.float32))
    g_vol = torch.tensor(rng.randn(1, G, G, G, 1).astype(np.float32))
    adv = O.advect(d, vel)
    ga_d, ga_v = torch.autograd.grad(adv, (d, vel), g_vol)
    rot = O.rotate(d, R)
    g_rot = torch.tensor(rng.randn(2, G, G, G, 1).astype(np.float32))
    (gr_d,) = torch.autograd.grad(rot, d, g_rot, retain_graph=True)
    sm = O.smooth3d_relu(d - 0.5, 3)
    (gs_d,) = torch.autograd.grad(sm, d, g_vol)
    img = torch.cat([O.render(rot[v:v + 1], 0.2) for v in range(2)])
    g_img = torch.tensor(rng.randn(2, G, G, 1).astype(np.float32))
    (gi_d,) = torch.autograd.grad(img, d, g_img)
    save("ops_grid12.npz", seed=7, d=d.detach(), vel=vel.detach(), R=R, g_vol=g_vol, g_rot=g_rot, g_img=g_img,
         advect=adv.detach(), advect_gd=ga_d, advect_gvel=ga_v, rotate=rot.detach(), rotate_gd=gr_d,
         smooth=sm.detach(), smooth_gd=gs_d, render=img.detach(), render_gd=gi_d)

    p = torch.tensor(rng.uniform(-0.03, 1.03, (1, 200, 3)).astype(np.float32), requires_grad=True)
    x = torch.tensor(rng.rand(1, 200, 1).astype(np.float32), requires_grad=True)
    dens = O.p2g(p, [G, G, G], [G, G, G], 0.5, 1000., 1, is_2d=False, clip=False)
    wav = O.p2g_wavg(p, x, [G, G, G], [G, G, G], 0.5, 1, is_2d=False, clip=False)
    (gp_d,) = torch.autograd.grad(dens, p, g_vol)
    gp_w, gx_w = torch.autograd.grad(wav, (p, x), g_vol)
    save("ops_splat200.npz", seed=7, p=p.detach(), x=x.detach(), g=g_vol, p2g=dens.detach(), p2g_gp=gp_d,
         wavg=wav.detach(), wavg_gp=gp_w, wavg_gx=gx_w)

    # (3) TF-Adam 5 steps incl. tiny gradients
    rng = np.random.RandomState(11)
    x0 = rng.randn(64).astype(np.float32)
    gs = [(rng.randn(64) * (1e-7 if t % 2 else 1.0)).astype(np.float32) for t in range(5)]
    opt = O.TFAdam(); xt = torch.tensor(x0); traj = []
    for g in gs:
        xt = opt.step(xt, torch.tensor(g), 0.1); traj.append(xt.numpy().copy())
    save("adam_tf5.npz", seed=11, x0=x0, grads=np.stack(gs), traj=np.stack(traj), lr=0.1)

    # (4) end-to-end: 5-iteration Adam trajectory, 24^3 grid, 2 views, conv1_1+conv2_1 at width/8
    rng = np.random.RandomState(123)
    G, V, K = 24, 2, 5
    from neural_flow_style_amd import synthetic as S
    d0 = S.blob_density(G, rng)
    v0 = (rng.randn(G, G, G, 3) * 0.3 / (G - 1)).astype(np.float32)
    simg = S.style_image(G, G, rng)
    mats = np.asarray(S.uniform_views(V), np.float32)
    layers = ["conv1_1", "conv2_1"]
    w = O.synthetic_vgg19_weights(123, upto="conv2_1")
    sfe = O.style_target_features(torch.tensor(simg)[None], w, layers, upto="conv2_1")
    cfg = dict(k=3, transmit=0.05, style_layer=layers, w_style_layer=[1.0, 1.0], w_style=1.0, upto="conv2_1")
    vel = torch.tensor(v0)[None]; opt = O.TFAdam(); losses = []
    for _ in range(K):
        vv = vel.clone().requires_grad_()
        tot, _, _ = O.grid_forward(torch.tensor(d0)[None, ..., None], vv, torch.tensor(mats), cfg, w, sfe)
        (g,) = torch.autograd.grad(tot, vv)
        if not losses:
            g_first = g[0].numpy().copy()
        vel = opt.step(vel, g, 0.002); losses.append(float(tot))
    d_fin = O.smooth3d_relu(O.advect(torch.tensor(d0)[None, ..., None], vel), 3)[0, ..., 0]
    save("e2e_grid24.npz", seed=123, d0=d0, vel0=v0, style=simg, rot=mats,
         losses=np.array(losses), grad_first_step=g_first.astype(np.float32)[::2, ::2, ::2],
         d_final=d_fin.numpy().astype(np.float32)[::2, ::2, ::2], lr=0.002, transmit=0.05, layers=np.array(layers))
    make_g2p()


if __name__ == "__main__":
    main()
