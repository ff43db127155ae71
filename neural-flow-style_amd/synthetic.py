"""Seeded synthetic inputs for the benchmark / parity runs.
No dataset, checkpoint or style image ships with the reference and there is no network,
so every input is generated from ``np.random.RandomState`` in a fixed, documented order."""
from __future__ import annotations

import numpy as np

from . import transform as T


def blob_density(G, rng, n_blobs=32):
    """smoke-like density [G,G,G] in [0,1]: sum of Gaussian blobs (centres U[.2,.8]^3, sigma
    U[.03,.10]*G, amplitude U[.2,1]), clipped to [0,1], values < 0.02 zeroed."""
    ax = np.arange(G, dtype=np.float64)
    d = np.zeros((G, G, G), np.float64)
    for _ in range(n_blobs):
        c = rng.uniform(0.2, 0.8, 3) * (G - 1)
        s = rng.uniform(0.03, 0.10) * G
        a = rng.uniform(0.2, 1.0)
        g = [np.exp(-0.5 * ((ax - c[k]) / s) ** 2) for k in range(3)]
        d += a * np.einsum("i,j,k->ijk", g[0], g[1], g[2])
    d = np.clip(d, 0.0, 1.0)
    d[d < 0.02] = 0.0
    return d.astype(np.float32)


def curl_velocity(G, rng, max_cells=2.0):
    """divergence-free velocity [G,G,G,3] in the advect() units (normalised, component k along
    array axis k): curl of a smooth 3-component potential (white noise, Gaussian sigma=G/16),
    scaled so that the largest displacement is ``max_cells`` cells = 2*max_cells/(G-1)."""
    from scipy.ndimage import gaussian_filter
    pot = [gaussian_filter(rng.randn(G, G, G), sigma=G / 16.0) for _ in range(3)]
    d = lambda f, ax: np.gradient(f, axis=ax)
    v = np.stack([d(pot[2], 1) - d(pot[1], 2), d(pot[0], 2) - d(pot[2], 0), d(pot[1], 0) - d(pot[0], 1)], -1)
    v *= (2.0 * max_cells / (G - 1)) / np.abs(v).max()
    return v.astype(np.float32)


def uniform_views(V):
    """V=8: the lattice phi in {-5,5}, theta in {-10,-10/3,10/3,10} (config.py:63-68 defaults);
    other V: theta evenly spaced in [-10,10] at phi 0 (odd) or phi in {-5,5} (even)."""
    if V == 1:
        return [np.eye(3)]
    if V % 2 == 0:
        phis, thetas = [-5.0, 5.0], np.linspace(-10, 10, V // 2) if V > 2 else [0.0]
    else:
        phis, thetas = [0.0], np.linspace(-10, 10, V)
    return [np.matmul(T.rot_y_3d(t), T.rot_z_3d(p)) for p in phis for t in thetas]


def style_image(H, W, rng):
    """127.5 + 80 sin(2 pi (3x+5y)) [1,.7,.4] + N(0,10), float32 [H,W,3] in 0..255"""
    y, x = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    s = np.sin(2 * np.pi * (3 * x + 5 * y))[..., None] * np.array([1.0, 0.7, 0.4])
    img = 127.5 + 80.0 * s + rng.randn(H, W, 3) * 10.0
    return np.clip(img, 0, 255).astype(np.float32)


def blob_particles(N, rng, n_blobs=16):
    """particles [N,3] ordered (z,y,x) in [0.05,0.95]^3 sampled from a blob mixture"""
    c = rng.uniform(0.25, 0.75, (n_blobs, 3))
    s = rng.uniform(0.04, 0.10, n_blobs)
    k = rng.randint(0, n_blobs, N)
    p = c[k] + rng.randn(N, 3) * s[k, None]
    return np.clip(p, 0.05, 0.95).astype(np.float32)


def dambreak_particles(n_side, rng, n=None):
    """2-D jittered lattice filling the lower-left 35% x 60% box, [N,2] ordered (y,x) in [0,1].  ``n``: keep the first n
    particles (row-major from the floor up): SURVEY 8(d)'s configs[0] workload is ``dambreak_particles(280, rng, 16384)``
    (scene/dambreak2d.py:96-103 at two particles per cell and dimension on the 128 x 128 grid)"""
    ny, nx = int(n_side * 0.60), int(n_side * 0.35)
    yy, xx = np.meshgrid((np.arange(ny) + 0.5) / n_side, (np.arange(nx) + 0.5) / n_side, indexing="ij")
    p = np.stack([yy.ravel(), xx.ravel()], -1)
    p += rng.uniform(-0.25, 0.25, p.shape) / n_side
    if n is not None:
        assert n <= p.shape[0], (n, p.shape[0])
        p = p[:n]
    return p.astype(np.float32)
