"""Second, INDEPENDENT restatement of the floating-point hot path -- TEST INFRASTRUCTURE ONLY (like nfs_oracle.py: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import anything under oracle/).

Why it exists: the TF-1.15 reference cannot run in this image, so the PyTorch oracle (nfs_oracle.py) is pinned to the
reference only where the reference holds vectors (warp KAT, view sampling, config, util helpers).  For VGG conv /
avg-pool, Gram + style loss, the render with its global max, the legacy resizes and ApplyAdam the oracle has one
author and one implementation; a shared misreading would go unnoticed.  This file restates the same reference lines a
second time, in float64 NumPy with explicit index loops and hand-derived adjoints -- no torch, no autograd, no
library convolution / pooling / cumsum-flip tricks -- so that ``tests/test_oracle_independent.py`` can assert
oracle == restatement (values AND gradients).  Parity stays "unpinned by the reference" for these ops (stated in
DESIGN.md); what this removes is the single-implementation risk.

Every function cites the reference lines it follows.
"""
import numpy as np


# ---- vgg.py:44-48, 89-108: slim.conv2d 3x3 SAME stride 1 + bias + ReLU; slim.avg_pool2d [2,2] (stride 2, VALID) -------
def conv3x3_same_bias_relu(x, w, b, relu=True):
    """x [B,H,W,Ci], w [3,3,Ci,Co] HWIO, b [Co] -> [B,H,W,Co].  SAME for a 3x3 stride-1 window = one zero pixel on
    every side; TF's Conv2D is a cross-correlation: out[y,x] = sum_{r,s} in[y+r-1, x+s-1] * w[r,s]."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64); b = np.asarray(b, np.float64)
    B, H, W, Ci = x.shape
    Co = w.shape[3]
    out = np.zeros((B, H, W, Co))
    for r in range(3):
        for s in range(3):
            for y in range(H):
                yy = y + r - 1
                if yy < 0 or yy >= H:
                    continue
                x_lo, x_hi = max(0, 1 - s), min(W, W + 1 - s)       # output columns whose tap stays inside
                out[:, y, x_lo:x_hi, :] += x[:, yy, x_lo + s - 1:x_hi + s - 1, :] @ w[r, s]
    out += b
    return np.maximum(out, 0.0) if relu else out


def conv3x3_dgrad(g_pre, w, shape_in):
    """gradient wrt the conv INPUT from the gradient wrt its pre-activation output (weights frozen: no wgrad):
    g_in[yy,xx,ci] = sum_{r,s,co} g_pre[yy-r+1, xx-s+1, co] w[r,s,ci,co]"""
    g_pre = np.asarray(g_pre, np.float64); w = np.asarray(w, np.float64)
    B, H, W, Ci = shape_in
    g_in = np.zeros(shape_in)
    for r in range(3):
        for s in range(3):
            for y in range(H):
                yy = y + r - 1
                if yy < 0 or yy >= H:
                    continue
                x_lo, x_hi = max(0, 1 - s), min(W, W + 1 - s)
                g_in[:, yy, x_lo + s - 1:x_hi + s - 1, :] += g_pre[:, y, x_lo:x_hi, :] @ w[r, s].T
    return g_in


def avgpool2_valid(x):
    """[B,H,W,C] -> [B,H//2,W//2,C]: mean of the 2x2 window, stride 2, VALID (an odd last row/column is dropped)"""
    x = np.asarray(x, np.float64)
    B, H, W, C = x.shape
    out = np.zeros((B, H // 2, W // 2, C))
    for y in range(H // 2):
        for xx in range(W // 2):
            out[:, y, xx] = (x[:, 2 * y, 2 * xx] + x[:, 2 * y, 2 * xx + 1] + x[:, 2 * y + 1, 2 * xx]
                             + x[:, 2 * y + 1, 2 * xx + 1]) / 4.0
    return out


def avgpool2_valid_bwd(g, shape_in):
    g = np.asarray(g, np.float64)
    gi = np.zeros(shape_in)
    for y in range(g.shape[1]):
        for xx in range(g.shape[2]):
            for dy in (0, 1):
                for dx in (0, 1):
                    gi[:, 2 * y + dy, 2 * xx + dx] += g[:, y, xx] / 4.0
    return gi


VGG_MEAN = (0.485 * 255, 0.456 * 255, 0.406 * 255)   # vgg.py:18-20 (mean only; the std lines are commented out)
VGG19 = (("conv1", 2), ("conv2", 2), ("conv3", 4), ("conv4", 4), ("conv5", 4))


def vgg19_forward(d_img, weights, upto):
    """d_img [B,H,W,3] 0..255 -> {name: post-ReLU activation}, plus the tape needed by vgg19_backward (vgg.py:50-66)"""
    x = np.asarray(d_img, np.float64) - np.asarray(VGG_MEAN)
    feats, tape = {}, []
    for blk, reps in VGG19:
        for i in range(reps):
            name = "%s_%d" % (blk, i + 1)
            w, b = weights[name]
            tape.append(("conv", name, x.shape))
            x = conv3x3_same_bias_relu(x, w, b)
            feats[name] = x
            if name == upto:
                return feats, tape
        tape.append(("pool", blk, x.shape))
        x = avgpool2_valid(x)
    return feats, tape


def vgg19_backward(feats, tape, weights, g_feats):
    """sum of the given gradients wrt the post-ReLU end points, chained back to d_img"""
    g = None
    for kind, name, shape_in in reversed(tape):
        if kind == "pool":
            g = avgpool2_valid_bwd(g, shape_in)
            continue
        if name in g_feats:
            g = g_feats[name] if g is None else g + g_feats[name]
        g_pre = g * (feats[name] > 0)                               # ReLU
        g = conv3x3_dgrad(g_pre, weights[name][0], shape_in)
    return g


# ---- styler_base.py:96-102, 152-185: Gram matrix and style loss -------------------------------------------------------
def gram(x):
    """x [h,w,C] -> F^T F with F = reshape(x, (hw, C)) (98-100)"""
    f = np.asarray(x, np.float64).reshape(-1, x.shape[-1])
    C = f.shape[1]
    G = np.zeros((C, C))
    for p in range(f.shape[0]):
        G += np.outer(f[p], f[p])
    return G


def style_loss_and_grad(feats, style_feats, layers, w_layers, w_style=1.0):
    """total = w_style * sum_l w_l * sum((G_l/denom - Gs_l/denom_s)^2), denom = 2 h w C (157, 176-183);
    returns (loss, {layer: dL/dfeature [1,h,w,C]}) for ONE image (batch index 0, styler_base.py:98)"""
    total, grads = 0.0, {}
    for name, wl in zip(layers, w_layers):
        x = np.asarray(feats[name], np.float64)[0]
        s = np.asarray(style_feats[name], np.float64)[0]
        h, w, C = x.shape
        hs, ws, _ = s.shape
        G = gram(x) / (2.0 * h * w * C)
        Gs = gram(s) / (2.0 * hs * ws * C)
        Dm = G - Gs
        total += w_style * wl * float((Dm ** 2).sum())
        # d/dF of sum(Dm^2) with G = F^T F / denom:  (2/denom) F (Dm + Dm^T)
        f = x.reshape(-1, C)
        grads[name] = (w_style * wl * (2.0 / (2.0 * h * w * C)) * (f @ (Dm + Dm.T))).reshape(1, h, w, C)
    return total, grads


# ---- styler_3p.py:147-158: transmittance render + global max -----------------------------------------------------------
def render(d, tau, liquid=False):
    """d [B,D,H,W] -> [B,H,W].  T[z] = exp(-tau * sum_{z' >= z} d[z']) (reverse cumsum INCLUDING the own cell, 155),
    I = sum_z d[z] T[z] (156-157), I /= max over the WHOLE tensor (158); liquid: 1 - exp(-tau sum_z d) (150-152)"""
    d = np.asarray(d, np.float64)
    B, D, H, W = d.shape
    img = np.zeros((B, H, W))
    for b in range(B):
        acc = np.zeros((H, W))
        for z in range(D - 1, -1, -1):                               # march from the far end
            acc = acc + d[b, z]
            if not liquid:
                img[b] += d[b, z] * np.exp(-tau * acc)
        if liquid:
            img[b] = 1.0 - np.exp(-tau * acc)
    return img if liquid else img / img.max()


def render_bwd(d, tau, g_norm, liquid=False):
    """adjoint of render: closed form dI/dd[k] = T[k] - tau * sum_{z <= k} d[z] T[z], and for the global max
    m = max(I): d(I/m) = dI/m - I dm/m^2 with dm routed to the arg-max pixels, split equally among ties (TF)."""
    d = np.asarray(d, np.float64); g_norm = np.asarray(g_norm, np.float64)
    B, D, H, W = d.shape
    if liquid:
        s = d.sum(axis=1)
        return (tau * np.exp(-tau * s) * g_norm)[:, None] * np.ones((1, D, 1, 1))
    T = np.zeros_like(d)
    acc = np.zeros((B, H, W))
    for z in range(D - 1, -1, -1):
        acc = acc + d[:, z]
        T[:, z] = np.exp(-tau * acc)
    I = (d * T).sum(axis=1)
    m = I.max()
    ties = (I == m)
    g_I = g_norm / m - ties * ((g_norm * I).sum() / (m * m) / ties.sum())
    g_d = np.zeros_like(d)
    pre = np.zeros((B, H, W))
    for k in range(D):
        pre = pre + d[:, k] * T[:, k]
        g_d[:, k] = (T[:, k] - tau * pre) * g_I
    return g_d


# ---- styler_base.py:35-38, 166: TF-1 legacy image.resize (align_corners=False, no half-pixel centres) -------------------
def tf1_resize_bilinear(x, oh, ow):
    """src = dst * in/out (float32 scale as the TF kernel computes it); lower = floor, upper = min(lower+1, in-1)"""
    x = np.asarray(x, np.float64)
    B, H, W, C = x.shape
    out = np.zeros((B, oh, ow, C))
    sy, sx = np.float32(H) / np.float32(oh), np.float32(W) / np.float32(ow)
    for y in range(oh):
        fy = float(np.float32(y) * sy)
        y0 = int(np.floor(fy)); y1 = min(y0 + 1, H - 1); ly = fy - y0
        for xx in range(ow):
            fx = float(np.float32(xx) * sx)
            x0 = int(np.floor(fx)); x1 = min(x0 + 1, W - 1); lx = fx - x0
            top = x[:, y0, x0] * (1 - lx) + x[:, y0, x1] * lx
            bot = x[:, y1, x0] * (1 - lx) + x[:, y1, x1] * lx
            out[:, y, xx] = top * (1 - ly) + bot * ly
    return out


def _bicubic_weights(frac_index):
    """TF's ResizeBicubic coefficient table: 1024 entries, A = -0.75 (Keys), looked up at round(frac * 1024)"""
    a = -0.75
    t = np.float32(frac_index) / np.float32(1024)
    t = float(t)

    def near(u):      # |u| <= 1
        return ((a + 2) * u - (a + 3)) * u * u + 1

    def far(u):       # 1 < |u| < 2
        return ((a * u - 5 * a) * u + 8 * a) * u - 4 * a
    return [far(t + 1), near(t), near(1 - t), far(2 - t)]


def tf1_resize_bicubic(x, oh, ow):
    """legacy ResizeBicubic: src = dst * in/out, taps floor-1..floor+2 clamped to the image, table weights"""
    x = np.asarray(x, np.float64)
    B, H, W, C = x.shape
    sy, sx = np.float32(H) / np.float32(oh), np.float32(W) / np.float32(ow)
    rows = np.zeros((B, oh, W, C))
    for y in range(oh):
        f = np.float32(y) * sy
        y0 = int(np.floor(f))
        wts = _bicubic_weights(int(np.rint((f - np.float32(y0)) * 1024)))
        for k in range(4):
            rows[:, y] += wts[k] * x[:, min(max(y0 - 1 + k, 0), H - 1)]
    out = np.zeros((B, oh, ow, C))
    for xx in range(ow):
        f = np.float32(xx) * sx
        x0 = int(np.floor(f))
        wts = _bicubic_weights(int(np.rint((f - np.float32(x0)) * 1024)))
        for k in range(4):
            out[:, :, xx] += wts[k] * rows[:, :, min(max(x0 - 1 + k, 0), W - 1)]
    return out


# ---- tf.compat.v1.train.AdamOptimizer (styler_3p.py:320): TF ApplyAdam -------------------------------------------------
def adam_tf_trajectory(x0, grads, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """x_{t+1} = x_t - lr_t m_t / (sqrt(v_t) + eps), lr_t = lr sqrt(1 - b2^t) / (1 - b1^t): epsilon is added to the
    UN-bias-corrected sqrt(v) (the documented TF kernel; torch.optim.Adam divides sqrt(v) by sqrt(1 - b2^t) first)"""
    x = np.asarray(x0, np.float64).copy()
    m = np.zeros_like(x); v = np.zeros_like(x)
    out = []
    for t, g in enumerate(grads, start=1):
        g = np.asarray(g, np.float64)
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
        x = x - lr_t * m / (np.sqrt(v) + eps)
        out.append(x.copy())
    return out


# ---- styler_3p.py:112-125: smoothing conv + tf.maximum --------------------------------------------------------------------
def smooth3d_relu(d, k):
    """kernel [1,k,1] (x) [1,k,1] (x) [1,k,1] / (k+2)^3, conv3d SAME (zero padding), then max(., 0)"""
    d = np.asarray(d, np.float64)
    D, H, W = d.shape
    k1 = np.array([1.0, float(k), 1.0])
    out = np.zeros_like(d)
    pad = np.zeros((D + 2, H + 2, W + 2)); pad[1:-1, 1:-1, 1:-1] = d
    for a in range(3):
        for b in range(3):
            for c in range(3):
                out += k1[a] * k1[b] * k1[c] * pad[a:a + D, b:b + H, c:c + W]
    return np.maximum(out / (k + 2.0) ** 3, 0.0)
