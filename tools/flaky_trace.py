"""Single-rank runs of the frame-sharded test's sequence with every optimiser step recorded (gradient, variable before /
after): runs are grouped by their final variables; the first two distinct groups are compared call by call."""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_drivers_gpu as T
HOOK = r'''
from neural_flow_style_amd import engine as _E
_rec = []
_orig = _E.TFAdamState.step
def _step(self, var, g, lr, *a, **k):
    before = var.detach().clone()
    out = _orig(self, var, g, lr, *a, **k)
    _rec.append((g.detach().cpu().numpy().copy(), before.cpu().numpy(), var.detach().cpu().numpy().copy()))
    return out
_E.TFAdamState.step = _step
'''
src = T._FRAMES_SCRIPT % {"root": ROOT, "mode": "sum"}
src = src.replace("st = Styler(cfg)", HOOK + "st = Styler(cfg)")
src = src.replace("if world > 1:\n    dist.barrier()", "np.savez(sys.argv[1] + '.trace.npz', g=np.stack([r[0] for r in _rec]), b=np.stack([r[1] for r in _rec]), a=np.stack([r[2] for r in _rec]))\nif world > 1:\n    dist.barrier()")
tmp = tempfile.mkdtemp()
script = os.path.join(tmp, "rank.py")
open(script, "w").write(src)
env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    env.pop(k, None)
groups = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    out = os.path.join(tmp, "r%d.npz" % i)
    subprocess.run([sys.executable, script, out], check=True, env=env, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    r, tr = np.load(out), np.load(out + ".trace.npz")
    for gr in groups:
        if np.linalg.norm(r["opt"] - gr[0]["opt"]) <= 1e-5 * np.linalg.norm(r["opt"]):
            gr[2].append(i)
            break
    else:
        groups.append((dict(opt=r["opt"]), dict(g=tr["g"], b=tr["b"], a=tr["a"]), [i]))
print("groups:", [g[2] for g in groups])
if len(groups) > 1:
    A, B = groups[0][1], groups[1][1]
    for c in range(A["g"].shape[0]):
        dg = np.abs(A["g"][c] - B["g"][c]).max() / max(np.abs(A["g"][c]).max(), 1e-30)
        db = np.abs(A["b"][c] - B["b"][c]).max()
        da = np.abs(A["a"][c] - B["a"][c]).max()
        flag = "  <--" if (dg > 1e-4 or da > 1e-6) else ""
        print("call %3d: grad rel-max diff %.2e   var before %.2e  after %.2e   |g|max %.3e%s" % (c, dg, db, da, np.abs(A["g"][c]).max(), flag))

    # the first call whose gradient differs: which particles, and where they sit (a ReLU flip of one activation reaches
    # the particles along the rays through a few pixels of ONE view: a thin tube through the volume)
    sys.path.insert(0, ROOT)
    from neural_flow_style_amd import synthetic as S
    rng = np.random.RandomState(17)
    p0 = S.blob_particles(900, rng)
    drift = rng.randn(900, 3).astype(np.float32) * 0.004
    for c in range(A["g"].shape[0]):
        d = np.abs(A["g"][c] - B["g"][c]).max(axis=1)
        if d.max() > 1e-5 * np.abs(A["g"][c]).max():
            idx = np.nonzero(d > 1e-6 * np.abs(A["g"][c]).max())[0]
            t = c % 5
            pos = np.clip(p0 + drift * t, 0.05, 0.95)[idx] + A["b"][c][idx]
            q = pos - pos.mean(0)
            sv = np.linalg.svd(q, compute_uv=False) / np.sqrt(len(idx))
            print("call %d (frame %d): %d particles differ; principal std-devs of their positions (cells): %s; all particles: %s"
                  % (c, t, len(idx), np.round(sv * 16, 2), np.round(np.linalg.svd(p0 - p0.mean(0), compute_uv=False) / 30 * 16, 2)))
            gA, gB = A["g"][c].astype(np.float64), B["g"][c].astype(np.float64)
            dd = gA - gB
            print("  |dg| / |g| = %.3e; cos(dg, g) = %.4f; best scalar fit dg ~ s g: s = %.3e, residual after fit / |dg| = %.3f"
                  % (np.linalg.norm(dd) / np.linalg.norm(gA), (dd * gA).sum() / np.linalg.norm(dd) / np.linalg.norm(gA),
                     (dd * gA).sum() / (gA * gA).sum(), np.linalg.norm(dd - (dd * gA).sum() / (gA * gA).sum() * gA) / np.linalg.norm(dd)))
            order = np.argsort(-d)[:8]
            print("  largest |dg| particles:", [(int(i), "%.2e" % d[i], "%.2e" % np.abs(gA[i]).max()) for i in order])
            print("  quantiles of per-particle |dg|max / |g|max(all): ", ["%.1e" % v for v in np.quantile(d / np.abs(gA).max(), [0.1, 0.5, 0.9, 0.99, 1.0])])
            gA, gB = A["g"][c].astype(np.float64), B["g"][c].astype(np.float64)
            dd = gA - gB
            print("  |dg| / |g| = %.3e; cos(dg, g) = %.4f; best scalar fit dg ~ s g: s = %.3e, residual after fit / |dg| = %.3f"
                  % (np.linalg.norm(dd) / np.linalg.norm(gA), (dd * gA).sum() / np.linalg.norm(dd) / np.linalg.norm(gA),
                     (dd * gA).sum() / (gA * gA).sum(), np.linalg.norm(dd - (dd * gA).sum() / (gA * gA).sum() * gA) / np.linalg.norm(dd)))
            order = np.argsort(-d)[:8]
            print("  largest |dg| particles:", [(int(i), "%.2e" % d[i], "%.2e" % np.abs(gA[i]).max()) for i in order])
            print("  quantiles of per-particle |dg|max / |g|max(all): ", ["%.1e" % v for v in np.quantile(d / np.abs(gA).max(), [0.1, 0.5, 0.9, 0.99, 1.0])])
            break
