"""The reference drivers' own configurations at FULL scale on synthetic data (demo mode), a few iterations each: wall time
per iteration of Styler.run as a user of test_dambreak2d.py / test_chocolate.py / test_smokegun.py sees it (set-up and the
final inference excluded by differencing two run lengths)."""
import os, sys, time, tempfile, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from config import get_config

def run(drvname, argv, iters):
    import importlib
    drv = importlib.import_module(drvname)
    out = []
    for it in (iters[0],) + tuple(iters):                  # the first run builds the lazy state (filters, tuner, graphs): discarded
        av = argv + ["--iter", str(it)]
        sys.argv = [drvname + ".py"] + av
        cfg, _ = get_config()
        tmp = tempfile.mkdtemp()
        cfg.log_dir, cfg.data_dir = os.path.join(tmp, "log"), os.path.join(tmp, "nodata")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = drv.main(cfg)
        torch.cuda.synchronize()
        out.append((it, time.perf_counter() - t0, sum(len(l) for l in res["l"])))
    (i0, t0_, n0), (i1, t1_, n1) = out[1:]
    print("%-18s %-70s %.2f s for %d loss evaluations, %.2f s for %d  ->  %.2f ms per evaluation"
          % (drvname, " ".join(argv)[:70], t0_, n0, t1_, n1, 1e3 * (t1_ - t0_) / max(n1 - n0, 1)), flush=True)

which = sys.argv[1:] or ["dambreak", "dambreak_hist", "chocolate", "smokegun", "smokegun_vgg"]
if "dambreak" in which:
    run("test_dambreak2d", ["--num_frames", "1", "--target_frame", "150"], (10, 30))
if "dambreak_hist" in which:
    run("test_dambreak2d", ["--num_frames", "1", "--target_frame", "150", "--w_hist", "1"], (10, 30))
if "dambreak_batch" in which:
    run("test_dambreak2d", ["--num_frames", "4", "--batch_size", "4", "--target_frame", "150"], (10, 30))
if "chocolate" in which:
    run("test_chocolate", ["--num_frames", "1", "--target_frame", "70", "--w_style", "1", "--w_content", "0"], (10, 30))
if "smokegun" in which:
    run("test_smokegun", ["--num_frames", "1", "--target_frame", "70", "--synthetic_weights", "true"], (10, 30))
if "smokegun_vgg" in which:
    run("test_smokegun", ["--num_frames", "1", "--target_frame", "70", "--network", "vgg_19.ckpt", "--rotate", "true",
                          "--n_views", "8", "--w_style", "1", "--synthetic_weights", "true"], (10, 30))
