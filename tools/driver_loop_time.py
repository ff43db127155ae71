import contextlib, importlib, io, os, sys, tempfile, time
root = sys.argv[1]
sys.path.insert(0, root); os.chdir(root)
import torch
from config import get_config
drv = importlib.import_module("test_smokegun")
runs = []
for it in (4, 4, 28, 4, 28):
    sys.argv = ["test_smokegun.py", "--num_frames", "1", "--target_frame", "70", "--network", "vgg_19.ckpt", "--rotate", "true",
                "--n_views", "8", "--w_style", "1", "--synthetic_weights", "true", "--iter", str(it)]
    c3, _ = get_config()
    tmp = tempfile.mkdtemp()
    c3.log_dir, c3.data_dir = os.path.join(tmp, "log"), os.path.join(tmp, "nodata")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        res = drv.main(c3)
    torch.cuda.synchronize()
    runs.append((it, time.perf_counter() - t0))
ta = min(t for i, t in runs[1:] if i == 4); tb = min(t for i, t in runs[1:] if i == 28)
print(root, "ms/iter %.3f" % (1e3 * (tb - ta) / 24), runs)
