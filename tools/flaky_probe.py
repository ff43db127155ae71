"""How far apart are two runs of the frame-sharded particle sequence test (tests/test_drivers_gpu.py)?  Prints the four
quantities the test bounds, for one-rank vs one-rank and one-rank vs two-rank pairs."""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_drivers_gpu as T
mode = sys.argv[1] if len(sys.argv) > 1 else "sum"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tmp = tempfile.mkdtemp()
script = os.path.join(tmp, "rank.py")
open(script, "w").write(T._FRAMES_SCRIPT % {"root": ROOT, "mode": mode})
env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    env.pop(k, None)
def run(world, out, port):
    if world == 1:
        subprocess.run([sys.executable, script, out], check=True, env=env, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    else:
        subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), script, out], check=True, env=env,
                       stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    return np.load(out)
a = run(1, os.path.join(tmp, "a.npz"), 0)
def report(tag, b):
    l = np.max(np.abs(b["l"] - a["l"]) / np.abs(a["l"]))
    o = np.linalg.norm(b["opt"] - a["opt"]) / np.linalg.norm(a["opt"])
    p = np.max(np.abs(b["p"] - a["p"]))
    d = np.max(np.abs(b["d"] - a["d"]) - 1e-4 * np.abs(a["d"]))
    print("%s: loss rel %.2e (bar 2e-5)  opt relL2 %.2e (bar 1e-4)  p abs %.2e (bar 2e-5)  d excess %.2e (bar 1e-6)" % (tag, l, o, p, d), flush=True)
    if o > 1e-5:
        e = np.abs(b["opt"] - a["opt"]).max(axis=2)              # [frames, particles]
        print("    per frame max |d opt|:", ["%.1e" % v for v in e.max(axis=1)], " particles above 1e-6 per frame:", (e > 1e-6).sum(axis=1),
              " loss rel per entry:", ["%.1e" % v for v in (np.abs(b["l"] - a["l"]) / np.abs(a["l"])).ravel()], flush=True)
for i in range(reps):
    report("one vs one ", run(1, os.path.join(tmp, "b.npz"), 0))
    report("two vs one ", run(2, os.path.join(tmp, "c.npz"), 29760 + i))
