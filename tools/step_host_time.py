"""the headline step (200^3, 8 views): time the host needs to issue one step vs the time the GPU needs to run it, and
the Python-side cost centres of the issue path (cProfile over 100 steps)"""
import os, sys, time, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gs, rot, base = bench.build_problem(200, views, torch.device("cuda:0"), 0, 1)
for _ in range(10):
    gs.step(rot, loss_view=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(100):
    gs.step(rot, loss_view=True)
e1.record(); host = (time.perf_counter() - t0) / 100 * 1e3
torch.cuda.synchronize()
print("%d views: GPU %.3f ms per step, host issue %.3f ms per step" % (views, e0.elapsed_time(e1) / 100, host))
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    gs.step(rot, loss_view=True)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
