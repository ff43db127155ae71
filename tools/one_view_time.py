"""wall time of the per-rank step of an 8-GPU view-sharded run (ONE local view, field work replicated), hipGraph replay or
eager (ONE_VIEW_GRAPH=1/0): python tools/one_view_time.py [views=1]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
views = int(sys.argv[1]) if len(sys.argv) > 1 else 1
gs, rot, base = bench.build_problem(200, 8, torch.device("cuda:0"), 0, 1)
gs.use_graph = (os.environ.get("ONE_VIEW_GRAPH", "1") == "1")
rot = rot[:views].contiguous()
for _ in range(20):
    gs.step(rot, loss_view=True)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(100):
        gs.step(rot, loss_view=True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 100)
print("views %d graph %s: %.4f ms/step (best of 5 x 100)" % (views, gs.use_graph, 1e3 * best))
