"""Is the step host-bound?  Time N steps with and without waiting for the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

for V, graph in ((1, False), (2, False), (4, False), (8, False), (1, True), (8, True)):
    gs, rot, data = bench.build_problem(200, V, torch.device("cuda", 0), 0, 1)
    gs.use_graph = graph
    for _ in range(3):
        gs.step(rot, loss_view=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gs.step(rot, loss_view=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("graph", graph, "views %d: host issue %.3f ms/step, total %.3f ms/step" % (V, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t0) / 20))
