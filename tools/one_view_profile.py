"""the per-rank step of an 8-GPU view-sharded run (200^3, ONE local view) as a loop: run under
``rocprofv3 --kernel-trace --stats`` for the per-kernel table"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
views = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
gs, rot, base = bench.build_problem(200, 8, torch.device("cuda:0"), 0, 1)
gs.use_graph = (os.environ.get("ONE_VIEW_GRAPH", "0") == "1")
rot = rot[:views].contiguous()
for _ in range(n):
    gs.step(rot, loss_view=True)
torch.cuda.synchronize()
