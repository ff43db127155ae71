import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from neural_flow_style_amd import synthetic as S, transform as T
p = torch.tensor(S.blob_particles(500000, np.random.RandomState(0)), device="cuda")
for st in (True, False):
    T.grid_order(p, [200]*3, stable=st); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(20): o=T.grid_order(p, [200]*3, stable=st)
    torch.cuda.synchronize(); print("stable", st, "grid_order ms", (time.perf_counter()-t0)/20*1e3)
o = T.grid_order(p, [200] * 3)
x = torch.rand(1, 500000, 3, device="cuda", requires_grad=True)
w = torch.rand(1, 500000, 3, device="cuda")
for name, f in (("permute_particles", lambda: T.permute_particles(x, o)), ("x[:, order]", lambda: x[:, o])):
    y = f(); y.backward(w); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        x.grad = None; f().backward(w)
    torch.cuda.synchronize(); print(name, "gather + adjoint ms", (time.perf_counter() - t0) / 20 * 1e3)
