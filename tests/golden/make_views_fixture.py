#!/usr/bin/env python
"""Golden vectors for SURVEY row A3' (view sampling) from the REFERENCE ITSELF: its transform.py is imported in the
build container and its NumPy-only helpers are run (rot_[xyz]_3d, PoissonDisc, rot_mat_poisson, rot_mat with
sample_type='poisson').  transform.py does ``import tensorflow as tf`` at module level; an empty module of that name
is registered only to let that statement pass -- none of the functions exercised here touches TensorFlow.
(``rot_mat_uniform`` cannot be run: it passes a float sample count to np.linspace, which modern NumPy rejects;
SURVEY section 8.1 lists that defect.)   Run:  python tests/golden/make_views_fixture.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

sys.modules["tensorflow"] = types.ModuleType("tensorflow")
spec = importlib.util.spec_from_file_location("ref_transform", "/root/reference/transform.py")
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

out = {}
angles = np.array([-37.5, -10.0, -5.0, 0.0, 3.3333333, 10.0, 90.0, 123.0])
out["angles"] = angles
for ax in "xyz":
    out["rot_%s" % ax] = np.stack([np.asarray(getattr(R, "rot_%s_3d" % ax)(a), np.float64) for a in angles])
cases = [(-5, 5, 5, -10, 10, 10, 123, 8), (0, 0, 0, -30, 30, 5, 7, 5), (-20, 20, 2, -45, 45, 3, 1, 16)]
out["cases"] = np.array(cases, np.float64)
for i, (p0, p1, pu, t0, t1, tu, seed, nv) in enumerate(cases):
    views = R.rot_mat_poisson(p0, p1, pu, t0, t1, tu, np.random.RandomState(seed))
    out["poisson_views_%d" % i] = np.array([[v["theta"], v["phi"]] for v in views], np.float64)
    mats, vv = R.rot_mat(p0, p1, pu, t0, t1, tu, sample_type="poisson", rng=np.random.RandomState(seed), nv=nv)
    out["rot_mat_%d" % i] = np.asarray(mats, np.float64)
    out["rot_mat_views_%d" % i] = np.array([[v["theta"], v["phi"]] for v in vv], np.float64)
pd = R.PoissonDisc(np.random.RandomState(5), width=20, height=10, r=2.5)
out["disc_samples"] = np.asarray(pd.sample(), np.float64)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "views_reference.npz")
np.savez_compressed(path, **out)
print(path, {k: v.shape for k, v in out.items()})
