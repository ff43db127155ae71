"""The outer-loop optimisers as host logic (no device needed: elementwise torch ops on whatever device the variable is on):
TF ApplyAdam's update rule is pinned by tests/test_oracle_kat.py; L-BFGS (north_star's other outer loop) is held here to
``torch.optim.LBFGS`` called once per iteration."""
import numpy as np
import torch


def _rosen(x):
    return (100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2).sum()


def test_lbfgs_state_follows_torch_lbfgs_one_iteration_per_call():
    from neural_flow_style_amd.engine import LBFGSState, make_optimizer, TFAdamState
    assert isinstance(make_optimizer("adam"), TFAdamState) and isinstance(make_optimizer("lbfgs"), LBFGSState)
    rng = np.random.RandomState(0)
    x0 = torch.tensor(rng.uniform(-1.0, 1.0, 12), dtype=torch.float64)
    A = torch.tensor(rng.randn(12, 12), dtype=torch.float64)
    A = A @ A.t() + 0.5 * torch.eye(12, dtype=torch.float64)
    for fn, lr, steps in ((_rosen, 0.5, 25), (lambda x: 0.5 * x @ A @ x + x.sum(), 1.0, 25)):
        ref = x0.clone().requires_grad_()
        opt = torch.optim.LBFGS([ref], lr=lr, max_iter=1, history_size=10, line_search_fn=None,
                                tolerance_grad=0.0, tolerance_change=0.0)

        def closure():
            opt.zero_grad()
            l = fn(ref)
            l.backward()
            return l
        mine = x0.clone().reshape(3, 4)                     # (any shape: the state works on the flattened variable)
        st = LBFGSState(history=10)
        for k in range(steps):
            opt.step(closure)
            xv = mine.reshape(-1).clone().requires_grad_()
            (g,) = torch.autograd.grad(fn(xv), xv)
            st.step(mine, g.reshape(3, 4), lr)
            np.testing.assert_allclose(mine.reshape(-1).numpy(), ref.detach().numpy(), rtol=1e-9, atol=1e-12,
                                       err_msg="step %d" % k)
        assert float(fn(mine.reshape(-1))) < float(fn(x0))


def test_lbfgs_history_is_bounded_and_skips_non_positive_curvature():
    from neural_flow_style_amd.engine import LBFGSState
    st = LBFGSState(history=3)
    x = torch.zeros(5)
    for k in range(8):
        g = torch.full((5,), 1.0 + k)                      # gradients GROW along the step direction: y.s < 0 every time
        st.step(x, g, 0.1)
    assert len(st.S) == 0                                  # no pair with y.s <= 1e-10 is kept
    st = LBFGSState(history=3)
    x = torch.ones(5)
    for k in range(8):
        st.step(x, 2.0 * x.clone(), 0.3)                   # f = |x|^2: positive curvature
    assert len(st.S) == 3 and len(st.Y) == 3 and len(st.ro) == 3


def test_lbfgs_guards_and_one_state_per_variable():
    """a zero gradient on the first step (an empty or masked frame) must not divide by zero; a state must
    not be fed another variable; frames of one Adam group get their own L-BFGS states"""
    from neural_flow_style_amd.engine import LBFGSState, optimizer_slot
    st = LBFGSState()
    x = torch.ones(6)
    st.step(x, torch.zeros(6), 0.5)                         # nothing moves, nothing is recorded
    assert torch.equal(x, torch.ones(6)) and st.n == 0 and st.prev_g is None
    st.step(x, 2.0 * x.clone(), 0.3)
    assert st.n == 1 and not torch.equal(x, torch.ones(6))
    import pytest
    with pytest.raises(ValueError, match="one state per variable"):
        st.step(torch.ones(4), torch.ones(4), 0.3)
    # a direction that is not a descent direction moves nothing (torch.optim.LBFGS: gtd > -tolerance_change)
    st = LBFGSState()
    x = torch.ones(3)
    st.step(x, torch.tensor([1.0, 0.0, 0.0]), 0.1)
    st.H, st.S, st.Y, st.ro = -1.0, [], [], []              # force an ascent direction through a negative Hessian scale
    before = x.clone()
    st.prev_g = torch.tensor([1.0, 0.0, 0.0]); st.d = torch.zeros(3); st.t = 0.0
    st.step(x, torch.tensor([1.0, 0.0, 0.0]), 0.1)
    assert torch.equal(x, before)
    assert optimizer_slot("adam", 7, 5) == 1 and optimizer_slot(None, 7, 5) == 1
    assert optimizer_slot("lbfgs", 7, 5) != optimizer_slot("lbfgs", 8, 5)
