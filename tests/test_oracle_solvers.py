"""Anchors for the oracle's solver restatement (the torchdiffeq boundary is "parity unpinned").

torchdiffeq 0.2.2 is absent and the reference holds no solver fixtures, so the restatement is anchored by
independent facts available offline: scipy's Dormand-Prince tableau and single-step result, analytic ODEs
(convergence orders), tableau identities, the dense-output polynomial's defining conditions.
"""
import math

import numpy as np
import pytest
import torch
from scipy.integrate._ivp.rk import RK45

from oracle import solvers as S


def test_dp_tableau_matches_scipy_rk45():
    np.testing.assert_allclose(np.array(S.DP_ALPHA[:5]), RK45.C[1:6], rtol=0, atol=1e-16)
    for i, row in enumerate(S.DP_BETA[:5]):
        np.testing.assert_allclose(np.array(row), RK45.A[i + 1][: i + 1], rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.array(S.DP_C_SOL[:6]), RK45.B, rtol=0, atol=1e-16)
    # torchdiffeq's error weights = -(2/3) x scipy's E (scipy: E = b5 - b4 with opposite sign convention checked below)
    e = np.array(S.DP_C_ERR)
    assert abs(e.sum()) < 1e-15
    nz = RK45.E != 0
    assert e[~nz].tolist() == [0.0]
    ratio = e[nz] / RK45.E[nz]
    np.testing.assert_allclose(ratio, -2.0 / 3.0, rtol=1e-12)  # proportional: |2/3| x the textbook b5 - b4


def test_dp_tableau_identities():
    for a, row in zip(S.DP_ALPHA, S.DP_BETA):
        assert abs(sum(row) - a) < 1e-15
    assert S.DP_BETA[-1] == S.DP_C_SOL[:6] and S.DP_C_SOL[6] == 0.0  # FSAL
    assert abs(sum(S.DP_C_SOL) - 1) < 1e-15
    assert abs(sum(S.DP_C_MID) - 0.5) < 1e-12  # y_mid is exact for y' = const


def test_dp_single_step_equals_scipy():
    """Same (t0, dt, y0) => same y1 and same (rescaled) error estimate as scipy's RK45 step (fp64)."""
    A = torch.tensor([[-0.5, 2.0, 0.1], [-2.0, -0.3, 0.0], [0.3, 0.0, -1.0]], dtype=torch.float64)

    def f(t, y):
        return torch.tanh(y @ A.t()) + torch.sin(t)

    y0 = torch.tensor([[0.3, -0.7, 1.1]], dtype=torch.float64)
    t0, dt = torch.tensor(0.25, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64)
    tab = (S.DP_ALPHA, tuple(torch.tensor(b, dtype=torch.float64) for b in S.DP_BETA), torch.tensor(S.DP_C_ERR, dtype=torch.float64))
    rhs = S._Rhs(f)
    y1, f1, err, k = S._dp_attempt(rhs, y0, rhs(t0, y0), t0, dt, t0 + dt, tab)
    from scipy.integrate._ivp.rk import rk_step

    def fn(t, y):
        return f(torch.tensor(t, dtype=torch.float64), torch.from_numpy(y)[None])[0].numpy()

    K = np.empty((7, 3))
    y_new, f_new = rk_step(fn, float(t0), y0[0].numpy(), fn(float(t0), y0[0].numpy()), float(dt), RK45.A, RK45.B, RK45.C, K)
    # our alpha==1 stages sit at nextafter(t1, -inf): a 1-ulp time shift => ~1e-16 difference
    np.testing.assert_allclose(y1[0].numpy(), y_new, rtol=1e-13, atol=1e-15)
    scipy_err = K.T @ RK45.E * float(dt)
    np.testing.assert_allclose(err[0].numpy(), scipy_err * (S.DP_C_ERR[0] / RK45.E[0]), rtol=1e-9, atol=1e-16)


@pytest.mark.parametrize("method,order", [("euler", 1), ("midpoint", 2), ("rk4", 4)])
def test_fixed_grid_convergence_order(method, order):
    """y' = -y and the harmonic oscillator in fp64: error ratio under dt halving ~ 2^order."""
    def f(t, y):
        return torch.stack([-y[:, 0], y[:, 2], -y[:, 1]], dim=-1)

    y0 = torch.tensor([[1.0, 0.0, 1.0]], dtype=torch.float64)
    errs = []
    for n in (16, 32, 64):
        t = torch.linspace(0, 1, n + 1, dtype=torch.float64)
        h = S.odeint(f, y0, t, method=method)
        exact = torch.tensor([math.exp(-1), math.sin(1), math.cos(1)], dtype=torch.float64)
        errs.append((h[-1, 0] - exact).abs().max().item())
    for a, b in zip(errs[:-1], errs[1:]):
        assert abs(math.log2(a / b) - order) < 0.25, (errs, order)


def test_rk4_is_three_eighths_rule_not_classic():
    """One step on y' = t^3 ... both rules integrate cubics exactly; distinguish them on y' = y with the stability polynomial
    (identical) -> so use a non-autonomous non-polynomial rhs where the two rules differ at O(dt^5)."""
    def f(t, y):
        return torch.cos(3 * t) * y

    y0 = torch.ones(1, 1, dtype=torch.float64)
    t = torch.tensor([0.0, 0.5], dtype=torch.float64)
    got = S.odeint(f, y0, t, method="rk4")[-1, 0, 0].item()
    dt = 0.5

    def fl(tt, yy):
        return math.cos(3 * tt) * yy

    k1 = fl(0, 1.0)
    k2 = fl(dt / 3, 1 + dt * k1 / 3)
    k3 = fl(2 * dt / 3, 1 + dt * (k2 - k1 / 3))
    k4 = fl(dt, 1 + dt * (k1 - k2 + k3))
    three_eighths = 1 + dt * (k1 + 3 * (k2 + k3) + k4) / 8
    c2 = fl(dt / 2, 1 + dt * k1 / 2)
    c3 = fl(dt / 2, 1 + dt * c2 / 2)
    c4 = fl(dt, 1 + dt * c3)
    classic = 1 + dt * (k1 + 2 * c2 + 2 * c3 + c4) / 6
    assert abs(got - three_eighths) < 1e-15
    assert abs(got - classic) > 1e-6


def test_output_grid_conventions():
    def f(t, y):
        return -y

    y0 = torch.tensor([[2.0]])
    t = torch.tensor([0.0, 0.25, 0.5, 1.0])
    h = S.odeint(f, y0, t, method="rk4")
    assert h.shape == (4, 1, 1) and h[0, 0, 0] == 2.0
    # step_size option: internal grid finer than outputs, outputs linearly interpolated / hit exactly
    h2 = S.odeint(f, y0, t, method="rk4", options={"step_size": 0.125})
    assert abs(h2[-1, 0, 0].item() - 2 * math.exp(-1)) < 5e-6  # RK4 truncation at dt=0.125 is ~2e-6


def test_perturb_uses_one_sided_limits():
    """Heaviside forcing at t=0.5 on the grid: with perturb the step ENDING at 0.5 must not see the jump at all and
    the step STARTING at 0.5 must see it in every stage."""
    seen = []

    def f(t, y):
        seen.append(float(t))
        return (t >= 0.5).to(y.dtype) * torch.ones_like(y)

    y0 = torch.zeros(1, 1)
    t = torch.tensor([0.0, 0.5, 1.0])
    h = S.odeint(f, y0, t, method="rk4", options={"perturb": True})
    assert h[1, 0, 0].item() == 0.0
    assert abs(h[2, 0, 0].item() - 0.5) < 1e-7
    assert seen[0] == float(np.nextafter(np.float32(0.0), np.float32(1.0)))
    assert seen[3] == float(np.nextafter(np.float32(0.5), np.float32(0.0)))
    seen.clear()
    h = S.odeint(f, y0, t, method="rk4")
    assert h[1, 0, 0].item() == pytest.approx(0.5 / 8, abs=1e-7)  # k4 of the first step sees the jump


def test_dense_output_polynomial_conditions():
    torch.manual_seed(0)
    y0, y1, f0, f1, ym = (torch.randn(2, 3, dtype=torch.float64) for _ in range(5))
    dt = torch.tensor(0.37, dtype=torch.float64)
    k = torch.zeros(2, 3, 7, dtype=torch.float64)
    k[..., 0], k[..., -1] = f0, f1
    # choose c_mid so that y_mid == ym:  y_mid = y0 + k @ (dt*c_mid); use a fake c_mid hitting slot 1 with the needed value
    k[..., 1] = (ym - y0) / dt
    c_mid = torch.zeros(7, dtype=torch.float64)
    c_mid[1] = 1.0
    coef = S._interp_fit(y0, y1, k, dt, c_mid)
    t0, t1 = torch.tensor(1.0, dtype=torch.float64), torch.tensor(1.37, dtype=torch.float64)
    np.testing.assert_allclose(S._interp_eval(coef, t0, t1, t0).numpy(), y0.numpy(), atol=1e-14)
    np.testing.assert_allclose(S._interp_eval(coef, t0, t1, t1).numpy(), y1.numpy(), atol=1e-13)
    np.testing.assert_allclose(S._interp_eval(coef, t0, t1, (t0 + t1) / 2).numpy(), ym.numpy(), atol=1e-13)
    # derivative at the ends equals f0 / f1
    e, d, c, b, a = coef
    np.testing.assert_allclose((d / dt).numpy(), f0.numpy(), atol=1e-13)
    np.testing.assert_allclose(((d + 2 * c + 3 * b + 4 * a) / dt).numpy(), f1.numpy(), atol=1e-11)


def test_dopri5_accuracy_and_controller():
    def f(t, y):
        return torch.stack([y[:, 1], -y[:, 0]], dim=-1)

    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
    t = torch.linspace(0, 6, 13, dtype=torch.float64)
    st = {}
    h = S.odeint(f, y0, t, method="dopri5", rtol=1e-9, atol=1e-10, stats=st)
    exact = torch.stack([torch.sin(t), torch.cos(t)], dim=-1)[:, None]
    assert (h - exact).abs().max().item() < 5e-8
    assert st["n_accepted"] > 10 and st["nfe"] == 2 + 6 * (st["n_accepted"] + st["n_rejected"])
    # controller factor bounds
    dt = torch.tensor(0.1, dtype=torch.float64)
    assert S._next_dt(dt, torch.tensor(0.0)).item() == pytest.approx(1.0)
    assert S._next_dt(dt, torch.tensor(1e-12)).item() == pytest.approx(1.0)           # capped at x10
    assert S._next_dt(dt, torch.tensor(0.9)).item() >= 0.1                              # never shrinks after an accepted step
    assert S._next_dt(dt, torch.tensor(1e9)).item() == pytest.approx(0.02)             # floor x0.2
    assert S._next_dt(dt, torch.tensor(2.0)).item() == pytest.approx(0.1 * 0.9 / 2 ** 0.2)


def test_dopri5_fp32_state_fp64_clock_gradients_flow():
    w = torch.tensor([[0.0, 1.0], [-1.0, -0.1]], requires_grad=True)

    def f(t, y):
        return y @ w.t()

    y0 = torch.tensor([[1.0, 0.0]], requires_grad=True)
    t = torch.arange(0, 2.01, 0.25)
    h = S.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8)
    assert h.dtype == torch.float32
    h[-1].sum().backward()
    assert torch.isfinite(w.grad).all() and torch.isfinite(y0.grad).all()
    # compare with finite differences of the exact flow expm(W t) in fp64
    W = w.detach().double()
    def flow(Wm):
        return torch.matrix_exp(Wm * 2.0) @ torch.tensor([1.0, 0.0], dtype=torch.float64)
    g = torch.zeros(2, 2, dtype=torch.float64)
    for i in range(2):
        for j in range(2):
            d = torch.zeros(2, 2, dtype=torch.float64); d[i, j] = 1e-6
            g[i, j] = (flow(W + d).sum() - flow(W - d).sum()) / 2e-6
    np.testing.assert_allclose(w.grad.double().numpy(), g.numpy(), rtol=2e-3, atol=2e-4)


def test_dopri5_tape_replay_is_the_free_running_oracle_on_its_own_tape():
    """`odeint_dopri5_replay` driven along the tape the free-running oracle itself took: identical outputs and identical
    gradients (same ops in the same order), including the derivative of the first step size -- and detaching that
    derivative changes grad_y0 at the 1e-3 .. 1e-2 level on a problem with dose jumps (the size of the term the HIP backward
    has to reproduce, tests/test_hip_dopri5.py)."""
    from hode import synth
    from oracle.rhs import RocheRHS, THETA_NAMES
    from oracle.solvers import odeint, odeint_dopri5_replay
    N, T, D = 8, 12, 8
    inp = synth.solver_inputs(N, T, D, seed=5)
    torch.manual_seed(5)
    f = RocheRHS(D, synth.STEP)
    with torch.no_grad():
        f.ml_net[0].weight.mul_(2.0)
    f.set_action(inp["actions"])
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(3))

    def grads(h, y0):
        f.zero_grad()
        (h * cot).sum().backward()
        return [y0.grad.clone(), f.ml_net[0].weight.grad.clone(), torch.stack([getattr(f, n).grad for n in THETA_NAMES])]

    y0 = inp["z0"].clone().requires_grad_(True)
    st = {}
    h = odeint(f, y0, inp["t"], method="dopri5", rtol=1e-7, atol=1e-8, stats=st)
    g_full = grads(h, y0)
    assert st["first_attempt_accepted"] and st["n_rejected"] > st["n_accepted"] > T
    y0b = inp["z0"].clone().requires_grad_(True)
    hb = odeint_dopri5_replay(f, y0b, inp["t"], 1e-7, 1e-8, st["tape"], True)
    g_rep = grads(hb, y0b)
    assert torch.equal(h.detach(), hb.detach())
    for a, b in zip(g_rep, g_full):
        assert torch.equal(a, b)
    y0c = inp["z0"].clone().requires_grad_(True)
    g_det = grads(odeint_dopri5_replay(f, y0c, inp["t"], 1e-7, 1e-8, st["tape"], False), y0c)
    rel = float((g_det[0] - g_full[0]).norm() / g_full[0].norm())
    assert 1e-3 < rel < 5e-2, rel
