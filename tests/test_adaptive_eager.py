"""`tests/adaptive_eager.py` (dopri5 for the right-hand side without a fused adaptive kernel: NeuralODE) against
the CPU oracle's torchdiffeq-semantics dopri5.  The module is plain torch, so its arithmetic is pinned here on the CPU; the
model classes only hand it HIP tensors (`tests/test_hip_neural.py::test_neural_dopri5_through_the_mirror`)."""
import pytest
import torch

import model
import adaptive_eager as ae
from oracle.rhs import NeuralRHS, RocheRHS
from oracle.solvers import odeint as oracle_odeint


def _actions(T, B, scale=1.0):
    a = torch.zeros(T, B, 1)
    for b in range(B):
        a[(2 * b + 1) % (T - 1), b, 0] = scale * (1.0 + b)
    return a


def _grads(h, cot, y0, f):
    y0.grad = None
    for p in f.parameters():
        p.grad = None
    (h * cot).sum().backward()
    return [y0.grad.clone()] + [None if p.grad is None else p.grad.clone() for p in f.parameters()]


@pytest.mark.parametrize("rhs", ["neural", "roche"])
def test_step_sequence_outputs_and_gradients_match_the_oracle_on_smooth_problems(rhs):
    torch.manual_seed(0)
    D, B, T, step = 6, 5, 12, 0.125
    f = NeuralRHS(D, step) if rhs == "neural" else RocheRHS(D, step)
    f.set_action(_actions(T, B, 1.0 if rhs == "neural" else 1e-30))  # the neural impulse dose only fires on exact stage times
    y0 = (torch.rand(B, D) * 0.1).requires_grad_(True)
    t = torch.arange(T) * step
    cot = torch.randn(T, B, D)
    stats = {}
    h_o = oracle_odeint(f, y0, t, rtol=1e-6, atol=1e-8, method="dopri5", stats=stats)
    g_o = _grads(h_o, cot, y0, f)
    h = ae.odeint_dopri5(f, y0, t, rtol=1e-6, atol=1e-8)
    g = _grads(h, cot, y0, f)
    assert ae.last_stats["n_accepted"] == stats["n_accepted"] > 0
    assert ae.last_stats["n_rejected"] == stats["n_rejected"]
    assert ae.last_stats["nfe"] == stats["nfe"]
    assert (h - h_o).abs().max().item() <= 2e-6  # fp32 tolerance: same ops, the oracle differentiates one big graph
    for a, b in zip(g, g_o):
        if b is None or float(b.abs().max()) == 0.0:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        assert float((a - b).norm() / b.norm()) <= 1e-4


def test_dose_jumps_tape_adjoint_equals_autograd_over_the_same_tape():
    """With dose jumps inside the window the controller sits at ratio ~ 1 and the accept / reject sequence is chaotic in
    the last bit of the error norm (it differs between two CPUs for the ORACLE alone; tests/test_hip_dopri5.py has the
    same finding for the HIP kernels), so the oracle pins the trajectory only to the solver tolerance here.  The adjoint
    algebra is pinned exactly instead: the stepwise reverse sweep against one autograd graph over the same accepted tape
    (first step size detached on both sides), and -- torchdiffeq's graph differentiates the FIRST step size, Hairer's h0
    being a function of y0 and f0 outside `no_grad`, which with a jump in the window moves all later step boundaries
    relative to the jump: 2-5 % of grad_y0 here -- the default path against the oracle's step algebra replayed along the
    same tape with that term (oracle/solvers.py::odeint_dopri5_replay)."""
    import oracle.solvers as osol
    torch.manual_seed(0)
    D, B, T, step = 6, 5, 12, 0.125
    f = RocheRHS(D, step)
    f.set_action(_actions(T, B))
    y0 = (torch.rand(B, D) * 0.1).requires_grad_(True)
    t = torch.arange(T) * step
    cot = torch.randn(T, B, D)
    h = ae.odeint_dopri5(f, y0, t, rtol=1e-6, atol=1e-8, detach_first_step=True)
    steps = h.grad_fn.steps
    g = _grads(h, cot, y0, f)
    assert ae.last_stats["n_rejected"] > ae.last_stats["n_accepted"] > 20  # the regime described above

    tab, tt = ae._Tableau(y0), [float(v) for v in t.double()]
    y, f0, out = y0, f(ae._scalar(tt[0], y0), y0), [y0]
    for (t0, dt, _, _, j0, j1) in steps:
        y1, f1, k, dts = ae._attempt(f, tab, y, f0, t0, dt, t0 + dt)
        coef = ae._dense_coefficients(tab, y, y1, k, dts)
        out += [ae._dense_eval(coef, t0, t0 + dt, tt[j], y) for j in range(j0, j1)]
        y, f0 = y1, f1
    h_graph = torch.stack(out)
    g_graph = _grads(h_graph, cot, y0, f)
    assert torch.equal(h_graph, h.detach())
    for a, b in zip(g, g_graph):
        if b is None or float(b.abs().max()) == 0.0:
            continue
        assert float((a - b).norm() / b.norm()) <= 2e-5

    # the reference's graph: first step size differentiated.  Same tape, the oracle's algebra with that term.
    h_full = ae.odeint_dopri5(f, y0, t, rtol=1e-6, atol=1e-8)
    first = h_full.grad_fn.first_accepted
    g_full = _grads(h_full, cot, y0, f)
    rs = {}
    h_rep = osol.odeint_dopri5_replay(f, y0, t, 1e-6, 1e-8, [(s[0], s[1]) for s in steps], first, stats=rs)
    g_rep = _grads(h_rep, cot, y0, f)
    assert first and (h_full.detach() - h_rep.detach()).abs().max().item() <= 2e-6
    assert abs(ae.last_stats["sigma"] - float(rs["dt0"].grad)) <= 2e-3 * abs(float(rs["dt0"].grad))
    for a, b in zip(g_full, g_rep):
        if b is None or float(b.abs().max()) == 0.0:
            continue
        assert float((a - b).norm() / b.norm()) <= 1e-4
    assert float((g[0] - g_rep[0]).norm() / g_rep[0].norm()) >= 1e-3  # the term is material on this problem

    stats = {}
    h_o = osol.odeint(f, y0, t, rtol=1e-6, atol=1e-8, method="dopri5", stats=stats)
    assert (h.detach() - h_o.detach()).abs().max().item() <= 1e-4
    assert abs(stats["n_accepted"] - ae.last_stats["n_accepted"]) <= 0.25 * stats["n_accepted"]
    g_raw = _grads(h_o, cot, y0, f)[0]
    assert float((g[0] - g_raw).norm() / g_raw.norm()) < 0.2


def test_output_grid_coarser_and_finer_than_the_steps_and_no_grad_inputs():
    torch.manual_seed(1)
    D, B, step = 4, 3, 0.5
    f = NeuralRHS(D, step)
    for T in (2, 40):
        f.set_action(_actions(T, B))
        t = torch.arange(T) * (step if T == 2 else 0.01)
        y0 = torch.rand(B, D) * 0.1  # requires no grad: only the parameters are differentiated
        h = ae.odeint_dopri5(f, y0, t, rtol=1e-5, atol=1e-7)
        h_o = oracle_odeint(f, y0, t, rtol=1e-5, atol=1e-7, method="dopri5")
        assert h.shape == (T, B, D) and torch.equal(h[0], y0)
        assert (h - h_o).abs().max().item() <= 2e-6
        h.sum().backward()
        assert f.ml_net[0].weight.grad is not None and torch.isfinite(f.ml_net[0].weight.grad).all()


def test_failures_are_runtime_errors():
    class Blowup(torch.nn.Module):
        def forward(self, t, y):
            return y * y * 1e6

    with pytest.raises(RuntimeError):
        ae.odeint_dopri5(Blowup(), torch.ones(2, 3) * 10, torch.arange(5) * 1.0, rtol=1e-7, atol=1e-9)


def test_mirror_refuses_cpu_tensors_for_the_eager_path():
    dec = model.RocheExpertDecoder(6, 6, 1, 1.0, 0.125, roche=False, method="dopri5", device=torch.device("cpu"))
    import hode
    with pytest.raises(hode.HodeConfigError, match="HIP device"):
        dec(torch.rand(3, 6) * 0.1, _actions(9, 3))
