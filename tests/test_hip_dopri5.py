"""HIP dopri5 (through the C ABI) vs the CPU oracle's torchdiffeq-semantics dopri5.  GPU only.

Two kinds of comparison.  (1) Against the FREE-RUNNING oracle (its own controller): step counts, trajectories, and
gradients at the level the chaos of the accept / reject decisions allows (below).  (2) Against the oracle's step algebra
REPLAYED along the HIP run's own (t_n, dt_n) tape (oracle/solvers.py::odeint_dopri5_replay): the controller drops out and
the adjoint -- including the derivative of Hairer's first step size, which torchdiffeq's graph contains -- is pinned to
the rounding level, dose jumps and all (test_dopri5_gradients_follow_the_tape_replay_oracle).

The controller is batch-global and discrete (accept/reject).  On a SMOOTH problem (no dose) the HIP path reproduces
the oracle's step sequence (equal accepted/rejected counts) and gradients agree to ~5e-7 -- that pins the adjoint
algebra.  With dose jumps the controller sits at ratio ~ 1 for hundreds of attempts (2/3 of them rejected), last-bit
differences of the error norm flip individual decisions, and the two step sequences drift apart (164 vs 174 accepted
steps measured).  Both are valid dopri5 runs of the same ODE: trajectories agree to 4e-5 absolute, gradients through
the discontinuity to 1e-3 .. 2e-2 -- the same spread the CPU oracle shows against itself when only the encoder's
summation order changes (tests/test_oracle_golden.py, G5).  Tolerances below are set to those measured levels.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.rhs import RocheRHS, THETA_NAMES, dose_schedule, THETA_DEFAULT
from oracle.solvers import odeint as oracle_odeint


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _setup(N, T, D, seed, ablate=False, n_dose=1):
    from hode import synth
    inp = synth.solver_inputs(N, T, D, seed=seed, n_dose=n_dose)
    torch.manual_seed(seed)
    f = RocheRHS(D, synth.STEP, ablate=ablate)
    if D > 4:
        with torch.no_grad():
            f.ml_net[0].weight.mul_(2.0)
    return inp, f


def _hip(inp, f, dev, lanes, rtol, atol, cot=None, detach_first_step=False):
    from hode import adaptive
    from hode.solver import pack_theta
    names = list(THETA_NAMES) + (["theta_1", "theta_2"] if f.ablate else [])
    scal = [getattr(f, n).detach().clone().to(dev).requires_grad_(True) for n in names]
    y0 = inp["z0"].to(dev).requires_grad_(True)
    w = b = None
    if f.ml_dim > 0:
        w = f.ml_net[0].weight.detach().clone().to(dev).requires_grad_(True)
        b = f.ml_net[0].bias.detach().clone().to(dev).requires_grad_(True)
    dosage, times = dose_schedule(inp["actions"], f.step_size)
    h = adaptive.roche_dopri5(y0, pack_theta(scal, dev), w, b, inp["t"].to(dev), dosage.to(dev), times.to(dev), rtol=rtol,
                              atol=atol, ablate=f.ablate, lanes_per_patient=lanes, detach_first_step=detach_first_step)
    out = {"h": h.detach().cpu(), "stats": dict(adaptive.last_stats)}
    if cot is not None:
        (h * cot.to(dev)).sum().backward()
        out["gy0"] = y0.grad.cpu()
        if w is not None:
            out["gw"], out["gb"] = w.grad.cpu(), b.grad.cpu()
        out["gtheta"] = torch.stack([s.grad for s in scal]).cpu()
    return out


def _replay(inp, f, rtol, atol, cot, tape, first_accepted, double=False):
    """Gradients of the oracle's accepted-step algebra driven along ``tape`` (the HIP run's), in fp32 or fp64."""
    import copy
    from oracle.solvers import odeint_dopri5_replay
    f.set_action(inp["actions"])
    fm, y0, tt, ct = f, inp["z0"].clone(), inp["t"], cot
    if double:
        fm = copy.deepcopy(f).double()
        fm.dosage, fm.times = f.dosage.double(), f.times.double()
        y0, tt, ct = y0.double(), tt.double(), cot.double()
    y0.requires_grad_(True)
    fm.zero_grad()
    st = {}
    h = odeint_dopri5_replay(fm, y0, tt, rtol, atol, list(zip(tape["t"], tape["dt"])), first_accepted, stats=st)
    (h * ct).sum().backward()
    names = list(THETA_NAMES) + (["theta_1", "theta_2"] if f.ablate else [])
    zero = torch.zeros((), dtype=y0.dtype)
    out = {"h": h.detach(), "gy0": y0.grad,
           "gtheta": torch.stack([getattr(fm, n).grad if getattr(fm, n).grad is not None else zero for n in names]),
           "sigma": float(st["dt0"].grad) if "dt0" in st else 0.0}
    if f.ml_dim > 0:
        out["gw"], out["gb"] = fm.ml_net[0].weight.grad, fm.ml_net[0].bias.grad
    return out


def _oracle(inp, f, rtol, atol, cot=None):
    f.set_action(inp["actions"])
    y0 = inp["z0"].clone().requires_grad_(True)
    f.zero_grad()
    st = {}
    h = oracle_odeint(f, y0, inp["t"], method="dopri5", rtol=rtol, atol=atol, stats=st)
    out = {"h": h.detach(), "stats": st}
    if cot is not None:
        (h * cot).sum().backward()
        out["gy0"] = y0.grad
        if f.ml_dim > 0:
            out["gw"], out["gb"] = f.ml_net[0].weight.grad, f.ml_net[0].bias.grad
        names = list(THETA_NAMES) + (["theta_1", "theta_2"] if f.ablate else [])
        out["gtheta"] = torch.stack([getattr(f, n).grad if getattr(f, n).grad is not None else torch.zeros(()) for n in names])
    return out


@pytest.mark.parametrize("D,lanes", [(12, 4), (12, 1), (8, 4), (4, 1), (6, 1)])
def test_dopri5_forward_backward_vs_oracle(D, lanes):
    dev = _dev()
    N, T = 21, 20
    rtol, atol = 1e-7, 1e-8  # the reference's values (model.py:1079-1080)
    inp, f = _setup(N, T, D, seed=40 + D)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(3))
    hip, ora = _hip(inp, f, dev, lanes, rtol, atol, cot), _oracle(inp, f, rtol, atol, cot)
    na, no = hip["stats"]["n_accepted"], ora["stats"]["n_accepted"]
    assert abs(na - no) <= 0.15 * no, (hip["stats"], no, ora["stats"]["n_rejected"])
    scale = 1 + ora["h"].abs().max().item()
    assert torch.equal(hip["h"][0], ora["h"][0])
    assert (hip["h"] - ora["h"]).abs().max().item() <= 1e-4 * scale
    assert torch.mean((hip["h"] - ora["h"]) ** 2).item() <= 1e-9 * scale ** 2  # BASELINE target: 1e-5
    # gradients against the free-running oracle carry the step-sequence chaos (different accept / reject decisions at
    # ratio ~ 1 move every later step boundary relative to the dose jumps): 1e-3 .. 2e-2 measured, the spread the oracle
    # shows against itself.  The tight gradient check is the tape-replay test below.
    for k in ("gy0", "gw", "gb", "gtheta"):
        if k in ora:
            assert _rel(hip[k], ora[k]) <= 4e-2, (k, _rel(hip[k], ora[k]))


@pytest.mark.parametrize("D,lanes,N,T,ablate", [(12, 4, 21, 20, False), (12, 1, 21, 20, False), (8, 4, 21, 20, False),
                                                  (4, 1, 21, 20, False), (6, 1, 21, 20, False), (12, 0, 300, 30, False),
                                                  (8, 4, 23, 14, True)])
def test_dopri5_gradients_follow_the_tape_replay_oracle(D, lanes, N, T, ablate):
    """With dose jumps, rtol 1e-7 (the reference's): the oracle's step algebra is driven along the HIP run's own tape, so
    both sides differentiate the SAME step sequence.

    (a) first step size detached on both sides: every gradient to 1e-5 (5e-7 measured) -- the adjoint algebra of the
        accepted steps, through the discontinuities.
    (b) the reference's graph (dt_0 = Hairer's initial step differentiated when attempt 0 is the accepted one): the
        dt_0 term is material (1e-3 .. 1e-2 of grad_y0) and sigma = d loss / d dt_0 is a cancellation-heavy fp32 sum --
        the oracle's own fp32 evaluation sits 1e-5 .. 7e-4 from the fp64 evaluation of the same graph on the same tape.
        The HIP gradients are held to 1e-4 against the fp64 evaluation, or to twice the fp32 oracle's own distance from it
        where that is larger, and must recover the term (be much closer to the full graph than the detached one is).
    """
    from hode import adaptive
    dev = _dev()
    rtol, atol = 1e-7, 1e-8
    inp, f = _setup(N, T, D, seed=40 + D, ablate=ablate)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(3))
    adaptive.keep_workspace = True
    try:
        hip = _hip(inp, f, dev, lanes, rtol, atol, cot)
        tape = adaptive.read_tape()
        hip_det = _hip(inp, f, dev, lanes, rtol, atol, cot, detach_first_step=True)
    finally:
        adaptive.keep_workspace = False
    assert len(tape["t"]) == hip["stats"]["n_accepted"] and tape["t"][0] == 0.0
    first = bool(tape["init"]["first_accepted"])
    keys = [k for k in ("gy0", "gw", "gb", "gtheta") if k in hip]
    # (a)
    det32 = _replay(inp, f, rtol, atol, cot, tape, False)
    scale = 1 + det32["h"].abs().max().item()
    assert (hip_det["h"] - det32["h"]).abs().max().item() <= 2e-5 * scale
    for k in keys:
        assert _rel(hip_det[k], det32[k]) <= 1e-5, ("detached", k, _rel(hip_det[k], det32[k]))
    # (b)
    full32 = _replay(inp, f, rtol, atol, cot, tape, first)
    full64 = _replay(inp, f, rtol, atol, cot, tape, first, double=True)
    for k in keys:
        noise = _rel(full32[k], full64[k])
        err = _rel(hip[k], full64[k])
        assert err <= max(1e-4, 2.0 * noise), ("full graph", k, err, noise)
    if first and not ablate:
        effect = _rel(det32["gy0"], full64["gy0"])
        assert effect >= 5e-4, effect  # the problem exercises the term
        assert _rel(hip["gy0"], full64["gy0"]) <= 0.25 * effect
        sig64 = full64["sigma"]
        assert abs(tape["init"]["sigma"] - sig64) <= max(2.0 * abs(full32["sigma"] - sig64), 2e-3 * abs(sig64))


def test_dopri5_smooth_problem_tight_gradients():
    """No dose (smooth rhs): gradients agree tightly, which isolates the adjoint algebra from discontinuity noise."""
    dev = _dev()
    N, T, D = 18, 12, 12
    inp, f = _setup(N, T, D, seed=77)
    inp["actions"].zero_()
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(4))
    for lanes in (4, 1):
        hip, ora = _hip(inp, f, dev, lanes, 1e-6, 1e-8, cot), _oracle(inp, f, 1e-6, 1e-8, cot)
        # step sizes differ in the last bits (error-norm summation order), so the run may need one step more or less to
        # pass the final output time; everything else must agree to the solver tolerance
        assert abs(hip["stats"]["n_accepted"] - ora["stats"]["n_accepted"]) <= 1 and hip["stats"]["n_rejected"] <= ora["stats"]["n_rejected"] + 1
        assert (hip["h"] - ora["h"]).abs().max().item() <= 1e-5 * (1 + ora["h"].abs().max().item())
        for k in ("gy0", "gw", "gb", "gtheta"):
            assert _rel(hip[k], ora[k]) <= 1e-4, (k, _rel(hip[k], ora[k]))


def test_dopri5_matches_fine_rk4_and_tape_grows():
    """Property at a larger batch: the adaptive solution agrees with a much finer fixed-grid solution away from the
    dose jumps' first-order error, and the tape retry path works (tiny initial tape)."""
    from hode import adaptive
    dev = _dev()
    N, T, D = 300, 30, 12
    inp, f = _setup(N, T, D, seed=5)
    hip = _hip(inp, f, dev, 0, 1e-7, 1e-8)
    assert torch.isfinite(hip["h"]).all() and hip["stats"]["n_accepted"] >= T - 1
    ora = _oracle({"z0": inp["z0"][:8], "actions": inp["actions"][:, :8], "t": inp["t"]}, f, 1e-7, 1e-8)
    # different batch => different global controller, same ODE: solutions agree to the tolerance scale
    assert (hip["h"][:, :8] - ora["h"]).abs().max().item() <= 5e-4 * (1 + ora["h"].abs().max().item())


def test_dopri5_failure_is_a_runtime_error():
    import hode
    from hode import adaptive
    from hode.solver import pack_theta
    dev = _dev()
    inp, f = _setup(6, 8, 8, seed=9)
    dosage, times = dose_schedule(inp["actions"], f.step_size)
    theta = torch.tensor(THETA_DEFAULT + (0.0,) * 3, device=dev)
    y0 = inp["z0"].to(dev)
    y0[2, 1] = float("nan")
    with pytest.raises(RuntimeError):
        adaptive.roche_dopri5(y0, theta, f.ml_net[0].weight.detach().to(dev), f.ml_net[0].bias.detach().to(dev),
                              inp["t"].to(dev), dosage.to(dev), times.to(dev), rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("variant", ["ablate", "general_hill", "two_doses", "tail_lanes"])
def test_dopri5_owner_layout_kernel_variants(variant):
    """The quad-layout (lanes 4) attempt / adjoint kernels exist in three rhs specialisations (Hill exponents == 2 with one
    dose per patient, == 2 with a dose list, general exponents) times ablate; the cases above only reach the first.
    Smooth problems (no dose) are compared tightly, the dose-list case at the tolerance of the discontinuous problem."""
    from hode import synth
    dev = _dev()
    N, T, D = (23, 14, 8) if variant != "tail_lanes" else (67, 9, 12)  # 67: a last wave with idle quads
    torch.manual_seed(90)
    theta = (3.0, 1.5, 0.8, 1.3, 0.7, 0.9, 1.1, 0.6, 1.2, 0.5, 1.4, 0.75, 0.65) if variant == "general_hill" else THETA_DEFAULT
    inp = synth.solver_inputs(N, T, D, seed=91, n_dose=2 if variant == "two_doses" else 1)
    f = RocheRHS(D, synth.STEP, ablate=variant == "ablate", theta=theta)
    with torch.no_grad():
        f.ml_net[0].weight.mul_(2.0)
    if variant != "two_doses":
        inp["actions"].zero_()
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(5))
    rtol = 1e-7 if variant == "two_doses" else 1e-6  # at 1e-6 both kernel layouts sit 2e-3 from the oracle on this problem
    hip, ora = _hip(inp, f, dev, 4, rtol, 1e-8, cot), _oracle(inp, f, rtol, 1e-8, cot)
    scale = 1 + ora["h"].abs().max().item()
    if variant == "two_doses":
        assert abs(hip["stats"]["n_accepted"] - ora["stats"]["n_accepted"]) <= 0.15 * ora["stats"]["n_accepted"]
        assert (hip["h"] - ora["h"]).abs().max().item() <= 1e-4 * scale
        tol = 4e-2
    else:
        assert abs(hip["stats"]["n_accepted"] - ora["stats"]["n_accepted"]) <= 1
        assert (hip["h"] - ora["h"]).abs().max().item() <= 1e-5 * scale
        tol = 2e-4
    for k in ("gy0", "gw", "gb", "gtheta"):
        if float(ora[k].abs().max()) < 1e-12:
            continue
        assert _rel(hip[k], ora[k]) <= tol, (variant, k, _rel(hip[k], ora[k]))


@pytest.mark.parametrize("N,T", [(1, 6), (3, 1), (2, 2), (65, 4)])
@pytest.mark.parametrize("lanes", [4, 1])
def test_dopri5_edge_shapes(N, T, lanes):
    """One patient, a single output time (nothing to integrate), a batch one past a whole wave."""
    dev = _dev()
    D = 12
    inp, f = _setup(N, T, D, seed=7) if T > 1 else _setup(N, 2, D, seed=7)
    if T == 1:
        inp = {"z0": inp["z0"], "actions": inp["actions"][:1] * 0, "t": inp["t"][:1]}
    else:
        inp["actions"].zero_()
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(6))
    hip = _hip(inp, f, dev, lanes, 1e-6, 1e-8, cot)
    assert hip["h"].shape == (T, N, D) and torch.equal(hip["h"][0], inp["z0"])
    if T == 1:
        assert hip["stats"]["n_accepted"] == 0 and torch.equal(hip["gy0"], cot[0])
        return
    ora = _oracle(inp, f, 1e-6, 1e-8, cot)
    assert abs(hip["stats"]["n_accepted"] - ora["stats"]["n_accepted"]) <= 1
    assert (hip["h"] - ora["h"]).abs().max().item() <= 1e-5 * (1 + ora["h"].abs().max().item())
    for k in ("gy0", "gw", "gb"):
        assert _rel(hip[k], ora[k]) <= 2e-4, (k, _rel(hip[k], ora[k]))


@pytest.mark.skipif(not __import__("os").environ.get("HODE_TEST_DP_EXPERIMENTS"),
                    reason="the persistent attempt loop is compiled only into experiment builds (HODE_DP_FLAGS=-DHODE_DP_EXPERIMENTS python "
                           "build_hip.py; then HODE_TEST_DP_EXPERIMENTS=1): measured slower (DESIGN.md 5c), not selectable in the product")
def test_dopri5_persistent_attempt_loop_is_an_equivalent_opt_in(monkeypatch):
    """HODE_DP_PERSIST=1: the whole attempt loop in one launch (waves exchange the error-norm partials through memory with a
    bounded poll).  Slower than one launch per attempt on this part and therefore opt-in, but it must integrate the same
    problem: same tolerance-level trajectory, gradients pinned by the replay oracle along ITS tape."""
    from hode import adaptive
    dev = _dev()
    N, T, D = 100, 16, 12
    inp, f = _setup(N, T, D, seed=52)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(3))
    base = _hip(inp, f, dev, 4, 1e-7, 1e-8, cot, detach_first_step=True)
    monkeypatch.setenv("HODE_DP_PERSIST", "1")
    adaptive.keep_workspace = True
    try:
        per = _hip(inp, f, dev, 4, 1e-7, 1e-8, cot, detach_first_step=True)
        tape = adaptive.read_tape()
    finally:
        adaptive.keep_workspace = False
    assert abs(per["stats"]["n_accepted"] - base["stats"]["n_accepted"]) <= 0.1 * base["stats"]["n_accepted"]
    assert (per["h"] - base["h"]).abs().max().item() <= 1e-4 * (1 + base["h"].abs().max().item())
    rep = _replay(inp, f, 1e-7, 1e-8, cot, tape, False)
    for k in ("gy0", "gw", "gb", "gtheta"):
        assert _rel(per[k], rep[k]) <= 1e-5, (k, _rel(per[k], rep[k]))


def test_dopri5_without_a_tape_is_the_same_forward():
    """HODE_FLAG_NO_TAPE (no input needs a gradient: evaluate() under no_grad): two state rows addressed by step parity
    instead of (max_steps + 1) rows -- same attempts, bit-identical trajectory, both rhs families; and the workspace does
    not grow with the step bound."""
    from hode import adaptive, synth
    from hode.solver import pack_theta
    dev = _dev()
    N, T, D = 150, 24, 12
    inp, f = _setup(N, T, D, seed=61)
    with_tape = _hip(inp, f, dev, 0, 1e-7, 1e-8)
    scal = [getattr(f, n).detach().to(dev) for n in THETA_NAMES]
    dosage, times = dose_schedule(inp["actions"], f.step_size)
    args = (inp["z0"].to(dev), pack_theta(scal, dev), f.ml_net[0].weight.detach().to(dev), f.ml_net[0].bias.detach().to(dev),
            inp["t"].to(dev), dosage.to(dev), times.to(dev))
    adaptive.keep_workspace = True
    try:
        with torch.no_grad():
            h = adaptive.roche_dopri5(*args, rtol=1e-7, atol=1e-8)
        ws_bytes = adaptive._last_ws[0].numel()
    finally:
        adaptive.keep_workspace = False
    assert adaptive.last_stats["no_tape"] is True
    assert all(adaptive.last_stats[k] == with_tape["stats"][k] for k in ("n_accepted", "n_rejected")) and torch.equal(h.cpu(), with_tape["h"])
    assert ws_bytes < 40 * N * D * 4 + (1 << 20) * 24 + (1 << 20)  # state rows + 24 B of time records per allowed step
    # neural rhs
    from oracle.rhs import NeuralRHS
    torch.manual_seed(3)
    g = NeuralRHS(D, synth.STEP)
    prm = [p.detach().to(dev) for p in (g.ml_net[0].weight, g.ml_net[0].bias, g.ml_net[2].weight, g.ml_net[2].bias)]
    y0 = (inp["z0"] * 30).to(dev)
    h_tape = adaptive.neural_dopri5(y0.clone().requires_grad_(True), *prm, args[4], args[5], args[6], rtol=1e-6, atol=1e-8)
    st = dict(adaptive.last_stats)
    with torch.no_grad():
        h_ring = adaptive.neural_dopri5(y0, *prm, args[4], args[5], args[6], rtol=1e-6, atol=1e-8)
    assert st["no_tape"] is False and adaptive.last_stats["no_tape"] is True
    assert all(adaptive.last_stats[k] == st[k] for k in ("n_accepted", "n_rejected")) and torch.equal(h_ring, h_tape.detach())
