"""HIP fixed-grid solver (through the C ABI) vs the CPU oracle on identical seeded inputs.  GPU only.

Tolerances (fp32): forward trajectory max-abs <= 2e-5 * (1 + max|h|) and MSE <= 1e-9 (BASELINE target is 1e-5);
gradients rel-L2 <= 1e-4 (SURVEY 8d asks <= 1e-3).  The arithmetic differs from the oracle only by fma contraction
in the stage updates and <= 3 ulp transcendentals.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.rhs import RocheRHS, dose_schedule, THETA_DEFAULT
from oracle.solvers import odeint as oracle_odeint


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _case(N, T, D, seed, n_dose=1, ablate=False, theta=None):
    import hode
    from hode import synth
    inp = synth.solver_inputs(N, T, D, seed=seed, n_dose=n_dose)
    torch.manual_seed(seed)
    f = RocheRHS(D, synth.STEP, ablate=ablate, theta=theta or THETA_DEFAULT)
    if D > 4:
        with torch.no_grad():  # larger weights than default init so that the learned block matters
            f.ml_net[0].weight.mul_(2.0)
    return inp, f


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _run_hip(inp, f, method, lanes, dev, perturb=False, need_theta=True, cot=None):
    import hode
    from hode.solver import pack_theta, roche_solve
    from oracle.rhs import THETA_NAMES
    names = list(THETA_NAMES) + (["theta_1", "theta_2"] if f.ablate else [])
    scal = [getattr(f, n).detach().clone().to(dev).requires_grad_(need_theta) for n in names]
    theta = pack_theta(scal, dev)
    y0 = inp["z0"].to(dev).requires_grad_(True)
    w = b = None
    if f.ml_dim > 0:
        w = f.ml_net[0].weight.detach().clone().to(dev).requires_grad_(True)
        b = f.ml_net[0].bias.detach().clone().to(dev).requires_grad_(True)
    dosage, times = dose_schedule(inp["actions"], f.step_size)
    h = roche_solve(y0, theta, w, b, inp["t"].to(dev), dosage.to(dev), times.to(dev), method=method, ablate=f.ablate,
                    perturb=perturb, lanes_per_patient=lanes)
    out = {"h": h.detach().cpu()}
    if cot is not None:
        (h * cot.to(dev)).sum().backward()
        out["gy0"] = y0.grad.cpu()
        if w is not None:
            out["gw"], out["gb"] = w.grad.cpu(), b.grad.cpu()
        if need_theta:
            out["gtheta"] = torch.stack([s.grad if s.grad is not None else torch.zeros(()) .to(dev) for s in scal]).cpu()
    return out


def _run_oracle(inp, f, method, perturb=False, cot=None):
    f.set_action(inp["actions"])
    y0 = inp["z0"].clone().requires_grad_(True)
    f.zero_grad()
    h = oracle_odeint(f, y0, inp["t"], method=method, options={"perturb": perturb} if perturb else None)
    out = {"h": h.detach()}
    if cot is not None:
        (h * cot).sum().backward()
        out["gy0"] = y0.grad
        if f.ml_dim > 0:
            out["gw"], out["gb"] = f.ml_net[0].weight.grad, f.ml_net[0].bias.grad
        from oracle.rhs import THETA_NAMES
        names = list(THETA_NAMES) + (["theta_1", "theta_2"] if f.ablate else [])
        out["gtheta"] = torch.stack([getattr(f, n).grad if getattr(f, n).grad is not None else torch.zeros(()) for n in names])
    return out


def _compare(hip, ora, grads=True, h_tol=2e-5, g_tol=1e-4):
    h, ho = hip["h"], ora["h"]
    assert h.shape == ho.shape
    assert torch.equal(h[0], ho[0])  # h[0] == y0 exactly
    scale = 1 + ho.abs().max().item()
    assert (h - ho).abs().max().item() <= h_tol * scale, (h - ho).abs().max().item()
    assert torch.mean((h - ho) ** 2).item() <= 1e-9 * scale ** 2
    if grads:
        for k in ("gy0", "gw", "gb", "gtheta"):
            if k in ora and k in hip:
                assert _rel(hip[k], ora[k]) <= g_tol, (k, _rel(hip[k], ora[k]), hip[k].flatten()[:4], ora[k].flatten()[:4])


@pytest.mark.parametrize("D", [4, 6, 8, 12, 20])
@pytest.mark.parametrize("lanes", [1, 4])
def test_rk4_forward_backward_vs_oracle(D, lanes):
    dev = _dev()
    N, T = 77, 40  # ragged: not a multiple of 16 or 64
    inp, f = _case(N, T, D, seed=11 + D)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(5))
    _compare(_run_hip(inp, f, "rk4", lanes, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("perturb", [False, True])
def test_methods_and_perturb(method, perturb):
    dev = _dev()
    N, T, D = 33, 25, 12
    inp, f = _case(N, T, D, seed=3)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(6))
    for lanes in (1, 4):
        _compare(_run_hip(inp, f, method, lanes, dev, perturb=perturb, cot=cot), _run_oracle(inp, f, method, perturb=perturb, cot=cot))


@pytest.mark.parametrize("D", [4, 12])
def test_ablate_rhs(D):
    dev = _dev()
    inp, f = _case(50, 30, D, seed=8, ablate=True)
    cot = torch.randn(30, 50, D, generator=torch.Generator().manual_seed(7))
    for lanes in (1, 4):
        _compare(_run_hip(inp, f, "rk4", lanes, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))


def test_multiple_doses_per_patient():
    dev = _dev()
    inp, f = _case(40, 30, 12, seed=9, n_dose=3)
    cot = torch.randn(30, 40, 12, generator=torch.Generator().manual_seed(8))
    for lanes in (1, 4):
        _compare(_run_hip(inp, f, "rk4", lanes, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))


def test_general_hill_exponents_and_random_theta():
    dev = _dev()
    theta = (3.0, 1.5, 0.8, 1.3, 0.7, 0.9, 1.1, 0.6, 1.2, 0.5, 1.4, 0.75, 0.65)
    inp, f = _case(45, 30, 8, seed=10, theta=theta)
    cot = torch.randn(30, 45, 8, generator=torch.Generator().manual_seed(9))
    for lanes in (1, 4):
        _compare(_run_hip(inp, f, "rk4", lanes, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot), g_tol=3e-4)


@pytest.mark.parametrize("N,T", [(1, 10), (2, 2), (5, 1), (64, 3), (65, 7)])
def test_edge_shapes(N, T):
    dev = _dev()
    inp, f = _case(N, T, 12, seed=20 + N)
    if T < 3:  # no room for a dose index in [0, T-2]: zero actions need K=0 handling
        inp["actions"].zero_()
        if T >= 2:
            inp["actions"][0, :, 0] = 1.5
    cot = torch.randn(T, N, 12, generator=torch.Generator().manual_seed(10))
    if float(inp["actions"].abs().sum()) == 0.0:
        pytest.skip("K=0 (no doses) is covered by test_no_dose")
    for lanes in (1, 4):
        _compare(_run_hip(inp, f, "rk4", lanes, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))


def test_no_dose():
    dev = _dev()
    inp, f = _case(19, 12, 12, seed=31)
    inp["actions"].zero_()
    cot = torch.randn(12, 19, 12, generator=torch.Generator().manual_seed(11))
    _compare(_run_hip(inp, f, "rk4", 4, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))


def test_config1_dim8_100x50_and_full_size_properties():
    """BASELINE configs[0] shape against the oracle, then configs[1] full size through size-independent properties."""
    dev = _dev()
    inp, f = _case(100, 50, 8, seed=666)
    cot = torch.randn(50, 100, 8, generator=torch.Generator().manual_seed(12))
    _compare(_run_hip(inp, f, "rk4", 0, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))
    # full size: (1) batch-slice invariance: patients are independent, so any sub-batch solved alone is bit-identical;
    # (2) lane-layout agreement: LPP=1 and LPP=4 variants agree to rounding; (3) gradient linearity in the cotangent.
    N, T, D = 10000, 100, 12
    inp, f = _case(N, T, D, seed=666)
    big = _run_hip(inp, f, "rk4", 4, dev)
    sub = {"z0": inp["z0"][1234:1300], "actions": inp["actions"][:, 1234:1300], "t": inp["t"]}
    small = _run_hip(sub, f, "rk4", 4, dev)
    assert torch.equal(big["h"][:, 1234:1300], small["h"])
    assert torch.isfinite(big["h"]).all()
    alt = _run_hip(inp, f, "rk4", 1, dev)
    assert (alt["h"] - big["h"]).abs().max().item() <= 2e-5 * (1 + big["h"].abs().max().item())
    ora = _run_oracle({"z0": inp["z0"][:256], "actions": inp["actions"][:, :256], "t": inp["t"]}, f, "rk4")
    assert (big["h"][:, :256] - ora["h"]).abs().max().item() <= 2e-5 * (1 + ora["h"].abs().max().item())
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(13))
    g1 = _run_hip(inp, f, "rk4", 4, dev, cot=cot)
    g2 = _run_hip(inp, f, "rk4", 4, dev, cot=2.0 * cot)
    assert _rel(g2["gw"], 2.0 * g1["gw"]) <= 1e-6 and _rel(g2["gy0"], 2.0 * g1["gy0"]) <= 1e-6
    # determinism: parameter gradients are folded in a fixed order
    g3 = _run_hip(inp, f, "rk4", 4, dev, cot=cot)
    assert torch.equal(g1["gw"], g3["gw"]) and torch.equal(g1["gtheta"], g3["gtheta"])


def test_errors_are_loud():
    import hode
    from hode.solver import pack_theta, roche_solve
    dev = _dev()
    inp, f = _case(8, 6, 12, seed=1)
    dosage, times = dose_schedule(inp["actions"], f.step_size)
    theta = torch.tensor(THETA_DEFAULT + (0.0,) * 3)
    with pytest.raises(hode.HodeConfigError):  # CPU tensors: no fallback
        roche_solve(inp["z0"], theta, f.ml_net[0].weight, f.ml_net[0].bias, inp["t"], dosage, times)
    with pytest.raises(hode.HodeConfigError):  # unsupported latent dim: a configuration error, not divergence
        roche_solve(torch.zeros(4, 7, device=dev), theta.to(dev), torch.zeros(3, 7, device=dev), torch.zeros(3, device=dev),
                    inp["t"].to(dev), torch.zeros(4, device=dev), torch.zeros(4, 1, device=dev))


# ------------------------------------------------------------------------------------------------ MFMA layout (lanes = 16)
@pytest.mark.parametrize("D", [8, 12])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_mfma_layout_vs_oracle(D, method):
    """hode_rk_mf.hip: same contract as the quad layout, matvecs and weight gradients on the matrix pipe."""
    dev = _dev()
    N, T = 77, 30
    inp, f = _case(N, T, D, seed=50 + D)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(15))
    _compare(_run_hip(inp, f, method, 16, dev, cot=cot), _run_oracle(inp, f, method, cot=cot))


def test_mfma_layout_variants():
    dev = _dev()
    cot = torch.randn(25, 40, 12, generator=torch.Generator().manual_seed(16))
    # perturb, ablate, several doses, general Hill exponents / random theta
    inp, f = _case(40, 25, 12, seed=61)
    _compare(_run_hip(inp, f, "rk4", 16, dev, perturb=True, cot=cot), _run_oracle(inp, f, "rk4", perturb=True, cot=cot))
    inp, f = _case(40, 25, 12, seed=62, ablate=True)
    _compare(_run_hip(inp, f, "rk4", 16, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))
    inp, f = _case(40, 25, 12, seed=63, n_dose=3)
    _compare(_run_hip(inp, f, "rk4", 16, dev, cot=cot), _run_oracle(inp, f, "rk4", cot=cot))
    theta = (3.0, 1.5, 0.8, 1.3, 0.7, 0.9, 1.1, 0.6, 1.2, 0.5, 1.4, 0.75, 0.65)
    inp, f = _case(40, 25, 8, seed=64, theta=theta)
    cot8 = torch.randn(25, 40, 8, generator=torch.Generator().manual_seed(17))
    _compare(_run_hip(inp, f, "rk4", 16, dev, cot=cot8), _run_oracle(inp, f, "rk4", cot=cot8), g_tol=3e-4)
    # edge shapes
    for N, T in ((1, 5), (16, 2), (17, 3), (5, 1)):
        inp, f = _case(N, T, 12, seed=70 + N)
        if T < 3:
            inp["actions"].zero_()
            if T >= 2:
                inp["actions"][0, :, 0] = 1.5
        c = torch.randn(T, N, 12, generator=torch.Generator().manual_seed(18))
        if float(inp["actions"].abs().sum()) == 0.0:
            continue
        _compare(_run_hip(inp, f, "rk4", 16, dev, cot=c), _run_oracle(inp, f, "rk4", cot=c))


def test_mfma_layout_agrees_with_quad_layout_at_full_size():
    dev = _dev()
    N, T, D = 10000, 100, 12
    inp, f = _case(N, T, D, seed=666)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(19))
    a = _run_hip(inp, f, "rk4", 16, dev, cot=cot)
    b = _run_hip(inp, f, "rk4", 4, dev, cot=cot)
    assert (a["h"] - b["h"]).abs().max().item() <= 2e-5 * (1 + b["h"].abs().max().item())
    for k in ("gy0", "gw", "gb", "gtheta"):
        assert _rel(a[k], b[k]) <= 1e-4, (k, _rel(a[k], b[k]))
    a2 = _run_hip(inp, f, "rk4", 16, dev, cot=cot)
    assert torch.equal(a["gw"], a2["gw"]) and torch.equal(a["gtheta"], a2["gtheta"])  # deterministic fold


# ------------------------------------------------------------------------------------------------ split layout (lanes = 48)
@pytest.mark.parametrize("D", [8, 12])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_split_layout_forward_vs_oracle(D, method):
    """hode_rk_split.hip: expert wave + learned waves through an LDS ring; forward kernel (backward = quad layout)."""
    dev = _dev()
    for N, T, perturb in ((77, 30, False), (48, 3, True), (1, 9, False), (150, 2, False)):
        inp, f = _case(N, T, D, seed=80 + D + N)
        if T < 3:
            inp["actions"].zero_()
            inp["actions"][0, :, 0] = 1.5
        cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(21))
        _compare(_run_hip(inp, f, method, 48, dev, perturb=perturb, cot=cot), _run_oracle(inp, f, method, perturb=perturb, cot=cot))
    inp, f = _case(40, 20, D, seed=90, ablate=True)
    cot = torch.randn(20, 40, D, generator=torch.Generator().manual_seed(22))
    _compare(_run_hip(inp, f, method, 48, dev, cot=cot), _run_oracle(inp, f, method, cot=cot))
    inp, f = _case(40, 20, D, seed=91, n_dose=2)
    _compare(_run_hip(inp, f, method, 48, dev, cot=cot), _run_oracle(inp, f, method, cot=cot))


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [4, 16, 48])
def test_plan_overwrite_flag_matches_accumulating_backward(lanes):
    """HODE_FLAG_OVERWRITE_GRADS (what RocheRKPlan sets): the fold STORES the parameter gradients -- same values as the
    accumulate-into-zeroed-buffers path of roche_solve, and a second backward does not double them."""
    dev = _dev()
    from hode.plan import RocheRKPlan
    from hode.solver import pack_theta
    from oracle.rhs import THETA_NAMES
    N, T, D = 130, 12, 12
    inp, f = _case(N, T, D, seed=123)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(5))
    ref = _run_hip(inp, f, "rk4", lanes, dev, cot=cot)
    theta = pack_theta([getattr(f, n).detach().to(dev) for n in THETA_NAMES], dev)
    dosage, times = dose_schedule(inp["actions"], f.step_size)
    plan = RocheRKPlan(inp["z0"].to(dev), theta, f.ml_net[0].weight.detach().to(dev), f.ml_net[0].bias.detach().to(dev),
                       inp["t"].to(dev), dosage.to(dev), times.to(dev), method="rk4", lanes_per_patient=lanes)
    plan.grad_flat.fill_(7.0)  # stale contents must not leak into the result
    plan.grad_h.copy_(cot.to(dev))
    for _ in range(2):
        plan.forward()
        gy0, flat = plan.backward()
    torch.cuda.synchronize()
    assert torch.equal(plan.h.cpu(), ref["h"])
    assert torch.equal(gy0.cpu(), ref["gy0"])
    assert torch.equal(plan.grad_w.cpu(), ref["gw"]) and torch.equal(plan.grad_b.cpu(), ref["gb"])
    assert torch.equal(plan.grad_theta[:13].cpu(), ref["gtheta"])


@pytest.mark.gpu
@pytest.mark.parametrize("D", [8, 12])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("N,T", [(101, 9), (1, 2), (47, 3), (49, 4), (96, 5)])
def test_split_tape_matches_recompute(D, method, N, T):
    """HODE_FLAG_TAPE: the forward leaves the expert stage states (rk4: and the learned block's last two stage derivatives)
    in the workspace, the backward reads them instead of re-integrating and runs the theta gradients on a fifth wave -- the
    same numbers in the same order, so every output must be bit-identical to the tape-less 4-wave backward.  Short grids
    exercise the pipelines' prologues / epilogues (the theta wave lags two iterations, the learned tape is fetched two
    ahead), batch sizes around the 48-patient workgroup the spare lanes."""
    dev = _dev()
    from hode.plan import RocheRKPlan
    from hode.solver import pack_theta
    from oracle.rhs import THETA_NAMES
    inp, f = _case(N, T, D, seed=7 + D)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(6)).to(dev)
    theta = pack_theta([getattr(f, n).detach().to(dev) for n in THETA_NAMES], dev)
    dosage, times = dose_schedule(inp["actions"], f.step_size)
    outs = []
    for tape in (False, True):
        plan = RocheRKPlan(inp["z0"].to(dev), theta, f.ml_net[0].weight.detach().to(dev), f.ml_net[0].bias.detach().to(dev),
                           inp["t"].to(dev), dosage.to(dev), times.to(dev), method=method, lanes_per_patient=48, tape=tape)
        plan.grad_h.copy_(cot)
        plan.forward()
        gy0, flat = plan.backward()
        torch.cuda.synchronize()
        outs.append((plan.h.clone(), gy0.clone(), flat.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_long_grids_fall_back_to_the_quad_layout():
    """The split kernels keep the time grid in LDS (<= 8192 points); longer grids must silently take the quad layout."""
    dev = _dev()
    from hode.solver import pack_theta, roche_solve
    from oracle.rhs import THETA_NAMES
    N, D = 5, 12
    inp, f = _case(N, 16, D, seed=3)
    theta = pack_theta([getattr(f, n).detach().to(dev) for n in THETA_NAMES], dev)
    w, b = f.ml_net[0].weight.detach().to(dev), f.ml_net[0].bias.detach().to(dev)
    dosage = torch.full((N,), 2.0, device=dev)
    times = torch.full((N, 1), 0.5, device=dev)
    for T, same in ((8192, False), (8193, True)):
        t = (torch.arange(T, dtype=torch.float32) * 1e-3).to(dev)
        outs = []
        for lanes in (0, 4):
            y0 = inp["z0"].to(dev).requires_grad_(True)
            h = roche_solve(y0, theta, w, b, t, dosage, times, method="rk4", lanes_per_patient=lanes)
            h[-1].sum().backward()
            outs.append((h.detach(), y0.grad.clone()))
        assert torch.isfinite(outs[0][0]).all()
        if same:  # T > 8192: `auto` IS the quad layout
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        else:
            assert _rel(outs[0][0], outs[1][0]) < 1e-5 and _rel(outs[0][1], outs[1][1]) < 1e-4
