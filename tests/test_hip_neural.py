"""HIP NeuralODE-rhs fixed-grid solver (C ABI) vs the CPU oracle (NeuralRHS pinned by golden G2).  GPU only.
Tolerances as for the Roche kernels: trajectory 2e-5*(1+max|h|), gradients rel-L2 <= 1e-4."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.rhs import NeuralRHS, dose_schedule
from oracle.solvers import odeint as oracle_odeint


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("layout", ["matrix-cores", "lane-per-patient"])
@pytest.mark.parametrize("D", [6, 8, 12])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_neural_forward_backward_vs_oracle(D, method, layout, monkeypatch):
    """Both kernel families behind HODE_RHS_NEURAL: hode_neural_mf.hip (default) and hode_neural.hip (HODE_NEURAL_LAYOUT=t);
    N = 70 leaves the last 16-patient wave of the matrix-core layout partly empty."""
    from hode import synth
    from hode.neural import neural_solve
    dev = _dev()
    if layout == "lane-per-patient":
        monkeypatch.setenv("HODE_NEURAL_LAYOUT", "t")
    else:
        monkeypatch.delenv("HODE_NEURAL_LAYOUT", raising=False)
    N, T = 70, 14
    inp = synth.solver_inputs(N, T, D, seed=D)
    inp["z0"] = inp["z0"] * 30.0
    torch.manual_seed(D)
    f = NeuralRHS(D, synth.STEP)
    f.set_action(inp["actions"])
    y0 = inp["z0"].clone().requires_grad_(True)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(1))
    ho = oracle_odeint(f, y0, inp["t"], method=method)
    (ho * cot).sum().backward()
    prm = [p.detach().clone().to(dev).requires_grad_(True) for p in (f.ml_net[0].weight, f.ml_net[0].bias, f.ml_net[2].weight, f.ml_net[2].bias)]
    y0g = inp["z0"].to(dev).requires_grad_(True)
    dosage, times = dose_schedule(inp["actions"], synth.STEP)
    h = neural_solve(y0g, *prm, inp["t"].to(dev), dosage.to(dev), times.to(dev), method=method)
    assert torch.equal(h[0].cpu(), ho[0].detach())
    assert (h.detach().cpu() - ho.detach()).abs().max().item() <= 2e-5 * (1 + ho.abs().max().item())
    (h * cot.to(dev)).sum().backward()
    assert _rel(y0g.grad, y0.grad) <= 1e-4
    for q, ref in zip(prm, (f.ml_net[0].weight, f.ml_net[0].bias, f.ml_net[2].weight, f.ml_net[2].bias)):
        assert _rel(q.grad, ref.grad) <= 1e-4, _rel(q.grad, ref.grad)


def test_neural_decoder_mirror_and_dose_impulse():
    """The dose enters only when a stage time equals a dose time exactly: a run with the dose moved off-grid differs."""
    import model
    from hode import synth
    dev = _dev()
    D, obs, T, B = 8, 40, 12, 20
    torch.manual_seed(0)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, roche=False, method="rk4", device=dev)
    assert dec.model_name == "NeuralODEDecoder" and list(dec.state_dict())[2:] == ["ode.kel", "ode.ml_net.0.weight", "ode.ml_net.0.bias", "ode.ml_net.2.weight", "ode.ml_net.2.bias"]
    inp = synth.solver_inputs(B, T, D, seed=2)
    z = inp["z0"].to(dev)
    a = inp["actions"].to(dev)
    x_hat, h = dec(z, a)
    assert x_hat.shape == (T, B, obs) and torch.isfinite(h).all()
    _, h0 = dec(z, torch.zeros_like(a))
    assert (h - h0).abs().max().item() > 1e-4  # the impulse at the on-grid dose time changes the trajectory


def _neural_case(N, T, D, seed, scale=30.0):
    from hode import synth
    inp = synth.solver_inputs(N, T, D, seed=seed)
    inp["z0"] = inp["z0"] * scale
    torch.manual_seed(seed)
    f = NeuralRHS(D, synth.STEP)
    with torch.no_grad():
        f.ml_net[2].weight.mul_(2.0)
    f.set_action(inp["actions"])
    return inp, f


def _neural_hip_dopri5(inp, f, dev, cot, rtol, atol, detach=False):
    from hode import adaptive, synth
    prm = [p.detach().clone().to(dev).requires_grad_(True) for p in (f.ml_net[0].weight, f.ml_net[0].bias, f.ml_net[2].weight, f.ml_net[2].bias)]
    y0 = inp["z0"].to(dev).requires_grad_(True)
    dosage, times = dose_schedule(inp["actions"], synth.STEP)
    h = adaptive.neural_dopri5(y0, *prm, inp["t"].to(dev), dosage.to(dev), times.to(dev), rtol=rtol, atol=atol, detach_first_step=detach)
    (h * cot.to(dev)).sum().backward()
    return {"h": h.detach().cpu(), "g": [y0.grad.cpu()] + [q.grad.cpu() for q in prm], "stats": dict(adaptive.last_stats)}


def _neural_ref_grads(f, y0):
    return [y0.grad] + [p.grad for p in (f.ml_net[0].weight, f.ml_net[0].bias, f.ml_net[2].weight, f.ml_net[2].bias)]


@pytest.mark.parametrize("D,N", [(12, 70), (8, 33), (6, 16)])
def test_neural_dopri5_fused_kernels_vs_oracle_and_tape_replay(D, N):
    """HODE_RHS_NEURAL through hode_dopri5_fwd / _bwd (csrc/hode_neural_dopri5.hip): against the free-running oracle (the
    rhs is smooth -- the impulse dose never fires off the grid -- so the two controllers take the same steps up to the
    last bit of the error norm), against the oracle's step algebra replayed along the run's own tape (both with and without
    the derivative of the first step size), and the on-chip weight gradients against autograd."""
    from hode import adaptive
    from oracle.solvers import odeint_dopri5_replay
    dev = _dev()
    T, rtol, atol = 14, 1e-6, 1e-8
    inp, f = _neural_case(N, T, D, seed=D)
    cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(1))
    adaptive.keep_workspace = True
    try:
        hip = _neural_hip_dopri5(inp, f, dev, cot, rtol, atol)
        tape = adaptive.read_tape()
    finally:
        adaptive.keep_workspace = False
    hip_det = _neural_hip_dopri5(inp, f, dev, cot, rtol, atol, detach=True)
    # free-running oracle
    y0 = inp["z0"].clone().requires_grad_(True)
    f.zero_grad()
    st = {}
    ho = oracle_odeint(f, y0, inp["t"], method="dopri5", rtol=rtol, atol=atol, stats=st)
    (ho * cot).sum().backward()
    assert abs(hip["stats"]["n_accepted"] - st["n_accepted"]) <= 1 and hip["stats"]["n_rejected"] <= st["n_rejected"] + 1
    assert torch.equal(hip["h"][0], ho[0].detach())
    assert (hip["h"] - ho.detach()).abs().max().item() <= 1e-5 * (1 + ho.abs().max().item())
    for a, b in zip(hip["g"], _neural_ref_grads(f, y0)):
        assert _rel(a, b) <= 2e-4, _rel(a, b)
    # the run's own tape, replayed by the oracle's step algebra
    pairs = list(zip(tape["t"], tape["dt"]))
    first = bool(tape["init"]["first_accepted"])
    for run, with_first in ((hip_det, False), (hip, first)):
        y0 = inp["z0"].clone().requires_grad_(True)
        f.zero_grad()
        hr = odeint_dopri5_replay(f, y0, inp["t"], rtol, atol, pairs, with_first)
        (hr * cot).sum().backward()
        assert (run["h"] - hr.detach()).abs().max().item() <= 5e-6 * (1 + hr.abs().max().item())
        for a, b in zip(run["g"], _neural_ref_grads(f, y0)):
            assert _rel(a, b) <= 1e-4, (with_first, _rel(a, b))


def test_neural_dopri5_matches_the_eager_cross_check_and_edge_shapes():
    """tests/adaptive_eager.py (torch launches per stage: torchdiffeq's semantics written out) stays as a cross-check of the
    fused kernels; one output time / one patient / a batch one past a wave."""
    import adaptive_eager
    from hode import adaptive, synth
    dev = _dev()
    D, T = 8, 12
    for N in (1, 17):
        inp, f = _neural_case(N, T, D, seed=3 + N)
        cot = torch.randn(T, N, D, generator=torch.Generator().manual_seed(2))
        hip = _neural_hip_dopri5(inp, f, dev, cot, 1e-6, 1e-8)
        fg = NeuralRHS(D, synth.STEP).to(dev)
        fg.load_state_dict(f.state_dict())
        fg.set_action(inp["actions"].to(dev))
        y0 = inp["z0"].to(dev).requires_grad_(True)
        he = adaptive_eager.odeint_dopri5(fg, y0, inp["t"].to(dev), rtol=1e-6, atol=1e-8)
        (he * cot.to(dev)).sum().backward()
        assert abs(adaptive_eager.last_stats["n_accepted"] - hip["stats"]["n_accepted"]) <= 1
        assert (hip["h"] - he.detach().cpu()).abs().max().item() <= 1e-5
        for a, b in zip(hip["g"], [y0.grad] + [p.grad for p in (fg.ml_net[0].weight, fg.ml_net[0].bias, fg.ml_net[2].weight, fg.ml_net[2].bias)]):
            assert _rel(a, b.cpu()) <= 2e-4
    inp, f = _neural_case(5, 2, D, seed=9)
    one = {"z0": inp["z0"], "actions": inp["actions"][:1] * 0, "t": inp["t"][:1]}
    f.set_action(one["actions"])
    cot = torch.randn(1, 5, D)
    hip = _neural_hip_dopri5(one, f, dev, cot, 1e-6, 1e-8)
    assert hip["stats"]["n_accepted"] == 0 and torch.equal(hip["h"][0], one["z0"]) and torch.equal(hip["g"][0], cot[0])
    assert all(float(g.abs().max()) == 0.0 for g in hip["g"][1:])


def test_neural_dopri5_through_the_mirror():
    """`run_simulation --method=neural` keeps the reference's default solver, dopri5 (sim_config.py:50): the mirror's
    NeuralODE integrates with the fused kernels at every compiled latent dimension (even, 4..14; the reference's configs
    use 6, 8, 12) and raises a configuration error elsewhere -- there is no torch-eager path in the product.  Same weights
    and inputs through the CPU oracle: trajectories and gradients."""
    import hode
    import model
    from hode import adaptive, synth
    from oracle import vi as ovi
    dev = _dev()
    obs, T, B = 40, 12, 20
    for D in (4, 6, 8, 10, 14):
        torch.manual_seed(0)
        dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, roche=False, method="dopri5", device=dev)
        dec_o = ovi.DecoderOracle(obs, D, (T - 1) * synth.STEP, synth.STEP, roche=False, method="dopri5")
        dec_o.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
        inp = synth.solver_inputs(B, T, D, seed=2)
        z = inp["z0"].to(dev).requires_grad_(True)
        zo = inp["z0"].clone().requires_grad_(True)
        cot = torch.randn(T, B, obs)
        adaptive.last_stats.update(n_accepted=-1)
        x_hat, h = dec(z, inp["actions"].to(dev))
        assert adaptive.last_stats["n_accepted"] > 0
        x_o, h_o = dec_o(zo, inp["actions"])
        assert (h.detach().cpu() - h_o.detach()).abs().max().item() <= 5e-6
        (x_hat * cot.to(dev)).sum().backward()
        (x_o * cot).sum().backward()
        assert _rel(z.grad.cpu(), zo.grad) <= 2e-4
        for (n, p), (_, po) in zip(dec.named_parameters(), dec_o.named_parameters()):
            if po.grad is None or float(po.grad.abs().max()) == 0.0:
                continue
            assert _rel(p.grad.cpu(), po.grad) <= 2e-4, (D, n)
    dec = model.RocheExpertDecoder(obs, 16, 1, (T - 1) * synth.STEP, synth.STEP, roche=False, method="dopri5", device=dev)
    inp = synth.solver_inputs(B, T, 16, seed=2)
    with pytest.raises(hode.HodeConfigError, match="4, 6, 8, 10, 12, 14"):
        dec(inp["z0"].to(dev), inp["actions"].to(dev))


@pytest.mark.parametrize("D", [4, 10, 14])
def test_neural_fixed_grid_at_the_added_latent_dimensions(D):
    """rk4 / midpoint / euler with the NeuralODE rhs at the latent dimensions added in round 3, through the mirror."""
    import model
    from hode import synth
    from oracle import vi as ovi
    dev = _dev()
    obs, T, B = 24, 10, 21
    for method in ("rk4", "midpoint", "euler"):
        torch.manual_seed(1)
        dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, roche=False, method=method, device=dev)
        dec_o = ovi.DecoderOracle(obs, D, (T - 1) * synth.STEP, synth.STEP, roche=False, method=method)
        dec_o.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
        inp = synth.solver_inputs(B, T, D, seed=4)
        z = inp["z0"].to(dev).requires_grad_(True)
        zo = inp["z0"].clone().requires_grad_(True)
        cot = torch.randn(T, B, obs)
        x_hat, h = dec(z, inp["actions"].to(dev))
        x_o, h_o = dec_o(zo, inp["actions"])
        assert (h.detach().cpu() - h_o.detach()).abs().max().item() <= 2e-5
        (x_hat * cot.to(dev)).sum().backward()
        (x_o * cot).sum().backward()
        assert _rel(z.grad.cpu(), zo.grad) <= 2e-4
        for (n, p), (_, po) in zip(dec.named_parameters(), dec_o.named_parameters()):
            if po.grad is None or float(po.grad.abs().max()) == 0.0:
                continue
            assert _rel(p.grad.cpu(), po.grad) <= 2e-4, (D, method, n)
