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


def test_neural_dopri5_through_the_mirror():
    """`run_simulation --method=neural` keeps the reference's default solver, dopri5 (sim_config.py:50): NeuralODE has no
    fused adaptive kernel, the mirror integrates with `hode.adaptive_eager` on the GPU and says so once.  Same weights and
    inputs through the CPU oracle: trajectories, step counts and gradients."""
    import warnings

    import model
    from hode import adaptive_eager, synth
    from oracle import vi as ovi
    dev = _dev()
    D, obs, T, B = 8, 40, 12, 20
    torch.manual_seed(0)
    model._EAGER_DOPRI5_ANNOUNCED.discard("NeuralODE")
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, roche=False, method="dopri5", device=dev)
    dec_o = ovi.DecoderOracle(obs, D, (T - 1) * synth.STEP, synth.STEP, roche=False, method="dopri5")
    dec_o.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    inp = synth.solver_inputs(B, T, D, seed=2)
    z = inp["z0"].to(dev).requires_grad_(True)
    zo = inp["z0"].clone().requires_grad_(True)
    cot = torch.randn(T, B, obs)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        x_hat, h = dec(z, inp["actions"].to(dev))
        dec(z.detach(), inp["actions"].to(dev))
    assert sum("no fused kernel" in str(w.message) for w in seen) == 1  # announced, once
    x_o, h_o = dec_o(zo, inp["actions"])
    assert adaptive_eager.last_stats["n_accepted"] > 0
    assert (h.detach().cpu() - h_o.detach()).abs().max().item() <= 5e-6
    (x_hat * cot.to(dev)).sum().backward()
    (x_o * cot).sum().backward()
    assert _rel(z.grad.cpu(), zo.grad) <= 1e-4
    for (n, p), (_, po) in zip(dec.named_parameters(), dec_o.named_parameters()):
        if po.grad is None or float(po.grad.abs().max()) == 0.0:
            continue
        assert _rel(p.grad.cpu(), po.grad) <= 2e-4, n
