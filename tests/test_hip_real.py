"""HIP real-data rhs solver (config 5: midpoint + perturb on the DDW-shaped problem) vs the CPU oracle.  GPU only.
RocheRealRHS is pinned by golden G3, EncoderLSTMReal by G4.  Tolerances: trajectory 2e-5*(1+max|h|), grads rel-L2 1e-4."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.rhs import RocheRealRHS
from oracle.solvers import odeint as oracle_odeint


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _flat(f):
    ps = [f.dx1_net[0].weight, f.dx1_net[0].bias, f.dx1_net[2].weight, f.dx1_net[2].bias,
          f.dx2_net[0].weight, f.dx2_net[0].bias, f.dx2_net[2].weight, f.dx2_net[2].bias]
    if f.ml_dim > 0:
        ps += [f.lin_hh.weight, f.lin_hz.weight, f.lin_hr.weight]
    return ps


@pytest.mark.parametrize("layout", ["matrix-cores", "lane-per-patient"])
@pytest.mark.parametrize("D,H", [(20, 43), (4, 9), (20, 17), (20, 64)])
@pytest.mark.parametrize("method,perturb", [("midpoint", True), ("rk4", True), ("euler", False)])
def test_real_forward_backward_vs_oracle(D, H, method, perturb, layout, monkeypatch):
    """Both kernel families behind HODE_RHS_ROCHE_REAL: hode_real_mf.hip (default at D = 20; hidden 17 / 43 / 64 exercise
    2, 3 and 4 hidden tiles) and hode_real.hip (HODE_REAL_LAYOUT=t; always for D = 4)."""
    from hode.real import real_solve
    dev = _dev()
    if layout == "lane-per-patient":
        monkeypatch.setenv("HODE_REAL_LAYOUT", "t")
    else:
        monkeypatch.delenv("HODE_REAL_LAYOUT", raising=False)
    B, Ta, t0 = 37, 30, 8
    gen = torch.Generator().manual_seed(D + H)
    torch.manual_seed(D)
    f = RocheRealRHS(D, H)
    a = (torch.rand(Ta, B, 1, generator=gen) < 0.2).float() * torch.rand(Ta, B, 1, generator=gen)
    f.set_action_static(a)
    t = torch.arange(t0 - 1, Ta, 1, dtype=torch.float32)
    y0 = (torch.randn(B, D, generator=gen) * 0.3).requires_grad_(True)
    cot = torch.randn(t.numel(), B, D, generator=gen)
    ho = oracle_odeint(f, y0, t, method=method, options={"perturb": perturb, "step_size": 1.0})
    (ho * cot).sum().backward()
    ps = _flat(f)
    wflat = torch.cat([p.detach().reshape(-1) for p in ps]).to(dev).requires_grad_(True)
    theta = torch.stack([f.k_immunity, f.kel, f.kel2]).detach().to(dev).requires_grad_(True)
    y0g = y0.detach().to(dev).requires_grad_(True)
    h = real_solve(y0g, theta, wflat, t.to(dev), a[..., 0].to(dev), H, method=method, perturb=perturb)
    assert (h.detach().cpu() - ho.detach()).abs().max().item() <= 2e-5 * (1 + ho.abs().max().item())
    (h * cot.to(dev)).sum().backward()
    assert _rel(y0g.grad, y0.grad) <= 1e-4
    want = torch.cat([p.grad.reshape(-1) for p in ps])
    assert _rel(wflat.grad, want) <= 1e-4, _rel(wflat.grad, want)
    assert _rel(theta.grad, torch.stack([f.k_immunity.grad, f.kel.grad, f.kel2.grad])) <= 1e-4


@pytest.mark.parametrize("ode_step_div", [1, 2, 3])
def test_config5_mirror_loss_matches_cpu_oracle_pipeline(ode_step_div):
    """DDW-shaped tensors (obs 24, statics 11, D 20, enc 37->44, dec hidden 43, t0 24): VariationalInferenceReal.loss and
    its gradients on the GPU vs the same modules evaluated on the CPU with the oracle solver injected (test only).
    `ode_step_div` > 1 (run_real.py:51, `--ode_step_div`) makes torchdiffeq integrate on a grid finer than the outputs:
    the kernels run over that grid and the outputs are read off it (`hode/substep.py`); 3 gives grid points that are
    not fp32-exact fractions of the hour."""
    import copy
    import model
    dev = _dev()
    obs, act, stat, D, T, t0, B = 24, 1, 11, 20, 40, 24, 33
    input_dim = obs + act + stat + 1
    hidden = int((obs + act + stat) * 1.2)
    torch.manual_seed(3)
    cpu = torch.device("cpu")
    enc_c = model.EncoderLSTMReal(input_dim, int(input_dim * 1.2), D, output_all=False, reverse=False, device=cpu)
    dec_c = model.DecoderReal(obs, D, act, stat, hidden, T, 1, method="midpoint", ode_step_size=1.0 / ode_step_div, ode_type="hybrid", t0=t0, device=cpu)
    dec_c._odeint = oracle_odeint
    enc_g, dec_g = copy.deepcopy(enc_c).to(dev), copy.deepcopy(dec_c).to(dev)
    enc_g.device = dec_g.device = dec_g.ode.device = dev
    dec_g.t = dec_g.t.to(dev)
    dec_g.options["step_t"] = dec_g.t
    dec_g._odeint = model.hode.odeint
    gen = torch.Generator().manual_seed(4)
    data = {"measurements": torch.randn(T, B, obs, generator=gen),
            "actions": (torch.rand(T, B, 1, generator=gen) < 0.1).float() * torch.rand(T, B, 1, generator=gen),
            "masks": (torch.rand(T, B, obs, generator=gen) < 0.5).float(),
            "statics": torch.rand(T, B, stat, generator=gen)}
    vi_c = model.VariationalInferenceReal(enc_c, dec_c, elbo=False, t0=t0)
    vi_g = model.VariationalInferenceReal(enc_g, dec_g, elbo=False, t0=t0)
    lc = vi_c.loss(data)
    lc.backward()
    lg = vi_g.loss({k: v.to(dev) for k, v in data.items()})
    lg.backward()
    assert vi_g.x_hat.shape == (T - t0, B, obs)
    assert abs(lg.item() - lc.item()) <= 1e-4 * abs(lc.item())
    for (n, pg), (_, pc) in zip(list(enc_g.named_parameters()) + list(dec_g.named_parameters()),
                                list(enc_c.named_parameters()) + list(dec_c.named_parameters())):
        if pc.grad is None or float(pc.grad.abs().max()) < 1e-10:
            continue
        assert _rel(pg.grad, pc.grad) <= 2e-3, (n, _rel(pg.grad, pc.grad))
