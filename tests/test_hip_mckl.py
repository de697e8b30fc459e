"""hode_mc_kl_exponential (through the C ABI) vs the reference arithmetic (model.py:1198-1214 restated with torch ops in
float64 on the CPU, gradients by autograd through the in-place clamp) on the SAME noise draws.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _reference(mu, log_var, eps, rate, clamp):
    """The reference loop, literally, in float64."""
    mu = mu.double().requires_grad_(True)
    lv = log_var.double().requires_grad_(True)
    terms = []
    for s in range(eps.shape[0]):
        std = torch.exp(0.5 * lv)
        z = eps[s].double() * std + mu
        z[z <= 0.0] = clamp
        log_p = torch.distributions.Exponential(rate=torch.tensor([rate], dtype=torch.float64)).log_prob(z)
        log_q = torch.distributions.Normal(mu, std).log_prob(z)
        terms.append(log_q - log_p)
    per_elem = torch.stack(terms, dim=-1).mean(dim=-1)        # (B, D): the kernel's output
    return per_elem, mu, lv


@pytest.mark.parametrize("B,D,S", [(33, 12, 100), (7, 4, 1), (1000, 8, 17)])
def test_fused_mc_kl_matches_reference_loop(B, D, S):
    dev = _dev()
    from hode.mckl import mc_kl_exponential
    g = torch.Generator().manual_seed(B + S)
    # posterior means around the prior scale with a good share of non-positive draws (exercises the clamp branch)
    mu = torch.randn(B, D, generator=g) * 0.02 + 0.01
    lv = torch.randn(B, D, generator=g) * 0.5 - 8.0
    eps = torch.randn(S, B, D, generator=g)
    clamp = float(torch.finfo(torch.float32).eps)
    ref, mu_r, lv_r = _reference(mu, lv, eps, 100.0, clamp)
    w = torch.randn(B, D, generator=g).double()
    (ref * w).sum().backward()
    mu_g, lv_g = mu.to(dev).requires_grad_(True), lv.to(dev).requires_grad_(True)
    out = mc_kl_exponential(mu_g, lv_g, eps.to(dev), 100.0, clamp)
    (out * w.float().to(dev)).sum().backward()
    frac_clamped = float(((eps * torch.exp(0.5 * lv) + mu) <= 0).float().mean())
    assert 0.02 < frac_clamped < 0.98 or S == 1
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(mu_g.grad.cpu().numpy(), mu_r.grad.numpy(), rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(lv_g.grad.cpu().numpy(), lv_r.grad.numpy(), rtol=2e-4, atol=1e-4)


def test_vi_loss_uses_the_fused_kl_and_agrees_with_the_eager_path():
    """VariationalInference.loss with the Exponential prior: fused KL vs the eager torch ops (same seed -> same draws)."""
    dev = _dev()
    import model
    obs, D, T, B, step = 40, 8, 12, 64, 0.125
    torch.manual_seed(3)
    enc = model.EncoderLSTM(obs + 1, 2 * obs, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density)
    g = torch.Generator().manual_seed(5)
    data = {"measurements": torch.randn(T, B, obs, generator=g).to(dev),
            "masks": (torch.rand(T, B, obs, generator=g) < 0.5).float().to(dev),
            "actions": torch.zeros(T, B, 1).to(dev)}
    data["actions"][3, :, 0] = 2.0
    out = []
    for fused in (True, False):
        vi.fuse_mc_kl = fused
        for p in vi.parameters():
            p.grad = None
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        loss = vi.loss(data)
        loss.backward()
        out.append((loss.item(), enc.lin.weight.grad.clone(), enc.log_var.weight.grad.clone()))
    assert abs(out[0][0] - out[1][0]) <= 1e-5 * abs(out[1][0])
    for a, b in zip(out[0][1:], out[1][1:]):
        assert float((a - b).norm() / (b.norm() + 1e-30)) <= 1e-4
