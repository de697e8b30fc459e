"""Fused readout + masked-SSE kernel vs the plain torch expression of reference model.py:1120 + :1179.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("D,obs,T,B", [(12, 80, 7, 33), (8, 40, 5, 100), (6, 20, 4, 17), (4, 20, 3, 5), (12, 80, 100, 257)])
def test_loss_and_gradients(D, obs, T, B):
    from hode.readout import masked_sse_readout
    dev = _dev()
    gen = torch.Generator().manual_seed(D + B)
    h = torch.randn(T, B, D, generator=gen)
    x = torch.randn(T, B, obs, generator=gen)
    m = (torch.rand(T, B, obs, generator=gen) < 0.5).float()
    lin = torch.nn.Linear(D, obs)
    hr = h.clone().double().requires_grad_(True)
    w64, b64 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    ref = torch.sum((x.double() - (hr @ w64.t() + b64)) ** 2 * m.double()) / B
    (ref * 1.7).backward()
    hg = h.to(dev).requires_grad_(True)
    wg, bg = lin.weight.detach().to(dev).requires_grad_(True), lin.bias.detach().to(dev).requires_grad_(True)
    lik = masked_sse_readout(hg, x.to(dev), m.to(dev), wg, bg)
    (lik * 1.7).backward()
    assert abs(lik.item() - ref.item()) <= 2e-5 * abs(ref.item())
    for got, want in ((hg.grad, hr.grad), (wg.grad, w64.grad), (bg.grad, b64.grad)):
        rel = float((got.cpu().double() - want).norm() / want.norm())
        assert rel <= 2e-5, rel


def test_vi_loss_uses_fused_path_and_matches_unfused():
    import model
    from hode import synth
    dev = _dev()
    obs, D, T, B = 80, 12, 16, 64
    torch.manual_seed(5)
    enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, elbo=False)
    sol = synth.solver_inputs(B, T, D, seed=6)
    ob = synth.observation_inputs(B, T, obs, seed=6)
    data = {k: v.to(dev) for k, v in {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}.items()}
    l1 = vi.loss(data)
    l1.backward()
    g1 = [p.grad.clone() for p in vi.parameters() if p.grad is not None]
    assert vi._x_hat is None and vi.x_hat.shape == (T, B, obs)  # lazily produced on access
    for p in vi.parameters():
        p.grad = None
    vi.fuse_likelihood = False
    l2 = vi.loss(data)
    l2.backward()
    g2 = [p.grad.clone() for p in vi.parameters() if p.grad is not None]
    assert abs(l1.item() - l2.item()) <= 2e-5 * abs(l2.item())
    assert len(g1) == len(g2)
    for a, b in zip(g1, g2):
        assert float((a - b).norm() / (b.norm() + 1e-20)) <= 2e-4
