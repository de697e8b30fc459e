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


@pytest.mark.parametrize("D,obs,T,B", [(12, 80, 9, 21), (12, 72, 3, 50), (12, 52, 2, 16), (8, 40, 6, 19), (8, 48, 4, 31), (8, 36, 2, 7)])
def test_matrix_core_variant_edges(D, obs, T, B):
    """The matrix-core kernel (D = 12 / 48 < obs <= 80, D = 8 / 32 < obs <= 48): partial last output tile, row counts that
    are not a multiple of 16, general (non 0/1) masks, the loss-only launch, and agreement with the lane-per-4-outputs
    kernel (HODE_READOUT_VALU) on the same inputs."""
    import os
    from hode.readout import masked_sse_readout
    dev = _dev()
    gen = torch.Generator().manual_seed(obs + B)
    h = torch.randn(T, B, D, generator=gen)
    x = torch.randn(T, B, obs, generator=gen)
    m = torch.rand(T, B, obs, generator=gen) * (torch.rand(T, B, obs, generator=gen) < 0.6).float()
    lin = torch.nn.Linear(D, obs)
    hr = h.clone().double().requires_grad_(True)
    w64, b64 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    ref = torch.sum((x.double() - (hr @ w64.t() + b64)) ** 2 * m.double()) / B
    ref.backward()

    def run():
        hg = h.to(dev).requires_grad_(True)
        wg, bg = lin.weight.detach().to(dev).requires_grad_(True), lin.bias.detach().to(dev).requires_grad_(True)
        lik = masked_sse_readout(hg, x.to(dev), m.to(dev), wg, bg)
        lik.backward()
        with torch.no_grad():
            lik0 = masked_sse_readout(hg.detach(), x.to(dev), m.to(dev), wg.detach(), bg.detach())
        return lik.item(), lik0.item(), hg.grad.cpu().double(), wg.grad.cpu().double(), bg.grad.cpu().double()

    got = run()
    os.environ["HODE_READOUT_VALU"] = "1"
    try:
        alt = run()
    finally:
        del os.environ["HODE_READOUT_VALU"]
    for res in (got, alt):
        assert abs(res[0] - ref.item()) <= 2e-5 * abs(ref.item())
        assert abs(res[1] - ref.item()) <= 2e-5 * abs(ref.item())
        for g, want in zip(res[2:], (hr.grad, w64.grad, b64.grad)):
            assert float((g - want).norm() / want.norm()) <= 2e-5


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


@pytest.mark.parametrize("D,T,B,weighted", [(20, 7, 33, False), (20, 5, 100, True), (4, 6, 17, True), (20, 96, 257, False), (20, 1, 1, False)])
def test_mlp_readout_loss_and_gradients(D, T, B, weighted):
    """hode_readout_mlp_sse (DecoderReal's Linear(D, D+1) -> ELU -> Linear(D+1, 24) readout fused with the masked,
    optionally time-weighted SSE of VariationalInferenceReal.loss, reference model.py:809-813 / :859 / :1243-1247) against
    the plain torch expression in float64; row counts that are not multiples of the 16-row tile, D = 4 (expert-only)."""
    from hode.readout import masked_sse_readout_mlp
    dev = _dev()
    obs = 24
    gen = torch.Generator().manual_seed(D + B)
    torch.manual_seed(D + T)
    h = torch.randn(T, B, D, generator=gen)
    x = torch.randn(T, B, obs, generator=gen)
    m = (torch.rand(T, B, obs, generator=gen) < 0.5).float()
    net = torch.nn.Sequential(torch.nn.Linear(D, D + 1), torch.nn.ELU(), torch.nn.Linear(D + 1, obs))
    tw = 1 / torch.arange(1, T + 1, dtype=torch.float32) if weighted else None
    net64 = torch.nn.Sequential(torch.nn.Linear(D, D + 1), torch.nn.ELU(), torch.nn.Linear(D + 1, obs)).double()
    net64.load_state_dict(net.state_dict())
    hr = h.clone().double().requires_grad_(True)
    wref = tw.double()[:, None, None] if weighted else 1.0
    ref = torch.sum((x.double() - net64(hr)) ** 2 * m.double() * wref) / B
    (ref * 1.3).backward()
    hg = h.to(dev).requires_grad_(True)
    prm = [p.detach().clone().to(dev).requires_grad_(True) for p in (net[0].weight, net[0].bias, net[2].weight, net[2].bias)]
    lik = masked_sse_readout_mlp(hg, x.to(dev), m.to(dev), *prm, None if tw is None else tw.to(dev))
    (lik * 1.3).backward()
    assert abs(lik.item() - ref.item()) <= 2e-5 * abs(ref.item())
    wants = (hr.grad, net64[0].weight.grad, net64[0].bias.grad, net64[2].weight.grad, net64[2].bias.grad)
    for got, want in zip([hg.grad] + [p.grad for p in prm], wants):
        rel = float((got.cpu().double() - want).norm() / want.norm())
        assert rel <= 3e-5, rel
    # loss only (no gradient requested): same value, nothing else touched
    with torch.no_grad():
        lik2 = masked_sse_readout_mlp(h.to(dev), x.to(dev), m.to(dev), *[p.detach() for p in prm], None if tw is None else tw.to(dev))
    assert abs(lik2.item() - lik.item()) <= 1e-6 * abs(lik.item())


@pytest.mark.parametrize("D,B", [(20, 33), (4, 17), (20, 257)])
def test_mlp_readout_skip_rows_equals_slicing(D, B):
    """skip_rows=1 (DecoderReal: the state at t0 - 1 is not read out) gives the same loss and gradients as passing h[1:]:
    bit for bit, with an exactly zero gradient row 0.  (4, 17): 68 floats per row -- aligned; the unaligned case falls back
    to slicing and is covered by the equality itself.)"""
    from hode.readout import masked_sse_readout_mlp
    dev = _dev()
    T, obs = 6, 24
    gen = torch.Generator().manual_seed(100 + B)
    h = torch.randn(T + 1, B, D, generator=gen).to(dev)
    x = torch.randn(T, B, obs, generator=gen).to(dev)
    m = (torch.rand(T, B, obs, generator=gen) < 0.5).float().to(dev)
    torch.manual_seed(D)
    net = torch.nn.Sequential(torch.nn.Linear(D, D + 1), torch.nn.ELU(), torch.nn.Linear(D + 1, obs)).to(dev)
    tw = (1 / torch.arange(1, T + 1, dtype=torch.float32)).to(dev)
    outs = []
    for skip in (True, False):
        hg = h.clone().requires_grad_(True)
        prm = [p.detach().clone().requires_grad_(True) for p in (net[0].weight, net[0].bias, net[2].weight, net[2].bias)]
        lik = masked_sse_readout_mlp(hg, x, m, *prm, tw, skip_rows=1) if skip else masked_sse_readout_mlp(hg[1:], x, m, *prm, tw)
        (0.7 * lik).backward()
        outs.append([lik.detach(), hg.grad] + [p.grad for p in prm])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert not outs[0][1][0].any()


def test_real_vi_loss_uses_fused_mlp_readout_and_matches_unfused():
    import model
    dev = _dev()
    obs, act, stat, D, T, t0, B = 24, 1, 11, 20, 40, 24, 70
    input_dim = obs + act + stat + 1
    torch.manual_seed(3)
    enc = model.EncoderLSTMReal(input_dim, int(input_dim * 1.2), D, output_all=False, reverse=False, device=dev)
    dec = model.DecoderReal(obs, D, act, stat, int((obs + act + stat) * 1.2), T, 1, method="midpoint", ode_step_size=1.0, ode_type="hybrid", t0=t0, device=dev)
    gen = torch.Generator().manual_seed(4)
    data = {k: v.to(dev) for k, v in {
        "measurements": torch.randn(T, B, obs, generator=gen),
        "actions": (torch.rand(T, B, 1, generator=gen) < 0.1).float() * torch.rand(T, B, 1, generator=gen),
        "masks": (torch.rand(T, B, obs, generator=gen) < 0.5).float(), "statics": torch.rand(T, B, stat, generator=gen)}.items()}
    for weight in (False, True):
        vi = model.VariationalInferenceReal(enc, dec, elbo=False, t0=t0, weight=weight)
        for p in vi.parameters():
            p.grad = None
        l1 = vi.loss(data)
        l1.backward()
        g1 = [p.grad.clone() for p in vi.parameters() if p.grad is not None]
        assert vi._x_hat is None and vi.x_hat.shape == (T - t0, B, obs)
        for p in vi.parameters():
            p.grad = None
        vi.fuse_likelihood = False
        l2 = vi.loss(data)
        l2.backward()
        g2 = [p.grad.clone() for p in vi.parameters() if p.grad is not None]
        assert abs(l1.item() - l2.item()) <= 2e-5 * abs(l2.item())
        assert len(g1) == len(g2)
        for a, b in zip(g1, g2):
            assert float((a - b).norm() / (b.norm() + 1e-20)) <= 3e-4
