"""Sizes at and past the limits of the layouts: the long-grid fallback (the split layout stages the time grid in LDS, 8 192 points),
a batch sixteen times the bench's, a ragged last workgroup at that size.  No oracle run at these sizes: the checks are properties --
two independent kernel layouts agree, every value is finite, patients are independent of the batch they sit in.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _solve(inp, w, b, theta, dev, lanes, cot, sl=slice(None)):
    from hode.solver import roche_solve
    from hode import synth
    chan = inp["actions"][..., 0][:, sl]
    dosage = chan.max(dim=0)[0]
    times = (torch.nonzero((chan != 0).t())[:, 1].reshape(chan.shape[1], -1) * synth.STEP).float()
    y0 = inp["z0"][sl].to(dev).requires_grad_(True)
    wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    h = roche_solve(y0, theta.to(dev), wg, bg, inp["t"].to(dev), dosage.to(dev), times.to(dev), method="rk4", lanes_per_patient=lanes)
    (h * cot[:, sl].to(dev)).sum().backward()
    return h.detach(), y0.grad, wg.grad, bg.grad


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


THETA = torch.tensor((2.0, 2.0) + (1.0,) * 11 + (0.0,) * 3)


def test_grid_longer_than_the_split_layouts_lds_stage_takes_the_quad_layout_and_agrees():
    """T = 8 193 > 8 192: lanes_per_patient = 0 (auto) must fall back to the quad layout and give what lanes_per_patient = 4 gives;
    T = 8 192 still runs the split layout and agrees with the quad layout to rounding."""
    from hode import synth
    dev = _dev()
    w, b = synth.default_ml_weights(12)
    for T in (8193, 8192):
        inp = synth.solver_inputs(64, T, 12, seed=5)
        inp["t"] = torch.arange(T, dtype=torch.float32) * (12.375 / (T - 1))   # the bench's time span on a fine grid
        cot = torch.randn(T, 64, 12, generator=torch.Generator().manual_seed(1)) / T
        auto = _solve(inp, w, b, THETA, dev, 0, cot)
        quad = _solve(inp, w, b, THETA, dev, 4, cot)
        assert all(bool(torch.isfinite(t).all()) for t in auto)
        if T > 8192:
            for a, q in zip(auto, quad):
                assert torch.equal(a, q)            # the same kernels ran
        else:
            assert _rel(auto[0], quad[0]) <= 1e-5 and _rel(auto[1], quad[1]) <= 1e-4 and _rel(auto[2], quad[2]) <= 1e-4


def test_batch_of_160_000_with_a_ragged_last_workgroup_is_finite_and_patientwise_identical_to_a_small_batch():
    """160 001 patients (3 334 workgroups, the last one holding a single patient): finite everywhere, and the first 96 and the last 49
    patients get bit for bit the trajectories and initial-state gradients they get in a batch of their own (patients are
    independent; the parameter gradients are sums over the batch and are compared with the small batches' only for finiteness)."""
    from hode import synth
    dev = _dev()
    N, T = 160001, 100
    w, b = synth.default_ml_weights(12)
    inp = synth.solver_inputs(N, T, 12, seed=9)
    cot = torch.randn(T, N, 12, generator=torch.Generator().manual_seed(2))
    h, gy0, gw, gb = _solve(inp, w, b, THETA, dev, 0, cot)
    assert bool(torch.isfinite(h).all()) and bool(torch.isfinite(gy0).all()) and bool(torch.isfinite(gw).all()) and bool(torch.isfinite(gb).all())
    for sl in (slice(0, 96), slice(N - 49, N)):
        hs, gs, _, _ = _solve(inp, w, b, THETA, dev, 0, cot, sl)
        assert torch.equal(hs, h[:, sl]) and torch.equal(gs, gy0[sl])


def test_encoder_at_100_000_patients_matches_itself_in_chunks():
    """LSTM encoder + BPTT at ten times the bench batch (T = 20): the final hidden state of every patient equals what the same
    patient gets in a batch of 48 (the kernels' tile), bit for bit, and the weight gradients are finite."""
    from hode.lstm import lstm_encode
    dev = _dev()
    T, B, obs, H = 20, 100003, 80, 160
    g = torch.Generator().manual_seed(3)
    x = torch.randn(T, B, obs, generator=g).to(dev)
    a = torch.rand(T, B, 1, generator=g).to(dev)
    m = (torch.rand(T, B, obs, generator=g) < 0.5).float().to(dev)
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(obs + 1, H).to(dev)
    prm = [lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0]
    h = lstm_encode(x, a, m, *prm, reverse=True)
    h.sum().backward()
    assert bool(torch.isfinite(h).all()) and all(bool(torch.isfinite(p.grad).all()) for p in prm)
    for s in (slice(0, 48), slice(B - 48, B), slice(B - 3, B)):
        with torch.no_grad():
            hs = lstm_encode(x[:, s].contiguous(), a[:, s].contiguous(), m[:, s].contiguous(), *prm, reverse=True)
        assert torch.equal(hs, h[s].detach())
