"""hode_ensemble_crps (through the C ABI) vs the CPU oracle; evaluate() end to end on the GPU.  GPU only.

Tolerance: the kernel sums <= 1225 fp32 terms per element in a fixed order; rel 2e-5 / abs 2e-6 against the fp64 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.evalmetrics import crps_field


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _oracle(h, truth, M, w=None, b=None):
    Tn, MB, Dv = h.shape
    B, obs = MB // M, truth.shape[-1]
    v = h.reshape(Tn, M, B, Dv).double()
    vals = (v @ w.double().t() + (b.double() if b is not None else 0.0)) if w is not None else v[..., :obs]
    return torch.from_numpy(crps_field(truth.double().numpy(), vals.permute(0, 2, 3, 1).numpy())).float()


@pytest.mark.parametrize("M", [1, 2, 10, 50])
@pytest.mark.parametrize("obs,D", [(80, 12), (40, 8), (4, 12), (24, 20)])
def test_linear_readout_crps_vs_oracle(M, obs, D):
    dev = _dev()
    from hode.crps import ensemble_crps
    g = torch.Generator().manual_seed(M * 100 + obs)
    Tn, B = 5, 7
    h = torch.randn(Tn, M * B, D, generator=g)
    w, b = torch.randn(obs, D, generator=g) * 0.5, torch.randn(obs, generator=g)
    truth = torch.randn(Tn, B, obs, generator=g)
    ref = _oracle(h, truth, M, w, b)
    full = ensemble_crps(h.to(dev), truth.to(dev), M, weight=w.to(dev), bias=b.to(dev), per_component=True).cpu()
    np.testing.assert_allclose(full.numpy(), ref.numpy(), rtol=2e-5, atol=2e-6)
    summed = ensemble_crps(h.to(dev), truth.to(dev), M, weight=w.to(dev), bias=b.to(dev)).cpu()
    np.testing.assert_allclose(summed.numpy(), ref.sum(-1).numpy(), rtol=2e-5, atol=2e-5)
    if M == 1:  # one member: plain absolute error
        x = h @ w.t() + b
        np.testing.assert_allclose(full.numpy(), (x - truth).abs().numpy(), rtol=1e-5, atol=1e-6)


def test_identity_readout_and_time_slice_view():
    dev = _dev()
    from hode.crps import ensemble_crps
    g = torch.Generator().manual_seed(1)
    M, B, D, Tn = 9, 11, 12, 6
    h = torch.randn(Tn, M * B, D, generator=g)
    truth = torch.randn(Tn, B, 4, generator=g)
    got = ensemble_crps(h.to(dev), truth.to(dev), M, per_component=True).cpu()       # first 4 components, no readout
    np.testing.assert_allclose(got.numpy(), _oracle(h, truth, M).numpy(), rtol=2e-5, atol=2e-6)
    hd = h.to(dev)
    got2 = ensemble_crps(hd[2:], truth[2:].to(dev), M, per_component=True).cpu()       # what evaluate() passes: h[t0:]
    np.testing.assert_allclose(got2.numpy(), got[2:].numpy(), rtol=0, atol=0)


def test_argument_errors():
    dev = _dev()
    import hode
    from hode.crps import ensemble_crps
    with pytest.raises(ValueError):
        ensemble_crps(torch.zeros(1, 7, 4, device=dev), torch.zeros(1, 2, 4, device=dev), 3)
    with pytest.raises(hode.HodeConfigError):
        ensemble_crps(torch.zeros(1, 2 * 129, 4, device=dev), torch.zeros(1, 2, 4, device=dev), 129)
    with pytest.raises(hode.HodeConfigError):  # identity readout asks for more components than a member vector has
        ensemble_crps(torch.zeros(1, 4, 3, device=dev), torch.zeros(1, 2, 5, device=dev), 2)


def test_evaluate_on_gpu_matches_oracle_crps_on_the_same_draws(monkeypatch, capsys):
    """The product path (HIP LSTM encoder, HIP solver over mc_itr * B latents, HIP CRPS) against the same flow with the
    CRPS kernel swapped for the oracle: identical draws (same seed), so every returned number agrees to fp32 noise."""
    dev = _dev()
    import model
    import training_utils
    from test_evaluate import FakeGenerator, oracle_ensemble_crps
    obs, D, T, step, t0 = 40, 8, 20, 0.125, 8
    dg = FakeGenerator(12, T, obs, D, seed=4, step=step)
    dg.data = {k: v.to(dev) for k, v in dg.data.items()}
    torch.manual_seed(21)
    enc = model.EncoderLSTM(obs + 1, 2 * obs, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, elbo=True)
    torch.manual_seed(5)
    torch.cuda.manual_seed(5)
    got = training_utils.evaluate(vi, dg, 6, t0, mc_itr=50)

    def oracle_on_gpu_tensors(h, truth, M, weight=None, bias=None, per_component=False):
        c = oracle_ensemble_crps(h.cpu(), truth.cpu(), M, None if weight is None else weight.cpu(),
                                 None if bias is None else bias.cpu(), per_component)
        return c.to(h.device)

    monkeypatch.setattr(training_utils, "_ensemble_crps", oracle_on_gpu_tensors)
    torch.manual_seed(5)
    torch.cuda.manual_seed(5)
    ref = training_utils.evaluate(vi, dg, 6, t0, mc_itr=50)
    capsys.readouterr()
    np.testing.assert_allclose(np.array(got), np.array(ref), rtol=2e-5)
    hz = training_utils.evaluate_horizon(vi, dg, 6, t0, mc_itr=10)
    assert hz["cprs_x"].shape == (T - t0,) and np.all(np.isfinite(hz["cprs_x"]))
