"""training_utils.evaluate / evaluate_horizon / bootstrap_RMSE (mirror of the reference's training_utils.py:100-279,
:568-577): host logic on the CPU with the solver and the CRPS kernel replaced by their oracles (test-only hooks), against
a literal restatement of the reference's per-sample loops on the SAME random draws."""
import numpy as np
import pytest
import torch

import model
import training_utils
from oracle.evalmetrics import crps_field, evaluate_reference
from oracle.solvers import odeint as oracle_odeint

CPU = torch.device("cpu")


def oracle_ensemble_crps(h, truth, n_members, weight=None, bias=None, per_component=False):
    """CPU stand-in with hode.crps.ensemble_crps' contract (member-major batch axis)."""
    Tn, MB, Dv = h.shape
    M, B, obs = n_members, MB // n_members, truth.shape[-1]
    v = h.reshape(Tn, M, B, Dv).double()
    vals = (v @ weight.double().t() + (bias.double() if bias is not None else 0.0)) if weight is not None else v[..., :obs]
    c = torch.from_numpy(crps_field(truth.double().numpy(), vals.permute(0, 2, 3, 1).numpy())).float()
    return c if per_component else c.sum(-1)


class FakeGenerator:
    expert_dim = 4

    def __init__(self, n, T, obs, D, seed, step):
        g = torch.Generator().manual_seed(seed)
        self.test_size = n
        self.data = {
            "measurements": torch.randn(T, n, obs, generator=g),
            "masks": (torch.rand(T, n, obs, generator=g) < 0.6).float(),
            "latents": torch.rand(T, n, D, generator=g) * 0.05,
            "actions": torch.zeros(T, n, 1),
        }
        idx = torch.randint(0, T - 1, (n,), generator=g)
        self.data["actions"][idx, torch.arange(n), 0] = torch.rand(n, generator=g) * 5 + 0.5

    def get_split(self, fold, bs, chunk=0):
        assert fold == "test"
        return {k: v[:, chunk * bs:(chunk + 1) * bs] for k, v in self.data.items()}


def _model(obs, D, T, step, method="rk4"):
    torch.manual_seed(11)
    enc = model.EncoderLSTM(obs + 1, 2 * obs, D, device=CPU)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, method=method, device=CPU)
    dec._odeint = oracle_odeint
    return model.VariationalInference(enc, dec, elbo=True, mc_size=3)


def test_posterior_scores_match_the_reference_loops(monkeypatch):
    obs, D, T, step, t0, M, bs = 6, 8, 10, 0.125, 4, 5, 3
    dg = FakeGenerator(6, T, obs, D, seed=2, step=step)
    vi = _model(obs, D, T, step)
    monkeypatch.setattr(training_utils, "_ensemble_crps", oracle_ensemble_crps)
    for chunk in range(2):
        data = dg.get_split("test", bs, chunk)
        with torch.no_grad():
            torch.manual_seed(77 + chunk)
            se_z0, sse_x, n_x, crps_z0, crps_x = training_utils._posterior_scores(vi, data, t0, M, False, dg.expert_dim)
            # the reference's flow, literally (training_utils.py:117-176): point estimate, then mc_itr x {draw, decode}
            torch.manual_seed(77 + chunk)
            enc_out = vi.encoder(data["measurements"][:t0], data["actions"][:t0], data["masks"][:t0])
            x_hat, _ = vi.decoder(enc_out[0], data["actions"])
            zs, xs = [], []
            for _ in range(M):
                z_ = vi.encoder.reparameterize(*enc_out)
                xh, _ = vi.decoder(z_, data["actions"])
                zs.append(z_)
                xs.append(xh[t0:])
            ref = evaluate_reference(data["latents"][0], enc_out[0], data["measurements"][t0:], data["masks"][t0:],
                                     x_hat[t0:], zs, xs, dg.expert_dim)
        np.testing.assert_allclose(se_z0.numpy(), ref["se_z0"].numpy(), rtol=1e-6)
        np.testing.assert_allclose((sse_x.sum(0) / n_x.sum(0)).numpy(), ref["mse_x"].numpy(), rtol=1e-5)
        np.testing.assert_allclose(crps_z0.numpy(), ref["crps_z0"].numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(crps_x.mean(0).numpy(), ref["crps_x"].numpy(), rtol=2e-5, atol=1e-7)


def test_evaluate_prints_and_returns_like_the_reference(monkeypatch, capsys):
    obs, D, T, step, t0 = 6, 8, 10, 0.125, 4
    dg = FakeGenerator(6, T, obs, D, seed=3, step=step)
    vi = _model(obs, D, T, step)
    monkeypatch.setattr(training_utils, "_ensemble_crps", oracle_ensemble_crps)
    torch.manual_seed(5)
    out = training_utils.evaluate(vi, dg, 3, t0, mc_itr=4)
    lines = capsys.readouterr().out.strip().split("\n")
    assert [l.split(",")[0] for l in lines] == ["rmse_z0", "rmse_x", "cprs_z0", "cprs_x"]
    assert len(out) == 6 and all(np.isfinite(v) for v in out)
    assert abs(float(lines[0].split(",")[1]) - out[0]) < 1e-4 and abs(float(lines[3].split(",")[1]) - out[5]) < 1e-4
    torch.manual_seed(5)
    hz = training_utils.evaluate_horizon(vi, dg, 3, t0, mc_itr=4)
    assert set(hz) == {"rmse_x", "rmse_x_sd", "cprs_x", "cprs_x_sd"} and hz["rmse_x"].shape == (T - t0,)
    assert hz["cprs_x"].shape == (T - t0,) and np.all(hz["cprs_x"] > 0)


def test_bootstrap_rmse_is_the_reference_estimator():
    err = torch.rand(200, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(9)
    got = training_utils.bootstrap_RMSE(err)
    torch.manual_seed(9)  # the reference's loop, literally (training_utils.py:568-577)
    ref = np.std(np.array([torch.sqrt(torch.mean(err[torch.randint(len(err), err.shape)])).item() for _ in range(500)]))
    assert got == ref
    assert abs(training_utils.bootstrap_RMSE(err.numpy()) - ref) < 0.2 * ref  # ndarray input accepted; different draws


def test_product_crps_refuses_cpu_tensors():
    import hode
    from hode import crps
    with pytest.raises(hode.HodeConfigError):
        crps.ensemble_crps(torch.zeros(1, 4, 3), torch.zeros(1, 2, 3), 2)
