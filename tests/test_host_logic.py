"""CPU-only tests of the host-side mirror (model.py / training_utils.py / hode.parallel): call surface, state_dict
keys, vectorised set_action, loss assembly (with the CPU oracle injected as the solver -- tests only), failure modes."""
import os

import numpy as np
import pytest
import torch

import model
from oracle.solvers import odeint as oracle_odeint

CPU = torch.device("cpu")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_state_dict_keys_and_param_order_match_reference_layout():
    torch.manual_seed(0)
    enc = model.EncoderLSTM(81, 160, 12, device=CPU)
    dec = model.RocheExpertDecoder(80, 12, 1, 12.375, 0.125, method="rk4", device=CPU)
    assert list(enc.state_dict()) == ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
                                      "lin.weight", "lin.bias", "log_var.weight", "log_var.bias"]
    keys = list(dec.state_dict())
    assert keys[:2] == ["output_function.0.weight", "output_function.0.bias"]
    assert keys[2:15] == ["ode." + n for n in ("HillCure", "HillPatho", "ec50_patho", "emax_patho", "k_dexa",
                                               "k_discure_immunereact", "k_discure_immunity", "k_disprog", "k_immune_disease",
                                               "k_immune_feedback", "k_immune_off", "k_immunity", "kel")]
    assert keys[15:] == ["ode.ml_net.0.weight", "ode.ml_net.0.bias"]
    assert sum(p.numel() for p in enc.parameters()) == 159384 + 0 and dec.output_function[0].weight.shape == (80, 12)
    vi = model.VariationalInference(enc, dec)
    assert vi.model_name == "VI_LSTMEncoder_HybridDecoder.pkl"
    assert model.RocheExpertDecoder(40, 4, 1, 6.125, 0.125, device=CPU).model_name == "ExpertDecoder"
    assert dec.t.shape == (100,) and dec.t.dtype == torch.float32


def test_set_action_matches_reference_golden(golden_dir):
    g = _load(golden_dir, "g1_roche_rhs.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        D, ablate, T, B = [int(v) for v in g[pre + "meta"]]
        ode = model.RocheODE(D, 1, (T - 1) * 0.125, float(g[pre + "step"]), ablate=bool(ablate), device=CPU)
        ode.set_action(torch.from_numpy(g[pre + "action"]))
        np.testing.assert_array_equal(ode.times.numpy(), g[pre + "times"])
        np.testing.assert_array_equal(ode.dosage.numpy(), g[pre + "dosage"])
        assert ode.times.dtype == torch.float32
    ode = model.RocheODE(8, 1, 14, 1, device=CPU)
    ode.set_action(torch.from_numpy(g["int_step_action"]))
    assert ode.times.dtype == torch.int64
    np.testing.assert_array_equal(ode.times.numpy(), g["int_step_times"])
    bad = torch.zeros(6, 2, 1)
    bad[1, 0, 0] = bad[2, 1, 0] = bad[4, 1, 0] = 1.0
    with pytest.raises(RuntimeError):
        ode.set_action(bad)


def test_rhs_as_torch_function_matches_reference_golden(golden_dir):
    g = _load(golden_dir, "g1_roche_rhs.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        D, ablate, T, B = [int(v) for v in g[pre + "meta"]]
        ode = model.RocheODE(D, 1, (T - 1) * 0.125, float(g[pre + "step"]), ablate=bool(ablate), device=CPU)
        sd = {k[len(pre + "sd_"):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + "sd_")}
        ode.load_state_dict(sd)
        ode.set_action(torch.from_numpy(g[pre + "action"]))
        y = torch.from_numpy(g[pre + "y"])
        for ti, t in enumerate(g[pre + "t"]):
            out = ode(torch.tensor(float(t)), y)
            np.testing.assert_allclose(out.detach().numpy(), g[pre + "f"][ti], rtol=2e-6, atol=1e-6, equal_nan=True)


def test_product_decoder_refuses_cpu_and_missing_action():
    dec = model.RocheExpertDecoder(10, 8, 1, 1.0, 0.125, method="rk4", device=CPU)
    import hode
    with pytest.raises(hode.HodeConfigError, match="no CPU fallback"):
        dec(torch.rand(3, 8), torch.zeros(9, 3, 1))
    dec = model.RocheExpertDecoder(10, 8, 1, 1.0, 0.125, method="rk4", device=CPU)
    with pytest.raises(RuntimeError, match="set_action"):
        dec.ode.hode_solve(torch.rand(3, 8), dec.t, 1e-7, 1e-8, "rk4", {})
    with pytest.raises(RuntimeError):
        model.EncoderLSTM(11, 20, 8, device=CPU)(torch.rand(5, 1, 10), torch.rand(5, 1, 1), torch.ones(5, 1, 10))


def test_vi_loss_against_reference_golden_with_oracle_solver_injected(golden_dir):
    """G5: everything around the solver (encoder, set_action, readout, masked SSE, analytic KL) equals the reference
    when the decoder's solver hook is pointed at the CPU oracle (test-only injection)."""
    g = _load(golden_dir, "g5_vi_loss.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        obs, D, T, B, seed = [int(v) for v in g[pre + "meta"]]
        method, mode = str(g[pre + "method"]), str(g[pre + "mode"])
        if mode == "kl_exp":
            continue  # MC-KL draws come in a different RNG order (batched); covered statistically below
        step = float(g[pre + "step"])
        enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=CPU)
        dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, method=method, device=CPU)
        dec._odeint = oracle_odeint
        enc.load_state_dict({k[len(pre + "enc_"):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + "enc_")})
        dec.load_state_dict({k[len(pre + "dec_"):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + "dec_")})
        vi = model.VariationalInference(enc, dec, elbo=(mode != "lik"), prior_log_pdf=None, mc_size=7)
        data = {k2: torch.from_numpy(g[pre + k]) for k, k2 in (("x", "measurements"), ("a", "actions"), ("mask", "masks"))}
        torch.manual_seed(seed)
        loss = vi.loss(data)
        np.testing.assert_allclose(loss.item(), float(g[pre + "loss"]), rtol=3e-5)
        np.testing.assert_allclose(vi.h_hat.detach().numpy(), g[pre + "h_hat"], rtol=1e-3, atol=2e-5)


def test_mc_kl_is_statistically_the_reference_estimator():
    torch.manual_seed(0)
    enc = model.EncoderLSTM(5, 8, 3, device=CPU)
    mu = torch.full((4000, 3), 0.05)
    log_var = torch.full((4000, 3), -7.0)
    v = model.VariationalInference.__new__(model.VariationalInference)
    v.encoder, v.prior_log_pdf = enc, model.ExponentialPrior.log_density
    est = v.mc_kl(mu, log_var, 50).mean().item()
    # closed form for draws that stay positive: E[log q] - E[log p] = -0.5*(1+log(2 pi s^2))*3 - 3*(log 100 - 100*mu)
    s2 = float(torch.exp(torch.tensor(-7.0)))
    want = 3 * (-0.5 * (1 + np.log(2 * np.pi * s2))) - 3 * (np.log(100.0) - 100 * 0.05)
    assert abs(est - want) < 0.05 * abs(want)


def test_prior_and_posterior_densities():
    z = torch.rand(5, 3) + 0.1
    want = torch.distributions.Exponential(torch.tensor([100.0])).log_prob(z).sum(-1)
    torch.testing.assert_close(model.ExponentialPrior.log_density(z), want)
    mu, lv = torch.randn(5, 3), torch.randn(5, 3)
    want = torch.distributions.Normal(mu, torch.exp(0.5 * lv)).log_prob(z).sum(-1)
    torch.testing.assert_close(model.GaussianReparam.log_density(mu, lv, z), want)
    want = torch.distributions.Normal(0.0, 1.0).log_prob(z).sum(-1)
    torch.testing.assert_close(model.StandardNormalPrior.log_density(z), want)


def test_training_loop_control_flow(tmp_path, capsys):
    """Early stop, best-on-disk checkpoint and reload, with a stub model (no solver involved)."""
    import training_utils

    class DG:
        train_size, val_size = 20, 10

        def get_mini_batch(self, fold, bs):
            return {"x": torch.ones(1)}

        def get_split(self, fold, bs, chunk=0):
            return {"x": torch.ones(1)}

    class Stub:
        model_name = "VI_stub.pkl"

        def __init__(self):
            self.encoder = torch.nn.Linear(1, 1)
            self.decoder = torch.nn.Linear(1, 1)
            self.calls = 0

        def loss(self, data):
            self.calls += 1
            return (self.encoder(data["x"]) ** 2).sum() + 1.0 + 0.0 * self.decoder(data["x"]).sum()

        def save(self, path, itr, best):
            torch.save({"itr": itr, "encoder_state_dict": self.encoder.state_dict(),
                        "decoder_state_dict": self.decoder.state_dict(), "best_loss": best}, path + self.model_name)

    m = Stub()
    opt = torch.optim.SGD(list(m.encoder.parameters()), lr=0.0)  # lr 0: loss never improves -> early stop
    _, best, _ = training_utils.variational_training_loop(50, DG(), m, 5, opt, test_freq=2, early_stop=2, path=str(tmp_path) + "/")
    out = capsys.readouterr().out
    assert "Iter 0002 | Total Loss" in out and "Overall best loss" in out
    assert out.count("Iter ") == 3  # improves once (first validation), then two stale validations -> stop
    assert os.path.exists(str(tmp_path) + "/VI_stub.pkl") and best < 1e9


def test_neural_and_real_mirrors_surface():
    """Constructor signatures, model names, state_dict layouts and CPU rhs arithmetic of the NeuralODE / real-data mirrors
    (reference model.py:969-1026, :570-657, :772-862, :180-242) against golden G2 / G3 / G4."""
    torch.manual_seed(0)
    dec = model.RocheExpertDecoder(40, 8, 1, 1.0, 0.125, roche=False, method="rk4", device=CPU)
    assert dec.model_name == "NeuralODEDecoder"
    assert list(dec.state_dict()) == ["output_function.0.weight", "output_function.0.bias", "ode.kel", "ode.ml_net.0.weight",
                                      "ode.ml_net.0.bias", "ode.ml_net.2.weight", "ode.ml_net.2.bias"]
    assert dec.ode.ml_net[0].weight.shape == (80, 9) and dec.ode.ml_net[2].weight.shape == (8, 80)
    dr = model.DecoderReal(24, 20, 1, 11, 43, 40, 1, method="midpoint", ode_step_size=1.0, ode_type="hybrid", t0=24, device=CPU)
    assert dr.model_name == "DecoderReal_hybrid" and dr.t.tolist() == [float(v) for v in range(23, 40)]
    keys = list(dr.state_dict())
    assert keys[:4] == ["output_function.0.weight", "output_function.0.bias", "output_function.2.weight", "output_function.2.bias"]
    assert keys[4:] == ["ode.k_immunity", "ode.kel", "ode.kel2", "ode.dx1_net.0.weight", "ode.dx1_net.0.bias", "ode.dx1_net.2.weight",
                        "ode.dx1_net.2.bias", "ode.dx2_net.0.weight", "ode.dx2_net.0.bias", "ode.dx2_net.2.weight", "ode.dx2_net.2.bias",
                        "ode.lin_hh.weight", "ode.lin_hz.weight", "ode.lin_hr.weight"]
    assert dr.ode.flat_weights().numel() == 9 * 43 + 2 + 3 * 16 * 16
    enc = model.EncoderLSTMReal(37, 44, 20, output_all=False, reverse=False, device=CPU)
    assert enc.model_name == "LSTMReal" and list(enc.state_dict())[:4] == ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0"]
    import hode
    with pytest.raises(hode.HodeConfigError):
        model.DecoderReal(24, 20, 1, 11, 43, 40, 1, ode_type="neural", device=CPU)


def test_real_mirror_cpu_arithmetic_matches_golden(golden_dir):
    g = _load(golden_dir, "g3_roche_real_rhs.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        D, H, T, B = [int(v) for v in g[pre + "meta"]]
        ode = model.RocheODEReal(D, 1, 11, H, T, 1, device=CPU)
        ode.load_state_dict({k[len(pre + "sd_"):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + "sd_")})
        ode.set_action_static(torch.from_numpy(g[pre + "action"]), None)
        y = torch.from_numpy(g[pre + "y"])
        for ti, t in enumerate(g[pre + "t"]):
            tt = torch.tensor(float(t))
            np.testing.assert_allclose(ode(tt, y).detach().numpy(), g[pre + "f"][ti], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(ode.dose_at_time(tt).detach().numpy(), g[pre + "dose"][ti], rtol=1e-6, atol=1e-7)
    g4 = _load(golden_dir, "g4_encoder.npz")
    obs, aw, H, Dz, T, B = [int(v) for v in g4["real_meta"]]
    enc = model.EncoderLSTMReal(obs + aw + 1, H, Dz, output_all=False, reverse=False, device=CPU)
    enc.load_state_dict({k[len("real_sd_"):].replace("__", "."): torch.from_numpy(g4[k]) for k in g4.files if k.startswith("real_sd_")})
    mu, lv = enc(torch.from_numpy(g4["real_x"]), torch.from_numpy(g4["real_a"]), torch.from_numpy(g4["real_mask"]))
    np.testing.assert_allclose(mu.detach().numpy(), g4["real_mu"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lv.detach().numpy(), g4["real_log_var"], rtol=2e-5, atol=2e-6)
    gn = _load(golden_dir, "g2_neural_rhs.npz")
    for ci in range(int(gn["n_cases"])):
        pre = "c%d_" % ci
        D, T, B = [int(v) for v in gn[pre + "meta"]]
        ode = model.NeuralODE(D, 1, (T - 1) * 0.125, float(gn[pre + "step"]), device=CPU)
        ode.load_state_dict({k[len(pre + "sd_"):].replace("__", "."): torch.from_numpy(gn[k]) for k in gn.files if k.startswith(pre + "sd_")})
        ode.set_action(torch.from_numpy(gn[pre + "action"]))
        y = torch.from_numpy(gn[pre + "y"])
        for ti, t in enumerate(gn[pre + "t"]):
            np.testing.assert_allclose(ode(torch.tensor(float(t)), y).detach().numpy(), gn[pre + "f"][ti], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("step_size", [0.0625, 0.05, 0.125, 0.3, 0.04])
def test_substep_grid_and_output_reads_equal_the_oracle(step_size):
    """`hode.substep` (options["step_size"]): solver grid + linear reads, with the oracle as the per-grid solver on both
    sides -- bit-equal to the oracle's own sub-stepping, including grids that do not hit the output times."""
    from hode import substep
    from oracle.rhs import RocheRHS
    torch.manual_seed(0)
    D, B, T, step = 6, 4, 9, 0.125
    f = RocheRHS(D, step)
    a = torch.zeros(T, B, 1)
    a[2, :, 0] = 1.0
    f.set_action(a)
    y0 = (torch.rand(B, D) * 0.1).requires_grad_(True)
    t = torch.arange(T) * step
    ref = oracle_odeint(f, y0, t, method="rk4", options={"step_size": step_size})
    got = substep.solve_with_step_size(lambda g: oracle_odeint(f, y0, g, method="rk4"), t, step_size)
    assert torch.equal(ref, got)
    g_ref, = torch.autograd.grad(ref.sum(), y0)
    g_got, = torch.autograd.grad(got.sum(), y0)
    assert torch.allclose(g_ref, g_got, rtol=1e-6, atol=1e-7)


def test_training_loop_lets_configuration_errors_through(tmp_path):
    """The mirrored loop catches RuntimeError around model.loss like the reference (solver divergence ends a restart);
    a missing / stale library, an unsupported shape or CPU tensors are HodeConfigError and must surface instead of
    being printed and followed by a checkpoint of the untrained model."""
    import hode
    import training_utils

    class _Gen:
        train_size = val_size = 4

        def get_mini_batch(self, fold, n):
            return {"measurements": torch.zeros(2, n, 3)}

        get_split = get_mini_batch

    class _Model:
        model_name = "m"

        def __init__(self, exc):
            self.exc, self.saved = exc, 0
            self.encoder = self.decoder = torch.nn.Linear(1, 1)

        def loss(self, data):
            raise self.exc

        def save(self, path, itr, best):
            self.saved += 1
            torch.save({"encoder_state_dict": self.encoder.state_dict(), "decoder_state_dict": self.decoder.state_dict(),
                        "best_loss": best}, path + self.model_name)

    opt = torch.optim.SGD(torch.nn.Linear(1, 1).parameters(), lr=0.1)
    m = _Model(hode.HodeConfigError("hode: libhode.so not found"))
    with pytest.raises(hode.HodeConfigError):
        training_utils.variational_training_loop(3, _Gen(), m, 4, opt, 1, path=str(tmp_path) + "/")
    assert m.saved == 0
    m = _Model(hode.HodeError("hode dopri5: underflow in dt"))  # numerical failure: handled like the reference does
    training_utils.variational_training_loop(3, _Gen(), m, 4, opt, 1, path=str(tmp_path) + "/")
    assert m.saved == 1
