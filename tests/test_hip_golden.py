"""The HIP path on the REFERENCE'S OWN numbers: every test here loads a fixture under tests/golden/ (recorded by
tests/golden/make_golden.py from the imported reference) and runs the kernels on the fixture's inputs through the C ABI.

The other GPU tests compare HIP with the oracle on random inputs and the CPU tests compare the oracle with the fixtures;
this file closes that two-hop chain: the fixtures' own inputs -- t exactly at / one ulp before / one
ulp after a dose, negative bases under `pow`, non-integer Hill exponents, the reference's seeded weights -- reach the
kernels, and the expected values are the reference's outputs, not the oracle's.

rhs values are read off ONE EULER STEP of the solver kernels: with the grid (t, t + 1) the kernel returns
h[1] = y + dt f(t, y) (torchdiffeq `euler`), so f = (h[1] - y) / dt up to the rounding of that sum,
and the discrete adjoint of the step returns cot + dt J^T cot and dt (df/dtheta)^T cot -- the fixture's VJPs (G6).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hode
import model
from test_golden_real import check_real_case, load_real_case


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _load_sd(module, g, prefix):
    sd = {k[len(prefix):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def _euler_rhs(ode, y, t, cot, dev):
    """f(t, y) and the VJPs of cot through one euler step of the HIP solver; returns (f, gy, {name: grad})."""
    t0 = np.float32(t)
    t1 = np.float32(t0 + np.float32(1.0))
    dt = float(np.float32(t1 - t0))                 # what the kernel forms from the fp32 grid
    grid = torch.tensor([t0, t1], dtype=torch.float32, device=dev)
    yy = y.to(dev).requires_grad_(True)
    for p in ode.parameters():
        p.grad = None
    h = hode.odeint(ode, yy, grid, method="euler")
    assert torch.equal(h[0].detach(), yy.detach())   # h[0] == y0 bit-exact
    f = (h[1].detach().double().cpu() - y.double()) / dt
    (h[1] * cot.to(dev)).sum().backward()
    gy = (yy.grad.double().cpu() - cot.double()) / dt
    gp = {n: (None if p.grad is None else p.grad.double().cpu() / dt) for n, p in ode.named_parameters()}
    return f.numpy(), gy.numpy(), gp


@pytest.mark.parametrize("lanes", [0, 4, 1])
def test_g1_roche_rhs_values_and_vjps_through_one_euler_step(golden_dir, lanes):
    """RocheODE.forward / dose_at_time / set_action (reference model.py:495-555) on G1: D in {4, 8, 12}, ablate, default /
    random rate constants, Hill exponents 3 and 1.5 with a negative base, t at / one ulp around a dose time.  All three
    lane layouts the library can pick (0 = its own choice: the split wave pipelines for D in {8, 12})."""
    dev = _dev()
    g = _g(golden_dir, "g1_roche_rhs.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        D, ablate, T, B = [int(v) for v in g[pre + "meta"]]
        step = float(g[pre + "step"])
        ode = model.RocheODE(D, 1, (T - 1) * step, step, ablate=bool(ablate), device=dev)
        _load_sd(ode, g, pre + "sd_")
        ode.lanes_per_patient = lanes
        ode.set_action(torch.from_numpy(g[pre + "action"]).to(dev))
        np.testing.assert_array_equal(ode.times.cpu().numpy(), g[pre + "times"])       # A2, bit-exact
        np.testing.assert_array_equal(ode.dosage.cpu().numpy(), g[pre + "dosage"])
        y, cot = torch.from_numpy(g[pre + "y"]), torch.from_numpy(g[pre + "cot"])
        for ti, t in enumerate(g[pre + "t"]):
            f, gy, gp = _euler_rhs(ode, y, t, cot, dev)
            want = g[pre + "f"][ti]
            assert np.array_equal(np.isnan(f), np.isnan(want)), (ci, ti)               # NaN where the reference has NaN
            # rounding of h[1] = y + dt f in fp32 bounds what one step can resolve: 2 ulp of max(|y|, |h1|)
            tol = 3e-7 * (1.0 + np.abs(y.numpy()) + np.abs(np.nan_to_num(want)))
            assert np.all(np.abs(np.nan_to_num(f) - np.nan_to_num(want)) <= tol), (ci, ti, np.abs(f - want).max())
            wgy = g[pre + "gy"][ti]
            ok = ~np.isnan(wgy)
            assert np.all(np.abs(gy[ok] - wgy[ok]) <= 2e-6 * (1.0 + np.abs(cot.numpy()[ok]) + np.abs(wgy[ok]))), (ci, ti)
            for n, got in gp.items():
                w = g[pre + "g_" + n.replace(".", "__")][ti]
                if np.isnan(w).any():
                    continue       # d pow(x, a) / da at a negative base: NaN in the reference's sum over patients
                if got is None:
                    assert np.abs(w).max() == 0.0, (ci, ti, n)
                    continue
                assert np.abs(got.numpy() - w).max() <= 3e-5 * (1.0 + np.abs(w).max()), (ci, ti, n, got, w)


def test_g2_neural_rhs_through_one_euler_step(golden_dir):
    """NeuralODE.forward (reference model.py:1019-1026; the impulse dose `times == t` needs exact fp32 equality)."""
    dev = _dev()
    g = _g(golden_dir, "g2_neural_rhs.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        D, T, B = [int(v) for v in g[pre + "meta"]]
        step = float(g[pre + "step"])
        ode = model.NeuralODE(D, 1, (T - 1) * step, step, device=dev)
        _load_sd(ode, g, pre + "sd_")
        ode.set_action(torch.from_numpy(g[pre + "action"]).to(dev))
        y, cot = torch.from_numpy(g[pre + "y"]), torch.from_numpy(g[pre + "cot"])
        for ti, t in enumerate(g[pre + "t"]):
            f, gy, _ = _euler_rhs(ode, y, t, cot, dev)
            want = g[pre + "f"][ti]
            assert np.abs(f - want).max() <= 3e-7 * (2.0 + np.abs(y.numpy()).max()), (ci, ti, np.abs(f - want).max())
            assert np.abs(gy - g[pre + "gy"][ti]).max() <= 3e-6 * (1.0 + np.abs(cot.numpy()).max()), (ci, ti)


def test_g3_roche_real_rhs_through_one_euler_step(golden_dir):
    """RocheODEReal.forward / dose_at_time (reference model.py:613-657) at the fixture's times, on and off the hourly grid
    (0.5, one ulp before 5, 5, 17.25, 31): the kernels evaluate the dose sum from a per-patient table of its value at
    the integer hours (csrc/hode_real_args.hpp::real_dose), the reference re-sums every past dose per call."""
    dev = _dev()
    g = _g(golden_dir, "g3_roche_real_rhs.npz")
    n_checked = 0
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        D, H, T, B = [int(v) for v in g[pre + "meta"]]
        ode = model.RocheODEReal(D, 1, 11, H, T, 1, device=dev)
        _load_sd(ode, g, pre + "sd_")
        ode.set_action_static(torch.from_numpy(g[pre + "action"]).to(dev), None)
        y, cot = torch.from_numpy(g[pre + "y"]), torch.from_numpy(g[pre + "cot"])
        for ti, t in enumerate(g[pre + "t"]):
            f, gy, _ = _euler_rhs(ode, y, t, cot, dev)
            want = g[pre + "f"][ti]
            assert np.abs(f - want).max() <= 4e-7 * (2.0 + np.abs(y.numpy()).max()), (ci, ti, np.abs(f - want).max())
            assert np.abs(gy - g[pre + "gy"][ti]).max() <= 3e-6 * (1.0 + np.abs(cot.numpy()).max()), (ci, ti)
            n_checked += 1
    assert n_checked == 12


def test_g4_encoders_on_the_mfma_kernels(golden_dir):
    """EncoderLSTM.forward (masked, reverse time; reference model.py:408-440) -> mu, log_var and every parameter gradient,
    and EncoderLSTMReal.forward (reference model.py:210-242), on the fp32-MFMA window / BPTT kernels."""
    dev = _dev()
    g = _g(golden_dir, "g4_encoder.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        obs, H, D, T, B = [int(v) for v in g[pre + "meta"]]
        enc = model.EncoderLSTM(obs + 1, H, D, device=dev)
        _load_sd(enc, g, pre + "sd_")
        x, a, m = (torch.from_numpy(g[pre + k]).to(dev) for k in ("x", "a", "mask"))
        mu, lv = enc(x, a, m)
        np.testing.assert_allclose(mu.detach().cpu().numpy(), g[pre + "mu"], rtol=5e-6, atol=1e-7)
        np.testing.assert_allclose(lv.detach().cpu().numpy(), g[pre + "log_var"], rtol=5e-6, atol=2e-6)
        ((mu * torch.from_numpy(g[pre + "cot_mu"]).to(dev)).sum() + (lv * torch.from_numpy(g[pre + "cot_lv"]).to(dev)).sum()).backward()
        for n, p in enc.named_parameters():
            assert _rel(p.grad.cpu().numpy(), g[pre + "g_" + n.replace(".", "__")]) <= 2e-5, (ci, n)
    obs, ast, H, D, T, B = [int(v) for v in g["real_meta"]]
    enc = model.EncoderLSTMReal(obs + ast + 1, H, D, reverse=False, device=dev)
    _load_sd(enc, g, "real_sd_")
    mu, lv = enc(*(torch.from_numpy(g["real_" + k]).to(dev) for k in ("x", "a", "mask")))
    np.testing.assert_allclose(mu.detach().cpu().numpy(), g["real_mu"], rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(lv.detach().cpu().numpy(), g["real_log_var"], rtol=1e-5, atol=2e-7)


class _HostDraws:
    """Draw `torch.randn` / `torch.randn_like` on the HOST generator and move the result to the device: the fixtures'
    reparameterisation / Monte-Carlo draws were taken from the CPU generator under a recorded seed."""

    def __enter__(self):
        self.randn, self.randn_like = torch.randn, torch.randn_like

        def randn(*size, **kw):
            dev = kw.pop("device", None)
            out = self.randn(*size, **kw)
            return out if dev is None else out.to(dev)

        def randn_like(x, **kw):
            return self.randn(tuple(x.shape), dtype=x.dtype).to(x.device)

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self.randn, self.randn_like
        return False


def test_g5_vi_loss_and_grads_through_the_kernels(golden_dir):
    """VariationalInference.loss (reference model.py:1150-1214) on G5: encoder (MFMA LSTM) -> reparameterisation ->
    set_action -> hode.odeint (rk4 / dopri5) -> readout + masked SSE -> analytic / Monte-Carlo KL, and every parameter
    gradient.  The fixture's solver was the oracle (torchdiffeq is absent), so this pins everything around the solver on the
    reference's numbers and the solver against the oracle, as the CPU test of the oracle does."""
    dev = _dev()
    g = _g(golden_dir, "g5_vi_loss.npz")
    for ci in range(int(g["n_cases"])):
        pre = "c%d_" % ci
        obs, D, T, B, seed = [int(v) for v in g[pre + "meta"]]
        step = float(g[pre + "step"])
        method, mode = str(g[pre + "method"]), str(g[pre + "mode"])
        enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
        dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, roche=True, method=method, device=dev)
        _load_sd(enc, g, pre + "enc_")
        _load_sd(dec, g, pre + "dec_")
        prior = model.ExponentialPrior.log_density if mode == "kl_exp" else None
        vi = model.VariationalInference(enc, dec, elbo=(mode != "lik"), prior_log_pdf=prior, mc_size=7)
        data = {k2: torch.from_numpy(g[pre + k]).to(dev) for k, k2 in (("x", "measurements"), ("a", "actions"), ("mask", "masks"))}
        torch.manual_seed(seed)
        with _HostDraws():
            loss = vi.loss(data)
        loss.backward()
        want = float(g[pre + "loss"])
        assert abs(loss.item() - want) <= 5e-5 * abs(want), (ci, method, mode, loss.item(), want)
        np.testing.assert_allclose(vi.z.detach().cpu().numpy(), g[pre + "z"], rtol=2e-5, atol=1e-7)
        hh = g[pre + "h_hat"]
        tol_h = 2e-5 if method == "rk4" else 2e-4    # dopri5: accept / reject decisions may flip on the last bit
        assert np.abs(vi.h_hat.detach().cpu().numpy() - hh).max() <= tol_h * (1 + np.abs(hh).max()), (ci, method)
        assert np.abs(vi.x_hat.detach().cpu().numpy() - g[pre + "x_hat"]).max() <= tol_h * (1 + np.abs(g[pre + "x_hat"]).max())
        a_scale = 5e-5 if method == "rk4" else 2e-3  # the CPU oracle's own bounds against this fixture (test_oracle_golden.py)
        for mod, tag in ((enc, "genc_"), (dec, "gdec_")):
            for n, p in mod.named_parameters():
                w = g[pre + tag + n.replace(".", "__")]
                got = p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros_like(w)
                np.testing.assert_allclose(got, w, rtol=5e-3, atol=a_scale * (1 + np.abs(w).max()), err_msg="%d %s %s" % (ci, method, n))


def test_g8_config5_pipeline_through_the_kernels(golden_dir):
    """VariationalInferenceReal.loss + DecoderReal.forward (reference model.py:833-862, :1223-1261) on G8: MFMA LSTM encoder
    on the first t0 steps, RocheODEReal midpoint / rk4 + perturb on the matrix cores (ode_step_div 1 and 2), the fused
    two-layer readout + time-weighted masked SSE, analytic KL, every parameter gradient -- against the reference's numbers
    (the CPU mirror is checked against the same fixture in tests/test_golden_real.py)."""
    dev = _dev()
    g = _g(golden_dir, "g8_vi_real.npz")
    for ci in range(int(g["n_cases"])):
        vi, enc, dec, data, seed = load_real_case(g, ci, dev)
        torch.manual_seed(seed)
        with _HostDraws():
            loss = vi.loss(data)
        loss.backward()
        check_real_case(g, ci, vi, enc, dec, loss, tol_loss=5e-5, tol_h=2e-5, tol_g=2e-3)
