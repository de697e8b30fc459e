"""The config-5 (real-data) pipeline of the host-side mirror against G8, the fixture recorded from the REFERENCE's
``EncoderLSTMReal`` + ``DecoderReal`` + ``VariationalInferenceReal`` (tests/golden/make_golden.py::gen_vi_real; reference
model.py:180-242, :772-862, :1217-1261).  CPU: the mirror classes with the oracle solver injected -- a mistake in the host
Python both config-5 GPU tests share (the time grid ``arange(t0 - 1, t_max)``, the ``[1:]`` slicing, the ELU readout, the
time-weighted masked SSE, ``t / max(mask)`` in the encoder input) shows here.  The GPU counterpart on the same inputs is
tests/test_hip_golden.py."""
import os

import numpy as np
import torch

import model
from oracle.solvers import odeint as oracle_odeint


def load_real_case(g, ci, device):
    """Mirror modules with the fixture's weights + the fixture's data dict (shared with tests/test_hip_golden.py)."""
    pre = "c%d_" % ci
    obs, act, stat, D, T, t0, B, div, weight, elbo, seed = [int(v) for v in g[pre + "meta"]]
    input_dim = obs + act + stat + 1
    enc = model.EncoderLSTMReal(input_dim, int(input_dim * 1.2), D, output_all=False, reverse=False, device=device)
    dec = model.DecoderReal(obs, D, act, stat, int((obs + act + stat) * 1.2), T, 1, method=str(g[pre + "method"]),
                            ode_step_size=1 / div, ode_type="hybrid", t0=t0, device=device)
    for mod, tag in ((enc, "enc_"), (dec, "dec_")):
        sd = {k[len(pre + tag):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + tag)}
        mod.load_state_dict(sd, strict=True)   # the reference's own state_dict keys
    vi = model.VariationalInferenceReal(enc, dec, elbo=bool(elbo), t0=t0, weight=bool(weight))
    data = {k: torch.from_numpy(g[pre + k]).to(device) for k in ("measurements", "actions", "masks", "statics")}
    return vi, enc, dec, data, seed


def check_real_case(g, ci, vi, enc, dec, loss, tol_loss, tol_h, tol_g):
    pre = "c%d_" % ci
    want = float(g[pre + "loss"])
    assert abs(loss.item() - want) <= tol_loss * abs(want), (ci, loss.item(), want)
    np.testing.assert_allclose(vi.h_hat.detach().cpu().numpy(), g[pre + "h_hat"], rtol=0, atol=tol_h * (1 + np.abs(g[pre + "h_hat"]).max()))
    np.testing.assert_allclose(vi.x_hat.detach().cpu().numpy(), g[pre + "x_hat"], rtol=0, atol=tol_h * (1 + np.abs(g[pre + "x_hat"]).max()))
    np.testing.assert_allclose(vi.z.detach().cpu().numpy(), g[pre + "z"], rtol=0, atol=tol_h)
    assert tuple(vi.x_hat.shape) == tuple(g[pre + "x_hat"].shape) and tuple(vi.h_hat.shape) == tuple(g[pre + "h_hat"].shape)
    np.testing.assert_array_equal(dec.t.cpu().numpy(), g[pre + "t"])
    for mod, tag in ((enc, "genc_"), (dec, "gdec_")):
        for n, p in mod.named_parameters():
            w = g[pre + tag + n.replace(".", "__")]
            got = p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros_like(w)
            scale = np.abs(w).max()
            if scale < 1e-12:
                assert np.abs(got).max() < 1e-9, n
                continue
            err = np.linalg.norm((got - w).ravel()) / np.linalg.norm(w.ravel())
            assert err <= tol_g, (ci, n, err)


def test_g8_mirror_pipeline_on_the_cpu_matches_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_vi_real.npz"), allow_pickle=False)
    for ci in range(int(g["n_cases"])):
        vi, enc, dec, data, seed = load_real_case(g, ci, torch.device("cpu"))
        dec._odeint = oracle_odeint          # tests only: the CPU solver the fixture was recorded with
        torch.manual_seed(seed)
        loss = vi.loss(data)
        loss.backward()
        check_real_case(g, ci, vi, enc, dec, loss, tol_loss=2e-5, tol_h=2e-6, tol_g=2e-4)
