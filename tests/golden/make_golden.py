#!/usr/bin/env python
"""Generate the golden vectors under ``tests/golden/`` from the REFERENCE implementation.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

What it does: imports the reference's ``model.py`` / ``dataloader.py`` from
``/root/reference`` and records inputs + outputs of the functions on the hot path
(G8 = the config-5 pipeline) as small ``.npz`` files.  The reference's ``model.py`` imports
the third-party ``torchdiffeq`` at module level, which is not installed; an in-memory
module of that name is registered whose ``odeint`` is this repo's CPU restatement
(``oracle.solvers.odeint``).  Therefore:

* G1-G4, G6, G7 (rhs, dose schedule, encoder, their autograd VJPs, generator batch) are
  pure reference arithmetic -- they PIN the oracle's rhs/encoder restatement.
* G5 (``VariationalInference.loss``) and G8 (``VariationalInferenceReal.loss`` with ``DecoderReal``) pin everything AROUND the solver (set_action,
  readout, masked SSE, KL) with the oracle solver in the loop; it does not pin the
  solver itself, which stays "parity unpinned" (see ``oracle/solvers.py``).

Only data (inputs / expected outputs) is written; no reference source travels.
"""

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle.solvers import odeint as oracle_odeint  # noqa: E402

_stub = types.ModuleType("torchdiffeq")
_stub.odeint = oracle_odeint
sys.modules["torchdiffeq"] = _stub
_ps = types.ModuleType("properscoring")  # only imported by training_utils (CRPS); never called here
sys.modules["properscoring"] = _ps
sys.path.insert(0, REF)

import dataloader  # noqa: E402  (reference)
import model  # noqa: E402  (reference)
import sim_config  # noqa: E402  (reference)

CPU = torch.device("cpu")


def npy(x):
    return x.detach().cpu().numpy()


def sd_arrays(module, prefix):
    return {prefix + k.replace(".", "__"): npy(v) for k, v in module.state_dict().items()}


def one_dose_actions(T, B, gen, dose_max=10.0, idx=None):
    a = torch.zeros(T, B, 1)
    if idx is None:
        idx = torch.randint(0, T - 1, (B,), generator=gen)
    amt = torch.rand(B, generator=gen) * dose_max
    a[idx, torch.arange(B), 0] = amt
    return a, idx


# ----------------------------------------------------------------------------- G1 + G6 (rhs and VJP)
def gen_roche_rhs():
    out = {}
    gen = torch.Generator().manual_seed(1234)
    cases = []
    step = 0.125
    T, B = 24, 7
    for D in (4, 8, 12):
        for ablate in (False, True):
            for theta_mode in ("default", "random", "hill"):
                if ablate and theta_mode != "default":
                    continue
                cases.append((D, ablate, theta_mode))
    for ci, (D, ablate, theta_mode) in enumerate(cases):
        torch.manual_seed(100 + ci)
        ode = model.RocheODE(D, 1, (T - 1) * step, step, ablate=ablate, device=CPU)
        with torch.no_grad():
            if theta_mode == "random":
                for name in ("ec50_patho", "emax_patho", "k_dexa", "k_discure_immunereact", "k_discure_immunity",
                             "k_disprog", "k_immune_disease", "k_immune_feedback", "k_immune_off", "k_immunity", "kel"):
                    getattr(ode, name).fill_(float(0.3 + 1.5 * torch.rand((), generator=gen)))
            if theta_mode == "hill":
                ode.HillCure.fill_(3.0)   # integer exponent: negative bases stay finite
                ode.HillPatho.fill_(1.5)  # non-integer exponent: negative base -> NaN (torch.pow semantics)
                ode.kel.fill_(0.7)
        idx = torch.tensor([3, 3, 8, 12, 0, 22, 8])
        a, idx = one_dose_actions(T, B, gen, idx=idx)
        ode.set_action(a)
        y = torch.rand(B, D, generator=gen) * 2.0
        if theta_mode != "hill":
            y[1, :] = -y[1, :]  # negative states: pow with exponent 2.0 must stay finite
        else:
            y[1, 2] = -y[1, 2]  # Immunity < 0 with HillCure = 3 (finite); ImmuneReact > 0 everywhere
        # times: before all doses, exactly at a dose, just before / after it, between, late
        t8 = 8 * step
        ts = [0.0, 3 * step, float(np.nextafter(np.float32(t8), np.float32(0))), t8,
              float(np.nextafter(np.float32(t8), np.float32(9))), t8 + step / 3, 2.9]
        pre = "c%d_" % ci
        out[pre + "meta"] = np.array([D, int(ablate), T, B], dtype=np.int64)
        out[pre + "step"] = np.float64(step)
        out[pre + "action"] = npy(a)
        out[pre + "y"] = npy(y)
        out[pre + "t"] = np.array(ts, dtype=np.float32)
        out[pre + "times"] = npy(ode.times)
        out[pre + "dosage"] = npy(ode.dosage)
        out.update(sd_arrays(ode, pre + "sd_"))
        fs, doses, gys, gparams = [], [], [], {}
        cot = torch.randn(B, D, generator=gen)
        out[pre + "cot"] = npy(cot)
        for t in ts:
            tt = torch.tensor(t, dtype=torch.float32)
            yy = y.clone().requires_grad_(True)
            f = ode(tt, yy)
            fs.append(npy(f))
            doses.append(npy(ode.dose_at_time(tt)))
            ode.zero_grad()
            (f * cot).sum().backward()
            gys.append(npy(yy.grad))
            for n, p in ode.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                gparams.setdefault(n, []).append(npy(g).copy())
        out[pre + "f"] = np.stack(fs)
        out[pre + "dose"] = np.stack(doses)
        out[pre + "gy"] = np.stack(gys)
        for n, g in gparams.items():
            out[pre + "g_" + n.replace(".", "__")] = np.stack(g)
        # fp64 time argument (what a solver with fp64 clocks would pass without casting)
        f64 = ode(torch.tensor(t8 + step / 3, dtype=torch.float64), y)
        out[pre + "f_t64"] = npy(f64.float())
        out[pre + "f_t64_is64"] = np.array(f64.dtype == torch.float64)
    out["n_cases"] = np.array(len(cases))
    # integer step_size => int64 dose times (dtype rule of set_action)
    ode = model.RocheODE(8, 1, 14, 1, device=CPU)
    a, _ = one_dose_actions(15, 4, gen, idx=torch.tensor([0, 5, 13, 2]))
    ode.set_action(a)
    out["int_step_times"] = npy(ode.times)
    out["int_step_action"] = npy(a)
    np.savez_compressed(os.path.join(HERE, "g1_roche_rhs.npz"), **out)


# ----------------------------------------------------------------------------- G2
def gen_neural_rhs():
    out = {}
    gen = torch.Generator().manual_seed(77)
    step, T, B = 0.125, 16, 5
    for ci, D in enumerate((8, 12)):
        torch.manual_seed(200 + ci)
        ode = model.NeuralODE(D, 1, (T - 1) * step, step, device=CPU)
        a, idx = one_dose_actions(T, B, gen, idx=torch.tensor([2, 2, 5, 0, 14]))
        ode.set_action(a)
        y = torch.randn(B, D, generator=gen)
        ts = [0.0, 2 * step, 2 * step + 1e-3, 5 * step, 1.0]
        pre = "c%d_" % ci
        out[pre + "meta"] = np.array([D, T, B], dtype=np.int64)
        out[pre + "step"] = np.float64(step)
        out[pre + "action"] = npy(a)
        out[pre + "y"] = npy(y)
        out[pre + "t"] = np.array(ts, dtype=np.float32)
        out.update(sd_arrays(ode, pre + "sd_"))
        cot = torch.randn(B, D, generator=gen)
        out[pre + "cot"] = npy(cot)
        fs, gys = [], []
        for t in ts:
            yy = y.clone().requires_grad_(True)
            f = ode(torch.tensor(t, dtype=torch.float32), yy)
            (f * cot).sum().backward()
            fs.append(npy(f))
            gys.append(npy(yy.grad))
        out[pre + "f"] = np.stack(fs)
        out[pre + "gy"] = np.stack(gys)
    out["n_cases"] = np.array(2)
    np.savez_compressed(os.path.join(HERE, "g2_neural_rhs.npz"), **out)


# ----------------------------------------------------------------------------- G3
def gen_roche_real_rhs():
    out = {}
    gen = torch.Generator().manual_seed(99)
    T, B = 32, 6
    for ci, (D, H) in enumerate(((20, 43), (4, 9))):
        torch.manual_seed(300 + ci)
        ode = model.RocheODEReal(D, 1, 11, H, T, 1, device=CPU)
        a = (torch.rand(T, B, 1, generator=gen) < 0.2).float() * torch.rand(T, B, 1, generator=gen)
        ode.set_action_static(a, None)
        y = torch.randn(B, D, generator=gen) * 0.5
        ts = [0.5, 1.0, float(np.nextafter(np.float32(5), np.float32(0))), 5.0, 17.25, 31.0]
        pre = "c%d_" % ci
        out[pre + "meta"] = np.array([D, H, T, B], dtype=np.int64)
        out[pre + "action"] = npy(a)
        out[pre + "y"] = npy(y)
        out[pre + "t"] = np.array(ts, dtype=np.float32)
        out.update(sd_arrays(ode, pre + "sd_"))
        cot = torch.randn(B, D, generator=gen)
        out[pre + "cot"] = npy(cot)
        fs, ds, gys = [], [], []
        for t in ts:
            tt = torch.tensor(t, dtype=torch.float32)
            yy = y.clone().requires_grad_(True)
            f = ode(tt, yy)
            (f * cot).sum().backward()
            fs.append(npy(f))
            ds.append(npy(ode.dose_at_time(tt)))
            gys.append(npy(yy.grad))
        out[pre + "f"] = np.stack(fs)
        out[pre + "dose"] = np.stack(ds)
        out[pre + "gy"] = np.stack(gys)
    out["n_cases"] = np.array(2)
    np.savez_compressed(os.path.join(HERE, "g3_roche_real_rhs.npz"), **out)


# ----------------------------------------------------------------------------- G4 (+ G6 encoder grads)
def gen_encoder():
    out = {}
    gen = torch.Generator().manual_seed(55)
    for ci, (obs, H, D, T, B) in enumerate(((12, 24, 8, 9, 5), (20, 40, 12, 6, 3))):
        torch.manual_seed(400 + ci)
        enc = model.EncoderLSTM(obs + 1, H, D, device=CPU)
        x = torch.randn(T, B, obs, generator=gen)
        a, _ = one_dose_actions(T, B, gen)
        m = (torch.rand(T, B, obs, generator=gen) < 0.5).float()
        mu, lv = enc(x, a, m)
        cot_mu = torch.randn(B, D, generator=gen)
        cot_lv = torch.randn(B, D, generator=gen)
        enc.zero_grad()
        ((mu * cot_mu).sum() + (lv * cot_lv).sum()).backward()
        pre = "c%d_" % ci
        out[pre + "meta"] = np.array([obs, H, D, T, B], dtype=np.int64)
        out[pre + "x"], out[pre + "a"], out[pre + "mask"] = npy(x), npy(a), npy(m)
        out[pre + "mu"], out[pre + "log_var"] = npy(mu), npy(lv)
        out[pre + "cot_mu"], out[pre + "cot_lv"] = npy(cot_mu), npy(cot_lv)
        out.update(sd_arrays(enc, pre + "sd_"))
        for n, p in enc.named_parameters():
            out[pre + "g_" + n.replace(".", "__")] = npy(p.grad)
    out["n_cases"] = np.array(2)
    # real-data encoder (forward only): 37 -> 44 -> 20, forward time order, no input masking
    torch.manual_seed(450)
    obs, act, stat, T, B = 24, 1, 11, 7, 4
    enc = model.EncoderLSTMReal(obs + act + stat + 1, 44, 20, reverse=False, device=CPU)
    x = torch.randn(T, B, obs, generator=gen)
    a_s = torch.rand(T, B, act + stat, generator=gen)
    m = (torch.rand(T, B, obs, generator=gen) < 0.5).float()
    mu, lv = enc(x, a_s, m)
    out["real_meta"] = np.array([obs, act + stat, 44, 20, T, B], dtype=np.int64)
    out["real_x"], out["real_a"], out["real_mask"] = npy(x), npy(a_s), npy(m)
    out["real_mu"], out["real_log_var"] = npy(mu), npy(lv)
    out.update(sd_arrays(enc, "real_sd_"))
    np.savez_compressed(os.path.join(HERE, "g4_encoder.npz"), **out)


# ----------------------------------------------------------------------------- G5
def gen_vi_loss():
    out = {}
    gen = torch.Generator().manual_seed(2021)
    obs, D, T, B, step = 10, 8, 12, 6, 0.125
    t_max = (T - 1) * step
    ci = 0
    for method in ("rk4", "dopri5"):
        for mode in ("lik", "kl_normal", "kl_exp"):
            torch.manual_seed(500 + ci)
            enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=CPU)
            dec = model.RocheExpertDecoder(obs, D, 1, t_max, step, roche=True, method=method, device=CPU)
            elbo = mode != "lik"
            prior = model.ExponentialPrior.log_density if mode == "kl_exp" else None
            vi = model.VariationalInference(enc, dec, elbo=elbo, prior_log_pdf=prior, mc_size=7)
            x = torch.randn(T, B, obs, generator=gen)
            a, _ = one_dose_actions(T, B, gen)
            m = (torch.rand(T, B, obs, generator=gen) < 0.5).float()
            data = {"measurements": x, "actions": a, "masks": m}
            torch.manual_seed(900 + ci)  # seeds the reparameterisation / MC-KL draws
            loss = vi.loss(data)
            for p in vi.parameters():
                p.grad = None
            loss.backward()
            pre = "c%d_" % ci
            out[pre + "method"] = np.array(method)
            out[pre + "mode"] = np.array(mode)
            out[pre + "meta"] = np.array([obs, D, T, B, 900 + ci], dtype=np.int64)
            out[pre + "step"] = np.float64(step)
            out[pre + "x"], out[pre + "a"], out[pre + "mask"] = npy(x), npy(a), npy(m)
            out[pre + "loss"] = npy(loss)
            out[pre + "z"] = npy(vi.z)
            out[pre + "h_hat"] = npy(vi.h_hat)
            out[pre + "x_hat"] = npy(vi.x_hat)
            out.update(sd_arrays(enc, pre + "enc_"))
            out.update(sd_arrays(dec, pre + "dec_"))
            for n, p in list(enc.named_parameters()):
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                out[pre + "genc_" + n.replace(".", "__")] = npy(g)
            for n, p in list(dec.named_parameters()):
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                out[pre + "gdec_" + n.replace(".", "__")] = npy(g)
            ci += 1
    out["n_cases"] = np.array(ci)
    np.savez_compressed(os.path.join(HERE, "g5_vi_loss.npz"), **out)


# ----------------------------------------------------------------------------- G8 (config 5: the real-data pipeline)
def gen_vi_real():
    """``DecoderReal.forward`` (model.py:833-862: ``t = arange(t0 - 1, t_max)``, ``output_function(h)[1:]``, ELU readout MLP)
    + ``VariationalInferenceReal.loss`` (model.py:1223-1261: encoder on the first t0 steps over cat(x, a, s, t / max(mask)),
    time-weighted masked SSE on x[t0:], analytic KL) + every parameter gradient, on DDW-shaped tensors (obs 24, statics 11,
    D 20, encoder 37 -> 44, decoder hidden 43, t0 24), constructed as run_real.py:38-72 constructs them.  The oracle solver
    is the torchdiffeq shim (same recipe as G5): this pins everything AROUND the solver on the real-data path."""
    out = {}
    gen = torch.Generator().manual_seed(808)
    obs, act, stat, D, T, t0, B = 24, 1, 11, 20, 34, 24, 5
    input_dim = obs + act + stat + 1
    hidden = int((obs + act + stat) * 1.2)
    cases = [("midpoint", 1, False, False), ("midpoint", 2, True, False), ("rk4", 1, False, False), ("midpoint", 1, True, True)]
    for ci, (method, div, weight, elbo) in enumerate(cases):
        torch.manual_seed(800 + ci)
        enc = model.EncoderLSTMReal(input_dim, int(input_dim * 1.2), D, output_all=False, reverse=False, device=CPU)
        dec = model.DecoderReal(obs, D, act, stat, hidden, T, 1, method=method, ode_step_size=1 / div, ode_type="hybrid",
                                t0=t0, device=CPU)
        vi = model.VariationalInferenceReal(enc, dec, elbo=elbo, t0=t0, weight=weight)
        data = {"measurements": torch.randn(T, B, obs, generator=gen),
                "actions": (torch.rand(T, B, act, generator=gen) < 0.15).float() * torch.rand(T, B, act, generator=gen),
                "masks": (torch.rand(T, B, obs, generator=gen) < 0.5).float(),
                "statics": torch.rand(1, B, stat, generator=gen).expand(T, B, stat).contiguous()}
        torch.manual_seed(880 + ci)  # seeds the reparameterisation draw of the elbo case
        loss = vi.loss(data)
        for p in vi.parameters():
            p.grad = None
        loss.backward()
        # what the decoder returned inside loss() (deterministic given z): run it once more for the record
        with torch.no_grad():
            a_in = torch.cat([data["actions"], data["statics"]], dim=-1)
            mu, log_var = enc(data["measurements"][:t0], a_in[:t0], data["masks"][:t0])
        pre = "c%d_" % ci
        out[pre + "method"] = np.array(method)
        out[pre + "meta"] = np.array([obs, act, stat, D, T, t0, B, div, int(weight), int(elbo), 880 + ci], dtype=np.int64)
        for k, v in data.items():
            out[pre + k] = npy(v)
        out[pre + "loss"] = npy(loss)
        out[pre + "mu"], out[pre + "log_var"] = npy(mu), npy(log_var)
        if elbo:
            torch.manual_seed(880 + ci)
            z = enc.reparameterize(mu, log_var)
        else:
            z = mu
        with torch.no_grad():
            x_hat, h_hat = dec(z, data["actions"], data["statics"])
        out[pre + "z"], out[pre + "x_hat"], out[pre + "h_hat"] = npy(z), npy(x_hat), npy(h_hat)
        out[pre + "t"] = npy(dec.t)
        out.update(sd_arrays(enc, pre + "enc_"))
        out.update(sd_arrays(dec, pre + "dec_"))
        for mod, tag in ((enc, "genc_"), (dec, "gdec_")):
            for n, p in mod.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                out[pre + tag + n.replace(".", "__")] = npy(g)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "g8_vi_real.npz"), **out)


# ----------------------------------------------------------------------------- G7
def gen_generator_batch():
    """100-patient dim8 batch from the reference generator (scipy LSODA latents): schema + loose known answer."""
    np.random.seed(666)
    torch.manual_seed(666)
    cfg = sim_config.dim8_config
    n = 100
    dg = dataloader.DataGeneratorRoche(
        n, cfg.obs_dim, cfg.t_max, cfg.step_size, sim_config.RochConfig(kel=1), cfg.output_sigma, cfg.dose_max,
        cfg.latent_dim, cfg.sparsity, p_remove=cfg.p_remove, output_sparsity=cfg.output_sparsity, device=CPU,
        val_size=10, test_size=20,
    )
    dg.generate_data()
    dg.split_sample()
    out = {
        "meta": np.array([n, cfg.obs_dim, cfg.latent_dim, cfg.t_max, cfg.step_size], dtype=np.int64),
        "latents": npy(dg.latents),
        "actions": npy(dg.actions),
        "measurements": npy(dg.measurements),
        "masks": npy(dg.masks),
        "ml_coef": np.asarray(dg.ml_coef, dtype=np.float64),
        "train_shape": np.array(dg.data_train["measurements"].shape, dtype=np.int64),
        "val_shape": np.array(dg.data_val["measurements"].shape, dtype=np.int64),
        "test_shape": np.array(dg.data_test["measurements"].shape, dtype=np.int64),
    }
    np.savez_compressed(os.path.join(HERE, "g7_generator_dim8.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(1)  # deterministic reduction order
    gen_roche_rhs()
    gen_neural_rhs()
    gen_roche_real_rhs()
    gen_encoder()
    gen_vi_loss()
    gen_generator_batch()
    gen_vi_real()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
