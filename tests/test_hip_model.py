"""End-to-end parity of the host-side mirror on the GPU (encoder -> reparam off -> HIP solver -> readout -> loss ->
backward) against the CPU oracle pipeline with identical weights and data.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import model
from oracle import vi as ovi
from oracle.encoder import EncoderLSTMOracle


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("method", ["rk4", "dopri5"])
@pytest.mark.parametrize("obs,D", [(40, 8), (80, 12)])
def test_vi_loss_and_grads_match_cpu_oracle(method, obs, D):
    from hode import synth
    dev = _dev()
    T, B, step = 20, 48, synth.STEP
    torch.manual_seed(1)
    enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, method=method, device=dev)
    vi = model.VariationalInference(enc, dec, elbo=False)
    enc_o = EncoderLSTMOracle(obs + 1, obs * 2, D)
    dec_o = ovi.DecoderOracle(obs, D, (T - 1) * step, step, method=method)
    enc_o.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})
    dec_o.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    sol = synth.solver_inputs(B, T, D, seed=3)
    ob = synth.observation_inputs(B, T, obs, seed=3)
    data = {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}
    from hode import adaptive
    adaptive.keep_workspace = method == "dopri5"
    try:
        loss = vi.loss({k: v.to(dev) for k, v in data.items()})
        loss.backward()
        if method == "dopri5":
            # drive the oracle's step algebra along the HIP run's own (t_n, dt_n) tape: accept / reject decisions that flip
            # on the last bit of the error norm (tests/test_hip_dopri5.py) drop out and every gradient can be held tightly
            from oracle.solvers import odeint_dopri5_replay
            tape = adaptive.read_tape()
            pairs, first = list(zip(tape["t"], tape["dt"])), bool(tape["init"]["first_accepted"])
            dec_o.solve = lambda f, y0, t: odeint_dopri5_replay(f, y0, t, 1e-7, 1e-8, pairs, first)
    finally:
        adaptive.keep_workspace = False
    loss_o = ovi.vi_loss(enc_o, dec_o, data, elbo=False)
    loss_o.backward()
    tol_h, tol_g = 3e-5, 2e-3
    assert abs(loss.item() - loss_o.item()) <= 2e-4 * abs(loss_o.item())
    assert (vi.h_hat.detach().cpu() - odeint_h(dec_o, enc_o, data)).abs().max().item() <= tol_h * 10
    names = [n for n, _ in list(enc.named_parameters()) + list(dec.named_parameters())]
    g_hip = [p.grad for p in list(enc.parameters()) + list(dec.parameters())]
    g32 = [None if p.grad is None else p.grad.clone() for p in list(enc_o.parameters()) + list(dec_o.parameters())]
    noise = [0.0] * len(g32)
    if method == "dopri5":
        # The derivative of Hairer's first step size (part of the reference's graph) is a cancellation-heavy sum: the
        # oracle's OWN fp32 evaluation of d loss / d kel sits 5e-3..7e-3 from its fp64 evaluation on this problem.  So the
        # yardstick is the fp64 oracle along the same tape, and a gradient may deviate from it by the 2e-3 of the other
        # cases or by twice what the oracle loses in fp32, whichever is larger.
        enc_o.double(); dec_o.double()
        for p in list(enc_o.parameters()) + list(dec_o.parameters()):
            p.grad = None
        data64 = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
        ovi.vi_loss(enc_o, dec_o, data64, elbo=False).backward()
        g64 = [p.grad for p in list(enc_o.parameters()) + list(dec_o.parameters())]
        noise = [0.0 if a is None or b is None else _rel(a, b) for a, b in zip(g32, g64)]
    else:
        g64 = g32
    for n, g, go, nz in zip(names, g_hip, g64, noise):
        if go is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        assert g is not None, n
        if float(go.abs().max()) < 1e-12:
            continue
        assert _rel(g, go) <= max(tol_g, 2.0 * nz), (n, _rel(g, go), nz)


def odeint_h(dec_o, enc_o, data):
    with torch.no_grad():
        mu, _ = enc_o(data["measurements"], data["actions"], data["masks"])
        _, h = dec_o(mu, data["actions"])
    return h


def test_state_dict_roundtrip_and_training_step_changes_only_optimised_params(tmp_path):
    """run_simulation.py optimises encoder + readout + ml_net only (reference :125-129); the 13 rate constants stay."""
    from hode import synth
    dev = _dev()
    obs, D, T, B = 40, 8, 16, 32
    torch.manual_seed(2)
    enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density, mc_size=10)
    params = list(enc.parameters()) + list(dec.output_function.parameters()) + list(dec.ode.ml_net.parameters())
    opt = torch.optim.Adam(params, lr=0.01)
    sol = synth.solver_inputs(B, T, D, seed=4)
    ob = synth.observation_inputs(B, T, obs, seed=4)
    data = {k: v.to(dev) for k, v in {"measurements": ob["measurements"], "actions": sol["actions"], "masks": ob["masks"]}.items()}
    before = {k: v.clone() for k, v in dec.state_dict().items()}
    l0 = vi.loss(data)
    l0.backward()
    opt.step()
    after = dec.state_dict()
    assert torch.equal(before["ode.kel"], after["ode.kel"]) and not torch.equal(before["ode.ml_net.0.weight"], after["ode.ml_net.0.weight"])
    vi.save(str(tmp_path) + "/", 1, float(l0))
    ck = torch.load(str(tmp_path) + "/" + vi.model_name)
    assert set(ck) == {"itr", "encoder_state_dict", "decoder_state_dict", "best_loss"}
    dec.load_state_dict(ck["decoder_state_dict"])
    assert torch.isfinite(l0)


@pytest.mark.parametrize("noise_level", [0.2, 0.4, 0.6, 0.8, 1.0])
@pytest.mark.parametrize("method_name,D,roche", [("hybrid", 12, True), ("expert", 4, True), ("neural", 12, False)])
def test_noise_level_sweep_loss_and_adjoint_grads(noise_level, method_name, D, roche):
    """BASELINE config 4: the noise-level sweep (`experiments/run_noise_level.sh`, `generated_data/generate_data_noise.py`:
    measurements += randn * (level - 0.2), seed 666) through loss + discrete-adjoint backward for the three model
    families the sweep trains (`--method` neural / expert / hybrid), rk4, against the CPU oracle."""
    from hode import synth
    dev = _dev()
    obs, T, B, step = 80, 24, 40, synth.STEP
    torch.manual_seed(2)
    enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, roche=roche, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, elbo=False)
    enc_o = EncoderLSTMOracle(obs + 1, obs * 2, D)
    dec_o = ovi.DecoderOracle(obs, D, (T - 1) * step, step, roche=roche, method="rk4")
    enc_o.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})
    dec_o.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    sol = synth.solver_inputs(B, T, D, seed=5)
    ob = synth.observation_inputs(B, T, obs, seed=5)
    torch.manual_seed(666)
    x = ob["measurements"] + torch.randn_like(ob["measurements"]) * (noise_level - 0.2)
    data = {"measurements": x, "actions": sol["actions"], "masks": ob["masks"]}
    loss = vi.loss({k: v.to(dev) for k, v in data.items()})
    loss.backward()
    loss_o = ovi.vi_loss(enc_o, dec_o, data, elbo=False)
    loss_o.backward()
    assert abs(loss.item() - loss_o.item()) <= 2e-4 * abs(loss_o.item())
    assert (vi.h_hat.detach().cpu() - odeint_h(dec_o, enc_o, data)).abs().max().item() <= 3e-4
    for (n, p), (_, po) in zip(list(enc.named_parameters()) + list(dec.named_parameters()),
                               list(enc_o.named_parameters()) + list(dec_o.named_parameters())):
        if po.grad is None or float(po.grad.abs().max()) < 1e-12:
            continue
        assert p.grad is not None, n
        assert _rel(p.grad, po.grad) <= 2e-3, (n, noise_level, _rel(p.grad, po.grad))


@pytest.mark.parametrize("roche", [True, False])
@pytest.mark.parametrize("step_size", [0.0625, 0.05])
def test_odeint_step_size_option_sub_steps_like_torchdiffeq(roche, step_size):
    """`hode.odeint(..., options={"step_size": s})` on the mirror's rhs modules: the kernels integrate torchdiffeq's own
    grid t0 + k s and the outputs are read off it (`hode/substep.py`); vs the oracle's sub-stepping on the CPU."""
    import hode
    from hode import synth
    from oracle.solvers import odeint as oracle_odeint
    dev = _dev()
    obs, D, T, B, step = 40, 8, 12, 37, synth.STEP
    torch.manual_seed(4)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, roche=roche, method="rk4", device=dev)
    dec_o = ovi.DecoderOracle(obs, D, (T - 1) * step, step, roche=roche, method="rk4")
    dec_o.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    sol = synth.solver_inputs(B, T, D, seed=6)
    z = sol["z0"].to(dev).requires_grad_(True)
    zo = sol["z0"].clone().requires_grad_(True)
    dec.ode.set_action(sol["actions"].to(dev))
    dec_o.ode.set_action(sol["actions"])
    h = hode.odeint(dec.ode, z, dec.t, method="rk4", options={"step_size": step_size})
    h_o = oracle_odeint(dec_o.ode, zo, dec_o.t, method="rk4", options={"step_size": step_size})
    assert h.shape == h_o.shape == (T, B, D)
    assert (h.detach().cpu() - h_o.detach()).abs().max().item() <= 3e-5
    cot = torch.randn(T, B, D)
    (h * cot.to(dev)).sum().backward()
    (h_o * cot).sum().backward()
    assert _rel(z.grad, zo.grad) <= 1e-4
    for (n, p), (_, po) in zip(dec.ode.named_parameters(), dec_o.ode.named_parameters()):
        if po.grad is None or float(po.grad.abs().max()) < 1e-12:
            continue
        assert _rel(p.grad, po.grad) <= 2e-3, (n, _rel(p.grad, po.grad))


def test_training_and_evaluation_run_from_device_resident_folds(tmp_path, capsys):
    """hode.batches.DeviceFolds behind the mirrored loops: every batch is produced on the device (one
    gather per field, contiguous), the training loop, evaluate and evaluate_horizon consume it unchanged."""
    import training_utils
    from hode import synth
    from hode.batches import DeviceFolds
    dev = _dev()
    T, obs, D = 16, 40, 8
    folds = DeviceFolds.synthetic(192, T, obs, D, 32, 32, dev, seed=4)
    b = folds.get_mini_batch("train", 64)
    assert all(v.is_cuda and v.is_contiguous() and v.shape[1] == 64 for v in b.values())
    torch.manual_seed(1)
    enc = model.EncoderLSTM(obs + 1, obs * 2, D, device=dev)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * synth.STEP, synth.STEP, method="rk4", device=dev)
    vi = model.VariationalInference(enc, dec, prior_log_pdf=model.ExponentialPrior.log_density)
    opt = torch.optim.Adam(list(enc.parameters()) + list(dec.output_function.parameters()) + list(dec.ode.ml_net.parameters()), lr=1e-3)
    w0 = dec.ode.ml_net[0].weight.detach().clone()
    vi, best, _ = training_utils.variational_training_loop(4, folds, vi, 64, opt, 2, path=str(tmp_path) + "/")
    assert best < 1e9 and not torch.equal(dec.ode.ml_net[0].weight.detach(), w0)
    out = training_utils.evaluate(vi, folds, 16, 8, mc_itr=5)
    assert len(out) == 6 and all(v == v for v in out)
    hz = training_utils.evaluate_horizon(vi, folds, 16, 8, mc_itr=3)
    assert hz["rmse_x"].shape == (T - 8,)


@pytest.mark.parametrize("roche,D", [(True, 12), (True, 4), (False, 12)])
def test_dopri5_under_no_grad_with_parameters_takes_the_tape_less_path(roche, D):
    """evaluate() and the validation pass integrate under `torch.no_grad()` with `ml_net`'s Parameters as inputs:
    `needs_input_grad` is still True there (it mirrors requires_grad, not the grad mode), so the wrapper samples the grad
    mode itself.  Under no_grad the solve must run with HODE_FLAG_NO_TAPE (two state rows instead of (16 T + 65) B D 4
    bytes), give the same trajectory bit for bit, and with grad enabled keep the tape."""
    from hode import adaptive, synth
    dev = _dev()
    obs, T, B, step = 40, 20, 64, synth.STEP
    torch.manual_seed(3)
    dec = model.RocheExpertDecoder(obs, D, 1, (T - 1) * step, step, roche=roche, method="dopri5", device=dev)
    sol = synth.solver_inputs(B, T, D, seed=8)
    z0, a = sol["z0"].to(dev), sol["actions"].to(dev)
    assert any(p.requires_grad for p in dec.ode.parameters())
    _, h_grad = dec(z0, a)
    st_grad = dict(adaptive.last_stats)
    with torch.no_grad():
        _, h_eval = dec(z0, a)
    st_eval = dict(adaptive.last_stats)
    assert st_grad["no_tape"] is False and st_eval["no_tape"] is True
    # with a backward to follow: (16 T + 64 + 1) state rows; without: two rows + 24 bytes per possible step (2^20 of them),
    # whatever the batch size
    assert st_grad["workspace_bytes"] >= (16 * T + 64) * B * D * 4, st_grad
    assert st_eval["workspace_bytes"] <= 24 * (1 << 20) + 64 * B * D * 4 + (1 << 16), st_eval
    assert st_eval["n_accepted"] == st_grad["n_accepted"] and st_eval["n_rejected"] == st_grad["n_rejected"]
    assert torch.equal(h_grad.detach(), h_eval)


def test_analytic_kl_function_matches_the_reference_expression():
    """model.analytic_kl (hand-written backward, fewer launches) against the expression of model.py:1188 in fp64."""
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    mu = torch.randn(1000, 12, generator=g).to(dev).requires_grad_(True)
    lv = (0.5 * torch.randn(1000, 12, generator=g)).to(dev).requires_grad_(True)
    out = model.analytic_kl(mu, lv)
    gm, gl = torch.autograd.grad(3.0 * out, (mu, lv))
    mu64, lv64 = mu.detach().double().requires_grad_(True), lv.detach().double().requires_grad_(True)
    assert isinstance(out.grad_fn, torch.autograd.function.BackwardCFunction)   # the fused function ran, not the expression
    ref = torch.mean(-0.5 * torch.sum(1 + lv64 - mu64 ** 2 - lv64.exp(), dim=1), dim=0)
    rm, rl = torch.autograd.grad(3.0 * ref, (mu64, lv64))
    assert abs(out.item() - ref.item()) <= 1e-6 * abs(ref.item())
    assert (gm.double() - rm).abs().max().item() <= 1e-6 * rm.abs().max().item()
    assert (gl.double() - rl).abs().max().item() <= 1e-6 * rl.abs().max().item()


@pytest.mark.parametrize("rows,n,k", [(8192, 45, 44), (10000, 12, 160), (997, 20, 21)])
def test_split_weight_gradient_product_matches_fp64(rows, n, k):
    """model._rows_tn (the heads' weight gradient as a batched product over row slices) against g.T @ x in fp64."""
    dev = _dev()
    gen = torch.Generator().manual_seed(rows)
    g, x = torch.randn(rows, n, generator=gen).to(dev), torch.randn(rows, k, generator=gen).to(dev)
    out = model._rows_tn(g, x)
    ref = g.double().t() @ x.double()
    assert out.shape == (n, k)
    assert _rel(out, ref) <= 2e-6


def test_split_column_sum_matches_fp64():
    dev = _dev()
    g = torch.randn(8192, 45, generator=torch.Generator().manual_seed(3)).to(dev)
    assert _rel(model._rows_sum(g), g.double().sum(dim=0)) <= 2e-6
    g = torch.randn(997, 20, generator=torch.Generator().manual_seed(4)).to(dev)
    assert _rel(model._rows_sum(g), g.double().sum(dim=0)) <= 2e-6
