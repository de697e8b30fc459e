"""Host-side mirror of the reference's ``model.py`` call surface for the hybrid-ODE hot path.

Same class names, constructor signatures, attribute names, parameter creation order and ``state_dict`` keys as the
reference (so ``run_simulation.py`` / ``training_utils.py`` style drivers and reference checkpoints drop in), but the
latent ODE is integrated by the gfx950 kernels of ``libhode.so`` (``hode.odeint``) instead of ``torchdiffeq``:

    reference                                  here
    RocheODE            model.py:446-555       RocheODE            (+ ``hode_solve``: kernel dispatch)
    RocheExpertDecoder  model.py:1030-1121     RocheExpertDecoder  (forward = set_action -> hode.odeint -> readout)
    EncoderLSTM         model.py:383-440       EncoderLSTM
    GaussianReparam / ExponentialPrior / StandardNormalPrior  model.py:18-45
    VariationalInference model.py:1124-1214    VariationalInference

    EncoderLSTMReal / RocheODEReal / DecoderReal / VariationalInferenceReal  model.py:180-242, 570-657, 772-862, 1217-1261

Out of scope: flow encoders, baselines (NeuralODEReal*, DecoderRealBenchmark).
"""

from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

import hode
import sim_config
from global_config import DTYPE, get_device

_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


class GaussianReparam:
    """Diagonal-Gaussian posterior helpers (reference model.py:18-31)."""

    @staticmethod
    def reparameterize(mu, log_var):
        sigma = torch.exp(0.5 * log_var)
        return torch.randn_like(sigma) * sigma + mu

    @staticmethod
    def log_density(mu, log_var, z):
        sigma = torch.exp(0.5 * log_var)
        lp = -((z - mu) ** 2) / (2 * sigma ** 2) - sigma.log() - _LOG_SQRT_2PI
        return lp.sum(dim=-1)


class _AnalyticKL(torch.autograd.Function):
    """mean_b( -0.5 sum_d(1 + log_var - mu^2 - exp(log_var)) ) (model.py:1188) with a hand-written backward: 8 launches
    instead of the 18 the expression and its autograd graph take -- on a (B, D) operand every one of them is launch latency."""

    @staticmethod
    def forward(ctx, mu, log_var):
        e = torch.exp(log_var)
        ctx.save_for_backward(mu, e)
        B, D = mu.shape
        return (torch.sum(torch.addcmul(log_var, mu, mu, value=-1.0) - e) + float(B * D)) * (-0.5 / B)

    @staticmethod
    def backward(ctx, g):
        mu, e = ctx.saved_tensors
        c = g / mu.shape[0]
        return c * mu, (e - 1.0) * (0.5 * c)


def analytic_kl(mu, log_var):
    if mu.is_cuda and mu.dim() == 2:
        return _AnalyticKL.apply(mu, log_var)
    return torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)


class StandardNormalPrior:
    @staticmethod
    def log_density(z):
        return (-0.5 * z ** 2 - _LOG_SQRT_2PI).sum(dim=-1)


class ExponentialPrior:
    """Exponential(rate 100) prior on the initial latents (reference model.py:41-45)."""

    rate = 100.0

    @staticmethod
    def log_density(z):
        return (math.log(ExponentialPrior.rate) - ExponentialPrior.rate * z).sum(dim=-1)


class EncoderLSTM(nn.Module, GaussianReparam):
    """Masked, reverse-time single-layer LSTM over the observation window -> (mu, log_var) of z0 (model.py:383-440)."""

    def __init__(self, input_dim, hidden_dim, output_dim, normalize=True, device=None):
        super().__init__()
        self.device = get_device() if device is None else device
        self.hidden_dim = hidden_dim
        self.normalize = normalize
        self.model_name = "LSTMEncoder"
        # creation order lstm -> lin -> log_var fixes both the seeded init and the state_dict key order
        self.lstm = nn.LSTM(input_dim, hidden_dim).to(self.device)
        self.lin = nn.Linear(hidden_dim, output_dim).to(self.device)
        self.log_var = nn.Linear(hidden_dim, output_dim).to(self.device)

    def final_hidden(self, x, a, mask):
        """h after walking t = T-1 .. 0 on cat(x,a)*cat(mask,1) (reference model.py:415-422).

        HIP tensors: the fp32-MFMA window kernel (``hode.lstm``: mask/concat fused, h and c stay on chip, BPTT kernel
        for the backward).  CPU tensors (host-logic tests): one ``nn.LSTM`` sequence call on the flipped, pre-masked
        input -- the reference's own operator, same arithmetic as its T single-step calls."""
        if x.dim() == 3 and x.shape[1] == 1:
            raise RuntimeError("EncoderLSTM: batch size 1 is not supported (the reference's x.squeeze() drops the batch axis)")
        p = self.lstm
        if x.is_cuda:
            from hode.lstm import lstm_encode
            return lstm_encode(x, a, mask, p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0, reverse=True)
        seq = torch.cat([x * mask, a], dim=-1).flip(0)
        _, (h, _) = p(seq)
        return h[0]

    def forward(self, x, a, mask):
        h = self.final_hidden(x, a, mask)
        if h.is_cuda:   # same numbers as nn.Linear; the weight gradient is a split product over the patients (_rows_tn)
            mu = _TallLinear.apply(h, self.lin.weight, self.lin.bias)
            log_var = _TallLinear.apply(h, self.log_var.weight, self.log_var.bias)
        else:
            mu, log_var = self.lin(h), self.log_var(h)
        if self.normalize:
            mu = torch.exp(mu) / 10
            log_var = log_var - 5.0
        return mu, log_var


dose_schedule_index = hode.solver.dose_schedule_index


_THETA_FIELDS = sim_config.RochConfig._fields  # == parameter creation order of the reference (model.py:468-482)


class RocheODE(nn.Module):
    """Expert PK/PD block + ``tanh(W y + b)`` learned block; ``forward(t, y)`` is the rhs (model.py:446-555)."""

    def __init__(self, latent_dim, action_dim, t_max, step_size, ablate=False, device=None, dtype=DTYPE):
        super().__init__()
        assert action_dim == 1
        self.action_dim = action_dim
        self.latent_dim = int(latent_dim)
        self.expert_dim = 4
        self.ml_dim = self.latent_dim - self.expert_dim
        self.expanded = self.ml_dim > 0
        self.ablate = ablate
        self.device = get_device() if device is None else device
        self.t_max = t_max
        self.step_size = step_size
        cfg = sim_config.RochConfig()
        for name in _THETA_FIELDS:
            setattr(self, name, nn.Parameter(torch.tensor(getattr(cfg, name), device=self.device, dtype=dtype)))
        if self.ablate:
            self.theta_1 = nn.Parameter(torch.tensor(1, device=self.device, dtype=dtype))
            self.theta_2 = nn.Parameter(torch.tensor(2, device=self.device, dtype=dtype))
        if self.expanded:
            self.ml_net = nn.Sequential(nn.Linear(self.latent_dim, self.ml_dim), nn.Tanh()).to(self.device)
        else:
            self.ml_net = nn.Identity().to(self.device)
        self.times = None
        self.dosage = None
        # tuning knobs of the kernel dispatch (not part of the reference surface)
        self.lanes_per_patient = 0
        self.check_finite = False

    # -- dose schedule -------------------------------------------------------------------------------------
    def set_action(self, action):
        """dosage (B,) = max over time; times (B, K) = non-zero grid indices * step_size.  Vectorised: the
        reference's per-patient Python loop (model.py:500-507) costs 0.19 s at 10k patients.

        Finding the dose indices needs the dose count K on the host (one device read-back + ``nonzero``): three host
        synchronisations in the middle of a training step, behind which the host has to enqueue the rest of the step while
        the GPU waits (bench.py ``full_training_step.host_enqueue_ms``).  Two sync-free routes: a batch source that knows
        its data (``hode.batches.DeviceFolds``) attaches the precomputed schedule to the action tensor
        (``action.hode_schedule = (dosage, dose_index)``), and a tensor OBJECT that was analysed before and has not been written to
        since (same object, same version counter) reuses its result."""
        sched = getattr(action, "hode_schedule", None)
        if sched is not None:
            self.dosage, idx = sched
            self.times = idx * self.step_size
            return
        # identity, not address: the cache keeps the tensor alive, so its storage cannot be recycled for another batch
        cached = getattr(self, "_schedule_cache", None)
        if cached is not None and cached[0] is action and cached[1] == action._version:
            self.dosage, self.times = cached[2], cached[3] * self.step_size
            return
        dosage, idx = dose_schedule_index(action)
        self.dosage, self.times = dosage, idx * self.step_size
        self._schedule_cache = (action, action._version, dosage, idx)

    def dose_at_time(self, t):
        on = t >= self.times
        return self.dosage * torch.sum(torch.exp(self.kel * (self.times - t) * on) * on, dim=-1)

    # -- rhs as a torch function (API parity; the solver never calls this) ----------------------------------
    def forward(self, t, y):
        dis, ir, imm, dose2 = y[:, 0], y[:, 1], y[:, 2], y[:, 3]
        if self.ablate:
            cols = [ir, -1.0 * dis * self.theta_1, dose2, -1.0 * imm * self.theta_2]
        else:
            dose = self.dose_at_time(t)
            irp = ir ** self.HillPatho
            cols = [
                dis * self.k_disprog - dis * imm ** self.HillCure * self.k_discure_immunity - dis * ir * self.k_discure_immunereact,
                dis * self.k_immune_disease - ir * self.k_immune_off + dis * ir * self.k_immune_feedback
                + (irp * self.emax_patho) / (self.ec50_patho ** self.HillPatho + irp) - dose2 * ir * self.k_dexa,
                ir * self.k_immunity,
                self.kel * dose - self.kel * dose2,
            ]
        out = torch.stack(cols, dim=-1)
        return torch.cat([out, self.ml_net(y)], dim=-1) if self.expanded else out

    # -- kernel dispatch (what hode.odeint calls) -----------------------------------------------------------
    def theta_vector(self):
        names = list(_THETA_FIELDS) + (["theta_1", "theta_2"] if self.ablate else [])
        return hode.solver.pack_theta([getattr(self, n) for n in names], self.device)

    def hode_solve(self, y0, t, rtol, atol, method, options):
        if self.times is None:
            raise RuntimeError("RocheODE: call set_action(a) before integrating")
        w = self.ml_net[0].weight if self.expanded else None
        b = self.ml_net[0].bias if self.expanded else None
        if method == "dopri5":
            from hode import adaptive
            return adaptive.roche_dopri5(y0, self.theta_vector(), w, b, t, self.dosage, self.times, rtol=rtol, atol=atol,
                                         ablate=self.ablate)
        from hode import substep
        perturb = bool(options.pop("perturb", False))
        theta = self.theta_vector()
        return substep.solve_with_step_size(
            lambda grid: hode.roche_solve(y0, theta, w, b, grid, self.dosage, self.times, method=method, ablate=self.ablate,
                                          perturb=perturb, lanes_per_patient=self.lanes_per_patient,
                                          check_finite=self.check_finite),
            t, options.pop("step_size", None))


class NeuralODE(nn.Module):
    """Pure neural rhs ``tanh(W2 tanh(W1 [y, Dose(t)] + b1) + b2)`` with an impulse dose (model.py:969-1026)."""

    def __init__(self, latent_dim, action_dim, t_max, step_size, device=None, dtype=DTYPE):
        super().__init__()
        assert action_dim == 1
        self.action_dim = action_dim
        self.latent_dim = int(latent_dim)
        self.expert_dim = 4
        self.ml_dim = self.latent_dim
        self.device = get_device() if device is None else device
        self.t_max, self.step_size = t_max, step_size
        self.kel = nn.Parameter(torch.tensor(sim_config.RochConfig().kel, device=self.device, dtype=dtype))  # unused by forward
        d = self.latent_dim
        self.ml_net = nn.Sequential(nn.Linear(d + 1, d * 10), nn.Tanh(), nn.Linear(d * 10, d), nn.Tanh()).to(self.device)
        self.times = None
        self.dosage = None

    set_action = RocheODE.set_action

    def dose_at_time(self, t):
        return self.dosage * torch.sum(self.times == t, dim=-1)

    def forward(self, t, y):
        return self.ml_net(torch.cat([y, self.dose_at_time(t)[:, None]], dim=-1))

    def hode_solve(self, y0, t, rtol, atol, method, options):
        if self.times is None:
            raise RuntimeError("NeuralODE: call set_action(a) before integrating")
        from hode import neural
        if method == "dopri5":
            from hode import adaptive
            if self.latent_dim not in adaptive.NEURAL_DIMS:
                raise hode.HodeConfigError("hode: NeuralODE with method='dopri5' is compiled for latent dimensions %s (got %d); "
                                           "there is no torch-eager path in the product"
                                           % (", ".join(str(d) for d in adaptive.NEURAL_DIMS), self.latent_dim))
            # the reference's default for --method=neural (sim_config.py:50): fused MFMA attempt kernels
            return adaptive.neural_dopri5(y0, self.ml_net[0].weight, self.ml_net[0].bias, self.ml_net[2].weight,
                                          self.ml_net[2].bias, t, self.dosage, self.times, rtol=rtol, atol=atol)
        from hode import substep
        perturb = bool(options.pop("perturb", False))
        return substep.solve_with_step_size(
            lambda grid: neural.neural_solve(y0, self.ml_net[0].weight, self.ml_net[0].bias, self.ml_net[2].weight,
                                             self.ml_net[2].bias, grid, self.dosage, self.times, method=method, perturb=perturb),
            t, options.pop("step_size", None))


class RocheExpertDecoder(nn.Module):
    """z0 -> latent trajectory h (T,B,D) on the observation grid -> linear readout x_hat (model.py:1030-1121)."""

    def __init__(self, obs_dim, latent_dim, action_dim, t_max, step_size, roche=True, ablate=False, method="dopri5",
                 ode_step_size=None, device=None, dtype=DTYPE):
        super().__init__()
        self.time_dim = int(t_max / step_size)
        self.obs_dim, self.latent_dim, self.action_dim = obs_dim, latent_dim, action_dim
        self.t_max, self.step_size = t_max, step_size
        self.roche, self.ablate = roche, ablate
        self.model_name = ("ExpertDecoder" if latent_dim == 4 else "HybridDecoder") if roche else "NeuralODEDecoder"
        if ablate:
            self.model_name += "Ablate"
            print("Running ablation study")
        self.device = get_device() if device is None else device
        self.t = torch.arange(0, t_max + step_size, step_size, device=self.device, dtype=dtype)
        self.options = {"method": method, "rtol": 1e-7, "atol": 1e-8}
        # readout first, then the ode: same parameter creation order as the reference (seeded-init parity)
        self.output_function = nn.Sequential(nn.Linear(latent_dim, obs_dim, bias=True)).to(self.device)
        if roche:
            self.ode = RocheODE(latent_dim, action_dim, t_max, step_size, ablate=ablate, device=self.device)
        else:
            self.ode = NeuralODE(latent_dim, action_dim, t_max, step_size, self.device)
        self._odeint = hode.odeint  # tests swap in the CPU oracle here; the product path never does

    def latent(self, init, a):
        """Latent trajectory h (T, B, D) only."""
        self.ode.set_action(a)
        return self._odeint(self.ode, init, self.t, rtol=self.options["rtol"], atol=self.options["atol"],
                            method=self.options["method"])

    def forward(self, init, a):
        h = self.latent(init, a)
        return self.output_function(h), h

    def fused_likelihood_ok(self, x):
        from hode import readout
        return x.is_cuda and self._odeint is hode.odeint and readout.supported(self.latent_dim, self.obs_dim)

    def masked_sse(self, h, x, mask):
        """sum((x - output_function(h))^2 * mask) / B in one fused pass (x_hat never hits HBM)."""
        from hode import readout
        lin = self.output_function[0]
        return readout.masked_sse_readout(h, x, mask, lin.weight, lin.bias)


class EncoderLSTMReal(nn.Module, GaussianReparam):
    """Real-data encoder (model.py:180-242): LSTM over cat(x, a_in, t/max(mask)) with NO input masking, two-layer tanh
    heads; ``reverse`` flips the window first, ``output_all`` is not used on the hot path (run_real.py passes False)."""

    def __init__(self, input_dim, hidden_dim, output_dim, output_all=False, reverse=True, normalize=True, device=None):
        super().__init__()
        self.device = get_device() if device is None else device
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.normalize = normalize
        self.model_name = "LSTMReal"
        self.lstm = nn.LSTM(input_dim, hidden_dim).to(self.device)
        self.lin = nn.Sequential(nn.Linear(hidden_dim, hidden_dim + 1), nn.Tanh(), nn.Linear(hidden_dim + 1, output_dim), nn.Tanh()).to(self.device)
        self.log_var = nn.Sequential(nn.Linear(hidden_dim, hidden_dim + 1), nn.Tanh(), nn.Linear(hidden_dim + 1, output_dim), nn.Tanh()).to(self.device)
        self.reverse = reverse
        self.output_all = output_all

    def forward(self, x, a, m):
        if self.output_all:
            raise hode.HodeConfigError("EncoderLSTMReal(output_all=True) is outside the accelerated path")
        if self.reverse:
            x, a, m = torch.flip(x, [0]), torch.flip(a, [0]), torch.flip(m, [0])
        T, B = m.shape[0], m.shape[1]
        tt = (torch.arange(T, device=x.device, dtype=x.dtype) / m.max()).view(T, 1, 1).expand(T, B, 1)
        x_in = torch.cat([x, a, tt], dim=-1)
        p = self.lstm
        if x_in.is_cuda:
            from hode.lstm import lstm_encode
            out = lstm_encode(x_in, None, None, p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0, reverse=False)
        else:
            out = p(x_in)[1][0][0]
        if out.is_cuda:
            return _tall_mlp(self.lin, out), _tall_mlp(self.log_var, out)
        return self.lin(out), self.log_var(out)


class RocheODEReal(nn.Module):
    """Real-data hybrid rhs: two small MLPs for x1, x2, expert x3 / Dose2, GRU-ODE block on the rest (model.py:570-657)."""

    def __init__(self, latent_dim, action_dim, static_dim, hidden_dim, t_max, step_size, device=None, dtype=DTYPE):
        super().__init__()
        self.action_dim, self.latent_dim = int(action_dim), int(latent_dim)
        self.static_dim, self.hidden_dim = int(static_dim), int(hidden_dim)
        self.dosage = None
        self._times = None
        self.device = get_device() if device is None else device
        self.t_max, self.step_size = t_max, step_size
        h = self.hidden_dim
        self.dx1_net = nn.Sequential(nn.Linear(3, h), nn.Tanh(), nn.Linear(h, 1), nn.Tanh())
        self.dx2_net = nn.Sequential(nn.Linear(2, h), nn.Tanh(), nn.Linear(h, 1), nn.Tanh())
        self.expert_dim = 4
        self.expert_only = self.latent_dim == self.expert_dim
        if not self.expert_only:
            m = self.latent_dim - self.expert_dim
            self.lin_hh = nn.Linear(m, m, bias=False)
            self.lin_hz = nn.Linear(m, m, bias=False)
            self.lin_hr = nn.Linear(m, m, bias=False)
        self.k_immunity = nn.Parameter(torch.tensor(1, device=self.device, dtype=dtype))
        self.kel = nn.Parameter(torch.tensor(0.2, device=self.device, dtype=dtype))
        self.kel2 = nn.Parameter(torch.tensor(0.2, device=self.device, dtype=dtype))
        self.to(self.device)  # the reference leaves the sub-networks on the default device; everything lives on one here

    def set_action_static(self, action, static):
        self.dosage = action
        self._times = None

    @property
    def times(self):
        """1, 2, .., T along the time axis, shaped like the action (model.py:650-651).  Only the eager rhs (`forward`,
        `dose_at_time`) reads it -- the kernels index time themselves -- so it is formed on first use, not per step."""
        if self._times is None and self.dosage is not None:
            self._times = torch.cumsum(torch.ones_like(self.dosage), dim=0)
        return self._times

    @times.setter
    def times(self, value):
        self._times = value

    def dose_at_time(self, t):
        on = t >= self.times
        return torch.sum(self.dosage * torch.exp(self.kel * (self.times - t) * on) * on, dim=(0, 2))

    def forward(self, t, y):
        dose = self.dose_at_time(t)
        cols = [self.dx1_net(y[:, :3]), self.dx2_net(y[:, :2]), (y[:, 1] * self.k_immunity)[..., None],
                (self.kel * dose - self.kel2 * y[:, 3])[..., None]]
        if not self.expert_only:
            hv = y[..., 4:]
            r = torch.sigmoid(self.lin_hr(hv))
            z = torch.sigmoid(self.lin_hz(hv))
            u = torch.tanh(self.lin_hh(r * hv))
            cols.append((1 - z) * (u - hv))
        return torch.cat(cols, dim=-1)

    def flat_weights(self):
        ps = [self.dx1_net[0].weight, self.dx1_net[0].bias, self.dx1_net[2].weight, self.dx1_net[2].bias,
              self.dx2_net[0].weight, self.dx2_net[0].bias, self.dx2_net[2].weight, self.dx2_net[2].bias]
        if not self.expert_only:
            ps += [self.lin_hh.weight, self.lin_hz.weight, self.lin_hr.weight]
        return torch.cat([p.reshape(-1) for p in ps])

    def hode_solve(self, y0, t, rtol, atol, method, options):
        if self.dosage is None:
            raise RuntimeError("RocheODEReal: call set_action_static(a, s) before integrating")
        from hode import real
        if method == "dopri5":
            # DecoderReal hands dopri5 a `step_t` grid (model.py:826: steps are cut at the hourly dose switches); that
            # option's semantics are not restated by the oracle.  real.sh:15 uses midpoint.
            raise hode.HodeConfigError("hode: RocheODEReal is built for the fixed-grid methods (euler, midpoint, rk4); "
                                 "dopri5 with options['step_t'] is not supported")
        from hode import substep
        step_size = options.pop("step_size", None)
        if step_size is not None and t.numel() > 1:
            # torchdiffeq builds its own grid t0 + k*step_size; when that IS the output grid (run_real.py's default,
            # ode_step_div = 1) nothing has to be interpolated
            # (a host read-back: cached per grid tensor, the decoder hands over the same `t` at every call)
            cache = getattr(self, "_uniform_grid_cache", None)
            if cache is None or cache[0] is not t or cache[1] != (t._version, float(step_size)):
                cache = (t, (t._version, float(step_size)), bool(torch.allclose(t[1:] - t[:-1], torch.full_like(t[1:], float(step_size)))))
                self._uniform_grid_cache = cache
            if cache[2]:
                step_size = None
        options.pop("step_t", None)  # ignored by fixed-grid solvers (torchdiffeq only warns)
        theta = torch.stack([self.k_immunity, self.kel, self.kel2])
        wflat, perturb = self.flat_weights(), bool(options.pop("perturb", False))
        return substep.solve_with_step_size(
            lambda grid: real.real_solve(y0, theta, wflat, grid, self.dosage[..., 0], self.hidden_dim, method=method, perturb=perturb),
            t, step_size)


class _TallLinear(torch.autograd.Function):
    """``x @ W.T + b`` for x with ~1e6 rows and ~20 columns.  Same numbers as ``nn.Linear``; the backward forms the bias
    gradient as a (1 x rows) GEMM -- torch's column sum over such a shape takes 2 ms at 0.8 M x 21 (config 5), two orders
    of magnitude off the HBM rate."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        return g @ weight, _rows_tn(g, x), _rows_sum(g)


_ONES = {}


def _ones(*shape, like):
    """Constant all-ones GEMM operand, kept per (shape, device, dtype): a fill kernel per use is 5 us of launch latency."""
    key = (shape, like.device, like.dtype)
    if key not in _ONES:
        _ONES[key] = torch.ones(shape, device=like.device, dtype=like.dtype)
    return _ONES[key]


def _row_slices(t):
    rows = t.shape[0]
    if t.is_cuda:
        for cand in range(64, 1, -1):
            if rows % cand == 0 and rows // cand >= 128:
                return cand
    return 1


def _rows_sum(g):
    """Column sums of a (rows x n) matrix as GEMMs with a ones vector, over row slices like `_rows_tn`."""
    rows, P = g.shape[0], _row_slices(g)
    if P == 1:
        return (_ones(1, rows, like=g) @ g).reshape(-1)
    part = torch.bmm(_ones(P, 1, rows // P, like=g), g.reshape(P, rows // P, -1))
    return (_ones(1, P, like=g) @ part.view(P, -1)).reshape(-1)


def _rows_tn(g, x):
    """``g.T @ x`` for (rows x n), (rows x k) with rows in the thousands and n, k in the tens: the result is a handful of
    tiles, so the library runs the whole contraction on as many workgroups (57 us for 45 x 44 over 8 192 rows, config 5's
    encoder heads).  The rows are cut into P slices, one batched product forms P partial results on P times as many
    workgroups and a (1 x P) product folds them."""
    rows, P = g.shape[0], _row_slices(g)
    if P == 1:
        return g.t() @ x
    part = torch.bmm(g.reshape(P, rows // P, -1).transpose(1, 2), x.reshape(P, rows // P, -1))   # reshape: x may be a strided view
    return (_ones(1, P, like=g) @ part.view(P, -1)).view(part.shape[1], part.shape[2])


def _tall_mlp(seq, x):
    """Apply an ``nn.Sequential`` of Linear / activation layers to (..., k) rows with ``_TallLinear`` for the Linear ones."""
    shape = x.shape[:-1]
    y = x.reshape(-1, x.shape[-1])
    for layer in seq:
        y = _TallLinear.apply(y, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(y)
    return y.reshape(*shape, y.shape[-1])


class DecoderReal(nn.Module):
    """z0 -> h over t = t0-1 .. t_max-1 -> MLP readout, first output row dropped (model.py:772-862; hybrid ode_type)."""

    def __init__(self, obs_dim, latent_dim, action_dim, static_dim, hidden_dim, t_max, step_size, t0=0, method="dopri5",
                 ode_step_size=None, ode_type="neural", device=None, dtype=DTYPE):
        super().__init__()
        self.time_dim = int(t_max / step_size)
        self.obs_dim, self.latent_dim, self.action_dim = obs_dim, latent_dim, action_dim
        self.t_max, self.t0 = t_max, t0
        self.static_dim, self.hidden_dim = int(static_dim), int(hidden_dim)
        self.model_name = "DecoderReal_" + ode_type
        self.device = get_device() if device is None else device
        self.output_function = nn.Sequential(nn.Linear(latent_dim, latent_dim + 1, bias=True), nn.ELU(),
                                             nn.Linear(latent_dim + 1, obs_dim, bias=True)).to(self.device)
        if ode_type in ("neural", "2nd"):
            raise hode.HodeConfigError("DecoderReal(ode_type=%r): NeuralODEReal baselines are outside the accelerated path" % ode_type)
        self.ode = RocheODEReal(latent_dim, action_dim, static_dim, hidden_dim, t_max, step_size, self.device)
        self.t = torch.arange(t0 - 1, t_max, step_size, device=self.device, dtype=dtype)
        self.options = {"step_t": self.t, "step_size": ode_step_size, "perturb": True}
        self.rtol, self.atol = 1e-7, 1e-8
        self.method = method
        self.step_size = ode_step_size
        self._odeint = hode.odeint

    def forward(self, init, a, s):
        self.ode.set_action_static(a, s)
        if init.dim() != 2:
            raise hode.HodeConfigError("DecoderReal: per-step initial states (3-D init) are outside the accelerated path")
        h = self.latent(init, a, s)
        if h.is_cuda and h.shape[0] * h.shape[1] >= 65536:
            return _tall_mlp(self.output_function, h)[1:], h
        return self.output_function(h)[1:], h

    def latent(self, init, a, s):
        """Latent trajectory h (T - t0 + 1, B, D) only (row 0 = the state at t0 - 1, which the readout drops)."""
        self.ode.set_action_static(a, s)
        if init.dim() != 2:
            raise hode.HodeConfigError("DecoderReal: per-step initial states (3-D init) are outside the accelerated path")
        return self._odeint(self.ode, init, self.t, method=self.method, options=dict(self.options), rtol=self.rtol, atol=self.atol)

    def fused_likelihood_ok(self, x):
        from hode import readout
        return (x.is_cuda and self._odeint is hode.odeint
                and readout.mlp_supported(self.latent_dim, self.output_function[0].out_features, self.obs_dim))

    def masked_sse(self, h, x, mask, time_weight=None):
        """sum((x - output_function(h[1:]))^2 * mask * time_weight) / B in one fused pass over the rows (x_hat never hits
        HBM; the two Linear layers, the ELU, the loss and all their gradients are one kernel)."""
        from hode import readout
        l0, l2 = self.output_function[0], self.output_function[2]
        return readout.masked_sse_readout_mlp(h, x, mask, l0.weight, l0.bias, l2.weight, l2.bias, time_weight, skip_rows=1)


class VariationalInference:
    """Negative ELBO: masked SSE likelihood + KL (analytic vs N(0,1), or Monte-Carlo vs a given prior) (model.py:1124-1214)."""

    epsilon = torch.finfo(DTYPE).eps

    def __init__(self, encoder, decoder, elbo=True, prior_log_pdf=None, mc_size=100):
        self.encoder, self.decoder = encoder, decoder
        self.prior_log_pdf = prior_log_pdf
        self.mc_size = mc_size
        self.elbo = elbo
        self.fuse_likelihood = True  # fused readout + masked-SSE kernel when the decoder supports the shape
        self.fuse_mc_kl = True       # fused Monte-Carlo KL kernel for the Exponential prior
        self.model_name = "VI_{}_{}.pkl".format(encoder.model_name, decoder.model_name)

    def save(self, path, itr, best_loss):
        path = path + self.model_name
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save({"itr": itr, "encoder_state_dict": self.encoder.state_dict(),
                    "decoder_state_dict": self.decoder.state_dict(), "best_loss": best_loss}, path)

    def parameters(self):
        return list(self.encoder.parameters()) + list(self.decoder.parameters())

    def loss(self, data):
        x, a, mask = data["measurements"], data["actions"], data["masks"]
        self.x, self.a, self.mask = x, a, mask
        mu, log_var = self.encoder(x, a, mask)
        self.mu, self.log_var = mu, log_var
        z = self.encoder.reparameterize(mu, log_var) if self.elbo else mu
        self.z = z
        fused = getattr(self.decoder, "fused_likelihood_ok", None)
        if self.fuse_likelihood and fused is not None and fused(x):
            # readout + masked SSE (+ gradients) in one HBM pass; x_hat is produced only if somebody reads vi.x_hat
            h_hat = self.decoder.latent(z, a)
            self.h_hat, self._x_hat = h_hat, None
            lik = self.decoder.masked_sse(h_hat, x, mask)
        else:
            x_hat, h_hat = self.decoder(z, a)
            self._x_hat, self.h_hat = x_hat, h_hat
            lik = torch.sum((x - x_hat) ** 2 * mask) / x.shape[1]
        if not self.elbo:
            return lik
        if self.prior_log_pdf is None:
            kld = analytic_kl(mu, log_var)
        else:
            kld = torch.mean(self.mc_kl(mu, log_var, self.mc_size), dim=0)
        return lik + kld

    @property
    def x_hat(self):
        if getattr(self, "_x_hat", None) is None and getattr(self, "h_hat", None) is not None:
            with torch.no_grad():
                self._x_hat = self.decoder.output_function(self.h_hat)
                if isinstance(self.decoder, DecoderReal):
                    self._x_hat = self._x_hat[1:]  # output_function(h)[1:], model.py:859
        return self._x_hat

    @x_hat.setter
    def x_hat(self, value):
        self._x_hat = value

    def mc_kl(self, mu, log_var, sample_size):
        """E_q[log q - log p] from `sample_size` draws, non-positive draws clamped to eps.  All draws in one batched
        tensor (sample axis first) instead of the reference's Python loop; same RNG stream order."""
        sigma = torch.exp(0.5 * log_var)
        eps = torch.randn((sample_size,) + tuple(sigma.shape), device=sigma.device, dtype=sigma.dtype)
        if (mu.is_cuda and self.prior_log_pdf is ExponentialPrior.log_density
                and type(self.encoder).log_density is GaussianReparam.log_density and self.fuse_mc_kl):
            from hode import mckl  # one kernel, analytic gradients (hode_mc_kl_exponential)
            return mckl.mc_kl_exponential(mu, log_var, eps, ExponentialPrior.rate, self.epsilon).sum(dim=-1)
        z = eps * sigma + mu
        z = torch.where(z <= 0.0, torch.full_like(z, self.epsilon), z)
        log_q = self.encoder.log_density(mu, log_var, z)
        return torch.mean(log_q - self.prior_log_pdf(z), dim=0)


class VariationalInferenceReal(VariationalInference):
    """Real-data objective (model.py:1217-1261): encode the first t0 steps, decode the rest, masked SSE on x[t0:]."""

    def __init__(self, encoder, decoder, elbo=True, prior_log_pdf=None, mc_size=100, t0=24, weight=False):
        super().__init__(encoder, decoder, elbo, prior_log_pdf, mc_size)
        self.t0 = t0
        self.weight = weight

    def loss(self, data):
        x, a, mask, s = data["measurements"], data["actions"], data["masks"], data["statics"]
        t0 = self.t0
        mu, log_var = self.encoder(x[:t0], torch.cat([a[:t0], s[:t0]], dim=-1), mask[:t0])   # only the window is assembled
        z = self.encoder.reparameterize(mu, log_var) if self.elbo else mu
        self.z = z
        fused = getattr(self.decoder, "fused_likelihood_ok", None)
        if self.fuse_likelihood and fused is not None and fused(x):
            # readout MLP + masked, time-weighted SSE (+ every gradient) in one pass; x_hat only if somebody reads vi.x_hat
            h_hat = self.decoder.latent(z, a, s)
            self.h_hat, self._x_hat = h_hat, None
            tw = 1 / torch.arange(1, self.decoder.t_max - t0 + 1, device=x.device, dtype=torch.float32) if self.weight else None
            lik = self.decoder.masked_sse(h_hat, x[t0:], mask[t0:], tw)
        else:
            x_hat, h_hat = self.decoder(z, a, s)
            self.x_hat, self.h_hat = x_hat, h_hat
            if self.weight:
                weight = 1 / torch.arange(1, self.decoder.t_max - t0 + 1, device=x.device)[:, None, None]
            else:
                weight = 1.0
            lik = torch.sum((x[t0:] - x_hat) ** 2 * mask[t0:] * weight) / x[t0:].shape[1]
        if not self.elbo:
            return lik
        if self.prior_log_pdf is None:
            kld = analytic_kl(mu, log_var)
        else:
            kld = torch.mean(self.mc_kl(mu, log_var, self.mc_size), dim=0)
        return lik + kld
