"""TEST INFRASTRUCTURE (not part of the product, never imported by it): Dormand-Prince 5(4) with the rhs as the module's own
torch launches per stage -- an independent cross-check of the fused adaptive kernels (tests/test_hip_neural.py,
tests/test_adaptive_eager.py).  Until round 3 this module served (NeuralODE, "dopri5") at latent dimensions without a fused
kernel from inside the product; the kernels now cover every even dimension 4..14 and the product raises HodeConfigError
elsewhere.

``torchdiffeq.odeint(func, y0, t, rtol=, atol=, method="dopri5")`` semantics (the reference's default solver,
sim_config.py:50, reached with the neural rhs by ``run_simulation --method=neural``; call site model.py:1116) on the
tensors' own device: the rhs is the module's torch ``forward(t, y)``, i.e. GPU element-wise / GEMM launches per stage --
the way the reference itself runs -- NOT a hand-written kernel.  The fused adaptive kernels (``hode.adaptive``,
``csrc/hode_dopri5*``) cover the Roche rhs; this module exists so that the remaining (rhs, "dopri5") combinations of the
reference's call surface integrate instead of raising.  Callers announce it once with a warning (model.py).

What is different from running torchdiffeq's autograd graph: the attempt loop runs under ``no_grad`` and only the accepted
steps' (t_n, dt_n, y_n, f_n) are kept; the backward re-evaluates one accepted step at a time under autograd and carries the
two cotangents (state, FSAL derivative) down the tape -- the same discrete adjoint the HIP backward kernel implements
(``dp_bwd_body``), O(1) graph memory instead of one graph node per stage of every step.  Step sizes dt_1, dt_2, ... are
constants for differentiation, as in torchdiffeq (its controller runs under ``no_grad``); dt_0 -- Hairer's initial step,
which torchdiffeq computes with autograd on -- is differentiated when attempt 0 is the accepted one: the sweep collects
sigma = d loss / d dt_0 (dt_0 itself in step 0, the shift of every later step boundary) and pushes it through
``_initial_step_ad`` (the same term ``hode_dopri5_bwd`` adds, csrc/hode_dopri5_kernels.hpp).

Semantics kept: time in float64, state / stages in the state dtype with the tableau rounded to it;
stage times formed in the state dtype; stages with alpha == 1 evaluated at ``nextafter(t1, -inf)``; one batch-global RMS
error ratio; ``dt *= min(10, max(0.9 ratio^(-1/5), dfactor))`` with ``dfactor = 1 if ratio < 1 else 0.2`` (x10 when the
ratio is 0); Hairer's initial step with order 4; quartic dense output through ``y_mid``; steps are not clipped to the
output grid; failures are ``HodeError`` (a ``RuntimeError``: non-finite state, dt underflow).
"""
import numpy as np
import torch

from hode import _lib as L

_ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_BETA = (
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
_C_ERR = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 - -12231 / 42400,
          11 / 84 - 649 / 6300, -1.0 / 60.0)
_C_MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
          187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)

last_stats = {"n_accepted": 0, "n_rejected": 0, "nfe": 0}


class _Tableau:
    def __init__(self, ref):
        cast = lambda v: torch.tensor(v, dtype=torch.float64).to(dtype=ref.dtype, device=ref.device)
        self.np_t = np.float32 if ref.dtype == torch.float32 else np.float64
        self.alpha = tuple(self.np_t(a) for a in _ALPHA)
        self.beta = tuple(cast(b) for b in _BETA)
        self.c_err = cast(_C_ERR)
        self.c_mid = cast(_C_MID)


def _scalar(v, ref):
    return torch.full((), float(v), dtype=ref.dtype, device=ref.device)


def _stage_times(tab, t0, dt, t1):
    """Stage times 2..7 in the STATE dtype (torchdiffeq casts t0, dt, t1 to it before forming t0 + alpha dt)."""
    f = tab.np_t
    t0s, dts, t1s = f(t0), f(dt), f(t1)
    return [float(np.nextafter(t1s, f(-np.inf))) if a == 1 else float(t0s + a * dts) for a in tab.alpha]


def _attempt(func, tab, y, f0, t0, dt, t1):
    """One attempt from (y, f0): y1, f1 (FSAL) and the stacked stage derivatives k (..., 7)."""
    dts = _scalar(tab.np_t(dt), y)
    ks = [f0]
    yi = y
    for ti, b in zip(_stage_times(tab, t0, dt, t1), tab.beta):
        yi = y + torch.stack(ks, dim=-1).matmul(b * dts).view_as(f0)
        ks.append(func(_scalar(ti, y), yi))
    return yi, ks[-1], torch.stack(ks, dim=-1), dts


def _dense_coefficients(tab, y, y1, k, dts):
    y_mid = y + k.matmul(dts * tab.c_mid).view_as(y)
    f0, f1 = k[..., 0], k[..., -1]
    a = 2 * dts * (f1 - f0) - 8 * (y1 + y) + 16 * y_mid
    b = dts * (5 * f0 - 3 * f1) + 18 * y + 14 * y1 - 32 * y_mid
    c = dts * (f1 - 4 * f0) - 11 * y - 5 * y1 + 16 * y_mid
    return (y, dts * f0, c, b, a)


def _dense_eval(coef, t0, t1, tj, ref):
    x = _scalar((tj - t0) / (t1 - t0), ref)
    total = coef[0] + x * coef[1]
    xp = x
    for c in coef[2:]:
        xp = xp * x
        total = total + xp * c
    return total


def _rms(x):
    return float(x.pow(2).mean().sqrt())


def _initial_step(func, tab, t0, y0, f0, rtol, atol):
    f = tab.np_t
    scale = atol + y0.abs() * rtol
    d0, d1 = f(_rms(y0 / scale)), f(_rms(f0 / scale))
    h0 = f(1e-6) if (d0 < 1e-5 or d1 < 1e-5) else f(f(0.01) * d0 / d1)
    y1 = y0 + _scalar(h0, y0) * f0
    f1 = func(_scalar(f(t0) + h0, y0), y1)
    d2 = f(f(_rms((f1 - f0) / scale)) / h0)
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(f(1e-6), f(h0 * f(1e-3)))
    else:
        h1 = f(f(f(0.01) / max(d1, d2)) ** f(1.0 / 5.0))
    return float(min(f(100) * h0, h1))


def _attempt_ad(func, tab, y, f0, t0, dt):
    """``_attempt`` with the step's start time and size as float64 0-dim TENSORS (the backward differentiates them).
    Same fp32 stage-time arithmetic as ``_stage_times``; alpha == 1 stages at nextafter(t1, -inf) with a pass-through
    gradient (torchdiffeq's _StitchGradient)."""
    t0s, dts, t1s = t0.to(y.dtype), dt.to(y.dtype), (t0 + dt).to(y.dtype)
    ks = [f0]
    yi = y
    for a, b in zip(tab.alpha, tab.beta):
        if a == 1:
            ti = t1s + (torch.nextafter(t1s.detach(), t1s.detach() - 1) - t1s.detach())
        else:
            ti = t0s + float(a) * dts
        yi = y + torch.stack(ks, dim=-1).matmul(b * dts).view_as(f0)
        ks.append(func(ti, yi))
    return yi, ks[-1], torch.stack(ks, dim=-1), dts


def _initial_step_ad(func, tab, t0, y0, f0, rtol, atol):
    """Hairer's initial step as a differentiable function of ``y0`` and the parameters (through ``f0``, ``f1``): what
    torchdiffeq's ``_select_initial_step`` leaves in the reference's autograd graph (oracle/solvers.py::_initial_step)."""
    rms = lambda x: x.pow(2).mean().sqrt()
    scale = atol + y0.abs() * rtol
    d0, d1 = rms(y0 / scale), rms(f0 / scale)
    if float(d0.detach()) < 1e-5 or float(d1.detach()) < 1e-5:
        h0 = torch.full((), 1e-6, dtype=y0.dtype, device=y0.device)
    else:
        h0 = 0.01 * d0 / d1
    f1 = func(_scalar(t0, y0) + h0, y0 + h0 * f0)
    d2 = rms((f1 - f0) / scale) / h0
    if float(d1.detach()) <= 1e-15 and float(d2.detach()) <= 1e-15:
        h1 = torch.max(torch.full((), 1e-6, dtype=y0.dtype, device=y0.device), h0 * 1e-3)
    else:
        h1 = (0.01 / (d2 if float(d2.detach()) > float(d1.detach()) else d1)) ** (1.0 / 5.0)
    return torch.min(100 * h0, h1)


def _next_dt(dt, ratio):
    if ratio == 0:
        return dt * 10.0
    dfactor = 1.0 if ratio < 1 else 0.2
    return dt * min(10.0, max(0.9 / ratio ** 0.2, dfactor))


class _Dopri5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, func, tt, rtol, atol, max_num_steps, detach_first_step, y0, *params):
        tab = _Tableau(y0)
        nfe = 0
        with torch.no_grad():
            y = y0.detach()
            f0 = func(_scalar(tt[0], y), y)
            dt = _initial_step(func, tab, tt[0], y, f0, rtol, atol)
            nfe += 2
            t_hi = tt[0]
            out = [y]
            steps = []  # accepted steps: (t0, dt, y_n, f_n, first output index, one past the last)
            n_rej = 0
            first_accepted = False  # attempt 0 (the one that ran with Hairer's dt_0) is on the tape
            j = 1
            while j < len(tt):
                if len(steps) + n_rej >= max_num_steps:
                    raise L.HodeError("hode dopri5: max_num_steps exceeded")
                if not t_hi + dt > t_hi:
                    raise L.HodeError("hode dopri5: underflow in dt %r" % dt)
                if not bool(torch.isfinite(y).all()):
                    raise L.HodeError("hode dopri5: non-finite values in state `y`")
                t_new = t_hi + dt
                y1, f1, k, dts = _attempt(func, tab, y, f0, t_hi, dt, t_new)
                nfe += 6
                tol = atol + rtol * torch.max(y.abs(), y1.abs())
                ratio = abs(_rms(k.matmul(dts * tab.c_err).view_as(y) / tol))
                if ratio <= 1:
                    coef = _dense_coefficients(tab, y, y1, k, dts)
                    j0 = j
                    while j < len(tt) and tt[j] <= t_new:
                        out.append(_dense_eval(coef, t_hi, t_new, tt[j], y))
                        j += 1
                    if not steps:
                        first_accepted = n_rej == 0
                    steps.append((t_hi, dt, y, f0, j0, j))
                    t_hi, y, f0 = t_new, y1, f1
                else:
                    n_rej += 1
                    if ratio != ratio:
                        raise L.HodeError("hode dopri5: non-finite error ratio")
                dt = _next_dt(dt, ratio)
        last_stats.update(n_accepted=len(steps), n_rejected=n_rej, nfe=nfe)
        ctx.func, ctx.tab, ctx.steps, ctx.tt, ctx.params = func, tab, steps, tt, params
        ctx.first_accepted, ctx.tol, ctx.y0 = first_accepted and not detach_first_step, (rtol, atol), y0.detach()
        return torch.stack(out, dim=0)

    @staticmethod
    def backward(ctx, grad_h):
        func, tab, steps, tt, params = ctx.func, ctx.tab, ctx.steps, ctx.tt, ctx.params
        wrt = [p for p in params if p.requires_grad]
        g_params = [torch.zeros_like(p) for p in wrt]
        lam_y = torch.zeros_like(grad_h[0])
        lam_f = torch.zeros_like(grad_h[0])
        sigma = 0.0  # d loss / d dt_0
        track = ctx.first_accepted
        for n in range(len(steps) - 1, -1, -1):
            t0, dt, y_n, f_n, j0, j1 = steps[n]
            with torch.enable_grad():
                y = y_n.detach().requires_grad_(True)
                # time leaves: dt_0 itself in step 0; for n >= 1 the shift of the step's start, t_n = t[0] + dt_0 + const
                tvar = torch.tensor(dt if n == 0 else 0.0, dtype=torch.float64, device=y.device, requires_grad=track)
                t0_t = torch.tensor(t0, dtype=torch.float64, device=y.device) + (0.0 if n == 0 else tvar)
                dt_t = tvar if n == 0 else torch.tensor(dt, dtype=torch.float64, device=y.device)
                if n == 0:
                    f0 = func(_scalar(t0, y), y)  # the first derivative is a function of y0 and the parameters
                    leaves = [y]
                else:
                    f0 = f_n.detach().requires_grad_(True)  # k7 of step n-1: its cotangent is handed down the tape
                    leaves = [y, f0]
                y1, f1, k, dts = _attempt_ad(func, tab, y, f0, t0_t, dt_t)
                total = (y1 * lam_y).sum() + (f1 * lam_f).sum()
                if j1 > j0:
                    coef = _dense_coefficients(tab, y, y1, k, dts)
                    t1_t = t0_t + dt_t
                    for j in range(j0, j1):
                        x = ((tt[j] - t0_t) / (t1_t - t0_t)).to(y.dtype)
                        out, xp = coef[0] + x * coef[1], x
                        for c in coef[2:]:
                            xp = xp * x
                            out = out + xp * c
                        total = total + (out * grad_h[j]).sum()
                grads = torch.autograd.grad(total, leaves + wrt + ([tvar] if track else []), allow_unused=True)
            if track:
                if grads[-1] is not None:
                    sigma += float(grads[-1])
                grads = grads[:-1]
            lam_y = grads[0] if grads[0] is not None else torch.zeros_like(lam_y)
            if n > 0:
                lam_f = grads[1] if grads[1] is not None else torch.zeros_like(lam_f)
            for acc, g in zip(g_params, grads[len(leaves):]):
                if g is not None:
                    acc.add_(g)
        grad_y0 = lam_y + grad_h[0]
        last_stats["sigma"] = sigma
        if track and sigma != 0.0 and steps:
            (rtol, atol) = ctx.tol
            with torch.enable_grad():
                y = ctx.y0.detach().requires_grad_(True)
                f0 = func(_scalar(tt[0], y), y)
                dt0 = _initial_step_ad(func, tab, tt[0], y, f0, rtol, atol)
                grads = torch.autograd.grad(dt0, [y] + wrt, allow_unused=True)
            if grads[0] is not None:
                grad_y0 = grad_y0 + sigma * grads[0]
            for acc, g in zip(g_params, grads[1:]):
                if g is not None:
                    acc.add_(sigma * g)
        it = iter(g_params)
        return (None, None, None, None, None, None, grad_y0) + tuple(next(it) if p.requires_grad else None for p in params)


def odeint_dopri5(func, y0, t, rtol=1e-7, atol=1e-9, max_num_steps=2 ** 31 - 1, detach_first_step=False):
    """``h (T, B, D)`` of ``dy/dt = func(t, y)`` on the output grid ``t``; differentiable w.r.t. ``y0`` and
    ``func.parameters()`` (``func`` is an ``nn.Module``).  Runs on the device of ``y0``.  ``detach_first_step`` treats
    Hairer's dt_0 as a constant (torchdiffeq's graph differentiates it; the default follows torchdiffeq)."""
    tt = [float(v) for v in t.detach().to(torch.float64).cpu()]  # the output grid, read back once
    return _Dopri5.apply(func, tt, float(rtol), float(atol), int(max_num_steps), bool(detach_first_step), y0,
                         *tuple(func.parameters()))
