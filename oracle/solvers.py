"""``odeint`` restated on PyTorch-CPU with ``torchdiffeq==0.2.2`` semantics.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

The reference delegates all solver arithmetic to the third-party package
``torchdiffeq`` pinned at 0.2.2 (reference ``requirements.txt:9``; call sites
``model.py:837``, ``:842``, ``:1116``).  That package is not installed here and cannot be
fetched, so this file restates its *published* algorithm.  **Parity unpinned** at this
boundary: the reference ships no test / golden vector for solver output.  What pins it
instead is listed in ``oracle/__init__.py`` and exercised by ``tests/test_oracle_solvers.py``.

Semantics reproduced (all as published for 0.2.2):

* result[0] = y0; fixed-grid methods step exactly on the output grid ``t`` when no
  ``step_size`` option is given (the reference never forwards one for the synthetic
  decoder, ``model.py:1116``); with ``step_size`` the grid is ``t0 + k*step_size`` clipped
  to ``t[-1]`` and outputs are linearly interpolated.
* the rhs always receives ``t`` converted to the state dtype (fp32); with
  ``perturb=True`` (only ``DecoderReal``, ``model.py:826``) the first stage of a step is
  evaluated at ``nextafter(t0, +inf)`` and the stage at ``t1`` at ``nextafter(t1, -inf)``.
* ``rk4`` is the 3/8-rule; ``midpoint`` and ``euler`` as usual.
* ``dopri5``: Dormand-Prince 5(4) with FSAL, time variables fp64, state/stages fp32,
  tableau rounded to fp32, stages with alpha == 1 evaluated at ``nextafter(t1, -inf)``,
  one batch-global RMS error ratio, controller ``dt *= min(10, max(0.9 ratio^-1/5, dfactor))``
  with ``dfactor = 1 if ratio < 1 else 0.2`` and ``dt*10`` when ``ratio == 0``, Hairer initial
  step with order 4, 4th-order dense output through ``y_mid``.
* gradients are ordinary autograd through every op (the reference imports plain
  ``odeint``, not ``odeint_adjoint``: ``model.py:9-10``); controller quantities are constants.
* failure: non-finite state / dt underflow raise ``AssertionError`` as torchdiffeq does.
"""

from __future__ import annotations

import math
import warnings

import torch

# ----------------------------------------------------------------------------- fixed grid

_ONE_THIRD = 1 / 3
_TWO_THIRDS = 2 / 3


class _Stitch(torch.autograd.Function):
    """Forward: the perturbed value; backward: identity to the unperturbed one (torchdiffeq does the same)."""

    @staticmethod
    def forward(ctx, x, out):
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def _nextafter(x: torch.Tensor, toward_up: bool) -> torch.Tensor:
    with torch.no_grad():
        out = torch.nextafter(x, x + 1 if toward_up else x - 1)
    return _Stitch.apply(x, out)


class _Rhs:
    """Wraps the user rhs: casts ``t`` to the state dtype and applies the optional perturbation."""

    NONE, NEXT, PREV = 0, 1, 2

    def __init__(self, func):
        self.func = func
        self.nfe = 0

    def __call__(self, t, y, perturb=0):
        t = t.to(y.dtype)
        if perturb == self.NEXT:
            t = _nextafter(t, True)
        elif perturb == self.PREV:
            t = _nextafter(t, False)
        self.nfe += 1
        return self.func(t, y)


def _step_euler(f, t0, dt, t1, y0, perturb):
    k1 = f(t0, y0, _Rhs.NEXT if perturb else _Rhs.NONE)
    return dt * k1


def _step_midpoint(f, t0, dt, t1, y0, perturb):
    half = 0.5 * dt
    k1 = f(t0, y0, _Rhs.NEXT if perturb else _Rhs.NONE)
    return dt * f(t0 + half, y0 + k1 * half)


def _step_rk4_38(f, t0, dt, t1, y0, perturb):
    k1 = f(t0, y0, _Rhs.NEXT if perturb else _Rhs.NONE)
    k2 = f(t0 + dt * _ONE_THIRD, y0 + dt * k1 * _ONE_THIRD)
    k3 = f(t0 + dt * _TWO_THIRDS, y0 + dt * (k2 - k1 * _ONE_THIRD))
    k4 = f(t1, y0 + dt * (k1 - k2 + k3), _Rhs.PREV if perturb else _Rhs.NONE)
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


_FIXED = {"euler": _step_euler, "midpoint": _step_midpoint, "rk4": _step_rk4_38}


def _fixed_grid(t, step_size):
    if step_size is None:
        return t
    t0, t_end = t[0], t[-1]
    n = int(torch.ceil((t_end - t0) / step_size + 1).item())
    grid = torch.arange(0, n, dtype=t.dtype, device=t.device) * step_size + t0
    grid[-1] = t_end
    return grid


def _odeint_fixed(func, y0, t, method, step_size=None, perturb=False):
    f = _Rhs(func)
    step = _FIXED[method]
    grid = _fixed_grid(t, step_size)
    out = [y0]
    j = 1
    y = y0
    for t0, t1 in zip(grid[:-1], grid[1:]):
        dt = t1 - t0
        y_next = y + step(f, t0, dt, t1, y, perturb)
        while j < len(t) and t1 >= t[j]:
            if t[j] == t0:
                out.append(y)
            elif t[j] == t1:
                out.append(y_next)
            else:
                out.append(y + (t[j] - t0) / (t1 - t0) * (y_next - y))
            j += 1
        y = y_next
    return torch.stack(out, dim=0)


# ----------------------------------------------------------------------------- dopri5

DP_ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
DP_BETA = (
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
DP_C_SOL = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
#: torchdiffeq's own error weights (= 2/3 of the textbook b5 - b4; sums to 0)
DP_C_ERR = (
    35 / 384 - 1951 / 21600,
    0.0,
    500 / 1113 - 22642 / 50085,
    125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400,
    11 / 84 - 649 / 6300,
    -1.0 / 60.0,
)
DP_C_MID = (
    6025192743 / 30085553152 / 2,
    0.0,
    51252292925 / 65400821598 / 2,
    -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2,
    -1776094331 / 19743644256 / 2,
    11237099 / 235043384 / 2,
)


def _rms(x):
    return x.pow(2).mean().sqrt()


def _initial_step(f, t0, y0, order, rtol, atol, f0):
    dtype = y0.dtype
    t0 = t0.to(dtype)
    scale = atol + torch.abs(y0) * rtol
    d0 = _rms(y0 / scale)
    d1 = _rms(f0 / scale)
    if d0 < 1e-5 or d1 < 1e-5:
        h0 = torch.tensor(1e-6, dtype=dtype)
    else:
        h0 = 0.01 * d0 / d1
    y1 = y0 + h0 * f0
    f1 = f(t0 + h0, y1)
    d2 = _rms((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6, dtype=dtype), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / float(order + 1))
    return torch.min(100 * h0, h1).to(torch.float64)


def _dp_attempt(f, y0, f0, t0, dt, t1, tab):
    """One Dormand-Prince attempt: returns y1, f1 (FSAL), error estimate, stage list."""
    alpha, beta, c_err = tab
    t0 = t0.to(y0.dtype)
    dt = dt.to(y0.dtype)
    t1 = t1.to(y0.dtype)
    ks = [f0]
    for a_i, b_i in zip(alpha, beta):
        if a_i == 1.0:
            ti, pert = t1, _Rhs.PREV
        else:
            ti, pert = t0 + a_i * dt, _Rhs.NONE
        # torchdiffeq: y0 + k[..., :i+1].matmul(beta_i * dt)
        yi = y0 + torch.stack(ks, dim=-1).matmul(b_i * dt).view_as(f0)
        ks.append(f(ti, yi, pert))
    k = torch.stack(ks, dim=-1)
    y1 = yi  # last beta row == solution weights (FSAL)
    return y1, ks[-1], k.matmul(dt * c_err), k


def _interp_fit(y0, y1, k, dt, c_mid):
    dt = dt.type_as(y0)
    y_mid = y0 + k.matmul(dt * c_mid).view_as(y0)
    f0, f1 = k[..., 0], k[..., -1]
    a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
    b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
    c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
    d = dt * f0
    return [y0, d, c, b, a]


def _interp_eval(coef, t0, t1, t):
    x = ((t - t0) / (t1 - t0)).to(coef[0].dtype)
    total = coef[0] + x * coef[1]
    xp = x
    for c in coef[2:]:
        xp = xp * x
        total = total + xp * c
    return total


@torch.no_grad()
def _next_dt(dt, ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5):
    if ratio == 0:
        return dt * ifactor
    if ratio < 1:
        dfactor = 1.0
    ratio = ratio.type_as(dt)
    expo = torch.tensor(order, dtype=dt.dtype).reciprocal()
    factor = torch.min(torch.tensor(ifactor, dtype=dt.dtype), torch.max(safety / ratio ** expo, torch.tensor(dfactor, dtype=dt.dtype)))
    return dt * factor


def _odeint_dopri5(func, y0, t, rtol, atol, max_num_steps=2 ** 31 - 1, stats=None):
    f = _Rhs(func)
    dtype = y0.dtype
    tab = (
        tuple(float(torch.tensor(a, dtype=torch.float64).to(dtype)) for a in DP_ALPHA),
        tuple(torch.tensor(b, dtype=torch.float64).to(dtype) for b in DP_BETA),
        torch.tensor(DP_C_ERR, dtype=torch.float64).to(dtype),
    )
    c_mid = torch.tensor(DP_C_MID, dtype=torch.float64).to(dtype)
    t = t.to(torch.float64)
    f0 = f(t[0], y0)
    dt = _initial_step(f, t[0], y0, 4, rtol, atol, f0)
    t_lo = t_hi = t[0]
    y, fy = y0, f0
    coef = [y0] * 5
    out = [y0]
    n_acc = n_rej = 0
    tape = []
    first_ok = False
    for j in range(1, len(t)):
        n_steps = 0
        while t[j] > t_hi:
            assert n_steps < max_num_steps, "max_num_steps exceeded"
            assert t_hi + dt > t_hi, "underflow in dt {}".format(dt.item())
            assert torch.isfinite(y).all(), "non-finite values in state `y`"
            t_new = t_hi + dt
            y1, f1, err, k = _dp_attempt(f, y, fy, t_hi, dt, t_new, tab)
            tol = atol + rtol * torch.max(y.abs(), y1.abs())
            ratio = _rms(err / tol).abs()
            if ratio <= 1:
                coef = _interp_fit(y, y1, k, dt, c_mid)
                if not tape:
                    first_ok = n_rej == 0
                tape.append((float(t_hi.detach()), float(dt.detach())))
                t_lo, t_hi = t_hi, t_new
                y, fy = y1, f1
                n_acc += 1
            else:
                t_lo = t_hi  # torchdiffeq stores (t0, t_next=t0) on reject; interp is never read then
                n_rej += 1
            dt = _next_dt(dt, ratio)
            n_steps += 1
        out.append(_interp_eval(coef, t_lo, t_hi, t[j]))
    if stats is not None:
        stats.update(n_accepted=n_acc, n_rejected=n_rej, nfe=f.nfe, tape=tape, first_attempt_accepted=first_ok)
    return torch.stack(out, dim=0)


def odeint_dopri5_replay(func, y0, t, rtol, atol, tape, first_attempt_accepted=True, stats=None):
    """The accepted-step algebra of ``_odeint_dopri5`` driven along a GIVEN tape of ``(t_n, dt_n)`` pairs instead of by
    the controller (test infrastructure: lets a parity test follow the step sequence another implementation took, so
    that accept / reject decisions that flip on the last bit of the error norm drop out of the comparison).

    What autograd sees is what it sees in ``_odeint_dopri5``: every ``dt_n`` with n >= 1 is a constant (the controller
    runs under ``no_grad``); ``dt_0`` is Hairer's differentiable initial step when the first attempt was the one that got
    accepted (``first_attempt_accepted``), a constant otherwise; step boundaries are ``t_{n+1} = t_n + dt_n`` and therefore
    carry ``dt_0``'s gradient.  Values are taken from the tape, gradients from the formulas (value-stitching), so the
    result is the gradient of torchdiffeq's graph evaluated on the tape's step sequence.
    """
    f = _Rhs(func)
    dtype = y0.dtype
    tab = (
        tuple(float(torch.tensor(a, dtype=torch.float64).to(dtype)) for a in DP_ALPHA),
        tuple(torch.tensor(b, dtype=torch.float64).to(dtype) for b in DP_BETA),
        torch.tensor(DP_C_ERR, dtype=torch.float64).to(dtype),
    )
    c_mid = torch.tensor(DP_C_MID, dtype=torch.float64).to(dtype)
    t = t.to(torch.float64)
    f0 = f(t[0], y0)
    out = [y0]
    if len(tape) == 0:
        return torch.stack(out + [y0] * (len(t) - 1), dim=0)
    dt0 = torch.tensor(float(tape[0][1]), dtype=torch.float64)
    if first_attempt_accepted:
        dt_init = _initial_step(f, t[0], y0, 4, rtol, atol, f0)
        dt0 = dt0 + (dt_init - dt_init.detach())
    if stats is not None and dt0.requires_grad:
        dt0.retain_grad()  # after backward: stats["dt0"].grad is d loss / d dt_0
        stats["dt0"] = dt0
    shift = dt0 - dt0.detach()  # zero-valued carrier of d/d(dt_0) for every later step boundary
    y, fy = y0, f0
    j = 1
    for n, (t_n, dt_n) in enumerate(tape):
        if j >= len(t):
            break
        if n == 0:
            t_lo, dt = t[0], dt0
        else:
            t_lo, dt = torch.tensor(float(t_n), dtype=torch.float64) + shift, torch.tensor(float(dt_n), dtype=torch.float64)
        t_hi = t_lo + dt
        y1, f1, _, k = _dp_attempt(f, y, fy, t_lo, dt, t_hi, tab)
        if t[j] <= t_hi:
            coef = _interp_fit(y, y1, k, dt, c_mid)
            while j < len(t) and t[j] <= t_hi:
                out.append(_interp_eval(coef, t_lo, t_hi, t[j]))
                j += 1
        y, fy = y1, f1
    assert j == len(t), "tape ends before the last output time"
    return torch.stack(out, dim=0)


# ----------------------------------------------------------------------------- public entry


def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, stats=None):
    """Drop-in restatement of ``torchdiffeq.odeint`` for the methods the reference uses.

    ``func(t, y) -> dy`` with y of shape (B, D); ``t`` strictly increasing 1-D.  Returns
    (len(t), B, D).  ``stats`` (a dict), if given, receives step counts for ``dopri5``.
    """
    method = method or "dopri5"
    options = dict(options or {})
    if method == "dopri5":
        for key in ("step_size", "perturb"):
            options.pop(key, None)
        if options.pop("step_t", None) is not None:
            raise NotImplementedError("step_t is not restated for dopri5 (unused on the reference's dopri5 path)")
        return _odeint_dopri5(func, y0, t, rtol, atol, stats=stats)
    if method not in _FIXED:
        raise ValueError("oracle odeint: method {!r} not restated".format(method))
    step_size = options.pop("step_size", None)
    perturb = bool(options.pop("perturb", False))
    for key in list(options):  # torchdiffeq warns on unused solver kwargs (e.g. step_t for fixed grids)
        warnings.warn("{}: Unexpected arguments {}".format(method, {key: options.pop(key)}))
    return _odeint_fixed(func, y0, t, method, step_size, perturb)
