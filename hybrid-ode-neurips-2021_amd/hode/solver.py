"""``odeint`` drop-in backed by the gfx950 kernels (fixed-grid path; dopri5 added in ``adaptive.py``).

Boundary mirrored: ``torchdiffeq.odeint(func, y0, t, *, rtol, atol, method, options)`` as called at reference
``model.py:1116`` / ``:837``.  ``func`` must be one of this package's rhs modules (``model.RocheODE`` ...), which
carry the dose schedule set by ``set_action`` exactly like the reference's modules do.
"""

from __future__ import annotations

import warnings

import torch

from . import _lib as L


def _require_gpu(*tensors):
    for x in tensors:
        if x is not None and not x.is_cuda:
            raise L.HodeConfigError(
                "hode: the solver path runs only on a HIP device (got a %s tensor); there is no CPU fallback" % x.device
            )


def _f32c(x):
    return x.detach().to(torch.float32).contiguous()


def _ptr(x):
    return 0 if x is None else x.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pack_theta(scalars, device):
    """Stack the rhs' 0-dim parameters into the [HODE_N_THETA] vector the kernels read (differentiable)."""
    vec = torch.stack([s.reshape(()) for s in scalars]).to(torch.float32)
    if vec.numel() < L.N_THETA:
        vec = torch.cat([vec, vec.new_zeros(L.N_THETA - vec.numel())])
    return vec


def dose_schedule_index(action):
    """(dosage (B,), dose_index (B, K) int64) of an action tensor (T, B, 1): what ``RocheODE.set_action`` (reference
    model.py:495-507) derives, with the grid indices left unscaled.  Requires equal dose counts per patient, like the
    reference's ``torch.stack``."""
    chan = action[..., 0]
    dosage = torch.max(chan, dim=0)[0]
    hit = (chan != 0).t()
    counts = hit.sum(dim=1)
    k = int(counts[0]) if counts.numel() else 0
    if counts.numel() and not bool((counts == k).all()):
        raise RuntimeError("stack expects each tensor to be equal size (patients have different dose counts)")
    return dosage, torch.nonzero(hit)[:, 1].reshape(hit.shape[0], k)


class _RocheFixedGrid(torch.autograd.Function):
    """h = odeint(RocheODE, y0, t, method) on the gfx950 kernel; backward = discrete adjoint kernel."""

    @staticmethod
    def forward(ctx, y0, theta, w, b, t, dosage, dose_times, method, ablate, perturb, lanes, check_finite):
        _require_gpu(y0, theta, t, dosage, dose_times)
        lib = L.lib()
        B, D = y0.shape
        T = t.numel()
        y0c, thc, tc = _f32c(y0), _f32c(theta), _f32c(t)
        dosc, dtc = _f32c(dosage), _f32c(dose_times)
        wc = _f32c(w) if w is not None else None
        bc = _f32c(b) if b is not None else None
        h = torch.empty((T, B, D), device=y0.device, dtype=torch.float32)
        status = torch.zeros(1, device=y0.device, dtype=torch.int32) if check_finite else None
        d = L.new_solve_desc()
        d.rhs_kind = L.RHS_ROCHE_ABLATE if ablate else L.RHS_ROCHE
        d.method, d.perturb, d.batch, d.latent_dim, d.n_times = method, int(perturb), B, D, T
        d.n_dose = dtc.shape[1] if dtc.dim() == 2 else 0
        d.lanes_per_patient = lanes
        d.t, d.y0, d.dosage, d.dose_times, d.theta = tc.data_ptr(), y0c.data_ptr(), dosc.data_ptr(), _ptr(dtc), thc.data_ptr()
        d.w1, d.b1, d.h, d.status = _ptr(wc), _ptr(bc), h.data_ptr(), _ptr(status)
        # a backward will follow: let the forward leave its stage tape in the buffer the backward gets (HODE_FLAG_TAPE)
        ws = None
        if any(ctx.needs_input_grad[:4]):
            d.flags = L.FLAG_TAPE
            nbytes = lib.hode_workspace_bytes(d, L.WS_RK_FWD)
            if nbytes:
                ws = torch.empty(nbytes, device=y0.device, dtype=torch.uint8)
                d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
            else:
                d.flags = 0
        ctx.tape_ws = ws
        with torch.cuda.device(y0.device):
            L.check(lib.hode_rk_fwd(d, _stream()), "hode_rk_fwd")
        if check_finite and int(status.item()) & L.STATUS_NONFINITE:
            raise L.HodeError("hode: non-finite values in state `y` (fixed-grid solve)")
        ctx.save_for_backward(h, thc, wc if wc is not None else thc, bc if bc is not None else thc, tc, dosc, dtc)
        ctx.meta = (method, ablate, int(perturb), lanes, w is not None)
        return h

    @staticmethod
    def backward(ctx, grad_h):
        h, thc, wc, bc, tc, dosc, dtc = ctx.saved_tensors
        method, ablate, perturb, lanes, has_w = ctx.meta
        lib = L.lib()
        T, B, D = h.shape
        gh = grad_h.to(torch.float32).contiguous()
        need_th = bool(ctx.needs_input_grad[1])
        gy0 = torch.empty((B, D), device=h.device, dtype=torch.float32)
        gth = torch.zeros(L.N_THETA, device=h.device, dtype=torch.float32)
        gw = torch.empty_like(wc) if has_w else None  # HODE_FLAG_OVERWRITE_GRADS: the kernels store, no memset needed
        gb = torch.empty_like(bc) if has_w else None
        d = L.new_solve_desc()
        d.rhs_kind = L.RHS_ROCHE_ABLATE if ablate else L.RHS_ROCHE
        d.method, d.perturb, d.batch, d.latent_dim, d.n_times = method, perturb, B, D, T
        d.n_dose = dtc.shape[1] if dtc.dim() == 2 else 0
        d.lanes_per_patient = lanes
        d.need_theta_grad = int(need_th)
        d.t, d.y0, d.dosage, d.dose_times, d.theta = tc.data_ptr(), h.data_ptr(), dosc.data_ptr(), _ptr(dtc), thc.data_ptr()
        d.w1, d.b1, d.h = (_ptr(wc), _ptr(bc), h.data_ptr()) if has_w else (0, 0, h.data_ptr())
        d.grad_h, d.grad_y0 = gh.data_ptr(), gy0.data_ptr()
        d.grad_w1, d.grad_b1, d.grad_theta = _ptr(gw), _ptr(gb), gth.data_ptr()
        d.flags = L.FLAG_OVERWRITE_GRADS | (L.FLAG_TAPE if ctx.tape_ws is not None else 0)
        nbytes = lib.hode_workspace_bytes(d, L.WS_RK_BWD)
        ws = ctx.tape_ws if ctx.tape_ws is not None else torch.empty(max(nbytes, 4), device=h.device, dtype=torch.uint8)
        if ws.numel() < nbytes:
            raise L.HodeConfigError("hode: tape buffer of the forward (%d B) is smaller than the backward needs (%d B)" % (ws.numel(), nbytes))
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        with torch.cuda.device(h.device):
            L.check(lib.hode_rk_bwd(d, _stream()), "hode_rk_bwd")
        return gy0, (gth if need_th else None), gw, gb, None, None, None, None, None, None, None, None


def roche_solve(y0, theta, w, b, t, dosage, dose_times, method="rk4", ablate=False, perturb=False,
                lanes_per_patient=0, check_finite=False):
    """Functional form: integrate the Roche rhs over grid ``t`` from ``y0``; returns h (T, B, D).

    ``theta`` is the [16] packed vector of expert constants (``pack_theta``), ``w``/``b`` the ``ml_net.0`` weight
    (D-4, D) and bias (D-4,) or ``None`` when D == 4, ``dosage`` (B,), ``dose_times`` (B, K) fp32.
    """
    if method not in L.METHODS:
        raise ValueError("hode.roche_solve: method %r is not a fixed-grid method" % (method,))
    if dose_times.dim() != 2:
        dose_times = dose_times.reshape(y0.shape[0], -1)
    return _RocheFixedGrid.apply(y0, theta, w, b, t, dosage, dose_times.to(torch.float32), L.METHODS[method], bool(ablate),
                                 bool(perturb), int(lanes_per_patient), bool(check_finite))


def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None):
    """Same signature as ``torchdiffeq.odeint``; ``func`` is an rhs module of this package (has ``hode_solve``)."""
    solve = getattr(func, "hode_solve", None)
    if solve is None:
        raise L.HodeConfigError(
            "hode.odeint: %s is not a hode rhs module (no hode_solve); arbitrary Python rhs callables are outside "
            "the accelerated path" % type(func).__name__
        )
    return solve(y0, t, rtol=rtol, atol=atol, method=method or "dopri5", options=dict(options or {}))
