"""Fixed-grid solve of the pure neural latent ODE (reference ``NeuralODE``, ``model.py:969-1026``) on the gfx950 kernels.

The matrix-core backward kernel returns ``grad_y0`` and accumulates the weight gradients on chip (outer products over the
wave's patients as MFMAs, one partial block per wave, fixed-order fold).  The one-patient-per-lane kernels
(``HODE_NEURAL_LAYOUT=t``) tape the operands of those outer products instead and they are contracted here with batched
BLAS GEMMs."""

from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from .solver import _f32c, _ptr, _require_gpu, _stream

_STAGES = {L.METHOD_EULER: 1, L.METHOD_MIDPOINT: 2, L.METHOD_RK4_38: 4}


def _desc(y0, t, dosage, dose_times, w1, b1, w2, b2, h, method, perturb):
    B, D = y0.shape
    d = L.new_solve_desc()
    d.rhs_kind, d.method, d.perturb = L.RHS_NEURAL, method, int(perturb)
    d.batch, d.latent_dim, d.n_times, d.hidden_dim = B, D, t.numel(), w1.shape[0]
    d.n_dose = dose_times.shape[1] if dose_times.dim() == 2 else 0
    d.t, d.y0, d.dosage, d.dose_times = t.data_ptr(), y0.data_ptr(), dosage.data_ptr(), _ptr(dose_times)
    d.w1, d.b1, d.w2, d.b2, d.h = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), h.data_ptr()
    return d


class _NeuralFixedGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, w1, b1, w2, b2, t, dosage, dose_times, method, perturb):
        _require_gpu(y0, w1, t, dosage, dose_times)
        lib = L.lib()
        y0c, tc, dosc, dtc = _f32c(y0), _f32c(t), _f32c(dosage), _f32c(dose_times)
        w1c, b1c, w2c, b2c = _f32c(w1), _f32c(b1), _f32c(w2), _f32c(b2)
        B, D = y0c.shape
        h = torch.empty((tc.numel(), B, D), device=y0.device, dtype=torch.float32)
        d = _desc(y0c, tc, dosc, dtc, w1c, b1c, w2c, b2c, h, method, perturb)
        n = lib.hode_workspace_bytes(d, L.WS_RK_FWD)
        ws = torch.empty(max(n, 4), device=y0.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), n
        with torch.cuda.device(y0.device):
            L.check(lib.hode_rk_fwd(d, _stream()), "hode_rk_fwd[neural]")
        ctx.save_for_backward(h, tc, dosc, dtc, w1c, b1c, w2c, b2c)
        ctx.meta = (method, int(perturb))
        return h

    @staticmethod
    def backward(ctx, grad_h):
        h, tc, dosc, dtc, w1c, b1c, w2c, b2c = ctx.saved_tensors
        method, perturb = ctx.meta
        lib = L.lib()
        T, B, D = h.shape
        HD = w1c.shape[0]
        gh = grad_h.to(torch.float32).contiguous()
        gy0 = torch.empty((B, D), device=h.device, dtype=torch.float32)
        d = _desc(h[0], tc, dosc, dtc, w1c, b1c, w2c, b2c, h, method, perturb)
        d.grad_h, d.grad_y0 = gh.data_ptr(), gy0.data_ptr()
        onchip = os.environ.get("HODE_NEURAL_LAYOUT", "")[:1] != "t"
        if onchip:
            gw1, gb1, gw2, gb2 = (torch.zeros_like(x) for x in (w1c, b1c, w2c, b2c))
            d.grad_w1, d.grad_b1, d.grad_w2, d.grad_b2 = gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr()
        n = lib.hode_workspace_bytes(d, L.WS_RK_BWD)
        ws = torch.empty(max(n, 4), device=h.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), n
        with torch.cuda.device(h.device):
            L.check(lib.hode_rk_bwd(d, _stream()), "hode_rk_bwd[neural]")
        if onchip:
            return gy0, gw1, gb1, gw2, gb2, None, None, None, None, None
        off = (C.c_size_t * 4)()
        L.check(lib.hode_neural_tape_offsets(d, off), "hode_neural_tape_offsets")
        inst = (T - 1) * _STAGES[method]

        def tape(k, rows):
            return ws[off[k]: off[k] + inst * rows * B * 4].view(torch.float32).view(inst, rows, B)

        if inst == 0:
            z = torch.zeros_like
            return gy0, z(w1c), z(b1c), z(w2c), z(b2c), None, None, None, None, None
        a1t, u1t, yet, u2t = tape(0, HD), tape(1, HD), tape(2, D + 1), tape(3, D)
        # sums over the instance axis as (1 x inst) GEMMs, over patients as GEMVs with a ones vector: torch's strided
        # reductions over these shapes run at a fraction of the HBM rate
        ones_i = torch.ones((1, inst), device=h.device, dtype=torch.float32)
        ones_b = torch.ones((B, 1), device=h.device, dtype=torch.float32)

        def fold(part):  # (inst, m, n) -> (m, n)
            return (ones_i @ part.reshape(inst, -1)).view(part.shape[1], part.shape[2])

        gw1 = fold(torch.bmm(u1t, yet.transpose(1, 2)))
        gw2 = fold(torch.bmm(u2t, a1t.transpose(1, 2)))
        gb1 = fold(u1t @ ones_b).reshape(-1)
        gb2 = fold(u2t @ ones_b).reshape(-1)
        return gy0, gw1, gb1, gw2, gb2, None, None, None, None, None


def neural_solve(y0, w1, b1, w2, b2, t, dosage, dose_times, method="rk4", perturb=False):
    """h (T, B, D) for dy/dt = tanh(W2 tanh(W1 [y, Dose(t)] + b1) + b2); fixed-grid methods only."""
    if method not in L.METHODS:
        raise L.HodeConfigError("hode: the neural rhs is built for the fixed-grid methods (euler, midpoint, rk4); got %r" % (method,))
    if dose_times.dim() != 2:
        dose_times = dose_times.reshape(y0.shape[0], -1)
    return _NeuralFixedGrid.apply(y0, w1, b1, w2, b2, t, dosage, dose_times.to(torch.float32), L.METHODS[method], bool(perturb))
