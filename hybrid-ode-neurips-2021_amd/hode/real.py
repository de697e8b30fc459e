"""Fixed-grid solve of the real-data hybrid ODE (reference ``RocheODEReal``, ``model.py:570-657``) on the gfx950 kernels.

The backward kernel returns ``grad_y0`` and the gradients of the three scalars, and tapes the operands of every linear
layer's weight gradient; they are contracted here with batched BLAS GEMMs and returned as one flat gradient matching
the flat weight buffer (parameter creation order of the reference)."""

from __future__ import annotations

import os

import torch

from . import _lib as L
from .solver import _f32c, _require_gpu, _stream

_STAGES = {L.METHOD_EULER: 1, L.METHOD_MIDPOINT: 2, L.METHOD_RK4_38: 4}


def _desc(y0, t, act, theta, wflat, h, method, perturb, hidden):
    B, D = y0.shape
    d = L.new_solve_desc()
    d.rhs_kind, d.method, d.perturb = L.RHS_ROCHE_REAL, method, int(perturb)
    d.batch, d.latent_dim, d.n_times, d.hidden_dim, d.n_action_times = B, D, t.numel(), hidden, act.shape[0]
    d.t, d.y0, d.dosage, d.theta, d.w1, d.h = t.data_ptr(), y0.data_ptr(), act.data_ptr(), theta.data_ptr(), wflat.data_ptr(), h.data_ptr()
    return d


class _RealFixedGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, theta, wflat, t, act, method, perturb, hidden):
        _require_gpu(y0, theta, wflat, t, act)
        lib = L.lib()
        y0c, thc, wc, tc, ac = _f32c(y0), _f32c(theta), _f32c(wflat), _f32c(t), _f32c(act)
        B, D = y0c.shape
        M = D - 4
        assert wc.numel() == 9 * hidden + 2 + 3 * M * M, "flat weight buffer has the wrong length"
        h = torch.empty((tc.numel(), B, D), device=y0.device, dtype=torch.float32)
        d = _desc(y0c, tc, ac, thc, wc, h, method, perturb, hidden)
        n = lib.hode_workspace_bytes(d, L.WS_RK_FWD)  # the per-patient dose table
        ws = torch.empty(max(n, 4), device=y0.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), n
        with torch.cuda.device(y0.device):
            L.check(lib.hode_rk_fwd(d, _stream()), "hode_rk_fwd[real]")
        ctx.save_for_backward(h, thc, wc, tc, ac)
        ctx.meta = (method, int(perturb), int(hidden))
        return h

    @staticmethod
    def backward(ctx, grad_h):
        h, thc, wc, tc, ac = ctx.saved_tensors
        method, perturb, H = ctx.meta
        lib = L.lib()
        T, B, D = h.shape
        M = D - 4
        gh = grad_h.to(torch.float32).contiguous()
        gy0 = torch.empty((B, D), device=h.device, dtype=torch.float32)
        gth = torch.zeros(L.N_THETA, device=h.device, dtype=torch.float32)
        d = _desc(h[0], tc, ac, thc, wc, h, method, perturb, H)
        d.grad_h, d.grad_y0, d.grad_theta = gh.data_ptr(), gy0.data_ptr(), gth.data_ptr()
        # matrix-core kernels (D = 20, hidden <= 64): the weight gradients are accumulated on chip into this flat buffer;
        # the lane-per-patient kernels (HODE_REAL_LAYOUT=t, D = 4) tape the GEMM operands for the contractions below
        onchip = D == 20 and H <= 64 and os.environ.get("HODE_REAL_LAYOUT", "")[:1] != "t"
        if onchip:
            gw = torch.zeros_like(wc)
            d.grad_w1 = gw.data_ptr()
        n = lib.hode_workspace_bytes(d, L.WS_RK_BWD)
        ws = torch.empty(max(n, 4), device=h.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), n
        with torch.cuda.device(h.device):
            L.check(lib.hode_rk_bwd(d, _stream()), "hode_rk_bwd[real]")
        if onchip:
            return gy0, gth[:3].clone(), gw, None, None, None, None, None
        inst = (T - 1) * _STAGES[method]
        rows = 5 + 4 * H + 5 * M
        if inst == 0:
            return gy0, gth[:3].clone(), torch.zeros_like(wc), None, None, None, None, None
        tape = ws[: inst * rows * B * 4].view(torch.float32).view(inst, rows, B)
        o = 0

        def take(k):
            nonlocal o
            v = tape[:, o:o + k]
            o += k
            return v

        y3, a11, u11, u12, a21, u21, u22 = take(3), take(H), take(H), take(1), take(H), take(H), take(1)
        hh, rh, ur, uz, uh = (take(M) for _ in range(5)) if M > 0 else (None,) * 5

        # sums over instances are (1 x inst) GEMMs: torch's strided sum over these shapes runs at a fraction of HBM speed
        ones_i = torch.ones((1, inst), device=h.device, dtype=torch.float32)
        ones_b = torch.ones((B, 1), device=h.device, dtype=torch.float32)

        def fold(part):  # (inst, m, n) -> (m, n)
            return (ones_i @ part.reshape(inst, -1)).view(part.shape[1], part.shape[2])

        def outer(u, x):  # sum over instances and patients of u x^T
            return fold(torch.bmm(u, x.transpose(1, 2)))

        def colsum(u):   # (inst, k, B) -> (k,)
            return fold(u @ ones_b).reshape(-1)

        parts = [outer(u11, y3).reshape(-1), colsum(u11), outer(u12, a11).reshape(-1), colsum(u12),
                 outer(u21, y3[:, :2]).reshape(-1), colsum(u21), outer(u22, a21).reshape(-1), colsum(u22)]
        if M > 0:
            parts += [outer(uh, rh).reshape(-1), outer(uz, hh).reshape(-1), outer(ur, hh).reshape(-1)]
        return gy0, gth[:3].clone(), torch.cat(parts), None, None, None, None, None


def real_solve(y0, theta, wflat, t, act, hidden, method="midpoint", perturb=True):
    """h (T, B, D).  ``theta`` = (k_immunity, kel, kel2); ``wflat`` = all weights in creation order; ``act`` (Ta, B) doses."""
    if method not in L.METHODS:
        raise L.HodeConfigError("hode: the real-data rhs is built for the fixed-grid methods (euler, midpoint, rk4); got %r" % (method,))
    return _RealFixedGrid.apply(y0, theta, wflat, t, act, L.METHODS[method], bool(perturb), int(hidden))
