"""Fused linear readout + masked SSE likelihood (``hode_readout_sse``): ``sum((x - Linear(h))^2 * mask) / B`` without
materialising ``x_hat`` (reference ``model.py:1120`` + ``:1179``); one HBM pass computes the loss and all gradients."""

from __future__ import annotations

import torch

from . import _lib as L
from .solver import _f32c, _require_gpu, _stream

SUPPORTED_LATENT = (4, 6, 8, 12)


def supported(latent_dim: int, obs_dim: int) -> bool:
    return latent_dim in SUPPORTED_LATENT and obs_dim % 4 == 0 and 0 < obs_dim <= 128


class _ReadoutSSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, x, mask, w, b):
        _require_gpu(h, x, mask, w, b)
        lib = L.lib()
        T, B, D = h.shape
        obs = x.shape[-1]
        hc, xc, mc, wc, bc = _f32c(h), _f32c(x), _f32c(mask), _f32c(w), _f32c(b)
        need_grad = any(ctx.needs_input_grad)
        lik = torch.empty(1, device=h.device, dtype=torch.float32)
        d = L.ReadoutDesc()
        d.struct_size = L.C.sizeof(L.ReadoutDesc)
        d.latent_dim, d.obs_dim, d.scale, d.rows = D, obs, 1.0 / B, T * B
        d.h, d.x, d.mask, d.w, d.b, d.lik = hc.data_ptr(), xc.data_ptr(), mc.data_ptr(), wc.data_ptr(), bc.data_ptr(), lik.data_ptr()
        if need_grad:
            gh = torch.empty_like(hc)
            gw = torch.zeros_like(wc)
            gb = torch.zeros_like(bc)
            d.grad_h, d.grad_w, d.grad_b = gh.data_ptr(), gw.data_ptr(), gb.data_ptr()
        n = lib.hode_readout_workspace_bytes(d)
        ws = torch.empty(max(n, 4), device=h.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), n
        with torch.cuda.device(h.device):
            L.check(lib.hode_readout_sse(d, _stream()), "hode_readout_sse")
        if need_grad:
            ctx.save_for_backward(gh, gw, gb)
        return lik[0] / B

    @staticmethod
    def backward(ctx, g):
        gh, gw, gb = ctx.saved_tensors
        return gh * g, None, None, gw * g, gb * g


def masked_sse_readout(h, x, mask, weight, bias):
    """``sum((x - (h @ weight.T + bias))^2 * mask) / B`` for h (T,B,D), x/mask (T,B,obs); differentiable in h, weight, bias."""
    return _ReadoutSSE.apply(h, x, mask, weight, bias)


# ----------------------------------------------------------------------------------------------------------------------
# two-layer readout of the real-data decoder (hode_readout_mlp_sse)

def mlp_supported(latent_dim: int, hidden_dim: int, obs_dim: int) -> bool:
    return latent_dim in (20, 4) and hidden_dim == latent_dim + 1 and obs_dim == 24


class _ReadoutMlpSSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, x, mask, w1, b1, w2, b2, time_weight, skip):
        _require_gpu(h, x, mask, w1, w2)
        lib = L.lib()
        T, B, D = h.shape
        T -= skip   # the first `skip` rows of h take no part (DecoderReal drops the state at t0 - 1): the kernel starts behind
        #             them and their gradient rows are zeroed here -- slicing h outside costs a zero fill + a copy of grad_h
        hc, xc, mc = _f32c(h), _f32c(x), _f32c(mask)
        w1c, b1c, w2c, b2c = _f32c(w1), _f32c(b1), _f32c(w2), _f32c(b2)
        twc = None if time_weight is None else _f32c(time_weight)
        need_grad = any(ctx.needs_input_grad)
        lik = torch.empty(1, device=h.device, dtype=torch.float32)
        d = L.ReadoutMlpDesc()
        d.struct_size = L.C.sizeof(L.ReadoutMlpDesc)
        d.latent_dim, d.hidden_dim, d.obs_dim, d.batch, d.scale, d.rows = D, w1c.shape[0], x.shape[-1], B, 1.0 / B, T * B
        off = skip * B * D * 4
        d.h, d.x, d.mask = hc.data_ptr() + off, xc.data_ptr(), mc.data_ptr()
        d.time_weight = 0 if twc is None else twc.data_ptr()
        d.w1, d.b1, d.w2, d.b2, d.lik = w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(), lik.data_ptr()
        if need_grad:
            gh = torch.empty_like(hc)
            if skip:
                gh[:skip].zero_()
            # the four parameter-gradient accumulators as views of ONE zeroed buffer: one fill here, one scaling in backward
            sizes = [t.numel() for t in (w1c, b1c, w2c, b2c)]
            gflat = torch.zeros(sum(sizes), device=h.device, dtype=torch.float32)
            gw1, gb1, gw2, gb2 = (v.view(t.shape) for v, t in zip(torch.split(gflat, sizes), (w1c, b1c, w2c, b2c)))
            d.grad_h, d.grad_w1, d.grad_b1, d.grad_w2, d.grad_b2 = (gh.data_ptr() + off, gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(),
                                                                    gb2.data_ptr())
        n = lib.hode_readout_mlp_workspace_bytes(d)
        ws = torch.empty(max(n, 4), device=h.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), n
        with torch.cuda.device(h.device):
            L.check(lib.hode_readout_mlp_sse(d, _stream()), "hode_readout_mlp_sse")
        if need_grad:
            ctx.save_for_backward(gh, gflat)
            ctx.shapes = (sizes, [t.shape for t in (w1c, b1c, w2c, b2c)])
        return lik[0] / B

    @staticmethod
    def backward(ctx, g):
        gh, gflat = ctx.saved_tensors
        sizes, shapes = ctx.shapes
        gw1, gb1, gw2, gb2 = (v.view(sh) for v, sh in zip(torch.split(gflat * g, sizes), shapes))
        return gh * g, None, None, gw1, gb1, gw2, gb2, None, None


def masked_sse_readout_mlp(h, x, mask, w1, b1, w2, b2, time_weight=None, skip_rows=0):
    """``sum((x - (W2 ELU(W1 h + b1) + b2))^2 * mask * time_weight[t]) / B`` for h (skip_rows + T,B,D), x / mask (T,B,obs);
    differentiable in h and the four parameters (reference model.py:809-813, :859, :1243-1247).  The first ``skip_rows``
    time rows of h are left out (their gradient is zero) -- the same as passing ``h[skip_rows:]``, without the copy."""
    skip = int(skip_rows)
    if skip and (skip * h.shape[1] * h.shape[2]) % 4:
        return _ReadoutMlpSSE.apply(h[skip:], x, mask, w1, b1, w2, b2, time_weight, 0)   # the kernel wants 16-byte aligned rows
    return _ReadoutMlpSSE.apply(h, x, mask, w1, b1, w2, b2, time_weight, skip)
