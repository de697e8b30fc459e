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
