"""Masked LSTM encoder on the gfx950 matrix cores (``hode_lstm_fwd`` / ``hode_lstm_bwd``).

Replaces the T single-step ``nn.LSTM`` calls of reference ``EncoderLSTM.forward`` (``model.py:420-422``) together with
the ``cat([x, a]) * cat([mask, 1])`` that feeds them.
"""

from __future__ import annotations

import os

import torch

from . import _lib as L
from .solver import _f32c, _ptr, _require_gpu, _stream


#: measurement hook (bench.py): set to a list and every encoder call appends (name, start_event, end_event) for its three
#: pieces -- "lstm_fwd" (pack kernels + lstm_fwd_kernel), "lstm_bwd" (BPTT kernel), "wgrad_gemm" (operand fill + split-K GEMM
#: + fold).  None (the default) records nothing.
timeline = None


class _Span:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if timeline is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if timeline is not None:
            self.e1.record()
            timeline.append((self.name, self.e0, self.e1))
        return False


def _desc(x, a, mask, w_ih, w_hh, b_ih, b_hh, reverse, save_tape):
    T, B, obs = x.shape
    ad = 0 if a is None else a.shape[-1]
    d = L.new_lstm_desc()
    d.seq_len, d.batch, d.input_dim, d.hidden_dim, d.obs_dim = T, B, obs + ad, w_hh.shape[1], obs
    d.reverse, d.save_tape = int(reverse), int(save_tape)
    d.x, d.a, d.mask = x.data_ptr(), _ptr(a), _ptr(mask)
    d.w_ih, d.w_hh, d.b_ih, d.b_hh = w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr()
    return d


def lstm_final_state(x, a, mask, w_ih, w_hh, b_ih, b_hh, reverse=True):
    """Forward only: final (h, c), each (B, H), after walking the window (reverse: t = T-1 .. 0).

    x (T, B, obs), a (T, B, A) or None, mask (T, B, obs) or None; weights as ``nn.LSTM`` stores them.
    """
    _require_gpu(x, w_ih)
    lib = L.lib()
    xc, ac, mc = _f32c(x), (None if a is None else _f32c(a)), (None if mask is None else _f32c(mask))
    wi, wh, bi, bh = _f32c(w_ih), _f32c(w_hh), _f32c(b_ih), _f32c(b_hh)
    B, H = xc.shape[1], wh.shape[1]
    h = torch.empty((B, H), device=x.device, dtype=torch.float32)
    c = torch.empty((B, H), device=x.device, dtype=torch.float32)
    d = _desc(xc, ac, mc, wi, wh, bi, bh, reverse, False)
    d.h_out, d.c_out = h.data_ptr(), c.data_ptr()
    nbytes = lib.hode_lstm_workspace_bytes(d)
    if nbytes == 0:
        L.check(lib.hode_lstm_fwd(d, _stream()), "hode_lstm_fwd")  # reports why the shape is unsupported
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    with torch.cuda.device(x.device):
        L.check(lib.hode_lstm_fwd(d, _stream()), "hode_lstm_fwd")
    return h, c


_ONES = {}


def _ones_row(n, like):
    """(1 x n) all-ones GEMM operand, kept per (n, device, dtype): a fill kernel per use is pure launch latency."""
    key = (n, like.device, like.dtype)
    if key not in _ONES:
        _ONES[key] = torch.ones((1, n), device=like.device, dtype=like.dtype)
    return _ONES[key]


def _splitk_tn(lhs, rhs):
    """lhs^T @ rhs for tall operands (K x M, K x N with K ~ 1e6, M, N ~ 1e2): the BLAS heuristics pick a kernel without
    split-K for this shape (2.4-3.5 ms measured); batching K into P slices and summing runs 2x faster (tools/gemm_probe.py)."""
    K = lhs.shape[0]
    P = 1
    for cand in range(256, 1, -1):
        if K % cand == 0 and K // cand >= 1024:
            P = cand
            break
    if P == 1:
        return lhs.t() @ rhs
    # The product is formed as (rhs^T lhs)^T: with M = 4H = 640 and N = 244 the library's 256 x 128 macro tile wastes 20 % on
    # the 640 x 244 result (3 x 2 tiles) and 5 % on 244 x 640 (1 x 5): 2.7 -> 2.3 ms at K = 1e6 in isolation
    # (tools/gemm_orient_probe.py), 0.05 ms inside the training step (same-call A/B, HODE_LSTM_GEMM_MN=1)
    if os.environ.get("HODE_LSTM_GEMM_MN"):   # A/B switch: the product in its natural orientation
        part = torch.bmm(lhs.view(P, K // P, -1).transpose(1, 2), rhs.view(P, K // P, -1))
        return (_ones_row(P, part) @ part.view(P, -1)).view(part.shape[1], part.shape[2])
    part = torch.bmm(rhs.view(P, K // P, -1).transpose(1, 2), lhs.view(P, K // P, -1))
    # fold the P partial products with a (1 x P) GEMM: torch's strided sum(0) over this shape reads at ~0.3 TB/s
    return (_ones_row(P, part) @ part.view(P, -1)).view(part.shape[1], part.shape[2]).t()


_SIDE = {}


def _side_stream(device):
    """One side stream per device for work that is independent of the kernel on the current stream."""
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class _LstmEncode(torch.autograd.Function):
    """h_final = LSTM(window); backward = BPTT kernel + one split-K BLAS GEMM over K = T*B for all weight and bias gradients."""

    @staticmethod
    def forward(ctx, x, a, mask, w_ih, w_hh, b_ih, b_hh, reverse):
        _require_gpu(x, w_ih)
        lib = L.lib()
        xc, ac, mc = _f32c(x), (None if a is None else _f32c(a)), (None if mask is None else _f32c(mask))
        wi, wh, bi, bh = _f32c(w_ih), _f32c(w_hh), _f32c(b_ih), _f32c(b_hh)
        B, H = xc.shape[1], wh.shape[1]
        h = torch.empty((B, H), device=x.device, dtype=torch.float32)
        c = torch.empty((B, H), device=x.device, dtype=torch.float32)
        d = _desc(xc, ac, mc, wi, wh, bi, bh, reverse, True)
        d.h_out, d.c_out = h.data_ptr(), c.data_ptr()
        nbytes = lib.hode_lstm_workspace_bytes(d)
        if nbytes == 0:
            L.check(lib.hode_lstm_fwd(d, _stream()), "hode_lstm_fwd")
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        with torch.cuda.device(x.device), _Span("lstm_fwd"):
            L.check(lib.hode_lstm_fwd(d, _stream()), "hode_lstm_fwd")
        ctx.save_for_backward(xc, ac if ac is not None else xc, mc if mc is not None else xc, wi, wh, bi, bh, ws)
        ctx.flags = (ac is not None, mc is not None, bool(reverse))
        return h

    @staticmethod
    def backward(ctx, grad_h):
        xc, ac, mc, wi, wh, bi, bh, ws = ctx.saved_tensors
        has_a, has_m, reverse = ctx.flags
        ac = ac if has_a else None
        mc = mc if has_m else None
        lib = L.lib()
        T, B, obs = xc.shape
        H = wh.shape[1]
        gh = grad_h.to(torch.float32).contiguous()
        dgates = torch.empty((T, B, 4 * H), device=xc.device, dtype=torch.float32)
        ad = 0 if ac is None else ac.shape[-1]
        I = obs + ad
        W = (I + H + 1 + 3) // 4 * 4
        hprev = torch.empty((T, B, W), device=xc.device, dtype=torch.float32)  # [x*mask | a | h_prev | 1 | 0-pad]
        d = _desc(xc, ac, mc, wi, wh, bi, bh, reverse, True)
        h_dummy = torch.empty(1, device=xc.device)
        d.h_out, d.c_out = h_dummy.data_ptr(), h_dummy.data_ptr()  # unused by the backward, must be non-NULL
        d.grad_h_out, d.grad_gates, d.h_prev = gh.data_ptr(), dgates.data_ptr(), hprev.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        # The first obs columns of the operand rows (x * mask) do not depend on the recurrence: they are filled on a side stream
        # WHILE the BPTT kernel runs (it occupies 209 of the 256 CUs at the bench shape and is bound by the matrix pipe, the fill
        # by HBM: off the critical path; HODE_LSTM_SERIAL_FILL=1 keeps it behind the kernel for A/Bs).
        side = None if os.environ.get("HODE_LSTM_SERIAL_FILL") else _side_stream(xc.device)

        def fill():   # the kernel leaves the first obs columns to the caller: x * mask, one streaming pass on the current stream
            with torch.cuda.device(xc.device):
                L.check(lib.hode_lstm_fill_operand(d, _stream()), "hode_lstm_fill_operand")

        if side is not None:
            side.wait_stream(torch.cuda.current_stream())     # hprev's allocation and whatever produced x / mask
            with torch.cuda.stream(side):
                fill()
        with torch.cuda.device(xc.device), _Span("lstm_bwd"):
            L.check(lib.hode_lstm_bwd(d, _stream()), "hode_lstm_bwd")
        dg2 = dgates.view(T * B, 4 * H)
        with _Span("wgrad_gemm"):
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            else:
                fill()
            g = _splitk_tn(dg2, hprev.view(T * B, W))  # ONE product: [grad_w_ih | grad_w_hh | grad_b | 0]
        g_wih = g[:, :I].contiguous()
        g_whh = g[:, I:I + H].contiguous()
        g_b = g[:, I + H].contiguous()
        return None, None, None, g_wih, g_whh, g_b, g_b.clone(), None


def lstm_encode(x, a, mask, w_ih, w_hh, b_ih, b_hh, reverse=True):
    """Differentiable (w.r.t. the four LSTM parameters) final hidden state (B, H) of the masked window LSTM."""
    return _LstmEncode.apply(x, a, mask, w_ih, w_hh, b_ih, b_hh, bool(reverse))
