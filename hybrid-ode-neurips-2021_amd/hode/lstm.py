"""Masked LSTM encoder on the gfx950 matrix cores (``hode_lstm_fwd`` / ``hode_lstm_bwd``).

Replaces the T single-step ``nn.LSTM`` calls of reference ``EncoderLSTM.forward`` (``model.py:420-422``) together with
the ``cat([x, a]) * cat([mask, 1])`` that feeds them.
"""

from __future__ import annotations

import torch

from . import _lib as L
from .solver import _f32c, _ptr, _require_gpu, _stream


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
