"""Masked reverse-time LSTM encoder restated on PyTorch-CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Follows reference ``EncoderLSTM``
(``model.py:383-440``): the observation window is walked from the last step to the first,
each step feeds ``cat(x, a)[t] * cat(mask, 1)[t]`` to a single-layer LSTM, and two linear
heads map the final hidden state to ``mu`` and ``log_var``; ``normalize`` applies
``mu = exp(mu)/10`` and ``log_var -= 5``.

The cell is written out gate by gate (PyTorch order i, f, g, o; two bias vectors) rather
than calling ``nn.LSTM`` so that the HIP kernel is checked against explicit arithmetic;
``tests/test_oracle_golden.py`` checks this against the reference's ``nn.LSTM`` outputs.
"""

from __future__ import annotations

import torch
import torch.nn as nn


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One LSTM step: gates = x W_ih^T + b_ih + h W_hh^T + b_hh, chunks (i, f, g, o)."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=-1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    return h_new, c_new


class EncoderLSTMOracle(nn.Module):
    """Parameter names match the reference ``state_dict`` (``lstm.*``, ``lin.*``, ``log_var.*``)."""

    def __init__(self, input_dim, hidden_dim, output_dim, normalize=True):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.normalize = normalize
        self.lstm = nn.LSTM(input_dim, hidden_dim)  # used as a parameter container (same init order as reference)
        self.lin = nn.Linear(hidden_dim, output_dim)
        self.log_var = nn.Linear(hidden_dim, output_dim)

    def final_hidden(self, x, a, mask):
        y_in = torch.cat([x, a], dim=-1)
        m_in = torch.cat([mask, torch.ones_like(a)], dim=-1)
        b = y_in.shape[1]
        h = y_in.new_zeros(b, self.hidden_dim)
        c = y_in.new_zeros(b, self.hidden_dim)
        p = self.lstm
        for t in reversed(range(y_in.shape[0])):
            h, c = lstm_cell(y_in[t] * m_in[t], h, c, p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0)
        return h, c

    def forward(self, x, a, mask):
        h, _ = self.final_hidden(x, a, mask)
        mu = self.lin(h)
        log_var = self.log_var(h)
        if self.normalize:
            mu = torch.exp(mu) / 10
            log_var = log_var - 5.0
        return mu, log_var
