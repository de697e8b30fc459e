"""Right-hand sides of the latent ODE, restated op-for-op on PyTorch-CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Each class cites the reference
lines it follows; evaluation order of every product / power is kept identical so
that fp32 results agree with the imported reference bit-for-bit on CPU
(``tests/test_oracle_golden.py``).

State layout (reference ``model.py:519-522``): ``y[:, 0]`` Disease, ``y[:, 1]``
ImmuneReact, ``y[:, 2]`` Immunity, ``y[:, 3]`` Dose2, ``y[:, 4:]`` learned latents.
"""

from __future__ import annotations

import torch
import torch.nn as nn

#: order of the 13 expert rate constants == parameter creation order in the
#: reference (``model.py:468-482``) == layout of ``theta[13]`` in ``include/hode.h``.
THETA_NAMES = (
    "HillCure",
    "HillPatho",
    "ec50_patho",
    "emax_patho",
    "k_dexa",
    "k_discure_immunereact",
    "k_discure_immunity",
    "k_disprog",
    "k_immune_disease",
    "k_immune_feedback",
    "k_immune_off",
    "k_immunity",
    "kel",
)

#: ``sim_config.RochConfig`` defaults (reference ``sim_config.py:4-18``): Hill exponents 2, rest 1.
THETA_DEFAULT = (2.0, 2.0) + (1.0,) * 11


def dose_schedule(action: torch.Tensor, step_size):
    """Vectorised restatement of ``RocheODE.set_action`` (reference ``model.py:495-507``).

    ``action`` is (T, B, 1).  Returns ``dosage`` (B,) = max over time of the dose channel
    and ``times`` (B, K) = grid indices of the non-zero entries times ``step_size``.  Like
    the reference (``torch.stack`` at ``model.py:507``) every patient must have the same
    number K of non-zero doses.  ``times`` is int64 when ``step_size`` is an ``int`` and
    float32 when it is a ``float`` -- the same dtype rule as ``LongTensor * python_scalar``.
    """
    chan = action[..., 0]
    dosage = torch.max(chan, dim=0)[0]
    hit = (chan != 0).t()  # (B, T)
    counts = hit.sum(dim=1)
    if counts.numel() and not bool((counts == counts[0]).all()):
        raise RuntimeError("stack expects each tensor to be equal size (unequal dose counts per patient)")
    k = int(counts[0]) if counts.numel() else 0
    idx = torch.nonzero(hit)[:, 1].reshape(hit.shape[0], k)
    return dosage, idx * step_size


class RocheRHS(nn.Module):
    """Expert PK/PD block + ``tanh(W y + b)`` learned block (reference ``RocheODE``, ``model.py:446-555``)."""

    def __init__(self, latent_dim: int, step_size, ablate: bool = False, theta=THETA_DEFAULT):
        super().__init__()
        self.latent_dim = int(latent_dim)
        self.ml_dim = self.latent_dim - 4
        self.step_size = step_size
        self.ablate = bool(ablate)
        for name, val in zip(THETA_NAMES, theta):
            setattr(self, name, nn.Parameter(torch.tensor(float(val), dtype=torch.float32)))
        if self.ablate:  # reference model.py:483-485
            self.theta_1 = nn.Parameter(torch.tensor(1.0))
            self.theta_2 = nn.Parameter(torch.tensor(2.0))
        # reference model.py:487-490
        self.ml_net = nn.Sequential(nn.Linear(self.latent_dim, self.ml_dim), nn.Tanh()) if self.ml_dim > 0 else nn.Identity()
        self.dosage = None
        self.times = None

    def set_action(self, action):
        self.dosage, self.times = dose_schedule(action, self.step_size)

    def dose_at_time(self, t):
        """``Dose(t) = dosage * sum_k 1[t>=tau_k] exp(kel (tau_k - t))`` (reference ``model.py:509-513``)."""
        on = t >= self.times
        return self.dosage * torch.sum(torch.exp(self.kel * (self.times - t) * on) * on, dim=-1)

    def forward(self, t, y):
        dis, ir, imm, dose2 = y[:, 0], y[:, 1], y[:, 2], y[:, 3]
        if not self.ablate:
            dose = self.dose_at_time(t)
            # reference model.py:527-531
            d1 = dis * self.k_disprog - dis * imm ** self.HillCure * self.k_discure_immunity - dis * ir * self.k_discure_immunereact
            # reference model.py:533-540
            ir_p = ir ** self.HillPatho
            d2 = (
                dis * self.k_immune_disease
                - ir * self.k_immune_off
                + dis * ir * self.k_immune_feedback
                + (ir_p * self.emax_patho) / (self.ec50_patho ** self.HillPatho + ir ** self.HillPatho)
                - dose2 * ir * self.k_dexa
            )
            d3 = ir * self.k_immunity  # model.py:542
            d4 = self.kel * dose - self.kel * dose2  # model.py:544
        else:  # linear oscillators, reference model.py:545-549
            d1 = ir
            d2 = -1.0 * dis * self.theta_1
            d3 = dose2
            d4 = -1.0 * imm * self.theta_2
        if self.ml_dim > 0:
            return torch.cat([d1[..., None], d2[..., None], d3[..., None], d4[..., None], self.ml_net(y)], dim=-1)
        return torch.stack([d1, d2, d3, d4], dim=-1)


class NeuralRHS(nn.Module):
    """Pure neural rhs ``tanh(W2 tanh(W1 [y, Dose] + b1) + b2)`` (reference ``NeuralODE``, ``model.py:969-1026``)."""

    def __init__(self, latent_dim: int, step_size):
        super().__init__()
        self.latent_dim = int(latent_dim)
        self.step_size = step_size
        self.kel = nn.Parameter(torch.tensor(1.0))  # created but unused by forward (model.py:989)
        d = self.latent_dim
        self.ml_net = nn.Sequential(nn.Linear(d + 1, d * 10), nn.Tanh(), nn.Linear(d * 10, d), nn.Tanh())
        self.dosage = None
        self.times = None

    def set_action(self, action):
        self.dosage, self.times = dose_schedule(action, self.step_size)

    def dose_at_time(self, t):
        # impulse only when t hits a dose time exactly (reference model.py:1017)
        return self.dosage * torch.sum(self.times == t, dim=-1)

    def forward(self, t, y):
        dose = self.dose_at_time(t)
        return self.ml_net(torch.cat([y, dose[:, None]], dim=-1))


class RocheRealRHS(nn.Module):
    """DDW variant: two tiny MLPs + GRU-ODE block (reference ``RocheODEReal``, ``model.py:570-657``)."""

    def __init__(self, latent_dim: int, hidden_dim: int):
        super().__init__()
        self.latent_dim = int(latent_dim)
        self.hidden_dim = int(hidden_dim)
        self.ml_dim = self.latent_dim - 4
        h = self.hidden_dim
        self.dx1_net = nn.Sequential(nn.Linear(3, h), nn.Tanh(), nn.Linear(h, 1), nn.Tanh())
        self.dx2_net = nn.Sequential(nn.Linear(2, h), nn.Tanh(), nn.Linear(h, 1), nn.Tanh())
        if self.ml_dim > 0:  # creation order hh, hz, hr as in model.py:599-607
            self.lin_hh = nn.Linear(self.ml_dim, self.ml_dim, bias=False)
            self.lin_hz = nn.Linear(self.ml_dim, self.ml_dim, bias=False)
            self.lin_hr = nn.Linear(self.ml_dim, self.ml_dim, bias=False)
        self.k_immunity = nn.Parameter(torch.tensor(1.0))
        self.kel = nn.Parameter(torch.tensor(0.2))
        self.kel2 = nn.Parameter(torch.tensor(0.2))
        self.dosage = None
        self.times = None

    def set_action_static(self, action, static=None):
        # every grid point is a potential dose at time index+1 (reference model.py:647-651)
        self.dosage = action
        self.times = torch.cumsum(torch.ones_like(action), dim=0)

    def dose_at_time(self, t):
        on = t >= self.times
        return torch.sum(self.dosage * torch.exp(self.kel * (self.times - t) * on) * on, dim=(0, 2))

    def forward(self, t, y):
        ir, dose2 = y[:, 1], y[:, 3]
        dose = self.dose_at_time(t)
        d1 = self.dx1_net(y[:, :3])
        d2 = self.dx2_net(y[:, :2])
        d3 = (ir * self.k_immunity)[..., None]
        d4 = (self.kel * dose - self.kel2 * dose2)[..., None]
        if self.ml_dim == 0:
            return torch.cat([d1, d2, d3, d4], dim=-1)
        h = y[..., 4:]
        r = torch.sigmoid(0 + self.lin_hr(h))
        z = torch.sigmoid(0 + self.lin_hz(h))
        u = torch.tanh(0 + self.lin_hh(r * h))
        return torch.cat([d1, d2, d3, d4, (1 - z) * (u - h)], dim=-1)
