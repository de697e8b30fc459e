"""Loss assembly around the solver, restated on PyTorch-CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Follows reference
``RocheExpertDecoder.forward`` (``model.py:1112-1121``) and ``VariationalInference.loss`` /
``mc_kl`` (``model.py:1150-1214``), ``GaussianReparam`` (``:18-31``), ``ExponentialPrior`` (``:41-45``).
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn

from .rhs import NeuralRHS, RocheRHS
from .solvers import odeint


class DecoderOracle(nn.Module):
    """``set_action`` -> ``odeint`` on the observation grid -> linear readout; returns (x_hat, h)."""

    def __init__(self, obs_dim, latent_dim, t_max, step_size, roche=True, ablate=False, method="dopri5"):
        super().__init__()
        self.method = method
        self.t = torch.arange(0, t_max + step_size, step_size, dtype=torch.float32)  # model.py:1072
        self.output_function = nn.Sequential(nn.Linear(latent_dim, obs_dim, bias=True))  # created first, model.py:1097
        self.ode = RocheRHS(latent_dim, step_size, ablate=ablate) if roche else NeuralRHS(latent_dim, step_size)
        #: tests may swap the solver call, e.g. for ``solvers.odeint_dopri5_replay`` along another run's step tape
        self.solve = None

    def forward(self, init, a):
        self.ode.set_action(a)
        if self.solve is not None:
            h = self.solve(self.ode, init, self.t)
        else:
            h = odeint(self.ode, init, self.t, rtol=1e-7, atol=1e-8, method=self.method)
        return self.output_function(h), h


def reparameterize(mu, log_var):
    std = torch.exp(0.5 * log_var)
    return torch.randn_like(std) * std + mu


def gaussian_log_density(mu, log_var, z):
    std = torch.exp(0.5 * log_var)
    return torch.sum(-((z - mu) ** 2) / (2 * std ** 2) - torch.log(std) - math.log(math.sqrt(2 * math.pi)), dim=-1)


def exponential_log_density(z, rate=100.0):
    return torch.sum(math.log(rate) - rate * z, dim=-1)


def masked_sse(x, x_hat, mask):
    return torch.sum((x - x_hat) ** 2 * mask) / x.shape[1]  # model.py:1179


def kl_standard_normal(mu, log_var):
    return torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)  # model.py:1188


def mc_kl(mu, log_var, sample_size, eps=torch.finfo(torch.float32).eps):
    """Monte-Carlo KL(q || Exp(100)) with non-positive samples clamped to machine eps (model.py:1198-1214)."""
    acc = []
    for _ in range(sample_size):
        z = reparameterize(mu, log_var)
        z[z <= 0.0] = eps
        acc.append(gaussian_log_density(mu, log_var, z) - exponential_log_density(z))
    return torch.mean(torch.stack(acc, dim=-1), dim=-1)


def vi_loss(encoder, decoder, data, elbo=True, exponential_prior=True, mc_size=100):
    x, a, mask = data["measurements"], data["actions"], data["masks"]
    mu, log_var = encoder(x, a, mask)
    z = reparameterize(mu, log_var) if elbo else mu
    x_hat, _ = decoder(z, a)
    lik = masked_sse(x, x_hat, mask)
    if not elbo:
        return lik
    kld = torch.mean(mc_kl(mu, log_var, mc_size), dim=0) if exponential_prior else kl_standard_normal(mu, log_var)
    return lik + kld
