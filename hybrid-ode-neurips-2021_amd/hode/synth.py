"""Seeded synthetic inputs of the benchmark shapes (generator conventions of reference
``dataloader.py:202-222,261-266``): no files, no reference code -- just tensors of the right distribution.

* grid: ``step_size = 0.125`` (dyadic, so dose-time comparisons are exact), ``t = arange(T) * step``
* ``z0 ~ Exponential(rate 100)`` (scale 0.01), shape (N, D)
* one dose per patient at a grid index ~ U{0..T-2}, amount ~ U(0, 10)
* ``x ~ N(0,1)`` (the generator z-scores measurements), masks ~ Bernoulli(0.5)
"""

from __future__ import annotations

import torch

STEP = 0.125
SEED = 666  # the reference's seed (run_simulation.py:162)


def grid(T, device="cpu", step=STEP):
    return torch.arange(T, dtype=torch.float32, device=device) * step


def one_dose_actions(T, N, gen, dose_max=10.0, n_dose=1):
    """(T, N, 1) action tensor with exactly ``n_dose`` non-zero entries per patient at distinct grid indices < T-1."""
    a = torch.zeros(T, N, 1)
    scores = torch.rand(N, max(T - 1, 1), generator=gen)
    idx = scores.topk(min(n_dose, max(T - 1, 1)), dim=1).indices  # distinct indices per patient
    amt = torch.rand(N, idx.shape[1], generator=gen) * dose_max + 1e-3
    a[idx, torch.arange(N)[:, None].expand_as(idx), 0] = amt
    return a


def solver_inputs(N, T, D, seed=SEED, n_dose=1):
    gen = torch.Generator().manual_seed(seed)
    z0 = torch.empty(N, D).exponential_(100.0, generator=gen)
    a = one_dose_actions(T, N, gen, n_dose=n_dose)
    return {"z0": z0, "actions": a, "t": grid(T)}


def observation_inputs(N, T, obs, seed=SEED):
    gen = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(T, N, obs, generator=gen)
    mask = (torch.rand(T, N, obs, generator=gen) < 0.5).float()
    return {"measurements": x, "masks": mask}


def default_ml_weights(D, seed=SEED):
    """``nn.Linear(D, D-4)`` default init under the seed (what the reference's fresh ``ml_net`` holds)."""
    torch.manual_seed(seed)
    lin = torch.nn.Linear(D, D - 4)
    return lin.weight.detach().clone(), lin.bias.detach().clone()
