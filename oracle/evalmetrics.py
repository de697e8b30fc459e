"""CPU oracle for the evaluation metrics of the reference's ``training_utils.evaluate`` -- TEST INFRASTRUCTURE.

``crps_ensemble`` restates the third-party ``properscoring.crps_ensemble`` (``properscoring`` in the reference's
``requirements.txt``; imported at ``training_utils.py:4``, called at ``:161``, ``:174``, ``:263``).  The package is
NOT installed in the container and the reference holds no fixtures for it => **parity unpinned** for this function.
It follows the published definition (Gneiting & Raftery 2007, eq. 21; the estimator properscoring documents for equal
member weights):

    CRPS(F_ens, y) = 1/M sum_i |x_i - y|  -  1/(2 M^2) sum_i sum_j |x_i - x_j|

and is anchored by closed-form cases in ``tests/test_oracle_evalmetrics.py`` (one member -> absolute error; a member
equal to the truth; shift invariance; the sorted-ensemble CDF-integral form, which is how properscoring evaluates it).

``evaluate_reference`` restates ``training_utils.evaluate`` (``training_utils.py:100-201``) with the posterior samples
passed in, so that the mirror's batched GPU evaluation can be compared on identical draws.
"""
import numpy as np
import torch


def crps_ensemble(truth, pred):
    """Scalar truth, 1-D ensemble ``pred`` -> CRPS (float64), pairwise form."""
    x = np.asarray(pred, dtype=np.float64).ravel()
    y = float(truth)
    m = x.size
    return np.abs(x - y).mean() - np.abs(x[:, None] - x[None, :]).sum() / (2.0 * m * m)


def crps_ensemble_cdf(truth, pred):
    """The same quantity as the integral of (F_ens(u) - 1[u >= y])^2 du over the sorted ensemble (the form
    properscoring's vectorised implementation integrates piecewise)."""
    x = np.sort(np.asarray(pred, dtype=np.float64).ravel())
    y = float(truth)
    m = x.size
    pts = np.concatenate([x, [y]])
    pts.sort()
    total = 0.0
    for lo, hi in zip(pts[:-1], pts[1:]):
        if hi == lo:
            continue
        mid = 0.5 * (lo + hi)
        cdf = np.searchsorted(x, mid, side="right") / m
        step = 1.0 if mid >= y else 0.0
        total += (cdf - step) ** 2 * (hi - lo)
    return total


def crps_field(truth, ens):
    """truth (...), ens (..., M) -> CRPS (...) with the triple Python loop of training_utils.py:168-175 vectorised."""
    t = np.asarray(truth, dtype=np.float64)
    e = np.asarray(ens, dtype=np.float64)
    m = e.shape[-1]
    s1 = np.abs(e - t[..., None]).mean(-1)
    s2 = np.abs(e[..., :, None] - e[..., None, :]).sum((-1, -2)) / (2.0 * m * m)
    return s1 - s2


def evaluate_reference(z0, z0_hat, x_test, mask_test, x_hat_point, z_samples, x_hat_samples, expert_dim):
    """Per-patient pieces of training_utils.evaluate for one chunk (training_utils.py:127-176).

    z0 (B, D) truth, z0_hat (B, D) point estimate, x_test / mask_test / x_hat_point (T', B, obs),
    z_samples (M, B, D), x_hat_samples (M, T', B, obs).  Returns dict of per-patient tensors:
    se_z0 (B,), mse_x (B,), crps_z0 (B,), crps_x (B,)."""
    se_z0 = torch.sum((z0[:, :expert_dim] - z0_hat[:, :expert_dim]) ** 2, dim=1)
    mse_x = torch.sum((x_test - x_hat_point) ** 2 * mask_test, dim=(0, 2)) / torch.sum(mask_test, dim=(0, 2))
    z_mat = torch.stack(list(z_samples), dim=-1)                 # B, D, M
    crps_z = crps_field(z0[:, :expert_dim].numpy(), z_mat[:, :expert_dim].numpy()).mean(axis=1)
    x_mat = torch.stack(list(x_hat_samples), dim=-1)             # T', B, obs, M
    crps_x = crps_field(x_test.numpy(), x_mat.numpy()).mean(axis=(0, 2))
    return {"se_z0": se_z0, "mse_x": mse_x, "crps_z0": torch.from_numpy(crps_z), "crps_x": torch.from_numpy(crps_x)}
