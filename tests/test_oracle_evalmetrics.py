"""The CRPS oracle (oracle/evalmetrics.py) against closed forms.  properscoring is absent => parity unpinned; these are
the anchors its docstring lists."""
import numpy as np

from oracle.evalmetrics import crps_ensemble, crps_ensemble_cdf, crps_field


def test_single_member_is_absolute_error():
    assert crps_ensemble(1.5, [0.25]) == 1.25
    assert crps_ensemble(-2.0, [-2.0]) == 0.0


def test_two_members_by_hand():
    # x = {0, 2}, y = 1: mean|x-y| = 1, sum_ij|xi-xj| = 4, 4/(2*4) = 0.5 -> 0.5
    assert abs(crps_ensemble(1.0, [0.0, 2.0]) - 0.5) < 1e-15
    # y outside the ensemble: x = {0, 2}, y = 5: 4 - 0.5 = 3.5
    assert abs(crps_ensemble(5.0, [0.0, 2.0]) - 3.5) < 1e-15


def test_pairwise_form_equals_cdf_integral():
    rng = np.random.default_rng(3)
    for m in (1, 2, 7, 50):
        for _ in range(20):
            x = rng.normal(size=m) * rng.uniform(0.1, 3.0)
            y = rng.normal() * 2.0
            assert abs(crps_ensemble(y, x) - crps_ensemble_cdf(y, x)) < 1e-12


def test_shift_and_scale_behaviour():
    rng = np.random.default_rng(4)
    x, y = rng.normal(size=50), 0.3
    base = crps_ensemble(y, x)
    assert abs(crps_ensemble(y + 7.0, x + 7.0) - base) < 1e-12
    assert abs(crps_ensemble(3.0 * y, 3.0 * x) - 3.0 * base) < 1e-12
    assert base >= 0.0


def test_field_matches_scalar_loop():
    rng = np.random.default_rng(5)
    truth = rng.normal(size=(3, 4, 5))
    ens = rng.normal(size=(3, 4, 5, 11))
    ref = np.zeros_like(truth)
    for i in range(3):
        for j in range(4):
            for k in range(5):
                ref[i, j, k] = crps_ensemble(truth[i, j, k], ens[i, j, k])
    assert np.abs(crps_field(truth, ens) - ref).max() < 1e-13
