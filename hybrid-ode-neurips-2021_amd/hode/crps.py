"""Ensemble CRPS on the GPU (``hode_ensemble_crps``): the metric loop of the reference's ``training_utils.evaluate``
(``training_utils.py:147-176``) as one kernel over posterior samples that were integrated in ONE solver launch."""

from __future__ import annotations

import torch

from . import _lib as L
from .solver import _f32c, _require_gpu, _stream


def ensemble_crps(h, truth, n_members, weight=None, bias=None, per_component=False):
    """CRPS of an M-member ensemble against ``truth`` (T', B, obs).

    ``h`` is (T', M * B, Dv) with the batch axis member-major (index m * B + b) -- exactly what a decoder returns for
    ``z.reshape(M * B, D)`` built from ``torch.stack`` of M draws.  With ``weight`` (obs, Dv) [and ``bias`` (obs,)] the
    scored value is the linear readout ``weight @ h_m + bias`` (never materialised); without, ``h_m[:obs]`` itself.
    Returns the per-(time, patient) SUM over components (T', B), or the full (T', B, obs) field if ``per_component``.
    """
    _require_gpu(h, truth)
    lib = L.lib()
    Tn, MB, Dv = h.shape
    M = int(n_members)
    if MB % M:
        raise ValueError("hode.ensemble_crps: batch axis %d is not a multiple of n_members %d" % (MB, M))
    B = MB // M
    obs = truth.shape[-1]
    if tuple(truth.shape) != (Tn, B, obs):
        raise ValueError("hode.ensemble_crps: truth shape %s != (%d, %d, obs)" % (tuple(truth.shape), Tn, B))
    hc, tc = _f32c(h), _f32c(truth)
    d = L.CrpsDesc()
    d.struct_size = L.C.sizeof(L.CrpsDesc)
    d.n_times, d.batch, d.n_members, d.latent_dim, d.obs_dim = Tn, B, M, Dv, obs
    d.time_stride, d.member_stride, d.patient_stride = MB * Dv, B * Dv, Dv
    d.h, d.truth = hc.data_ptr(), tc.data_ptr()
    keep = [hc, tc]
    if weight is not None:
        wc = _f32c(weight)
        if tuple(wc.shape) != (obs, Dv):
            raise ValueError("hode.ensemble_crps: weight shape %s != (%d, %d)" % (tuple(wc.shape), obs, Dv))
        d.w = wc.data_ptr()
        keep.append(wc)
        if bias is not None:
            bc = _f32c(bias)
            d.b = bc.data_ptr()
            keep.append(bc)
    if per_component:
        out = torch.empty((Tn, B, obs), device=h.device, dtype=torch.float32)
        d.crps = out.data_ptr()
    else:
        out = torch.empty((Tn, B), device=h.device, dtype=torch.float32)
        d.crps_sum = out.data_ptr()
    with torch.cuda.device(h.device):
        L.check(lib.hode_ensemble_crps(d, _stream()), "hode_ensemble_crps")
    return out
