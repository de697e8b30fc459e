"""dopri5 drop-in: ``torchdiffeq.odeint(method="dopri5")`` semantics (reference default, ``sim_config.py:50``) on the
gfx950 kernels (``hode_dopri5_fwd`` / ``hode_dopri5_bwd``).

One launch per attempted step with the controller resident on the device; the library reads the controller record
back once per chunk of attempts (the step count is data dependent).  Accepted steps form a tape inside the
workspace tensor, which the autograd node keeps alive for the backward sweep.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .solver import _f32c, _ptr, _require_gpu, _stream

#: last forward's step statistics (diagnostics; mirrors what torchdiffeq exposes through nfe counters)
last_stats = {"n_accepted": 0, "n_rejected": 0}
#: set to True to have the next forwards keep a handle on their workspace (``read_tape``); tests only
keep_workspace = False
_last_ws = None


def read_tape():
    """The accepted-step tape of the last forward run with ``keep_workspace = True``: ``{"t", "dt"}`` as float64 lists,
    ``"j"`` (first / one-past-last output index per step) and the initial-step record (``hode_dopri5_init_record``), read
    back from the workspace at the offsets ``hode_dopri5_tape_offsets`` reports.  After the backward ``init["sigma"]`` is
    d loss / d dt_0."""
    if _last_ws is None:
        raise L.HodeConfigError("hode.adaptive.read_tape: no workspace kept (set keep_workspace = True before the solve)")
    ws, d, n_acc = _last_ws
    off = (C.c_size_t * 5)()
    L.check(L.lib().hode_dopri5_tape_offsets(d, off), "hode_dopri5_tape_offsets")
    raw = ws.cpu().numpy()
    import numpy as np
    rec = L.Dopri5InitRecord.from_buffer_copy(raw[off[0]:off[0] + C.sizeof(L.Dopri5InitRecord)].tobytes())
    t = np.frombuffer(raw[off[1]:off[1] + 8 * n_acc].tobytes(), dtype=np.float64)
    dt = np.frombuffer(raw[off[2]:off[2] + 8 * n_acc].tobytes(), dtype=np.float64)
    j = np.frombuffer(raw[off[3]:off[3] + 8 * n_acc].tobytes(), dtype=np.int32).reshape(n_acc, 2)
    return {"t": t.tolist(), "dt": dt.tolist(), "j": j.tolist(),
            "init": {k: getattr(rec, k) for k, _ in L.Dopri5InitRecord._fields_ if k != "pad"}}


def _status_error(status):
    msgs = []
    if status & L.STATUS_NONFINITE:
        msgs.append("non-finite values in state `y`")
    if status & L.STATUS_DT_UNDERFLOW:
        msgs.append("underflow in dt")
    return "; ".join(msgs)


class _RocheDopri5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, theta, w, b, t, dosage, dose_times, rtol, atol, ablate, lanes, max_steps, detach_first_step,
                grad_enabled=True):
        _require_gpu(y0, theta, t, dosage, dose_times)
        lib = L.lib()
        B, D = y0.shape
        T = t.numel()
        y0c, thc, tc = _f32c(y0), _f32c(theta), _f32c(t)
        dosc, dtc = _f32c(dosage), _f32c(dose_times)
        wc = _f32c(w) if w is not None else None
        bc = _f32c(b) if b is not None else None
        h = torch.empty((T, B, D), device=y0.device, dtype=torch.float32)
        # Without a backward to follow (evaluate() integrates mc_itr * B latents under no_grad) the accepted-state tape
        # shrinks to two rows and the step bound costs 24 bytes per step; with one, the tape is (steps + 1) * B * D * 4
        # bytes and a run that outgrows it is repeated with a larger one -- as long as the device can hold it.
        # `needs_input_grad` reflects the inputs' requires_grad, not the grad mode (Parameters under torch.no_grad() still
        # say True), and inside forward() grad mode is always off: the wrapper samples it before .apply().
        no_tape = not (grad_enabled and any(ctx.needs_input_grad[:4]))
        steps = int(max_steps) if max_steps else ((1 << 20) if no_tape else 16 * T + 64)
        while True:
            status = torch.zeros(1, device=y0.device, dtype=torch.int32)
            n_acc, n_rej = C.c_int32(0), C.c_int32(0)
            d = L.new_solve_desc()
            d.rhs_kind = L.RHS_ROCHE_ABLATE if ablate else L.RHS_ROCHE
            d.batch, d.latent_dim, d.n_times = B, D, T
            d.n_dose = dtc.shape[1] if dtc.dim() == 2 else 0
            d.lanes_per_patient = lanes
            d.t, d.y0, d.dosage, d.dose_times, d.theta = tc.data_ptr(), y0c.data_ptr(), dosc.data_ptr(), _ptr(dtc), thc.data_ptr()
            d.w1, d.b1, d.h, d.status = _ptr(wc), _ptr(bc), h.data_ptr(), status.data_ptr()
            d.rtol, d.atol, d.max_steps = float(rtol), float(atol), steps
            d.flags = L.FLAG_NO_TAPE if no_tape else 0
            d.host_n_accepted, d.host_n_rejected = C.pointer(n_acc), C.pointer(n_rej)
            nbytes = lib.hode_workspace_bytes(d, L.WS_DOPRI5_FWD)
            free, _ = torch.cuda.mem_get_info(y0.device)
            if nbytes > free + torch.cuda.memory_reserved(y0.device) - torch.cuda.memory_allocated(y0.device):
                raise L.HodeError("hode dopri5: a tape of %d accepted steps needs %.1f GB, the device has %.1f GB free "
                                  "(max_num_steps exceeded)" % (steps, nbytes / 1e9, free / 1e9))
            ws = torch.empty(nbytes, device=y0.device, dtype=torch.uint8)
            d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
            with torch.cuda.device(y0.device):
                L.check(lib.hode_dopri5_fwd(d, _stream()), "hode_dopri5_fwd")
            st = int(status.item())
            if st & L.STATUS_MAX_STEPS and not (st & (L.STATUS_NONFINITE | L.STATUS_DT_UNDERFLOW)) and steps < (1 << 20):
                del ws
                steps *= 4  # tape too small: retry with a larger one (refused above once it no longer fits the device)
                continue
            break
        last_stats.update(n_accepted=n_acc.value, n_rejected=n_rej.value, no_tape=no_tape, workspace_bytes=int(nbytes))
        if st:
            # torchdiffeq raises AssertionError here, which the reference's training loop does NOT catch; a RuntimeError
            # subclass lets `except RuntimeError` (training_utils.py:45) end the diverged restart instead.
            raise L.HodeError("hode dopri5: " + (_status_error(st) or "max_num_steps exceeded"))
        ctx.save_for_backward(h, thc, wc if wc is not None else thc, bc if bc is not None else thc, tc, dosc, dtc, y0c, ws)
        ctx.meta = (float(rtol), float(atol), bool(ablate), int(lanes), w is not None, steps, n_acc.value,
                    bool(detach_first_step))
        if keep_workspace:
            global _last_ws
            _last_ws = (ws, d, n_acc.value)
        return h

    @staticmethod
    def backward(ctx, grad_h):
        h, thc, wc, bc, tc, dosc, dtc, y0c, ws = ctx.saved_tensors
        rtol, atol, ablate, lanes, has_w, steps, n_accepted, detach_first = ctx.meta
        lib = L.lib()
        T, B, D = h.shape
        gh = grad_h.to(torch.float32).contiguous()
        need_th = bool(ctx.needs_input_grad[1])
        gy0 = torch.empty((B, D), device=h.device, dtype=torch.float32)
        gth = torch.zeros(L.N_THETA, device=h.device, dtype=torch.float32)
        gw = torch.zeros_like(wc) if has_w else None
        gb = torch.zeros_like(bc) if has_w else None
        n_acc = C.c_int32(n_accepted)
        d = L.new_solve_desc()
        d.rhs_kind = L.RHS_ROCHE_ABLATE if ablate else L.RHS_ROCHE
        d.batch, d.latent_dim, d.n_times = B, D, T
        d.n_dose = dtc.shape[1] if dtc.dim() == 2 else 0
        d.lanes_per_patient = lanes
        d.need_theta_grad = int(need_th)
        d.t, d.y0, d.dosage, d.dose_times, d.theta = tc.data_ptr(), y0c.data_ptr(), dosc.data_ptr(), _ptr(dtc), thc.data_ptr()
        d.w1, d.b1, d.h = (_ptr(wc), _ptr(bc), h.data_ptr()) if has_w else (0, 0, h.data_ptr())
        d.grad_h, d.grad_y0 = gh.data_ptr(), gy0.data_ptr()
        d.grad_w1, d.grad_b1, d.grad_theta = _ptr(gw), _ptr(gb), gth.data_ptr()
        d.rtol, d.atol, d.max_steps = rtol, atol, steps
        d.flags = L.FLAG_DETACH_FIRST_STEP if detach_first else 0
        d.host_n_accepted = C.pointer(n_acc)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        with torch.cuda.device(h.device):
            L.check(lib.hode_dopri5_bwd(d, _stream()), "hode_dopri5_bwd")
        return gy0, (gth if need_th else None), gw, gb, None, None, None, None, None, None, None, None, None, None


def roche_dopri5(y0, theta, w, b, t, dosage, dose_times, rtol=1e-7, atol=1e-9, ablate=False, lanes_per_patient=0,
                 max_steps=0, detach_first_step=False):
    """Adaptive solve of the Roche rhs; returns h (T, B, D).  Arguments as ``hode.roche_solve``.

    The backward differentiates Hairer's first step size like torchdiffeq's graph does (reference model.py:1116 +
    training_utils.py:50); ``detach_first_step=True`` treats it as a constant (``HODE_FLAG_DETACH_FIRST_STEP``)."""
    if dose_times.dim() != 2:
        dose_times = dose_times.reshape(y0.shape[0], -1)
    return _RocheDopri5.apply(y0, theta, w, b, t, dosage, dose_times.to(torch.float32), rtol, atol, bool(ablate),
                              int(lanes_per_patient), int(max_steps), bool(detach_first_step), torch.is_grad_enabled())


class _NeuralDopri5(torch.autograd.Function):
    """NeuralODE rhs (reference model.py:969-1026) through ``hode_dopri5_fwd / _bwd`` with ``HODE_RHS_NEURAL``: the MFMA
    attempt kernels of csrc/hode_neural_dopri5.hip; the backward accumulates the weight gradients on chip."""

    @staticmethod
    def forward(ctx, y0, w1, b1, w2, b2, t, dosage, dose_times, rtol, atol, max_steps, detach_first_step, grad_enabled=True):
        _require_gpu(y0, w1, t, dosage, dose_times)
        lib = L.lib()
        B, D = y0.shape
        T = t.numel()
        y0c, tc, dosc, dtc = _f32c(y0), _f32c(t), _f32c(dosage), _f32c(dose_times)
        w1c, b1c, w2c, b2c = _f32c(w1), _f32c(b1), _f32c(w2), _f32c(b2)
        h = torch.empty((T, B, D), device=y0.device, dtype=torch.float32)
        no_tape = not (grad_enabled and any(ctx.needs_input_grad[:5]))  # see _RocheDopri5.forward
        steps = int(max_steps) if max_steps else ((1 << 20) if no_tape else 16 * T + 64)
        while True:
            status = torch.zeros(1, device=y0.device, dtype=torch.int32)
            n_acc, n_rej = C.c_int32(0), C.c_int32(0)
            d = L.new_solve_desc()
            d.rhs_kind = L.RHS_NEURAL
            d.batch, d.latent_dim, d.n_times, d.hidden_dim = B, D, T, w1c.shape[0]
            d.n_dose = dtc.shape[1] if dtc.dim() == 2 else 0
            d.t, d.y0, d.dosage, d.dose_times = tc.data_ptr(), y0c.data_ptr(), dosc.data_ptr(), _ptr(dtc)
            d.w1, d.b1, d.w2, d.b2 = w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr()
            d.h, d.status = h.data_ptr(), status.data_ptr()
            d.rtol, d.atol, d.max_steps = float(rtol), float(atol), steps
            d.flags = L.FLAG_NO_TAPE if no_tape else 0
            d.host_n_accepted, d.host_n_rejected = C.pointer(n_acc), C.pointer(n_rej)
            nbytes = lib.hode_workspace_bytes(d, L.WS_DOPRI5_FWD)
            free, _ = torch.cuda.mem_get_info(y0.device)
            if nbytes > free + torch.cuda.memory_reserved(y0.device) - torch.cuda.memory_allocated(y0.device):
                raise L.HodeError("hode dopri5: a tape of %d accepted steps needs %.1f GB, the device has %.1f GB free "
                                  "(max_num_steps exceeded)" % (steps, nbytes / 1e9, free / 1e9))
            ws = torch.empty(nbytes, device=y0.device, dtype=torch.uint8)
            d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
            with torch.cuda.device(y0.device):
                L.check(lib.hode_dopri5_fwd(d, _stream()), "hode_dopri5_fwd[neural]")
            st = int(status.item())
            if st & L.STATUS_MAX_STEPS and not (st & (L.STATUS_NONFINITE | L.STATUS_DT_UNDERFLOW)) and steps < (1 << 20):
                del ws
                steps *= 4
                continue
            break
        last_stats.update(n_accepted=n_acc.value, n_rejected=n_rej.value, no_tape=no_tape, workspace_bytes=int(nbytes))
        if st:
            raise L.HodeError("hode dopri5: " + (_status_error(st) or "max_num_steps exceeded"))
        ctx.save_for_backward(h, tc, dosc, dtc, y0c, w1c, b1c, w2c, b2c, ws)
        ctx.meta = (float(rtol), float(atol), steps, n_acc.value, bool(detach_first_step))
        if keep_workspace:
            global _last_ws
            _last_ws = (ws, d, n_acc.value)
        return h

    @staticmethod
    def backward(ctx, grad_h):
        h, tc, dosc, dtc, y0c, w1c, b1c, w2c, b2c, ws = ctx.saved_tensors
        rtol, atol, steps, n_accepted, detach_first = ctx.meta
        lib = L.lib()
        T, B, D = h.shape
        gh = grad_h.to(torch.float32).contiguous()
        gy0 = torch.empty((B, D), device=h.device, dtype=torch.float32)
        gw1, gb1, gw2, gb2 = (torch.zeros_like(x) for x in (w1c, b1c, w2c, b2c))
        n_acc = C.c_int32(n_accepted)
        d = L.new_solve_desc()
        d.rhs_kind = L.RHS_NEURAL
        d.batch, d.latent_dim, d.n_times, d.hidden_dim = B, D, T, w1c.shape[0]
        d.n_dose = dtc.shape[1] if dtc.dim() == 2 else 0
        d.t, d.y0, d.dosage, d.dose_times = tc.data_ptr(), y0c.data_ptr(), dosc.data_ptr(), _ptr(dtc)
        d.w1, d.b1, d.w2, d.b2, d.h = w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(), h.data_ptr()
        d.grad_h, d.grad_y0 = gh.data_ptr(), gy0.data_ptr()
        d.grad_w1, d.grad_b1, d.grad_w2, d.grad_b2 = gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr()
        d.rtol, d.atol, d.max_steps = rtol, atol, steps
        d.flags = L.FLAG_DETACH_FIRST_STEP if detach_first else 0
        d.host_n_accepted = C.pointer(n_acc)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        with torch.cuda.device(h.device):
            L.check(lib.hode_dopri5_bwd(d, _stream()), "hode_dopri5_bwd[neural]")
        return gy0, gw1, gb1, gw2, gb2, None, None, None, None, None, None, None, None


#: latent dimensions the fused neural dopri5 kernels are compiled for (csrc/hode_neural_dopri5.hip: [y, Dose, 1] must fit
#: one 16-row MFMA tile); the reference's simulation configs use 6, 8 and 12
NEURAL_DIMS = (4, 6, 8, 10, 12, 14)


def neural_dopri5(y0, w1, b1, w2, b2, t, dosage, dose_times, rtol=1e-7, atol=1e-9, max_steps=0, detach_first_step=False):
    """Adaptive solve of dy/dt = tanh(W2 tanh(W1 [y, Dose(t)] + b1) + b2); returns h (T, B, D)."""
    if dose_times.dim() != 2:
        dose_times = dose_times.reshape(y0.shape[0], -1)
    return _NeuralDopri5.apply(y0, w1, b1, w2, b2, t, dosage, dose_times.to(torch.float32), rtol, atol, int(max_steps),
                               bool(detach_first_step), torch.is_grad_enabled())
