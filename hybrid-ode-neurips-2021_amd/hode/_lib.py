"""ctypes binding of libhode.so (C ABI: include/hode.h).  Fails loudly when the library is missing."""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libhode.so"

HODE_ABI_VERSION = 1

RHS_ROCHE, RHS_ROCHE_ABLATE, RHS_NEURAL, RHS_ROCHE_REAL = 0, 1, 2, 3
METHOD_EULER, METHOD_MIDPOINT, METHOD_RK4_38 = 0, 1, 2
METHODS = {"euler": METHOD_EULER, "midpoint": METHOD_MIDPOINT, "rk4": METHOD_RK4_38}
N_THETA = 16
STATUS_NONFINITE, STATUS_DT_UNDERFLOW, STATUS_MAX_STEPS = 1, 2, 4
WS_RK_FWD, WS_RK_BWD, WS_DOPRI5_FWD, WS_DOPRI5_BWD = 0, 1, 2, 3
FLAG_SKIP_FOLD, FLAG_OVERWRITE_GRADS, FLAG_TAPE, FLAG_DETACH_FIRST_STEP, FLAG_NO_TAPE = 1, 2, 4, 8, 16


class HodeError(RuntimeError):
    """NUMERICAL failure of a solve, reported by the kernels' status word: non-finite state, dt underflow, step bound
    exceeded.  A RuntimeError so that the reference's ``except RuntimeError`` around ``model.loss``
    (training_utils.py:43-47) keeps ending a diverged restart."""


class HodeConfigError(Exception):
    """Everything that is NOT the numerics' fault: libhode.so missing / stale / ABI mismatch, an argument error or
    unsupported shape reported by an entry point (HODE_E_*), a HIP launch error, CPU tensors handed to the GPU-only
    path.  Deliberately not a RuntimeError: the mirrored training loop must not mistake it for solver divergence,
    print it, save the untrained model and carry on."""


_fp = C.c_void_p  # device pointers travel as integers


class SolveDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("rhs_kind", C.c_int32),
        ("method", C.c_int32),
        ("perturb", C.c_int32),
        ("batch", C.c_int32),
        ("latent_dim", C.c_int32),
        ("n_times", C.c_int32),
        ("n_dose", C.c_int32),
        ("hidden_dim", C.c_int32),
        ("n_action_times", C.c_int32),
        ("lanes_per_patient", C.c_int32),
        ("need_theta_grad", C.c_int32),
        ("t", _fp), ("y0", _fp), ("dosage", _fp), ("dose_times", _fp), ("theta", _fp),
        ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("h", _fp), ("status", _fp),
        ("grad_h", _fp), ("grad_y0", _fp), ("grad_w1", _fp), ("grad_b1", _fp), ("grad_w2", _fp),
        ("grad_b2", _fp), ("grad_theta", _fp),
        ("rtol", C.c_double), ("atol", C.c_double),
        ("max_steps", C.c_int32), ("flags", C.c_int32),
        ("host_n_accepted", C.POINTER(C.c_int32)), ("host_n_rejected", C.POINTER(C.c_int32)),
        ("workspace", _fp), ("workspace_bytes", C.c_size_t),
    ]


class Dopri5InitRecord(C.Structure):
    _fields_ = [("h0", C.c_float), ("d0", C.c_float), ("d1", C.c_float), ("d2", C.c_float), ("h1", C.c_float),
                ("first_accepted", C.c_int32), ("sigma", C.c_float), ("pad", C.c_int32)]


class LstmDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("seq_len", C.c_int32), ("batch", C.c_int32), ("input_dim", C.c_int32), ("hidden_dim", C.c_int32),
        ("obs_dim", C.c_int32), ("reverse", C.c_int32), ("save_tape", C.c_int32),
        ("x", _fp), ("a", _fp), ("mask", _fp), ("w_ih", _fp), ("w_hh", _fp), ("b_ih", _fp), ("b_hh", _fp),
        ("h_out", _fp), ("c_out", _fp), ("grad_h_out", _fp), ("grad_gates", _fp), ("h_prev", _fp),
        ("workspace", _fp), ("workspace_bytes", C.c_size_t),
    ]


class ReadoutDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("latent_dim", C.c_int32), ("obs_dim", C.c_int32), ("scale", C.c_float),
        ("rows", C.c_int64),
        ("h", _fp), ("x", _fp), ("mask", _fp), ("w", _fp), ("b", _fp), ("lik", _fp),
        ("grad_h", _fp), ("grad_w", _fp), ("grad_b", _fp),
        ("workspace", _fp), ("workspace_bytes", C.c_size_t),
    ]


class ReadoutMlpDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("latent_dim", C.c_int32), ("hidden_dim", C.c_int32), ("obs_dim", C.c_int32),
        ("batch", C.c_int32), ("scale", C.c_float), ("rows", C.c_int64),
        ("h", _fp), ("x", _fp), ("mask", _fp), ("time_weight", _fp), ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp),
        ("lik", _fp), ("grad_h", _fp), ("grad_w1", _fp), ("grad_b1", _fp), ("grad_w2", _fp), ("grad_b2", _fp),
        ("workspace", _fp), ("workspace_bytes", C.c_size_t),
    ]


class CrpsDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_times", C.c_int32), ("batch", C.c_int32), ("n_members", C.c_int32),
        ("latent_dim", C.c_int32), ("obs_dim", C.c_int32),
        ("time_stride", C.c_int64), ("member_stride", C.c_int64), ("patient_stride", C.c_int64),
        ("h", _fp), ("w", _fp), ("b", _fp), ("truth", _fp), ("crps", _fp), ("crps_sum", _fp),
    ]


class McKlDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_samples", C.c_int32), ("rows", C.c_int64), ("rate", C.c_float),
        ("clamp_value", C.c_float),
        ("mu", _fp), ("log_var", _fp), ("noise", _fp), ("kl", _fp), ("grad_mu", _fp), ("grad_log_var", _fp),
    ]


#: every symbol include/hode.h declares: (name, restype, argtypes)
EXPORTS = (
    ("hode_version", C.c_int, ()),
    ("hode_last_error_string", C.c_char_p, ()),
    ("hode_workspace_bytes", C.c_size_t, (C.POINTER(SolveDesc), C.c_int)),
    ("hode_lstm_workspace_bytes", C.c_size_t, (C.POINTER(LstmDesc),)),
    ("hode_neural_tape_offsets", C.c_int, (C.POINTER(SolveDesc), C.POINTER(C.c_size_t))),
    ("hode_rk_fwd", C.c_int, (C.POINTER(SolveDesc), C.c_void_p)),
    ("hode_rk_bwd", C.c_int, (C.POINTER(SolveDesc), C.c_void_p)),
    ("hode_dopri5_fwd", C.c_int, (C.POINTER(SolveDesc), C.c_void_p)),
    ("hode_dopri5_bwd", C.c_int, (C.POINTER(SolveDesc), C.c_void_p)),
    ("hode_dopri5_tape_offsets", C.c_int, (C.POINTER(SolveDesc), C.POINTER(C.c_size_t))),
    ("hode_readout_workspace_bytes", C.c_size_t, (C.POINTER(ReadoutDesc),)),
    ("hode_readout_sse", C.c_int, (C.POINTER(ReadoutDesc), C.c_void_p)),
    ("hode_readout_mlp_workspace_bytes", C.c_size_t, (C.POINTER(ReadoutMlpDesc),)),
    ("hode_readout_mlp_sse", C.c_int, (C.POINTER(ReadoutMlpDesc), C.c_void_p)),
    ("hode_ensemble_crps", C.c_int, (C.POINTER(CrpsDesc), C.c_void_p)),
    ("hode_mc_kl_exponential", C.c_int, (C.POINTER(McKlDesc), C.c_void_p)),
    ("hode_lstm_fwd", C.c_int, (C.POINTER(LstmDesc), C.c_void_p)),
    ("hode_lstm_bwd", C.c_int, (C.POINTER(LstmDesc), C.c_void_p)),
    ("hode_lstm_fill_operand", C.c_int, (C.POINTER(LstmDesc), C.c_void_p)),
)

_lib = None


def library_path() -> str:
    return os.environ.get("HODE_LIBRARY", os.path.join(_HERE, _LIB_NAME))


def lib():
    """Load (once) and return the ctypes handle; raises HodeConfigError if the library is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise HodeConfigError(
            "hode: %s not found -- build it with `python build_hip.py` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the solver path." % path
        )
    handle = C.CDLL(path)
    for name, restype, argtypes in EXPORTS:
        fn = getattr(handle, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = list(argtypes)
    if handle.hode_version() != HODE_ABI_VERSION:
        raise HodeConfigError("hode: ABI version %d != expected %d" % (handle.hode_version(), HODE_ABI_VERSION))
    _lib = handle
    return _lib


def check(code: int, what: str):
    if code != 0:
        msg = lib().hode_last_error_string().decode("utf-8", "replace")
        raise HodeConfigError("%s failed (code %d): %s" % (what, code, msg))


def new_solve_desc() -> SolveDesc:
    d = SolveDesc()
    d.struct_size = C.sizeof(SolveDesc)
    return d


def new_lstm_desc() -> LstmDesc:
    d = LstmDesc()
    d.struct_size = C.sizeof(LstmDesc)
    return d
