"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol include/hode.h declares, the ctypes
mirrors have the C struct sizes, and argument errors are reported (no kernel is launched without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hode.h")


@pytest.fixture(scope="module")
def lib():
    import hode
    if not os.path.exists(hode.library_path()):
        import build_hip
        build_hip.build(verbose=False)
    return hode.lib()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hode_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported_and_bound(lib):
    from hode import _lib as L
    declared = _declared_functions()
    assert {"hode_rk_fwd", "hode_rk_bwd", "hode_dopri5_fwd", "hode_dopri5_bwd", "hode_lstm_fwd", "hode_lstm_bwd",
            "hode_version", "hode_last_error_string", "hode_workspace_bytes"} <= set(declared)
    bound = {name for name, _, _ in L.EXPORTS}
    assert set(declared) == bound, (set(declared) ^ bound)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.hode_version() == L.HODE_ABI_VERSION


def test_struct_sizes_match_the_c_header(tmp_path):
    from hode import _lib as L
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(hode_solve_desc), '
                   'sizeof(hode_lstm_desc), offsetof(hode_solve_desc, workspace), sizeof(hode_readout_desc), '
                   'sizeof(hode_crps_desc), offsetof(hode_crps_desc, truth), sizeof(hode_mckl_desc), sizeof(hode_readout_mlp_desc), '
                   'sizeof(hode_dopri5_init_record));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    a, b, c, r, k, ko, mk, rm, ir = (int(v) for v in subprocess.check_output([str(exe)]).split())
    assert ctypes.sizeof(L.SolveDesc) == a and ctypes.sizeof(L.LstmDesc) == b
    assert L.SolveDesc.workspace.offset == c
    assert ctypes.sizeof(L.ReadoutDesc) == r and ctypes.sizeof(L.CrpsDesc) == k and L.CrpsDesc.truth.offset == ko
    assert ctypes.sizeof(L.McKlDesc) == mk and ctypes.sizeof(L.ReadoutMlpDesc) == rm and ctypes.sizeof(L.Dopri5InitRecord) == ir


def test_argument_errors_do_not_launch(lib):
    from hode import _lib as L
    assert lib.hode_rk_fwd(None, None) == -1 and b"NULL" in lib.hode_last_error_string()
    d = L.new_solve_desc()
    d.struct_size = 8
    assert lib.hode_rk_fwd(d, None) == -2 and b"struct_size" in lib.hode_last_error_string()
    d = L.new_solve_desc()
    d.rhs_kind, d.method, d.batch, d.latent_dim, d.n_times, d.n_dose = 0, 2, 4, 12, 5, 1
    assert lib.hode_rk_fwd(d, None) == -1  # required pointers missing
    d.rhs_kind = 7
    assert lib.hode_rk_fwd(d, None) == -3
    d.rhs_kind, d.batch = 0, 0
    assert lib.hode_rk_fwd(d, None) == -2


def test_workspace_query(lib):
    from hode import _lib as L
    d = L.new_solve_desc()
    d.batch, d.latent_dim, d.n_times = 10000, 12, 100
    P = 8 * 12 + 8 + 15
    assert lib.hode_workspace_bytes(d, L.WS_RK_FWD) == 0
    d.lanes_per_patient = 4  # quad layout: one partial row [w | b | theta] per wave
    n = lib.hode_workspace_bytes(d, L.WS_RK_BWD)
    assert n % (P * 4) == 0 and n // (P * 4) >= (10000 + 15) // 16
    d.lanes_per_patient = 0  # default at D = 12: split layout, 3 learned-wave rows + one theta row per 48 patients
    nblk = (10000 + 47) // 48
    n = lib.hode_workspace_bytes(d, L.WS_RK_BWD)
    assert n % 256 == 0 and n >= nblk * (3 * (8 * 12 + 8) + 15) * 4
    # HODE_FLAG_TAPE: forward and backward share one buffer = partials + 16 B per patient, interval and inner stage
    d.flags, d.method = L.FLAG_TAPE, L.METHODS["rk4"]
    nt = lib.hode_workspace_bytes(d, L.WS_RK_BWD)
    # expert stage states (3 x 16 B) + the learned block's last two stage derivatives (2 x 8 floats) per patient and interval
    assert nt == n + 99 * 3 * 10000 * 16 + 99 * 10000 * 8 * 2 * 4 and lib.hode_workspace_bytes(d, L.WS_RK_FWD) == nt
    d.method = L.METHODS["midpoint"]
    assert lib.hode_workspace_bytes(d, L.WS_RK_FWD) == n + 99 * 1 * 10000 * 16
    d.method = L.METHODS["euler"]  # single stage: nothing to tape
    assert lib.hode_workspace_bytes(d, L.WS_RK_FWD) == n
    d.lanes_per_patient = 4  # layouts without a tape ignore the flag
    assert lib.hode_workspace_bytes(d, L.WS_RK_FWD) == 0
    d.flags, d.lanes_per_patient = 0, 0
    d.latent_dim = 20  # no split layout for D = 20: quad
    n20 = lib.hode_workspace_bytes(d, L.WS_RK_BWD)
    assert n20 % ((16 * 20 + 16 + 15) * 4) == 0


def test_tape_flag_needs_its_workspace(lib):
    """hode_rk_fwd with HODE_FLAG_TAPE and no (or a short) workspace is an argument error, not a launch."""
    import ctypes as C
    from hode import _lib as L
    d = L.new_solve_desc()
    d.rhs_kind, d.method, d.batch, d.latent_dim, d.n_times, d.n_dose = 0, L.METHODS["rk4"], 96, 12, 5, 1
    buf = (C.c_float * 64)()
    ptr = C.addressof(buf)  # never dereferenced: the call must fail before any launch
    d.t = d.y0 = d.dosage = d.dose_times = d.theta = d.w1 = d.b1 = d.h = ptr
    d.flags = L.FLAG_TAPE
    assert lib.hode_rk_fwd(d, None) == -4 and b"workspace" in lib.hode_last_error_string()
    d.workspace, d.workspace_bytes = ptr, 128
    assert lib.hode_rk_fwd(d, None) == -4


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); import os; os.environ['HODE_LIBRARY'] = %r\n"
            "import hode\n"
            "try:\n    hode.lib()\nexcept hode.HodeConfigError as e:\n    print('RAISED', 'no CPU fallback' in str(e), isinstance(e, RuntimeError))\n"
            % (os.path.join(ROOT, "hybrid-ode-neurips-2021_amd"), str(tmp_path / "nope.so")))
    out = subprocess.check_output([sys.executable, "-c", code]).decode()
    assert "RAISED True False" in out  # a configuration error, not the RuntimeError the training loop treats as divergence


def test_library_was_built_from_the_sources_in_the_tree():
    """build_hip.py stamps libhode.so with a digest of the sources, header and flags it was built from; a library left
    over from other sources (an experiment reverted with `git checkout`, a forgotten rebuild) must not pass as current."""
    import build_hip
    stamp = build_hip.OUT + ".digest"
    assert os.path.exists(build_hip.OUT), "libhode.so missing: run `python build_hip.py`"
    assert os.path.exists(stamp), "libhode.so has no source digest: rebuild with `python build_hip.py`"
    assert open(stamp).read().strip() == build_hip.source_digest(), "libhode.so is stale: run `python build_hip.py`"
