"""Build-time guards that read the ISA of the built objects (no GPU): a compiler change that silently undoes a schedule the
source pins would pass every numerics test and cost 15 % of the BPTT kernel."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("tpw,nt", [(10, 3), (10, 1), (5, 3), (3, 3), (8, 2)])
def test_bptt_tile_loop_keeps_its_pinned_issue_order(tpw, nt):
    import check_lstm_isa
    obj = os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "csrc", "build", "hode_lstm_tpw%d.o" % tpw)
    if not os.path.exists(obj):
        pytest.skip("object files are not in the tree (library shipped pre-built)")
    assert check_lstm_isa.check(tpw, nt) == []


def test_split_forward_may_share_a_cu_and_backward_is_pinned():
    """Occupancy the kernel descriptors REQUEST (the backend pads the register allocation to enforce amdgpu_waves_per_eu):
    forward <= 2 workgroups per CU (measured +21-27 % past 12 288 patients, tools/scale_probe.py), backward 2 waves per SIMD
    (the theta wave shares one with a learned wave)."""
    import subprocess
    obj = os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "csrc", "build", "hode_rk_split.o")
    if not os.path.exists(obj):
        pytest.skip("object files are not in the tree")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_descriptor.py"), obj, "<12, 2, false"],
                         capture_output=True, text=True, check=True).stdout
    fwd = [l for l in out.splitlines() if "split_fwd_kernel<12, 2, false, true>" in l]
    bwd = [l for l in out.splitlines() if "split_bwd_kernel<12, 2, false, true, true>" in l]
    assert fwd and all("waves/SIMD <= 2" in l for l in fwd), out
    assert bwd and all("waves/SIMD <= 2" in l for l in bwd), out
