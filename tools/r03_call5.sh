#!/bin/bash
# Round-3 GPU call 5: c-wave adjoint -- parity tests of the rk kernels, then same-call A/B against the previous build.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c5
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_rk.py tests/test_hip_fullsize.py tests/test_hip_golden.py tests/test_hip_lstm.py tests/test_hip_dopri5.py tests/test_hip_model.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/tests.log | cut -c 1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 tools/ab_probe.py libhode_oldsplit.so libhode.so --reps 3 > $O/ab.txt 2>&1; echo "ab rc=$?"; grep rep $O/ab.txt
