#!/bin/bash
# Round-3 GPU call 6: which wave bounds the 6-wave adjoint (stamps), + neural dims tests
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c6
mkdir -p $O
HODE_LIBRARY=$PWD/hybrid-ode-neurips-2021_amd/hode/libhode_stamps.so timeout -k 10 300 python3 tools/split_stamp_probe.py > $O/stamps.txt 2>&1; echo "stamps rc=$?"; cat $O/stamps.txt | grep -v amdgpu.ids
timeout -k 10 900 python3 -m pytest tests/test_hip_neural.py tests/test_hip_dopri5.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log | cut -c 1-300
