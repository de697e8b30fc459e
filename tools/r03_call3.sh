#!/bin/bash
# Round-3 GPU call 3: LSTM hidden-size generalisation + golden-fixture tests, then the occupancy experiment.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c3
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_lstm.py tests/test_hip_golden.py tests/test_hip_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
timeout -k 10 600 python3 tools/scale_probe.py > $O/scale_probe.txt 2>&1; echo "scale rc=$?"; tail -40 $O/scale_probe.txt | cut -c 1-200
