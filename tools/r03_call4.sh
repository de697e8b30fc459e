#!/bin/bash
# Round-3 GPU call 4: the whole GPU suite after the LSTM refactor / occupancy change, then the driver's bench command.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c4
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log | cut -c 1-300
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo "bench rc=$?"; tail -3 $O/bench_20_5.err
