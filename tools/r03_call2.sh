#!/bin/bash
# Round-3 GPU call 2: new golden-fixture GPU tests + no_grad tape test, then the bench line under the driver's command,
# and the distributed code paths at world size 1 (graph-captured / overlap exchange).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c2
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_hip_golden.py tests/test_hip_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo "bench rc=$?"; tail -3 $O/bench_20_5.err
for mode in graph overlap; do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu --grad-exchange $mode > $O/bench_dist1_$mode.json 2> $O/bench_dist1_$mode.err; echo "dist1 $mode rc=$?"
done
