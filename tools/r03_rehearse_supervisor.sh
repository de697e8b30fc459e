#!/bin/bash
# Rehearsal of bench.py's N > 1 machinery on a ONE-GPU box, on RCCL: world size 1 through torch.distributed.run with the
# supervisor forced on -- (1) the default graph-captured exchange, (2) a forced failure of it -> fresh rank on the overlap
# path under a new store prefix of the launcher's agent store.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_supervisor
mkdir -p $O
run() {
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/$1.json 2> $O/$1.err
  echo "$1 rc=$?"; grep "supervisor\|guard\|communicator" $O/$1.err | cut -c 1-200
  python3 -c "
import json; d=json.load(open('$O/$1.json')); print('  value %.2f M, %s, fallback %s, full step %.2f ms (n_ranks %d), dopri5 %.1f ms' % (d['value']/1e6, d['config']['grad_exchange'], d['config']['grad_exchange_fallback'], d['full_training_step']['ms'], d['full_training_step']['n_ranks'], d['dopri5_step']['ms']))"
}
HODE_BENCH_FORCE_SUPERVISOR=1 run graph 29611
HODE_BENCH_FORCE_SUPERVISOR=1 HODE_BENCH_FAIL_GRAPH=1 run fallback 29612
