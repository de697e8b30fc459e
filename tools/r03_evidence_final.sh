#!/bin/bash
# Round-3 GPU call 10: evidence for the final kernels -- full GPU suite, the driver's bench command plain and under rocprofv3
# (kernel stats), PMC passes (separate: FETCH_SIZE / WRITE_SIZE / SQ groups) for the encoder and for the headline kernels.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_final
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log | cut -c 1-300
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "stats rc=$?"
pass() {
  tag=$1; name=$2; prog=$3; shift 3
  rocprofv3 --pmc "$@" -d $O/$tag/pmc_$name -o c --output-format csv -- python3 $prog > $O/${tag}_pmc_$name.log 2>&1 || { echo "pmc $tag $name failed"; tail -3 $O/${tag}_pmc_$name.log; return 1; }
  echo "pmc $tag $name done"
}
for tag in lstm split; do
  if [ $tag = lstm ]; then prog="tools/lstm_train_probe.py"; else prog="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --headline-only"; fi
  pass $tag fetch "$prog" FETCH_SIZE &&
  pass $tag write "$prog" WRITE_SIZE &&
  pass $tag sq1 "$prog" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU &&
  pass $tag sq2 "$prog" SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE &&
  pass $tag sq3 "$prog" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
done
rocprofv3 --kernel-trace --stats -d $O/lstm_stats -o s --output-format csv -- python3 tools/lstm_train_probe.py > $O/lstm_stats.log 2>&1; echo "lstm stats rc=$?"
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.csv" -size +30M -delete
du -sh $O
