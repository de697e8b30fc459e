#!/bin/bash
# Round-3 GPU call 1: (a) clock-ramp probe + the driver's 20/5 command next to 200/20; (b) rocprofv3 --pmc passes and
# kernel stats for the CURRENT encoder kernels (lstm_fwd_kernel<3,5,8,true>, lstm_bwd_kernel<3,10>).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_ramp
mkdir -p $O
python3 tools/ramp_probe.py > $O/ramp.txt 2> $O/ramp.err || { tail -5 $O/ramp.err; exit 1; }
echo "ramp done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --headline-only > $O/bench_20_5_a.json 2> $O/bench_20_5_a.err || exit 1
python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu --headline-only > $O/bench_200_20.json 2> $O/bench_200_20.err || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --headline-only > $O/bench_20_5_b.json 2> $O/bench_20_5_b.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats -d $O/lstm_stats -o s --output-format csv -- python3 tools/lstm_train_probe.py > $O/lstm_stats.log 2>&1 || exit 1
echo "stats done"
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" -d $O/pmc_$name -o c --output-format csv -- python3 tools/lstm_train_probe.py > $O/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $O/pmc_$name.log; return 1; }
  echo "pmc $name done"
}
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU &&
pass sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE &&
pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
find $O -name "*.csv" -size +20M -delete
ls -la $O
