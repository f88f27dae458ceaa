#!/bin/bash
# round 4, GPU call 18: scan_wide with the two stages one step apart; FSK modem: tone rings padded against bank conflicts, wait-loop naps
set -u
OUT=gpurun_out/r4r
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_fsk.py -q -m gpu -x > $OUT/pytest_fsk.log 2>&1
tail -3 $OUT/pytest_fsk.log
for rep in 1 2; do
  for mode in pipelined serial; do
    if [ $mode = serial ]; then export SONDE_SCAN_WIDE_SERIAL=1; else unset SONDE_SCAN_WIDE_SERIAL; fi
    timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide $mode', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'], d['config']['detections_last_step'][:4])"
  done
done | tee $OUT/scan_wide.txt
unset SONDE_SCAN_WIDE_SERIAL
LIBS="libsonde_hip exp_pad0 exp_nap2 exp_nap4" bash tools/ab_fsk_mixed.sh 2>&1 | tee $OUT/fsk_ab.txt
