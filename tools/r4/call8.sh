#!/bin/bash
set -u
OUT=gpurun_out/r4h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py -q -m gpu -x > $OUT/pytest_scan.log 2>&1
tail -6 $OUT/pytest_scan.log
for n in 32 512; do
  echo "== group form, $n channels"; SONDE_SP_PROF=1 timeout 300 python tools/scan_alone.py $n 2>&1 | grep "call 5\|prof" 
  echo "== pair form, $n channels"; SONDE_SP_OLD=1 timeout 300 python tools/scan_alone.py $n 2>&1 | grep "call 5"
done | tee $OUT/scan_alone.txt
for rep in 1 2; do
  timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'], d['config']['detections_last_step'][:4])"
done | tee $OUT/scan_wide.txt
