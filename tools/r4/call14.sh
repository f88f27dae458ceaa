#!/bin/bash
# round 4, GPU call 14: prefilter with the low-pass in place (half the window LDS), one reduction in front of the conversion; workgroups per CU A/B
set -u
OUT=gpurun_out/r4n
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py -q -m gpu -x > $OUT/pytest_scan.log 2>&1
tail -4 $OUT/pytest_scan.log
for rep in 1 2; do
for v in cur w4a32 w6a16 w6a8 w8a8; do
  if [ $v = cur ]; then unset SONDE_HIP_LIB; else export SONDE_HIP_LIB=$ROOT/radiosonde_auto_rx_amd/exp_$v.so; fi
  for n in 32 512; do
    echo "== $v $n channels: $(timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1)"
  done
  timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide $v', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'], d['config']['detections_last_step'][:4])"
done
done | tee $OUT/ab.txt
