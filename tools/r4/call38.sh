#!/bin/bash
# round 4, GPU call 38: prefilter at four workgroups per CU (64 registers, 8 KB of A fragments in LDS) against three (80 registers, 16 KB)
set -u
OUT=gpurun_out/r4zc
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
for rep in 1 2; do
for v in cur w8a8; do
  if [ $v = cur ]; then unset SONDE_HIP_LIB; else export SONDE_HIP_LIB=$ROOT/radiosonde_auto_rx_amd/exp_$v.so; fi
  for n in 32 512; do
    echo "== $v $n channels: $(timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1)"
  done
  SONDE_SCAN_WIDE_SERIAL=1 timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide (serial) $v', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'])"
done
done | tee $OUT/ab.txt
SONDE_HIP_LIB=$ROOT/radiosonde_auto_rx_amd/exp_w8a8.so timeout 600 python -m pytest tests/test_gpu_scan.py -q -m gpu 2>&1 | tail -2
