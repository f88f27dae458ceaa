#!/bin/bash
# round 4, GPU call 30: FSK modem, timing-only variants with one part of the pipeline removed each (results are garbage) — where a short-frame channel's time goes
set -u
OUT=gpurun_out/r4za
mkdir -p $OUT
export TMPDIR=/tmp
for lib in libsonde_hip exp_NOEST exp_NOWALK exp_NOWIN exp_NOSOFT libsonde_hip; do
  SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/$lib.so timeout 300 python bench.py --config fsk_mixed --steps 10 --no-cpu-baseline 2>$OUT/err_$lib.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['config']['kernel_ms_per_launch'])"
  grep -c "gave up" $OUT/err_$lib.txt
done | tee $OUT/fsk_ablation.txt
