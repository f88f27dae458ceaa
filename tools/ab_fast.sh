#!/bin/bash
# A/B of the decimator's sample loop in one session: hand-scheduled stream (default) vs the compiler's loop (SONDE_MD_NOFAST=1)
mkdir -p gpurun_out
for i in 1 2; do
  for v in fast nofast; do
    if [ $v = nofast ]; then export SONDE_MD_NOFAST=1; else unset SONDE_MD_NOFAST; fi
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_ms_avg'], d['config']['frames_ecc_ok'])"
  done
done 2>&1 | tee gpurun_out/ab_fast.txt
