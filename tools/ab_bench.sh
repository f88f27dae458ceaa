#!/bin/bash
# A/B: baseline lib vs current lib in one session, alternating
for i in 1 2 3; do
  for v in base cur; do
    if [ $v = base ]; then export SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/libsonde_hip_base.so; else unset SONDE_HIP_LIB; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['config']['kernel_ms_avg'])"
  done
done
