#!/bin/bash
# A/B in one session: radiosonde_auto_rx_amd/libsonde_hip_base.so (built from another revision) against the current library, alternating
for i in 1 2; do
  for v in base cur; do
    if [ $v = base ]; then export SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/libsonde_hip_base.so; else unset SONDE_HIP_LIB; fi
    python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['config']['kernels'].items()}, d['roofline']['frac'])"
  done
done
