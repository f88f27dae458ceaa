#!/bin/bash
# A/B/C in one session: bench the current library and every radiosonde_auto_rx_amd/libsonde_hip_<tag>.so given as argument, alternating, twice
for i in 1 2; do
  for v in cur "$@"; do
    if [ $v = cur ]; then unset SONDE_HIP_LIB; else export SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/libsonde_hip_$v.so; fi
    python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['config']['kernels'].items()}, d['roofline']['frac'])"
  done
done
