#!/bin/bash
# fsk_mixed A/B between builds (radiosonde_auto_rx_amd/<name>.so)
set -u
export TMPDIR=/tmp
for lib in ${LIBS:-libsonde_hip}; do
  for i in 1 2 3; do
  SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/$lib.so timeout 300 python bench.py --config fsk_mixed --steps 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['config']['kernel_ms_per_launch'])"
  done
done
