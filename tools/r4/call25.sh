#!/bin/bash
# round 4, GPU call 25: FSK modem, ring length of the oscillator's output (SONDE_FSK_RING) at auto_rx's short frames
set -u
OUT=gpurun_out/r4x
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for r in default 256 512 1024 2048 4096; do
  if [ $r = default ]; then unset SONDE_FSK_RING; else export SONDE_FSK_RING=$r; fi
  python bench.py --config fsk_mixed --steps 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ring $r', d['ms_per_step'], d['config'].get('kernel_ms_per_launch'), d['config']['verified_channels'])"
done
done | tee $OUT/fsk_ring.txt
