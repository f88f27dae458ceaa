#!/bin/bash
# experiment: does the spacing of the channel rows in HBM move the decimator's launch time?  (SONDE_BENCH_PAD = samples of padding per row)
for i in 1 2; do
  for pad in 0 1024 16384 262144 65537; do
    SONDE_BENCH_PAD=$pad python bench.py --steps 200 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad', d['ms_per_step'], d['config']['kernels']['mix_decimate']['ms_per_step'], d['roofline']['frac'])"
  done
done
