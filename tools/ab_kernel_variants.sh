#!/bin/bash
# kernel experiments: per-kernel table of a step with variant builds (radiosonde_auto_rx_amd/exp_<name>.so), phase profiles of the sync kernels
set -u
OUT=gpurun_out/r3i
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs --no-verify"
for v in ${VARIANTS:-base}; do
  SONDE_WF_PROF=1 SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/exp_$v.so SONDE_BENCH_NO_REPEAT=1 timeout 300 $B 2>$OUT/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['config']['kernels']; print('$v', d['ms_per_step'], {n: k[n]['ms_per_step'] for n in k})"
  grep "prof" $OUT/err.txt | cut -c1-300
done | tee $OUT/if_exp.txt
