#!/bin/bash
# Timing experiments on the decimator's sample loop (library built with EXTRA=-DSONDE_MD_EXPERIMENTS): SONDE_MD_VARIANT selects a
# generated stream with parts removed (tools/gen_md_fast.py --experiments ...).  Only variant 1 computes correct results.
mkdir -p gpurun_out
for i in 1 2; do
  for v in ${VARIANTS:-1 2 3 4 5 6 7}; do
    SONDE_MD_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_ms_avg']['mix_decimate'], d['config']['frames_ecc_ok'])"
  done
done 2>&1 | tee gpurun_out/ab_variants.txt
