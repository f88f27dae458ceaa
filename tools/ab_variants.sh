#!/bin/bash
# Timing experiments on the decimator (library built with EXTRA=-DSONDE_MD_EXPERIMENTS after
# `python3 tools/gen_md_fast.py --experiments <spec2> <spec3> ...`): SONDE_MD_VARIANT=k selects stream k (1 = production);
# "old" = SONDE_MD_NO50 (the kernel with one tile in flight).  Specs that remove work produce garbage results.
mkdir -p gpurun_out
for i in 1 2 3; do
  for v in ${VARIANTS:-1 old 2 3}; do
    if [ $v = old ]; then export SONDE_MD_NO50=1; unset SONDE_MD_VARIANT; else unset SONDE_MD_NO50; export SONDE_MD_VARIANT=$v; fi
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_ms_avg']['mix_decimate'], d['config']['frames_ecc_ok'])"
  done
done 2>&1 | tee gpurun_out/ab_variants.txt
