#!/bin/bash
# Round-3 A/B of the decimator's instruction mix: radiosonde_auto_rx_amd/exp_md.so = the library built from
#   python tools/gen_md_fast.py --experiments nosincos "nodc nocmul" nofir empty ; make -C radiosonde_auto_rx_amd/csrc OUT=../exp_md.so EXTRA=-DSONDE_MD_EXPERIMENTS
# SONDE_MD_VARIANT: 1 = production loop, 2 = without v_sin / v_cos, 3 = without IQ-DC sum and complex multiply, 4 = without the 7 tap FMAs,
# 5 = loads, LDS parking and stores only.  Variants 2..5 compute garbage: timing only (no verification, no frames).
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs --no-verify"
for i in 1 2 3; do
  for v in 1 2 3 4 5; do
    SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/exp_md.so SONDE_MD_VARIANT=$v SONDE_BENCH_NO_REPEAT=1 timeout 200 $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v  step %.3f ms  decimator %.3f ms' % (d['ms_per_step'], d['config']['kernels']['mix_decimate']['ms_per_step']))"
  done
done 2>&1 | tee gpurun_out/r3_md_variants.txt
