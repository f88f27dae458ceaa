#!/bin/bash
# A/B: frame fetch lag / second stream after the fetch copies moved to their own stream
set -u
OUT=gpurun_out/r3h
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify"
for rep in 1 2 3; do
  for v in "--lag 1" "--lag 0" "--lag 1 --two-streams" "--lag 2"; do
    timeout 300 $B $v 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'], d['config'].get('kernel_ms_per_launch'))"
  done
done | tee $OUT/ab_lag.txt
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -q -m gpu -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
