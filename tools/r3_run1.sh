#!/bin/bash
# round 3, GPU call 1: the new parity tests at the timed geometry, the multi-rank readiness test, the lag / two-stream A/B (kept: profiles/r3_ab_lag.txt)
# and one default bench line with every object
set -u
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multirank.py tests/test_gpu_polarity.py "tests/test_gpu_m10.py::test_m10_engine_many_channels" \
    "tests/test_gpu_parity.py::test_dfm_frames_match_golden_and_oracle" -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
: > $OUT/ab_lag.txt
for i in 1 2; do
  for v in "--lag 0" "--lag 1" "--lag 1 --two-streams"; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify $v 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', 'ms_per_step', d['ms_per_step'], 'timed_steps', d['config']['timed_steps'], {k: v['ms_per_step'] for k, v in d['config']['kernels'].items()}, 'md_avg_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'frames', d['config']['frames_decoded'])" >> $OUT/ab_lag.txt
  done
done
cat $OUT/ab_lag.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 6000 $OUT/bench_default.json; tail -5 $OUT/bench_default.err
