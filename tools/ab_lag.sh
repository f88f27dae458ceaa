#!/bin/bash
# A/B in one session: frame fetch in step (default), one step behind (--lag 1), and with the IF-rate kernels on a second stream (--lag 1 --two-streams)
for i in 1 2; do
  for v in "" "--lag 1" "--lag 1 --two-streams"; do
    python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-extras $v 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v]', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['config']['kernels'].items()}, d['roofline']['frac'], d['config']['frames_decoded'])"
  done
done
