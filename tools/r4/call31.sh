#!/bin/bash
# round 4, GPU call 31: FSK consumer: the next piece's samples and the piece's timing phasors requested ahead
set -u
OUT=gpurun_out/r4zb
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fsk.py -q -m gpu > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for i in 1 2 3; do
python bench.py --config fsk_mixed --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fsk_mixed', d['ms_per_step'], d['value'], d['config'].get('kernel_ms_per_launch'), d['config']['verified_channels'])"
done | tee $OUT/fsk_mixed.txt
