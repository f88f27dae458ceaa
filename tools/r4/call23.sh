#!/bin/bash
# round 4, GPU call 23: FSK channels whose pipeline gave up are repeated frame by frame; ragged calls through both scanner front-end forms
set -u
OUT=gpurun_out/r4w
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fsk.py tests/test_gpu_scan.py -q -m gpu > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for i in 1 2; do
python bench.py --config fsk_mixed --steps 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fsk_mixed', d['ms_per_step'], d['config'].get('kernel_ms_per_launch'), d['config']['verified_channels'])"
done
python bench.py --config fsk_mixed --channels 4096 --steps 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/fsk_mixed_4096.json
python -c "
import json; d=json.load(open('$OUT/fsk_mixed_4096.json')); print('fsk_mixed 4096', d['ms_per_step'], d['value'], d['config'].get('kernel_ms_per_launch'), d['config']['verified_channels'], d['config']['checked_channels'])"
