#!/bin/bash
set -u
OUT=gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
SONDE_FSK_PROF=1 timeout 300 python bench.py --config fsk_mixed --steps 10 --no-cpu-baseline > $OUT/fsk.json 2> $OUT/fsk.err
grep "fsk prof" $OUT/fsk.err
python -c "
import json; d=json.loads(open('$OUT/fsk.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms_per_launch'])"
timeout 600 python -m pytest "tests/test_gpu_scan.py::test_scan_windows_and_lines_match_reference" tests/test_gpu_fsk.py -q -m gpu > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
