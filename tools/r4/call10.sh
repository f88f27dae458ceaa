#!/bin/bash
set -u
OUT=gpurun_out/r4i
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py -q -m gpu -x > $OUT/pytest_scan.log 2>&1
tail -3 $OUT/pytest_scan.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/t32" -o t -- python "$ROOT/tools/scan_alone.py" 32 > /dev/null 2>&1
python "$ROOT/tools/timeline.py" "$(find $ROOT/$OUT/t32 -name '*results.db' | head -1)" --tail 24 > "$ROOT/$OUT/scan_alone32_timeline.txt" 2>&1
rm -rf "$ROOT/$OUT/t32"
cd $ROOT
cat $OUT/scan_alone32_timeline.txt
timeout 300 python tools/scan_alone.py 512 2>/dev/null | tail -1
timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'])"
