#!/bin/bash
# round 4, GPU call 6: prefilter with a fixed LDS budget; kernel profile of the small-batch scanner
set -u
OUT=gpurun_out/r4f
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py -q -m gpu -x > $OUT/pytest_scan.log 2>&1
tail -4 $OUT/pytest_scan.log
for n in 32 512; do
  echo "== $n channels"; timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -2
done | tee $OUT/scan_alone.txt
for rep in 1 2; do
  timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'], d['config']['detections_last_step'][:4])"
done | tee $OUT/scan_wide.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/t32" -o t -- python "$ROOT/tools/scan_alone.py" 32 > /dev/null 2>&1
python "$ROOT/tools/rocpd_summary.py" "$(find $ROOT/$OUT/t32 -name '*results.db' | head -1)" > "$ROOT/$OUT/scan_alone32_rocprofv3.txt" 2>&1
rm -rf "$ROOT/$OUT/t32"
cd $ROOT
head -40 $OUT/scan_alone32_rocprofv3.txt
