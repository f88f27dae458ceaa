#!/bin/bash
# round 4, GPU call 5: scanner — register-blocked k_scan_if, prefilter with LDS-resident A fragments; where the small-batch front end spends its time
set -u
OUT=gpurun_out/r4e
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
timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/tw" -o t -- python "$ROOT/bench.py" --config scan_wide --steps 5 --no-cpu-baseline > /dev/null 2>&1
python "$ROOT/tools/rocpd_summary.py" "$(find $ROOT/$OUT/tw -name '*results.db' | head -1)" > "$ROOT/$OUT/scan_wide_rocprofv3.txt" 2>&1
rm -rf "$ROOT/$OUT/t32" "$ROOT/$OUT/tw"
cd $ROOT
head -30 $OUT/scan_alone32_rocprofv3.txt
head -24 $OUT/scan_wide_rocprofv3.txt
timeout 1200 python -m pytest tests/test_gpu_chain.py -q -m gpu -x > $OUT/pytest_chain.log 2>&1
tail -3 $OUT/pytest_chain.log
