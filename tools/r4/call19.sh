#!/bin/bash
# round 4, GPU call 19: scanner front end in one pass (raw mix + block sums, the fold at the IF rate) against the two-pass form
set -u
OUT=gpurun_out/r4s
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py tests/test_gpu_chain.py -q -m gpu > $OUT/pytest_scan.log 2>&1
tail -6 $OUT/pytest_scan.log
for rep in 1 2; do
  for n in 32 128 512; do
    echo "== one pass, $n channels: $(timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1)"
    echo "== two passes, $n channels: $(SONDE_SCAN_TWO_PASS=1 timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1)"
  done
done | tee $OUT/scan_alone.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/t512" -o t -- python "$ROOT/tools/scan_alone.py" 512 > /dev/null 2>&1
python "$ROOT/tools/rocpd_summary.py" "$(find $ROOT/$OUT/t512 -name '*results.db' | head -1)" > "$ROOT/$OUT/scan_alone512_rocprofv3.txt" 2>&1
rm -rf "$ROOT/$OUT/t512"
cd $ROOT
head -30 $OUT/scan_alone512_rocprofv3.txt
