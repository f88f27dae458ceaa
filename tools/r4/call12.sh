#!/bin/bash
# round 4, GPU call 12: the scanner's base-rate front end in one pass (k_mix_decimate50w)
set -u
OUT=gpurun_out/r4l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_scan.py -q -m gpu -x > $OUT/pytest_scan.log 2>&1
tail -6 $OUT/pytest_scan.log
for n in 32 128 512; do
  echo "== one pass, $n channels"; timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1
  echo "== two passes, $n channels"; SONDE_SCAN_TWO_PASS=1 timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1
done | tee $OUT/scan_alone.txt
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_chan.py -q -m gpu -x > $OUT/pytest_chain.log 2>&1
tail -3 $OUT/pytest_chain.log
