#!/bin/bash
# round 3, GPU call 6: scanner with N_DFT 16384 / 32768, --br / --chk3, everything around the scanner again
set -u
OUT=gpurun_out/r3f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_edges.py tests/test_gpu_m10.py tests/test_gpu_chain.py -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
grep -E "^FAILED|passed|failed|rc |Error" $OUT/pytest.log | tail -30
