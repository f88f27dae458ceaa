#!/bin/bash
set -u
OUT=gpurun_out/r3j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_edges.py tests/test_gpu_polarity.py tests/test_gpu_m10.py tests/test_gpu_dc.py tests/test_gpu_ifiq.py tests/test_gpu_lowsnr.py tests/test_gpu_seam.py tests/test_gpu_softchains.py tests/test_gpu_zz_dfm_raw.py -q -m gpu -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
