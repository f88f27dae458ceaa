#!/bin/bash
set -u
OUT=gpurun_out/r3k
mkdir -p $OUT
export TMPDIR=/tmp
SONDE_WF_PROF=1 SONDE_BENCH_NO_REPEAT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs --no-verify 2>&1 | grep -E "prof" | cut -c1-400
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['config']['kernels']; print(d['ms_per_step'], d['roofline']['frac'], {n: k[n]['ms_per_step'] for n in k})"; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_scan.py tests/test_gpu_edges.py -q -m gpu -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
