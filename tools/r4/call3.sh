#!/bin/bash
# round 4, GPU call 3: record the CLI outputs for the caller-contract test; small frame-sync workgroups; CU reservation for the tail
set -u
OUT=gpurun_out/r4c
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
python tools/record_cli_outputs.py gpurun_out/cli_ours.npz > $OUT/record.txt 2>&1; tail -3 $OUT/record.txt
SONDE_CLI_RECORDED=gpurun_out/cli_ours.npz timeout 600 python -m pytest tests/test_gpu_cli_recorded.py -q -m gpu 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_cli_recorded.py -q -m gpu 2>&1 | tail -8 > $OUT/recorded_vs_reference_standin.txt; tail -3 $OUT/recorded_vs_reference_standin.txt
SONDE_FS_SMALL=1 timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_ecc_dev.py tests/test_gpu_edges.py -q -m gpu -x > $OUT/pytest_small.log 2>&1
tail -3 $OUT/pytest_small.log
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_00_reference_present.py -q -m gpu -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
Q="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify"
run() { local label="$1"; shift
  env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$label', d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], c['frames_decoded'], c['frames_ecc_ok'], c['frames_decoded_by_host_rs'])"
}
for rep in 1 2; do
  run "two-stream lag1" $Q --lag 1
  run "two-stream lag1 fs-small" SONDE_FS_SMALL=1 $Q --lag 1
  run "one-stream lag1" $Q --one-stream --lag 1
  run "one-stream lag1 fs-small" SONDE_FS_SMALL=1 $Q --one-stream --lag 1
  run "two-stream lag1 fs-small reserve16-top" SONDE_FS_SMALL=1 SONDE_DEC_RESERVE=16 $Q --lag 1
  run "two-stream lag1 fs-small reserve16-stride" SONDE_FS_SMALL=1 SONDE_DEC_RESERVE=16 SONDE_DEC_MASK=stride $Q --lag 1
  run "two-stream lag2 fs-small reserve16-stride" SONDE_FS_SMALL=1 SONDE_DEC_RESERVE=16 SONDE_DEC_MASK=stride $Q --lag 2
  run "two-stream lag2 fs-small reserve32-stride" SONDE_FS_SMALL=1 SONDE_DEC_RESERVE=32 SONDE_DEC_MASK=stride $Q --lag 2
  run "two-stream lag2 fs-small" SONDE_FS_SMALL=1 $Q --lag 2
done | tee $OUT/ab.txt
cd /tmp
SONDE_FS_SMALL=1 SONDE_DEC_RESERVE=16 SONDE_DEC_MASK=stride SONDE_BENCH_NO_REPEAT=1 timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/t" -o t -- python "$ROOT/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify --lag 2 > /dev/null 2>&1
python "$ROOT/tools/timeline.py" "$(find $ROOT/$OUT/t -name '*results.db' | head -1)" 40 3 > "$ROOT/$OUT/timeline_small_reserve16_lag2.txt" 2>&1
rm -rf "$ROOT/$OUT/t"
cd $ROOT
tail -60 $OUT/timeline_small_reserve16_lag2.txt
