#!/bin/bash
# round 4, GPU call 11: header search and frame sync in decimator-slot-sized workgroups
set -u
OUT=gpurun_out/r4k
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
SONDE_SMALL_TAIL=1 timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_ecc_dev.py -q -m gpu -x > $OUT/pytest_small.log 2>&1
tail -4 $OUT/pytest_small.log
timeout 900 python -m pytest tests/test_gpu_chain.py -q -m gpu -x -k "lms6_that_turns or wideband_receiver_finds" > $OUT/pytest_chain.log 2>&1
tail -3 $OUT/pytest_chain.log
Q="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify"
run() { local label="$1"; shift
  env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$label', d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], c['frames_decoded'], c['frames_ecc_ok'], c['frames_decoded_by_host_rs'])"
}
for rep in 1 2; do
  run "two-stream lag1 small-tail" $Q --lag 1
  run "two-stream lag1 big-tail" SONDE_SMALL_TAIL=0 $Q --lag 1
  run "two-stream lag2 small-tail" $Q --lag 2
  run "one-stream lag1 big-tail" $Q --one-stream --lag 1
  run "one-stream lag1 small-tail" SONDE_SMALL_TAIL=1 $Q --one-stream --lag 1
done | tee $OUT/ab.txt
cd /tmp
SONDE_BENCH_NO_REPEAT=1 timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/t" -o t -- python "$ROOT/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify --lag 2 > /dev/null 2>&1
python "$ROOT/tools/timeline.py" "$(find $ROOT/$OUT/t -name '*results.db' | head -1)" 40 3 > "$ROOT/$OUT/timeline_small_tail_lag2.txt" 2>&1
rm -rf "$ROOT/$OUT/t"
cd $ROOT
tail -45 $OUT/timeline_small_tail_lag2.txt
