#!/bin/bash
# round 4, GPU call 2: ECC kernel on its own stream, decimators decoupled from the tail (ev_if), fetch lag 1 / 2
set -u
OUT=gpurun_out/r4b
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_ecc_dev.py tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_multirank.py tests/test_gpu_dc.py -q -m gpu -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
Q="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify"
run() { # label, env..., -- args
  local label="$1"; shift
  env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$label', d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], c['frames_decoded'], c['frames_ecc_ok'], c['frames_decoded_by_host_rs'])"
}
for rep in 1 2; do
  run "one-stream lag1" $Q --one-stream --lag 1
  run "two-stream lag1" $Q --lag 1
  run "two-stream lag2" $Q --lag 2
  run "two-stream lag2 A-waits-tail" SONDE_A_WAITS_TAIL=1 $Q --lag 2
  run "two-stream lag2 ecc-inline" SONDE_ECC_INLINE=1 $Q --lag 2
  run "two-stream lag2 prio0" SONDE_B_PRIO=0 $Q --lag 2
done | tee $OUT/ab.txt
cd /tmp
for v in "--lag 2"; do
  tag=$(echo $v | tr -d ' -')
  SONDE_BENCH_NO_REPEAT=1 timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/t_$tag" -o t -- python "$ROOT/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify $v > /dev/null 2>&1
  python "$ROOT/tools/timeline.py" "$(find $ROOT/$OUT/t_$tag -name '*results.db' | head -1)" 40 3 > "$ROOT/$OUT/timeline_$tag.txt" 2>&1
  rm -rf "$ROOT/$OUT/t_$tag"
done
cd $ROOT
cat $OUT/timeline_lag2.txt | tail -70
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --lag 2 2>$OUT/bench_full.err | tail -1 > $OUT/bench_full.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4b/bench_full.json"))
c=d["config"]
print("full", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac"], c["timed_seconds"], c["frames_decoded"], c["frames_ecc_ok"], c["frames_repaired"], c["frames_decoded_by_host_rs"], c["verified_channels"], c.get("verify_mismatch_channels"))
print("ab", d.get("host_ecc_ab"))
print("kern", c["kernels"])
PY
