#!/bin/bash
# round 4, GPU call 1: device RS decoder parity; bench with the error mix, device vs host ECC; stream variants; kernel timeline of the two-stream form
set -u
OUT=gpurun_out/r4a
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_ecc_dev.py tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_edges.py -q -m gpu -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs"
timeout 600 $B 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4a/bench_default.json"))
c=d["config"]
print("default", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac"], c["timed_seconds"], c["frames_decoded"], c["frames_ecc_ok"], c["frames_repaired"], c["frames_decoded_by_host_rs"], c["verified_channels"], c.get("verify_mismatch_channels"))
print("ab", d.get("host_ecc_ab"))
print("kern", c["kernels"])
print("mix", c["error_mix"]["rs41_ecc_value_by_capture"])
print("det", {k:(v if not isinstance(v,dict) else v.get("ms_per_step")) for k,v in d.get("detect_in_step",{}).items() if k in ("ms_per_step","duty_1_4","duty_1_1")})
PY
Q="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify"
for rep in 1 2; do
  for v in "--lag 1" "--lag 1 --two-streams"; do
    for prio in 1 0; do
      SONDE_B_PRIO=$prio timeout 300 $Q $v 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v prio=$prio', d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'])"
    done
  done
done | tee $OUT/ab_streams.txt
cd /tmp
for v in "--lag 1" "--lag 1 --two-streams"; do
  tag=$(echo $v | tr -d ' -')
  SONDE_BENCH_NO_REPEAT=1 timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/t_$tag" -o t -- python "$ROOT/bench.py" --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-configs --no-verify $v > /dev/null 2>&1
  python "$ROOT/tools/timeline.py" "$(find $ROOT/$OUT/t_$tag -name '*results.db' | head -1)" 40 3 > "$ROOT/$OUT/timeline_$tag.txt" 2>&1
  rm -rf "$ROOT/$OUT/t_$tag"
done
cd $ROOT
tail -25 $OUT/timeline_lag1twostreams.txt
