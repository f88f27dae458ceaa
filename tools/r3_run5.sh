#!/bin/bash
# round 3, GPU call 5: base-rate channel restart, the C wideband receiver, the prefilter with block-prefetched fragments, torch-first fixture
set -u
OUT=gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py tests/test_gpu_edges.py tests/test_gpu_chain.py tests/test_gpu_broker.py tests/test_gpu_batch.py -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
grep -E "^FAILED|passed|failed|rc " $OUT/pytest.log | tail -20
timeout 300 python bench.py --config scan_wide --steps 20 --no-cpu-baseline > $OUT/scan_wide.json 2> $OUT/scan_wide.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3e/scan_wide.json").read().strip().splitlines()[-1])
print("scan_wide", d["ms_per_step"], d["config"]["kernels_ms_per_launch"], d["roofline"]["frac"], d["roofline"]["exact_pairs_per_launch"])
PY
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs ) > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3e/bench_default.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "verified", d["config"]["verified_channels"])
print("detect_in_step", d.get("detect_in_step", {}).get("ms_per_step"))
PY
