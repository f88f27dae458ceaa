#!/bin/bash
# round 3, GPU call 3: scanner prefilter (MFMA) — parity tests, the receivers that sit on the scanner, scan_wide bench, default bench line
set -u
OUT=gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_broker.py tests/test_gpu_chain.py tests/test_gpu_edges.py tests/test_gpu_chan.py "tests/test_gpu_parity.py::test_dfm_frames_match_golden_and_oracle" tests/test_gpu_audio.py -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 300 python bench.py --config scan_wide --steps 20 > $OUT/scan_wide.json 2> $OUT/scan_wide.err
tail -c 3000 $OUT/scan_wide.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c/bench_default.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["config"]["kernels"], d["roofline"]["frac"], "verified", d["config"]["verified_channels"])
print("detect_in_step", d.get("detect_in_step"))
print("scan_wide", d["scan_wide"].get("ms_per_step"), d["scan_wide"].get("config", {}).get("kernels_ms_per_launch"), d["scan_wide"].get("roofline"))
PY
tail -4 $OUT/bench_default.err
