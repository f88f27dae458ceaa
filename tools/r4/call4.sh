#!/bin/bash
# round 4, GPU call 4: scanner — generated double-phase decimator for the base-rate front end, full Toeplitz fragments in the prefilter
set -u
OUT=gpurun_out/r4d
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py -q -m gpu -x > $OUT/pytest_scan.log 2>&1
tail -4 $OUT/pytest_scan.log
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_iqdec.py tests/test_gpu_u8.py -q -m gpu -x > $OUT/pytest_chain.log 2>&1
tail -4 $OUT/pytest_chain.log
for n in 32 512; do
  echo "== generated kernel, $n channels"; timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -2
  echo "== generic kernel (SONDE_MD_NO50=1), $n channels"; SONDE_MD_NO50=1 timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -2
done | tee $OUT/scan_alone.txt
for rep in 1 2; do
  timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'], d['config']['detections_last_step'][:4])"
done | tee $OUT/scan_wide.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench_full.err | tail -1 > $OUT/bench_full.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4d/bench_full.json"))
c=d["config"]
print("full", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac"], c["timed_seconds"], c["verified_channels"])
det=d.get("detect_in_step",{})
print("det", det.get("ms_per_step"), det.get("rs41_detections_per_scanned_channel"), {k:(det[k]["ms_per_step"], det[k]["rs41_detections_per_scanned_channel"]) for k in ("duty_1_4","duty_1_1") if k in det})
for k in ("scan_wide","fsk_mixed"):
    print(k, d[k].get("ms_per_step"), d[k].get("error"), d[k].get("config",{}).get("kernel_ms_per_launch"), d[k].get("config",{}).get("verified_channels"))
PY
