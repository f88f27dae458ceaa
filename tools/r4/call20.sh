#!/bin/bash
# round 4, GPU call 21: the fold inside k_scan_if (as it loads its tile), edge corrections tabulated
set -u
OUT=gpurun_out/r4u
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py -q -m gpu > $OUT/pytest_scan.log 2>&1
tail -3 $OUT/pytest_scan.log
for rep in 1 2; do
  for n in 32 512; do
    echo "== one pass, $n channels: $(timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1)"
  done
done | tee $OUT/scan_alone.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/t512" -o t -- python "$ROOT/tools/scan_alone.py" 512 > /dev/null 2>&1
python "$ROOT/tools/rocpd_summary.py" "$(find $ROOT/$OUT/t512 -name '*results.db' | head -1)" > "$ROOT/$OUT/scan_alone512_rocprofv3.txt" 2>&1
rm -rf "$ROOT/$OUT/t512"
cd $ROOT
head -14 $OUT/scan_alone512_rocprofv3.txt
timeout 900 python bench.py --no-cpu-baseline --no-configs > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4u/bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('step_frac'))
x=d.get("detect_in_step"); print("detect", json.dumps(x)[:600])
PY
