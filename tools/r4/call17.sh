#!/bin/bash
# round 4, GPU call 17: scanner call without the host wait behind k_scan_if, ordered behind the channelizer by an event; scan_wide timeline
set -u
OUT=gpurun_out/r4q
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py tests/test_gpu_chain.py -q -m gpu > $OUT/pytest_scan.log 2>&1
tail -4 $OUT/pytest_scan.log
for rep in 1 2 3; do
  timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'], d['config']['detections_last_step'][:4])"
done | tee $OUT/scan_wide.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/t" -o t -- python "$ROOT/bench.py" --config scan_wide --no-cpu-baseline --steps 12 > /dev/null 2>&1
python - "$(find $ROOT/$OUT/t -name '*results.db' | head -1)" > "$ROOT/$OUT/timeline_scan_wide.txt" 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end, name, queue_id, grid_x from kernels order by start"))
# the last three channelizer launches of the timed part
idx = [i for i, r in enumerate(rows) if 'k_channelize' in r[2]]
i0 = idx[-6]; t0 = rows[i0][0]
print("    start       end      dur  queue  kernel")
for r in rows[i0:idx[-3]]:
    print(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:8.1f} {r[3]:6d}  {r[2].split('(')[0][:40]} grid={r[4]}")
PY
rm -rf "$ROOT/$OUT/t"
cd $ROOT
head -60 $OUT/timeline_scan_wide.txt
