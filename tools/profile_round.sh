#!/bin/bash
# Round profile refresh, run on the GPU box from the repo root (gpurun):  bash tools/profile_round.sh
#   1. plain bench line                                   -> gpurun_out/prof/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same bench  -> gpurun_out/prof/t_results.db (+ bench line under the profiler)
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only (no sys/hip/hsa tracing with counters)
#   4. text summary of all three                           -> gpurun_out/prof/summary.txt
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > "$OUT/bench.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/t" -o t -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_under_rocprof.json"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f" -o f -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/w" -o w -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd "$ROOT"
T=$(find "$OUT/t" -name '*results.db' | head -1); F=$(find "$OUT/f" -name '*results.db' | head -1); W=$(find "$OUT/w" -name '*results.db' | head -1)
python tools/rocpd_summary.py "$T" "$F" "$W" > "$OUT/summary.txt" 2>&1
# databases are large: keep only the text
rm -rf "$OUT/t" "$OUT/f" "$OUT/w"
ls -la "$OUT"
