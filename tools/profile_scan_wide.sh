#!/bin/bash
# refresh of the scan_wide profile files alone (tools/profile_round.sh does all of them): bench line + kernel trace -> gpurun_out/prof/<tag>_bench_scan_wide*
set -u
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python bench.py --config scan_wide 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_scan_wide.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/k_scan_wide" -o k -- python "$ROOT/bench.py" --config scan_wide --steps 5 --no-cpu-baseline > /dev/null 2>&1
cd "$ROOT"
python tools/rocpd_summary.py "$(find "$OUT/k_scan_wide" -name '*results.db' | head -1)" > "$OUT/${TAG}_bench_scan_wide_rocprofv3.txt" 2>&1
rm -rf "$OUT/k_scan_wide"
