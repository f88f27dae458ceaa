#!/bin/bash
# bench line + rocprofv3 kernel trace of the same command in one session (kernel durations next to the HIP-event figures)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/qp; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench.json"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/t" -o t -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_rocprof.json"
cd "$ROOT"
T=$(find "$OUT/t" -name '*results.db' | head -1)
python tools/rocpd_summary.py "$T" "$T" "$T" 2>/dev/null | head -40 > "$OUT/summary.txt"
rm -rf "$OUT/t"
python - <<'PY'
import json
for f in ("bench.json", "bench_rocprof.json"):
    d = json.load(open("gpurun_out/qp/" + f)); print(f, d["ms_per_step"], d["roofline"]["frac"], d["config"]["kernel_ms_avg"])
PY
head -12 "$OUT/summary.txt"; grep "k_mix_decimate50" "$OUT/summary.txt" | head -5
