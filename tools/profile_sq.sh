#!/bin/bash
# SQ issue / stall counters of the bench kernels (one rocprofv3 pass, kernel trace + PMC only), run on the GPU box:
#   bash tools/profile_sq.sh  ->  gpurun_out/prof_sq/summary.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_sq
rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    -d "$OUT/a" -o a -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    -d "$OUT/b" -o b -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/b.log" 2>&1
cd "$ROOT"
A=$(find "$OUT/a" -name '*results.db' | head -1); B=$(find "$OUT/b" -name '*results.db' | head -1)
python tools/rocpd_summary.py "$A" "$A" "$B" > "$OUT/summary.txt" 2>&1
tail -3 "$OUT/a.log" "$OUT/b.log" >> "$OUT/summary.txt"
rm -rf "$OUT/a" "$OUT/b"
