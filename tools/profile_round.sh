#!/bin/bash
# Round profile refresh, run on the GPU box from the repo root (gpurun):  bash tools/profile_round.sh [tag]   (default tag r6)
#   demod     bench line as the driver runs it; rocprofv3 --kernel-trace --stats of the same command; FETCH_SIZE / WRITE_SIZE and SQ
#             counter passes (separate, kernel trace + PMC only) -> traffic json of the dominant kernel
#   scan_wide, fsk_mixed, mixed_2400k   bench line + kernel trace each; fsk: phase counters, SQ counters of k_fsk_wave (tools/fsk_multi.py)
# everything lands in gpurun_out/prof/ as <tag>_*; copy what is to be judged into profiles/.
set -u
TAG=${1:-r6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
export SONDE_BENCH_VERBOSE=1      # the full objects (bench.py prints the compact line by default)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/${TAG}_bench.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/t" -o t -- python "$ROOT/bench.py" --steps 200 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f" -o f -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/w" -o w -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    -d "$OUT/a" -o a -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    -d "$OUT/b" -o b -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
cd "$ROOT"
db() { find "$OUT/$1" -name '*results.db' | head -1; }
python tools/rocpd_summary.py "$(db t)" "$(db f)" "$(db w)" > "$OUT/${TAG}_bench_rocprofv3.txt" 2>&1
python tools/rocpd_summary.py "$(db a)" "$(db a)" "$(db b)" > "$OUT/${TAG}_mix_decimate_sq.txt" 2>&1
python tools/traffic_json.py "$(db f)" "$(db w)" k_mix_decimate50 1572864 4915200000 > "$OUT/${TAG}_mix_decimate_traffic.json" 2>/dev/null
# phase profile of the two sync kernels (cycles of workgroup 0 / channel 0 per phase)
SONDE_WF_PROF=1 SONDE_BENCH_NO_REPEAT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-configs --no-verify 2>&1 >/dev/null | grep " prof " > "$OUT/${TAG}_sync_phases.txt"
for cfg in scan_wide fsk_mixed mixed_2400k; do
  timeout 300 python bench.py --config $cfg 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_${cfg}.json"
  if [ $cfg = fsk_mixed ]; then      # the phase counters slow the kernel down: a run of their own, not the bench line's
    SONDE_FSK_PROF=1 timeout 300 python bench.py --config $cfg --steps 5 --no-cpu-baseline 2>&1 >/dev/null | grep "fsk prof" > "$OUT/${TAG}_fsk_phases.txt"
    timeout 300 python bench.py --config $cfg --channels 4096 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_${cfg}_4096.json"
  fi
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/k_$cfg" -o k -- python "$ROOT/bench.py" --config $cfg --steps 5 --no-cpu-baseline > /dev/null 2>&1
  cd "$ROOT"
  python tools/rocpd_summary.py "$(db k_$cfg)" > "$OUT/${TAG}_bench_${cfg}_rocprofv3.txt" 2>&1
done
# HBM traffic of the three modem launches of the fsk_mixed line (FETCH_SIZE / WRITE_SIZE, separate passes) -> one json per configuration
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/ff" -o f -- python "$ROOT/bench.py" --config fsk_mixed --steps 5 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/fw" -o w -- python "$ROOT/bench.py" --config fsk_mixed --steps 5 --no-cpu-baseline > /dev/null 2>&1
# the scanner's correlation kernel is bound by the matrix cores, not by memory: its MFMA instruction count beside the kernel trace
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS -d "$OUT/sm" -o m -- python "$ROOT/bench.py" --config scan_wide --steps 5 --no-cpu-baseline > /dev/null 2>&1
cd "$ROOT"
python tools/traffic_json.py "$(db ff)" "$(db fw)" "k_fsk_wave<2, 7" 65664 65664000 > "$OUT/${TAG}_fsk_rs41_traffic.json" 2>/dev/null
python tools/traffic_json.py "$(db ff)" "$(db fw)" "k_fsk_wave<2, 8" 87296 68200000 > "$OUT/${TAG}_fsk_dfm_traffic.json" 2>/dev/null
python tools/traffic_json.py "$(db ff)" "$(db fw)" "k_fsk_wave<2, 6" 87296 65581120 > "$OUT/${TAG}_fsk_m10_traffic.json" 2>/dev/null
python tools/rocpd_summary.py "$(db sm)" "$(db sm)" "$(db sm)" > "$OUT/${TAG}_scan_pre_mfma.txt" 2>&1
# SQ counters of the modem kernel (three configurations of 342 channels submitted together — the 1024-channel mix — and 4 x that), three passes
for n in 342 1366; do
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT/fa$n" -o a -- python "$ROOT/tools/fsk_multi.py" rs41,m10,dfm $n > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d "$OUT/fb$n" -o b -- python "$ROOT/tools/fsk_multi.py" rs41,m10,dfm $n > /dev/null 2>&1
  cd "$ROOT"
  { python tools/fsk_multi.py rs41,m10,dfm $n 2>/dev/null | tail -n 1; python tools/rocpd_summary.py "$(db fa$n)" "$(db fa$n)" "$(db fb$n)"; } > "$OUT/${TAG}_fsk_wave_sq_${n}x3.txt" 2>&1
done
# databases are large: keep only the text
rm -rf "$OUT/t" "$OUT/f" "$OUT/w" "$OUT/a" "$OUT/b" "$OUT"/k_* "$OUT"/fa* "$OUT"/fb* "$OUT/ff" "$OUT/fw" "$OUT/sm"
ls -la "$OUT"
