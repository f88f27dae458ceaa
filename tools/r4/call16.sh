#!/bin/bash
# round 4, GPU call 16: prefilter score loop without branches, DPP reductions
set -u
OUT=gpurun_out/r4p
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_chan.py -q -m gpu > $OUT/pytest_scan.log 2>&1
tail -4 $OUT/pytest_scan.log
for rep in 1 2; do
for v in cur prev; do
  if [ $v = cur ]; then unset SONDE_HIP_LIB; else export SONDE_HIP_LIB=$ROOT/radiosonde_auto_rx_amd/exp_$v.so; fi
  for n in 32 512; do
    echo "== $v $n channels: $(timeout 300 python tools/scan_alone.py $n 2>/dev/null | tail -1)"
  done
  timeout 600 python bench.py --config scan_wide --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan_wide $v', d['ms_per_step'], d['roofline']['frac'], d['config']['kernels_ms_per_launch'], d['config']['detections_last_step'][:4])"
done
done | tee $OUT/ab.txt
cd /tmp
for v in cur prev; do
  if [ $v = cur ]; then unset SONDE_HIP_LIB; else export SONDE_HIP_LIB=$ROOT/radiosonde_auto_rx_amd/exp_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d "$ROOT/$OUT/p_$v" -o p -- python "$ROOT/tools/scan_alone.py" 512 > /dev/null 2>&1
  python "$ROOT/tools/rocpd_summary.py" "$(find $ROOT/$OUT/p_$v -name '*results.db' | head -1)" "$(find $ROOT/$OUT/p_$v -name '*results.db' | head -1)" 2>&1 | grep -E "k_scan_pre" > "$ROOT/$OUT/pmc_$v.txt"
  rm -rf "$ROOT/$OUT/p_$v"
  echo "== pmc $v"; cat "$ROOT/$OUT/pmc_$v.txt"
done
