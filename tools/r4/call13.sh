#!/bin/bash
# round 4, GPU call 13: full GPU suite (no -x) + SQ counters of the FSK modem kernels in fsk_mixed
set -u
OUT=gpurun_out/r4m
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -8 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --config fsk_mixed --steps 10 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/fsk_mixed.json
python -c "
import json; d=json.load(open('$OUT/fsk_mixed.json')); print('fsk_mixed', d['ms_per_step'], d['config'].get('kernel_ms_per_launch'))"
cd /tmp
B="python $ROOT/bench.py --config fsk_mixed --steps 3 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$ROOT/$OUT/a" -o a -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d "$ROOT/$OUT/b" -o b -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES SQ_INSTS_FLAT -d "$ROOT/$OUT/c" -o c -- $B > /dev/null 2>&1
cd $ROOT
A=$(find $OUT/a -name '*results.db' | head -1); Bd=$(find $OUT/b -name '*results.db' | head -1); Cd=$(find $OUT/c -name '*results.db' | head -1)
python tools/rocpd_summary.py "$A" "$A" "$Bd" "$Cd" > $OUT/fsk_sq.txt 2>&1
rm -rf $OUT/a $OUT/b $OUT/c
grep -i "fsk" $OUT/fsk_sq.txt | head -80
