#!/bin/bash
set -u
OUT=gpurun_out/r4g
mkdir -p $OUT
export TMPDIR=/tmp
SONDE_SP_PROF=1 timeout 300 python tools/scan_alone.py 512 2>&1 | grep "scan_pre prof\|call 5" | tee $OUT/sp_prof.txt
SONDE_SP_PROF=1 timeout 300 python tools/scan_alone.py 32 2>&1 | grep "scan_pre prof\|call 5" | tee -a $OUT/sp_prof.txt
