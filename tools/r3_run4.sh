#!/bin/bash
# round 3, GPU call 4: why did torch.cuda fail to initialise behind the scanner / broker tests?  then the whole GPU suite, then SQ counters of k_scan_pre
set -u
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
t() { name=$1; shift; timeout 600 python -m pytest "$@" -q -m gpu -p no:cacheprovider > $OUT/$name.log 2>&1; echo "$name rc $? : $(tail -1 $OUT/$name.log)"; }
t chan_alone tests/test_gpu_chan.py -k channelizer_matches
t scan_then_chan tests/test_gpu_scan.py tests/test_gpu_chan.py -k "prefilter or channelizer_matches"
t stall_then_chan tests/test_gpu_broker.py tests/test_gpu_chan.py -k "paused or channelizer_matches"
t shims_then_chan tests/test_gpu_broker.py tests/test_gpu_chan.py -k "decoder_shims or channelizer_matches"
t fskbroker_then_chan tests/test_gpu_broker.py tests/test_gpu_chan.py -k "(not decoder_shims and not paused) or channelizer_matches"
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/full.log 2>&1; echo "full rc $? : $(tail -1 $OUT/full.log)"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 \
   -d $GRAFT_REPO_ROOT/$OUT/pmc_a -o a -- python $GRAFT_REPO_ROOT/bench.py --config scan_wide --steps 3 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY \
   -d $GRAFT_REPO_ROOT/$OUT/pmc_b -o b -- python $GRAFT_REPO_ROOT/bench.py --config scan_wide --steps 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
db() { find "$OUT/$1" -name '*results.db' | head -1; }
python tools/rocpd_summary.py "$(db pmc_a)" "$(db pmc_a)" "$(db pmc_b)" > $OUT/scan_pre_sq.txt 2>&1
grep -E "k_scan_pre|k_scan_corr" $OUT/scan_pre_sq.txt | head -40
rm -rf $OUT/pmc_a $OUT/pmc_b
