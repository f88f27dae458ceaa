#!/bin/bash
set -u
export TMPDIR=/tmp
for C in 342 1024; do
echo "== C=$C old"; SONDE_HIP_LIB=$PWD/radiosonde_auto_rx_amd/exp_old.so python tools/fsk_kernel_ab.py $C 2>&1 | tail -1
for a in 0 1 2 3; do echo "== C=$C new ahead=$a"; SONDE_FSK_AHEAD=$a python tools/fsk_kernel_ab.py $C 2>&1 | tail -1; done
done
