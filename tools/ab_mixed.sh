#!/bin/bash
# A/B of the mixed-type engine's stream layout and fetch lag, one process per arm, all in one gpurun call (same box):  bash tools/ab_mixed.sh > gpurun_out/r6a_mixed_lag_ab.txt
for rep in 1 2; do
  echo "== shared launches (default), rep $rep";            LAGS=1,2 python tools/mixed_host.py 2>/dev/null | tail -2
  echo "== SONDE_MIXED_SPLIT=1 (a stream per group), rep $rep"; SONDE_MIXED_SPLIT=1 LAGS=1,2 python tools/mixed_host.py 2>/dev/null | tail -2
done
echo "== shared launches, SONDE_ECC_INLINE=1";  SONDE_ECC_INLINE=1 LAGS=1,2 python tools/mixed_host.py 2>/dev/null | tail -2
echo "== shared launches, SONDE_B_PRIO=0";      SONDE_B_PRIO=0 LAGS=1,2 python tools/mixed_host.py 2>/dev/null | tail -2
