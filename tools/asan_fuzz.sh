#!/bin/bash
# AddressSanitizer + UBSan build of the host-only half of the library (block codes, framers, telemetry and the bit-rate tiers: everything
# that parses bytes received over RF) and of the ten decoder front ends, then tests/fuzz/fuzz_family.py against the compiled reference with them.
# No GPU and no HIP runtime involved: the soft-bit / hex-line / hard-bit input forms never create an engine.
#     tools/asan_fuzz.sh [seed] [iterations]
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/asan
mkdir -p $OUT
SRC="sonde_ecc sonde_frame sonde_softin sonde_design sonde_rs41_fields sonde_dfm_fields sonde_m10_fields sonde_m20_fields sonde_lms6_fields sonde_meisei_fields sonde_imet54_fields sonde_mrz_fields sonde_mts01_fields sonde_rs92_fields sonde_gpsnav"
FLAGS="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
for f in $SRC; do g++ -std=c++17 $FLAGS -c radiosonde_auto_rx_amd/csrc/$f.cpp -o $OUT/$f.o; done
g++ -shared -fsanitize=address,undefined -o $OUT/libsonde_hip.so $OUT/*.o
for c in rs41mod dfm09mod m10mod m20mod lms6Xmod meisei100mod imet54mod mp3h1mod mts01mod rs92mod; do
    gcc $FLAGS -Iinclude -Ihost -o $OUT/$c host/$c.c -L$OUT -lsonde_hip -Wl,-rpath,'$ORIGIN' -Wl,--unresolved-symbols=ignore-all -lm
done
FUZZ_BIN_DIR=$OUT python tests/fuzz/fuzz_family.py "${1:-1}" "${2:-270}"
