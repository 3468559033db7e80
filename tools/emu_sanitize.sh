#!/bin/bash
# The kernel source as lane loops (tests/emu) under AddressSanitizer + UBSan (array bounds, shifts, overflow), CPU only:
# builds build/san/libmp2emu.so and runs the CPU test-suite against it.   usage: tools/emu_sanitize.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
mkdir -p build/san
g++ -O1 -g -std=c++17 -fPIC -mfma -ffp-contract=off -fno-strict-aliasing -Wno-unused-function -Wno-unused-variable -Wno-unknown-pragmas \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -o build/san/libmp2emu.so \
    tests/emu/mp2_emu.cpp odr-audioenc_amd/csrc/mp2_host.cpp -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    TL_EMU_LIB=$PWD/build/san/libmp2emu.so python -m pytest tests -q -m "not gpu" "$@"
