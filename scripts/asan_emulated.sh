#!/bin/bash
# The UNMODIFIED kernel sources on the CPU emulator (tests/hipemu) under AddressSanitizer: out-of-bounds reads and writes of
# global buffers by any kernel show up here without a GPU.  Builds an instrumented emulator library into /tmp, swaps it in for
# the run, restores the normal one.
set -e
cd "$(dirname "$0")/.."
g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address -fno-omit-frame-pointer -I tests/hipemu -I ka9q-radio_amd/csrc \
    tests/hipemu/emu_kernels.cpp -o /tmp/libchz_emu_asan.so
[ -f tests/hipemu/libchz_emu.so ] && cp tests/hipemu/libchz_emu.so /tmp/libchz_emu_backup.so
cp /tmp/libchz_emu_asan.so tests/hipemu/libchz_emu.so
trap '[ -f /tmp/libchz_emu_backup.so ] && cp /tmp/libchz_emu_backup.so tests/hipemu/libchz_emu.so' EXIT
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    python -m pytest tests/test_kernels_emulated.py -q -p no:cacheprovider "$@"
