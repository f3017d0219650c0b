#!/bin/bash
# Every device parity test that the CPU can carry, against the engine's HOST code built for the CPU (tests/test_engine_emulated.py
# runs the fast ones as part of `-m "not gpu"`; this runs the long demodulator scenarios and the full-size masters too: ~12 minutes).
cd "$(dirname "$0")/.." || exit 1
python -m pytest tests/test_engine_emulated.py -q -k refuses || exit 1        # builds tests/hipemu/libchz_hip_emu.so
LIBEMU=$PWD/tests/hipemu/libchz_hip_emu.so
if [ "$TSAN" = 1 ]; then     # the engine's threading under ThreadSanitizer (tests/c/engine_driver.c; ~6 min)
  CHZ_TEST_TSAN_ENGINE=1 python -m pytest tests/test_engine_emulated.py -q -k "thread_sanitizer"; exit $?
fi
if [ "$ASAN" = 1 ]; then     # the engine's host code AND the kernels under AddressSanitizer (61 tests clean in round 2; leave the long ones out with -k)
  g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address -fno-omit-frame-pointer -DHIPEMU -DHIPEMU_HOST -I tests/hipemu -I ka9q-radio_amd/csrc \
      -x c++ ka9q-radio_amd/csrc/chz_engine.hip -o /tmp/libchz_hip_emu_asan.so -lpthread -ldl || exit 1
  LIBEMU=/tmp/libchz_hip_emu_asan.so
  export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
fi
CHZ_LIB=$LIBEMU CHZ_ALLOW_EMULATED_ENGINE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_golden.py \
  -m gpu -q --timeout 600 -p no:cacheprovider \
  -k "not (soak or rccl or comm_rendezvous or graph or runs_out or noise_and_conversion or random_operations_on_the_device)" "$@"
