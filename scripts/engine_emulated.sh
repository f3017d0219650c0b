#!/bin/bash
# Every device parity test that the CPU can carry, against the engine's HOST code built for the CPU (tests/test_engine_emulated.py
# runs the fast ones as part of `-m "not gpu"`; this runs the long demodulator scenarios and the full-size masters too: ~12 minutes).
cd "$(dirname "$0")/.." || exit 1
python -m pytest tests/test_engine_emulated.py -q -k refuses || exit 1        # builds tests/hipemu/libchz_hip_emu.so
CHZ_LIB=$PWD/tests/hipemu/libchz_hip_emu.so CHZ_ALLOW_EMULATED_ENGINE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_golden.py \
  -m gpu -q --timeout 600 -p no:cacheprovider \
  -k "not (soak or rccl or comm_rendezvous or graph or runs_out or noise_and_conversion)" "$@"
