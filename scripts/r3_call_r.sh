#!/bin/bash
# round 3, call r: the |X|^2 image for large banks' noise windows -- parity, then noise_est per channel with and without it
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "noise" -p no:cacheprovider 2>&1 | tail -3
for m in 0 auto; do
  if [ $m = auto ]; then unset CHZ_NOISE_ENERGY; else export CHZ_NOISE_ENERGY=$m; fi
  echo "CHZ_NOISE_ENERGY=$m"
  timeout 300 python scripts/scale_kernels_probe.py 1.5 linear 2>&1 | tail -1 | tee -a gpurun_out/r3_noise_energy.jsonl
done
