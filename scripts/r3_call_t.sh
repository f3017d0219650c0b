#!/bin/bash
# round 3, call t: what the FM demodulator costs per channel -- plain, with the PL-tone detector, with the PLL demodulator
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for m in fm fmtone fmpll; do
  timeout 300 python scripts/scale_kernels_probe.py 0.3 $m 2>&1 | tail -1 | tee -a gpurun_out/r3_fm_probe.jsonl
done
