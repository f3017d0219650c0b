#!/bin/bash
# round 3, call aa: instruction counters of the chain's kernels at 0.3 M channels (linear mode)
cd "$(dirname "$0")/.." || exit 1
R=$PWD
mkdir -p gpurun_out/pmc_demod
cd /tmp && export TMPDIR=/tmp
export CHZ_NOTCH_ORDER=event
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv -d $R/gpurun_out/pmc_demod/sq1 -o sq1 -- python $R/scripts/scale_kernels_probe.py 0.3 linear > $R/gpurun_out/pmc_demod/sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -f csv -d $R/gpurun_out/pmc_demod/sq2 -o sq2 -- python $R/scripts/scale_kernels_probe.py 0.3 linear > $R/gpurun_out/pmc_demod/sq2.log 2>&1
cd $R
for d in sq1 sq2; do python scripts/rocprof_summary.py gpurun_out/pmc_demod/$d; done > gpurun_out/r3_pmc_demod.txt 2>&1
find gpurun_out/pmc_demod -type f ! -name "*.log" -size +200k -delete
cat gpurun_out/r3_pmc_demod.txt | head -120
