#!/bin/bash
# Collect rocprofv3 counters for the bench's kernels, one --pmc set per pass (never combined
# with trace domains), results under gpurun_out/pmc_<tag>/.   usage: scripts/pmc_passes.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# Counter collection serialises kernel dispatches: a kernel that waits on the device for a kernel of another stream (the notch
# ticket) can then wait for something that is not allowed to start.  Rounds 2-4 ordered the recurrence by HIP events under --pmc,
# which also UNFOLDS the notch from fwd_rows (a separate notch_fix launch): the counters were then not the shipped kernel's.  Round 5:
# ONE lane instead (CHZ_STREAMS=1) -- no ticket is taken with one stream, the notch stays folded inside fwd_rows exactly as shipped,
# and per-dispatch counters do not depend on how many streams the dispatches came from.  Every pass has its own time limit.
export CHZ_STREAMS=1
export BENCH_LEG_BUDGET_SCALE=10      # (bench.py's per-leg watchdog: counter passes serialise the dispatches)
# (rocprofv3 itself crashed once in a FETCH_SIZE pass of round 5 -- SIGSEGV inside its dispatch callback: a pass that fails is repeated once)
run() {
  name=$1; shift
  for attempt in 1 2; do
    rm -rf $R/gpurun_out/pmc_$TAG/$name
    timeout 240 rocprofv3 --pmc "$@" -f csv -d $R/gpurun_out/pmc_$TAG/$name -o $name -- python $R/bench.py --steps 160 --warmup 16 --min-seconds 0.05 --quick $BENCH_ARGS > $R/gpurun_out/pmc_$TAG/$name.log 2>&1 && break
    echo "pmc pass $name: attempt $attempt failed (rc $?)" >> $R/gpurun_out/pmc_$TAG/retries.txt
  done
}
mkdir -p $R/gpurun_out/pmc_$TAG
BENCH_ARGS="$@"
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
