#!/bin/bash
# scripts/gpu/call.sh <name> -- one parametrised GPU-box session (replaces the per-call scripts of earlier rounds).
# Usage on the dev box:   gpurun --timeout 1500 -- 'bash scripts/gpu/call.sh scale'
# Everything a call writes goes to gpurun_out/<name>/ ; summaries worth keeping are copied into profiles/ by hand.
set -u
name=${1:-tests}
out=gpurun_out/$name
mkdir -p "$out"
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5"
case "$name" in
  scale)     # round 4, first call: large-bank parity at default dispatch, the folded notch on hardware, headline A/B
    timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q --timeout 600 > "$out/scale.txt" 2>&1; echo "scale rc=$?" >> "$out/rc.txt"
    timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -q --timeout 300 -k "notch" > "$out/notch.txt" 2>&1; echo "notch rc=$?" >> "$out/rc.txt"
    $B --quick > "$out/bench_fold.json" 2> "$out/bench_fold.err"
    CHZ_NOTCH_FOLD=0 $B --quick > "$out/bench_nofold.json" 2> "$out/bench_nofold.err"
    $B --quick > "$out/bench_fold2.json" 2>> "$out/bench_fold.err"
    ;;
  full)      # the driver's command with every leg + the new GPU tests of the round
    $B > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -c 600 "$out/bench.err"
    timeout 900 python -m pytest tests/test_dropin.py tests/test_gpu_scale.py -m gpu -x -q --timeout 600 > "$out/dropin_scale.txt" 2>&1; echo "dropin+scale rc=$?" >> "$out/rc.txt"
    ;;
  paced)     # only the legs through filter.h (free-running + wall-clock paced), with the host's view of itself
    nproc > "$out/host.txt"; cat /sys/fs/cgroup/cpu.max >> "$out/host.txt" 2>&1; cat /sys/fs/cgroup/cpu.stat >> "$out/host.txt" 2>&1; uptime >> "$out/host.txt"
    $B --no-crt --no-next-rows --no-crt-pcie --no-cpu-baseline > "$out/bench_paced.json" 2> "$out/bench_paced.err"; echo "bench rc=$?" >> "$out/rc.txt"
    cat /sys/fs/cgroup/cpu.stat >> "$out/host.txt" 2>&1
    ;;
  tests)     # the whole GPU suite, as the driver runs it
    timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > "$out/gpu_suite.txt" 2>&1; echo "suite rc=$?" >> "$out/rc.txt"
    ;;
  bench)     # the driver's own command
    $B > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    ;;
  *)
    shift
    bash -c "$*" > "$out/custom.txt" 2>&1; echo "custom rc=$?" >> "$out/rc.txt"
    ;;
esac
tail -n 3 "$out"/*.txt 2>/dev/null | tail -n 40
