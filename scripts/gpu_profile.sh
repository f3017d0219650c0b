#!/bin/bash
# round profile: the driver's bench command under rocprofv3 --kernel-trace --stats, the same command with
# CHZ_STREAMS=1 (one kernel at a time: solo durations), then the PMC passes; summaries under gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export BENCH_LEG_BUDGET_SCALE=10      # (bench.py's per-leg watchdog: a tracer slows the legs down)
TAG=${1:-r04}
FAST="--quick"
cd /tmp
# the same command unprofiled, on this box, first: what bench.py's HIP events say here (boxes differ by 10 %)
timeout 300 python $R/bench.py --steps 20 --warmup 5 $FAST --detail $R/gpurun_out/${TAG}_plain_detail.json > $R/gpurun_out/${TAG}_plain.json 2> $R/gpurun_out/${TAG}_plain.err
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_4streams -o t -- python $R/bench.py --steps 20 --warmup 5 $FAST > $R/gpurun_out/${TAG}_prof4.json 2> $R/gpurun_out/${TAG}_prof4.err
CHZ_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_1stream -o t -- python $R/bench.py --steps 20 --warmup 5 $FAST > $R/gpurun_out/${TAG}_prof1.json 2> $R/gpurun_out/${TAG}_prof1.err
cd $R
[ "$SKIP_PMC" = 1 ] || timeout 1300 bash scripts/pmc_passes.sh $TAG
{
  echo "## the same command UNPROFILED on this box right before (bench.py's HIP-event durations, us): $(python -c "import json; r=json.load(open('gpurun_out/${TAG}_plain_detail.json'))['roofline']; print(r['kernels_us'], 'forward', round(r['forward_us_per_block'],2), 'frac', round(r['frac'],4))" 2>/dev/null)"
  echo "## rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 $FAST   (default: 4 HIP streams)"
  python scripts/rocprof_summary.py gpurun_out/prof_${TAG}_4streams
  echo; echo "## the same with CHZ_STREAMS=1 (one kernel at a time: solo kernel durations, comparable with roofline.kernels_us)"
  python scripts/rocprof_summary.py gpurun_out/prof_${TAG}_1stream
  echo; echo "## PMC passes (scripts/pmc_passes.sh), per-dispatch averages"
  [ "$SKIP_PMC" = 1 ] || for d in sq1 sq2 fetch write tcc; do python scripts/rocprof_summary.py gpurun_out/pmc_$TAG/$d; done
} > gpurun_out/${TAG}_rocprofv3_summary.txt 2>&1
python scripts/rocprof_summary.py --kernels-json gpurun_out/prof_${TAG}_4streams gpurun_out/prof_${TAG}_1stream gpurun_out/${TAG}_rocprof_kernels.json $TAG gpurun_out/${TAG}_plain_detail.json
[ "$SKIP_PMC" = 1 ] || python scripts/rocprof_summary.py --json gpurun_out/pmc_$TAG/fetch gpurun_out/pmc_$TAG/write gpurun_out/${TAG}_pmc_forward.json
# the raw traces are tens of MB: only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf gpurun_out/prof_${TAG}_4streams gpurun_out/prof_${TAG}_1stream
[ "$SKIP_PMC" = 1 ] || find gpurun_out/pmc_$TAG -type f ! -name "*.log" -size +200k -delete
head -40 gpurun_out/${TAG}_rocprofv3_summary.txt
