#!/bin/bash
# scripts/first_multigpu_run.sh [N] -- the FIRST run on a node with N (default 8) MI355X: everything the one-GPU runner could never execute, in the
# order "safest first", every step under its own time-out, every result into profiles/ (named r06_multigpu_*).  UNMEASURED ON HARDWARE: no multi-GPU
# node was available in rounds 1-6; the same code paths run on the CPU over stand-in devices and a stand-in RCCL
# (tests/test_dropin_stub.py, tests/test_engine_fake_rccl.py) and with one rank / one device on the MI355X.
#   1  RCCL sanity: N ranks, one all-reduce through the engine's own communicator (chz_comm_create / chz_comm_allreduce_max)
#   2  bench.py --gpus N, config 5 (replicas only: no collective anywhere)                          -> the weak-scaling floor
#   3  bench.py --gpus N, config 4: replicate leg first, then every exchange (auto, samples, broadcast, subband) -- bench.py's own
#      watchdog ends a leg that hangs and still prints the line (leg_timeouts, headline_from)         -> the north star's figure
#   4  the same with CHZ_NOTCH_ORDER=event (no device-side ticket wait beside RCCL's kernels) if step 3 reported a time-out
#   5  filter.h drop-in sharded over the N devices at wall-clock pace, KA9Q_HIP_EXCHANGE=samples, then =broadcast (its ladder falls back to samples)
# usage: bash scripts/first_multigpu_run.sh 8 2>&1 | tee profiles/r06_multigpu_log.txt
set -u
N=${1:-8}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
P=profiles
mkdir -p $P gpurun_out
step() { echo; echo "=== $(date +%T) $*"; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1"

step "0: devices"
rocm-smi --showtopo > $P/r06_multigpu_topology.txt 2>&1 || true
python -c "import torch; print(torch.cuda.device_count(), 'devices visible')"

step "1: RCCL sanity through the engine's communicator ($N ranks)"
timeout 300 $TR --master-port 29701 scripts/rccl_sanity.py > $P/r06_multigpu_rccl_sanity.txt 2>&1; echo "rc=$?"
tail -$N $P/r06_multigpu_rccl_sanity.txt 2>/dev/null

step "2: config 5, replicas only, N = 1 2 4 $N"
for n in 1 2 4 $N; do
  [ $n -gt $N ] && continue
  timeout 900 python bench.py --gpus $n --config 5 --steps 200 --warmup 50 --detail $P/r06_multigpu_config5_n$n.detail.json > $P/r06_multigpu_config5_n$n.json 2> $P/r06_multigpu_config5_n$n.err
  echo "config5 n=$n rc=$? $(tail -c 400 $P/r06_multigpu_config5_n$n.json | head -c 400)"
done

step "3: config 4 (the metric), every exchange leg, N = 2 4 $N"
for n in 2 4 $N; do
  [ $n -gt $N ] && continue
  timeout 1200 python bench.py --gpus $n --steps 200 --warmup 50 --detail $P/r06_multigpu_config4_n$n.detail.json > $P/r06_multigpu_config4_n$n.json 2> $P/r06_multigpu_config4_n$n.err
  rc=$?
  echo "config4 n=$n rc=$rc"; python - $P/r06_multigpu_config4_n$n.json <<'PY'
import json, sys
try:
    h = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   value", h.get("value"), "headline_from", h.get("headline_from"), "leg_timeouts", h.get("leg_timeouts"), "exchange_errors", h.get("exchange_errors"), "legs", h.get("legs"), "c_rt", (h.get("c_rt") or {}).get("channels"))
except Exception as ex:
    print("   no line:", ex)
PY
  if grep -q '"leg_timeouts":\[' $P/r06_multigpu_config4_n$n.json 2>/dev/null; then
    step "4: config 4 again at n=$n with the notch ordered by HIP events"
    CHZ_NOTCH_ORDER=event timeout 1200 python bench.py --gpus $n --steps 200 --warmup 50 --detail $P/r06_multigpu_config4_event_n$n.detail.json > $P/r06_multigpu_config4_event_n$n.json 2> $P/r06_multigpu_config4_event_n$n.err
    echo "config4 (event) n=$n rc=$?"
  fi
done

step "5: the drop-in sharded over $N devices at wall-clock pace (1024 x 24 kHz channels per device), both exchanges"
DEVS=$(python -c "print(','.join(str(i) for i in range($N)))")
for ex in samples broadcast; do
  KA9Q_HIP_EXCHANGE=$ex BLOCK0_ONLY=1 BLOCK0_CONFIG4=1 BLOCK0_EXTRA="[[\"sharded over $N devices, exchange $ex\", $((1024 * N)), {\"KA9Q_HIP_DEVICES\": \"$DEVS\", \"KA9Q_HIP_EXCHANGE\": \"$ex\"}]]" \
    timeout 600 python scripts/block0_probe.py 250 > $P/r06_multigpu_dropin_$ex.jsonl 2> $P/r06_multigpu_dropin_$ex.err
  echo "dropin $ex rc=$? $(cut -c1-400 $P/r06_multigpu_dropin_$ex.jsonl)"
done
echo; echo "done: profiles/r06_multigpu_*"
