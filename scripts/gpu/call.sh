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
  r6y)       # round 6: the driver's bench command alone (the committed record: profiles re-collected on the final kernel sources first)
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; tail -1 "$out/smoke.txt"
    timeout 600 $B > "$out/bench_stdout.txt" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -n 1 "$out/bench_stdout.txt" > "$out/bench_headline.json"; cp gpurun_out/bench_detail.json "$out/bench_detail.json" 2>/dev/null
    wc -c "$out/bench_headline.json"; cat "$out/rc.txt"
    ;;
  r6x)       # round 6, final tree: the GPU tier three times in a row on one box (flakiness of the new mini-radiod tables), then the mini-radiod file five more times
    for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -p no:cacheprovider > "$out/suite_$i.txt" 2>&1; echo "suite $i rc=$? $(tail -1 $out/suite_$i.txt)" | tee -a "$out/rc.txt"; done
    for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_mini_radiod.py -m gpu -q --timeout 200 -x -p no:cacheprovider > "$out/mr_$i.txt" 2>&1; echo "mini-radiod $i rc=$? $(tail -1 $out/mr_$i.txt)" | tee -a "$out/rc.txt"; done
    ;;
  r6w)       # round 6: src/wfm.c's demod_wfm() joins the mini-radiod: a P = 9600 slave + a REAL inline master (N = 15,360) with REAL / shifted COMPLEX slaves per WFM channel
    KA9Q_HIP_PROFILE=1 timeout 600 python -m pytest tests/test_mini_radiod.py -m gpu -q --timeout 300 -s -k "wfm or spectrum" > "$out/mini_radiod.txt" 2>&1; echo "rc=$?" >> "$out/rc.txt"
    grep -a "mini-radiod\|passed\|failed\|Error\|assert\|filter_hip" "$out/mini_radiod.txt" | cut -c1-1800 | tail -30; cat "$out/rc.txt"
    ;;
  r6v)       # round 6: src/spectrum.c's demod_spectrum() joins the mini-radiod (narrowband analysers: any-length COMPLEX slaves + plan_complex; a wideband one: a SPECTRUM slave); the hip link now carries an FFTW provider
    timeout 1200 python -m pytest tests/test_mini_radiod.py -m gpu -q --timeout 600 -s > "$out/mini_radiod.txt" 2>&1; echo "rc=$?" >> "$out/rc.txt"
    grep -a "mini-radiod\|passed\|failed\|Error\|assert" "$out/mini_radiod.txt" | cut -c1-1500 | tail -30; cat "$out/rc.txt"
    ;;
  r6u)       # round 6: do the non-temporal output stores hurt MID-SIZE banks whose outputs the demodulator reads back out of the Infinity Cache?  the linear chain at 20 k ... 1.5 M channels, shipped (NT) against cnt0 (plain)
    for n in 20000 40000 70000 130000 300000 1500000; do for v in default cnt0; do
      L=""; [ $v != default ] && L=$PWD/ka9q-radio_amd/libchz_hip_$v.so
      for rep in 1 2; do CHZ_LIB=$L timeout 200 python scripts/chain_profile.py linear $n 2>> "$out/err.txt" | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$v', r['channels'], round(r['pipelined_ms_per_block'], 4), {k: round(v, 3) for k, v in r['ns_per_channel'].items()}, r['pcm_mismatches'])"; done
    done; done | tee "$out/chain_sizes.txt"
    ;;
  r6t)       # round 6: non-temporal output stores in chan_ifft's staged path (cnt2) against the shipped build: the C_rt search, the 8f chain, the PCIe probes
    for rep in 1 2; do for v in default cnt2; do
      L=""; [ $v != default ] && L=$PWD/ka9q-radio_amd/libchz_hip_$v.so
      CHZ_LIB=$L BENCH_NO_STREAMED=1 timeout 400 $B --no-dropin --no-dropin-paced --no-cpu-baseline --next-rows-modes linear,fm --detail "$out/full_${v}_$rep.json" > /dev/null 2>> "$out/err.txt"
    done; done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/full_*.json")):
    j = json.load(open(f)); c = j["c_rt"]
    print(os.path.basename(f), "c_rt", c.get("channels"), c.get("worst_block_ms"), c.get("mean_block_ms"), c.get("mean_crossing_channels"), [(p["channels"], p["sustained"], round(p["mean_block_ms"], 2)) for p in c.get("probes", [])],
          "pcie", [(x["channels"], x["sustained"], round(x["worst_block_ms"], 2), round(x["mean_block_ms"], 2)) for x in (j.get("c_rt_pcie") or []) if "error" not in x],
          "chain", [(x["mode"], round(x["pipelined_ms_per_block"], 3), x["pcm_mismatches"]) for x in (j.get("next_rows") or []) if "error" not in x], "us/step", round(j["ms_per_step"] * 1e3, 2))
PY
    ;;
  r6s)       # round 6: chan_ifft's response rows / staged output rows as non-temporal accesses (A/B builds cnt1 / cnt2 / cnt3): mean block time of a 20 M-channel bank
    cp gpurun_out/hbm_stream.txt "$out/" 2>/dev/null
    for rep in 1 2; do for v in default cnt1 cnt2 cnt3; do
      L=""; [ $v != default ] && L=$PWD/ka9q-radio_amd/libchz_hip_$v.so
      CHZ_LIB=$L BENCH_NO_STREAMED=1 timeout 300 $B --crt-channels 20000000 --crt-blocks 80 --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --no-next-rows --detail "$out/crt_${v}_$rep.json" > /dev/null 2>> "$out/err.txt"
    done; done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/crt_*.json")):
    c = json.load(open(f))["c_rt"]
    print(os.path.basename(f), [(p["channels"], p["blocks"], round(p["mean_block_ms"], 3), round(p["worst_block_ms"], 3), p.get("max_rel_err")) for p in c.get("probes", [])], c.get("error"))
PY
    ;;
  r6r)       # round 6: mini-radiod config 1 (complex front end) and config 4 shape (2000 channels, two shards) on the device; the null-stream soak in three queue modes
    timeout 900 python -m pytest tests/test_mini_radiod.py -m gpu -q -s --timeout 600 -k "config1 or config4" > "$out/mini_radiod.txt" 2>&1; echo "mini_radiod rc=$?" >> "$out/rc.txt"
    grep -E "A/B on the device|passed|failed|Error" "$out/mini_radiod.txt" | cut -c1-1500
    timeout 1300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 450 -k "null_stream" > "$out/null_stream.txt" 2>&1; echo "null_stream rc=$?" >> "$out/rc.txt"
    tail -5 "$out/null_stream.txt"
    for q in 1 2 4; do CHZ_OWN_QUEUES=$q timeout 400 python scripts/null_stream_soak.py 2>/dev/null | tail -1; done > "$out/null_stream_soak.jsonl"; cat "$out/null_stream_soak.jsonl"; cat "$out/rc.txt"
    ;;
  r6q)       # round 6: mini-radiod at BASELINE config 2 / config 3 scale on the device; block 0 inside 10 ms (tightened cold-start test)
    timeout 900 python -m pytest tests/test_mini_radiod.py -m gpu -q -s --timeout 600 > "$out/mini_radiod.txt" 2>&1; echo "mini_radiod rc=$?" >> "$out/rc.txt"
    grep -E "A/B on the device|passed|failed|Error" "$out/mini_radiod.txt" | cut -c1-2500
    timeout 600 python -m pytest tests/test_dropin.py -m gpu -q --timeout 600 -k "cold_start or wall_clock" > "$out/dropin_cold.txt" 2>&1; echo "cold rc=$?" >> "$out/rc.txt"
    tail -5 "$out/dropin_cold.txt"; cat "$out/rc.txt"
    ;;
  r6p)       # round 6: the profiles of the final kernels (kernel trace with 4 and 1 streams, PMC passes), RCCL sanity with one rank
    SKIP_PMC=0 timeout 1500 bash scripts/gpu_profile.sh r06 > "$out/profile.txt" 2>&1; echo "profile rc=$?" >> "$out/rc.txt"
    cat gpurun_out/r06_pmc_forward.json; cat gpurun_out/pmc_r06/retries.txt 2>/dev/null
    timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 scripts/rccl_sanity.py > "$out/rccl_sanity.txt" 2>&1; echo "rccl rc=$?" >> "$out/rc.txt"
    tail -2 "$out/rccl_sanity.txt"; cat "$out/rc.txt"
    ;;
  r6h)       # round 6: demod_fm_lanes' discriminator phases 1 / 2 / 4 (shipped) / 8 samples side by side, and 2 wavefronts per SIMD: the FM chain at 1.5 M channels
    for rep in 1 2; do for v in default fmd1 fmd2 fmd8 fmw2; do
      L=""; [ $v != default ] && L=$PWD/ka9q-radio_amd/libchz_hip_$v.so
      CHZ_LIB=$L timeout 200 python scripts/chain_profile.py fm 1500000 >> "$out/fm_$v.jsonl" 2>> "$out/err.txt"
    done; done
    for v in default fmd1 fmd2 fmd8 fmw2; do echo "$v: $(cat $out/fm_$v.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print(round(r['pipelined_ms_per_block'], 3), round(r['ns_per_channel']['demodulator_and_pcm'], 3), r['pcm_mismatches'], end=' | ')
")"; done
    ;;
  r6g)       # round 6: the whole GPU suite + smoke + the driver's command after the options / desc_push / watchdog changes
    timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > "$out/gpu_suite.txt" 2>&1; echo "suite rc=$?" >> "$out/rc.txt"
    tail -8 "$out/gpu_suite.txt"
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; tail -1 "$out/smoke.txt"
    timeout 600 $B > "$out/bench_stdout.txt" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -n 1 "$out/bench_stdout.txt" > "$out/bench_headline.json"; cp gpurun_out/bench_detail.json "$out/bench_detail.json" 2>/dev/null
    wc -c "$out/bench_headline.json"; grep "^bench.py \[" "$out/bench.err" | tail -14; cat "$out/rc.txt"
    ;;
  r6f)       # round 6: the demodulator / PCM copy streams on a hardware queue of their own WITHOUT being blocking streams: priority streams (3 = high, 4 = low)
             # against CU-masked (1, round 5's default) and plain (0): the 8f chain at 1.5 M channels and the double-buffered PCIe probe; the bench watchdog test
    PC="--no-crt --no-dropin --no-dropin-paced --no-cpu-baseline --no-next-rows"
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes linear"
    for rep in 1 2; do for q in 0 1 3 4; do
      CHZ_OWN_QUEUES=$q BENCH_NO_STREAMED=1 timeout 200 $B $NR --detail "$out/chain_q${q}_$rep.json" > /dev/null 2>> "$out/err.txt"
      CHZ_OWN_QUEUES=$q BENCH_PCIE_PROBES=2 BENCH_NO_STREAMED=1 timeout 200 $B $PC --detail "$out/pcie_q${q}_$rep.json" > /dev/null 2>> "$out/err.txt"
    done; done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pcie_*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), [(x["channels"], round(x["worst_block_ms"], 2), round(x["mean_block_ms"], 2), x.get("blocks_over_20ms"), round(x.get("p99_block_ms") or 0, 2)) if "error" not in x else x for x in d["c_rt_pcie"]])
for f in sorted(glob.glob(out + "/chain_*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), [(x.get("mode"), round(x.get("pipelined_ms_per_block", 0), 3), x.get("pcm_mismatches")) if "error" not in x else x for x in (j.get("next_rows") or [])], "headline us/step", round(j["ms_per_step"] * 1e3, 2))
PY
    timeout 400 python -m pytest tests/test_bench_contract.py -m gpu -q -k "never_comes_back" --timeout 300 2>&1 | tail -5
    ;;
  r6e)       # round 6: descriptors refreshed by a kernel (desc_push) instead of hipMemcpyAsync: is block 1's 8 ms stall gone?  mini-radiod with fading signals
    timeout 300 python scripts/block0_probe.py 60 > "$out/block0_probe.jsonl" 2> "$out/block0_probe.err"; echo "block0 rc=$?" >> "$out/rc.txt"
    python -c "
import json,sys
for ln in open('$out/block0_probe.jsonl'):
    r=json.loads(ln); print(r['label'], 'block0', r['block0_ms'], 'first8', r['first_8_ms'], 'p99', r['p99_ms'], 'max', r['max_ms'], 'drops', r['drops']); print('   ', (r['first_blocks_profile'] or '')[:1000])"
    timeout 600 python -m pytest tests/test_mini_radiod.py -m gpu -q -s --timeout 300 > "$out/mini_radiod.txt" 2>&1; echo "mini_radiod rc=$?" >> "$out/rc.txt"
    tail -5 "$out/mini_radiod.txt" | cut -c1-1800
    timeout 120 python scripts/mini_radiod_ab.py --paced --blocks 100 --seed 6 --repeat 3 > "$out/mini_paced.txt" 2>&1; grep -v "^   call" "$out/mini_paced.txt" | cut -c1-700 | head
    cat "$out/rc.txt"
    ;;
  r6d)       # round 6: the 9 ms hipMemcpyAsync of block 1 in context (rocprofv3 --hip-trace of the harness, 12 paced blocks)
    cd /tmp && BLOCK0_ONLY=2 BENCH_DROPIN_WRAPPER="rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OLDPWD/$out/trace --" timeout 300 python $OLDPWD/scripts/block0_probe.py 12 > "$OLDPWD/$out/block0_traced.jsonl" 2>> "$OLDPWD/$out/err.txt"; cd "$OLDPWD"
    python scripts/slow_hip_calls.py "$out/trace" > "$out/slow_hip_calls.txt" 2>&1; grep -A 20 "^context" "$out/slow_hip_calls.txt" | head -120; grep -A 60 "^all copies" "$out/slow_hip_calls.txt" | head -70
    rm -rf "$out/trace"
    ;;
  r6c)       # round 6: (1) the paced mini-radiod run, frames side by side where the gain differs; (2) block 1 of the drop-in blocks 8 ms inside chz_bank_execute
             # with the spectrum copied back: which HIP call?  the harness under rocprofv3 --hip-trace; and the same leg with the spectrum copy off
    timeout 120 python scripts/mini_radiod_ab.py --paced --blocks 100 --seed 6 --repeat 2 > "$out/mini_paced.txt" 2>&1; echo "paced rc=$?" >> "$out/rc.txt"
    grep -v "^   call" "$out/mini_paced.txt" | cut -c1-600 | head -30
    BLOCK0_ONLY=1 BLOCK0_EXTRA='[["1024 threads, no spectrum copy, no noise", 1024, {"KA9Q_HIP_FDOMAIN": "0"}], ["2000 threads, spectrum copied back", 2000, {}]]' timeout 200 python scripts/block0_probe.py 40 > "$out/block0_extra.jsonl" 2>> "$out/err.txt"
    python -c "
import json,sys
for ln in open('$out/block0_extra.jsonl'):
    r=json.loads(ln); print(r['label'], r['block0_ms'], r['first_8_ms']); print('   ', (r['first_blocks_profile'] or '')[:900])"
    cd /tmp && BLOCK0_ONLY=2 BENCH_DROPIN_WRAPPER="rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OLDPWD/$out/trace --" timeout 300 python $OLDPWD/scripts/block0_probe.py 12 > "$OLDPWD/$out/block0_traced.jsonl" 2>> "$OLDPWD/$out/err.txt"; cd "$OLDPWD"
    python scripts/slow_hip_calls.py "$out/trace" > "$out/slow_hip_calls.txt" 2>&1; head -60 "$out/slow_hip_calls.txt"
    rm -rf "$out/trace"
    ;;
  r6b)       # round 6, first call: block 0 of the FIRST drop-in process on a fresh box (before anything else touches the GPU); the reference's own
             # callers on the drop-in (mini-radiod, tests/test_mini_radiod.py); is a CU-masked stream a blocking stream? (scripts/micro/masked_stream_blocking.hip)
    timeout 300 python scripts/block0_probe.py 100 > "$out/block0_probe.jsonl" 2> "$out/block0_probe.err"; echo "block0 rc=$?" >> "$out/rc.txt"
    cat "$out/block0_probe.jsonl"
    timeout 600 python -m pytest tests/test_mini_radiod.py -m gpu -q -s --timeout 300 > "$out/mini_radiod.txt" 2>&1; echo "mini_radiod rc=$?" >> "$out/rc.txt"
    tail -12 "$out/mini_radiod.txt"
    timeout 60 ./scripts/micro/masked_stream_blocking.bin > "$out/masked_stream_blocking.txt" 2>&1; echo "masked rc=$?" >> "$out/rc.txt"
    cat "$out/masked_stream_blocking.txt"; cat "$out/rc.txt"
    ;;
  r5t)       # round 5: how many channels the double-buffered PCIe probe carries now that its period is D2H-bound (12.1 ms at 1.43 M)
    PC="--no-crt --no-dropin --no-dropin-paced --no-cpu-baseline --no-next-rows"
    for n in 1800000 1950000 2050000; do
      BENCH_PCIE_PROBES=2 BENCH_PCIE_PIPELINED_CHANNELS=$n BENCH_NO_STREAMED=1 timeout 200 $B $PC --detail "$out/pcie_$n.json" > /dev/null 2>> "$out/err.txt"
    done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pcie_*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), [(x["channels"], round(x["worst_block_ms"], 2), round(x["mean_block_ms"], 2), x.get("blocks_over_20ms"), round(x.get("p99_block_ms") or 0, 2), round(x.get("worst_latency_ms") or 0, 1), x.get("pcm_mismatches"), round(x.get("d2h_GBps") or 0, 1)) if "error" not in x else x for x in d["c_rt_pcie"]])
PY
    ;;
  r5s)       # round 5, last validation: the GPU suite + smoke and the driver's command on the final tree (demodulator + PCM copy streams on their own queues)
    timeout 420 $B > "$out/bench_stdout.txt" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -n 1 "$out/bench_stdout.txt" > "$out/bench_headline.json"; cp gpurun_out/bench_detail.json "$out/bench_detail.json" 2>/dev/null
    wc -c "$out/bench_headline.json"; grep "^bench.py \[" "$out/bench.err" | tail -2
    timeout 600 python -m pytest tests -m gpu -q --timeout 400 > "$out/gpu_suite.txt" 2>&1; echo "suite rc=$?" >> "$out/rc.txt"
    tail -4 "$out/gpu_suite.txt"
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; tail -1 "$out/smoke.txt"
    ;;
  r5r)       # round 5: the double-buffered PCIe probe (mean period 13.3 ms with plain streams, 16.8 with the demodulator stream on its own queue) and the chain, per queue mode
    PC="--no-crt --no-dropin --no-dropin-paced --no-cpu-baseline --no-next-rows"
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes linear"
    for q in 0 1 3 2; do
      CHZ_OWN_QUEUES=$q BENCH_PCIE_PROBES=2 BENCH_NO_STREAMED=1 timeout 200 $B $PC --detail "$out/pcie_q${q}.json" > /dev/null 2>> "$out/err.txt"
      CHZ_OWN_QUEUES=$q BENCH_NO_STREAMED=1 timeout 200 $B $NR --detail "$out/chain_q${q}.json" > /dev/null 2>> "$out/err.txt"
    done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pcie_*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), [(x["channels"], round(x["worst_block_ms"], 2), round(x["mean_block_ms"], 2), x.get("blocks_over_20ms"), round(x.get("p99_block_ms") or 0, 2)) if "error" not in x else x for x in d["c_rt_pcie"]])
for f in sorted(glob.glob(out + "/chain_*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), [(x.get("mode"), round(x.get("pipelined_ms_per_block", 0), 3), x.get("pcm_mismatches")) if "error" not in x else x for x in (j.get("next_rows") or [])])
PY
    ;;
  r5q)       # round 5: does the demodulator stream's own hardware queue cost the synchronous PCIe probes a late block now and then?  3 x 3 probes of 500 blocks per setting
    PC="--no-crt --no-dropin --no-dropin-paced --no-cpu-baseline --no-next-rows"
    for rep in 1 2 3; do
      for q in 0 1; do
        CHZ_OWN_QUEUES=$q BENCH_NO_STREAMED=1 timeout 200 $B $PC --detail "$out/pcie_q${q}_$rep.json" > /dev/null 2>> "$out/err.txt"
      done
    done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pcie_*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), [(x["channels"], round(x["worst_block_ms"], 2), round(x["mean_block_ms"], 2), x.get("blocks_over_20ms"), round(x.get("p99_block_ms") or 0, 2)) if "error" not in x else x for x in d["c_rt_pcie"]])
PY
    ;;
  r5p)       # round 5: the profiles of the final tree once more (the first run's FETCH_SIZE pass died inside rocprofv3)
    SKIP_PMC=0 timeout 900 bash scripts/gpu_profile.sh r05 > "$out/profile.txt" 2>&1
    cat gpurun_out/r05_pmc_forward.json; cat gpurun_out/pmc_r05/retries.txt 2>/dev/null
    ;;
  r5i)       # round 5: the HBM-streamed forward figure; pll_lanes with 4 samples per LDS round trip (A/B build)
    $B --quick --detail "$out/quick.json" > "$out/quick.head" 2> "$out/quick.err"; echo "quick rc=$?" >> "$out/rc.txt"
    python -c "import json; r=json.load(open('$out/quick.json'))['roofline']; print('pipelined', r['pipelined']['forward_us_per_block'], r['pipelined']['frac'], 'streamed', r['streamed'])"
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes pll"
    P4=$PWD/ka9q-radio_amd/libchz_hip_pllu4.so
    for rep in 1 2; do
      timeout 200 $B $NR --detail "$out/pll_default_$rep.json" > /dev/null 2>> "$out/err.txt"
      CHZ_LIB=$P4 timeout 200 $B $NR --detail "$out/pll_u4_$rep.json" > /dev/null 2>> "$out/err.txt"
    done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pll_*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), [(x.get("mode"), round(x.get("pipelined_ms_per_block", 0), 3), x.get("pcm_mismatches"), {k: round(v, 3) for k, v in (x.get("ns_per_channel") or {}).items()}) if "error" not in x else x for x in (j.get("next_rows") or [])])
PY
    ;;
  r5h)       # round 5: the C_rt search with the 22 M-channel bank; why the all-streams-masked tree hung (stream callbacks on CU-masked streams?)
    $B --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --no-next-rows --detail "$out/crt22.json" > "$out/crt22.head" 2> "$out/crt22.err"; echo "crt22 rc=$?" >> "$out/rc.txt"
    python -c "import json; c=json.load(open('$out/crt22.json'))['c_rt']; print('c_rt', c.get('channels'), c.get('sustained'), c.get('worst_block_ms'), c.get('mean_crossing_channels'), [(p['channels'], p['blocks'], round(p['worst_block_ms'],2), p['sustained']) for p in c.get('probes', [])], c.get('error'))"
    for q in 0 1 2; do CHZ_OWN_QUEUES=$q timeout 60 python scripts/hostfunc_on_masked_stream.py >> "$out/hostfunc.txt" 2>&1; echo "CHZ_OWN_QUEUES=$q rc=$?" >> "$out/hostfunc.txt"; done
    cat "$out/hostfunc.txt"
    ;;
  r5g)       # round 5, final validation: the whole GPU suite + smoke, the driver's command, its profiles -- every step bounded
    timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > "$out/gpu_suite.txt" 2>&1; echo "suite rc=$?" >> "$out/rc.txt"
    tail -8 "$out/gpu_suite.txt"
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; tail -1 "$out/smoke.txt"
    timeout 480 $B > "$out/bench_stdout.txt" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -n 1 "$out/bench_stdout.txt" > "$out/bench_headline.json"; cp gpurun_out/bench_detail.json "$out/bench_detail.json" 2>/dev/null
    wc -c "$out/bench_headline.json"; grep "^bench.py \[" "$out/bench.err" | tail -3
    SKIP_PMC=0 timeout 900 bash scripts/gpu_profile.sh r05 > "$out/profile.txt" 2>&1
    ;;
  r5f)       # round 5, fifth call: the driver's command with the demodulator stream alone on its own queue (bounded: r5d's run with EVERY stream masked hung in one of the legs), then A/Bs of the chain
    timeout 420 $B > "$out/bench_stdout.txt" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -n 1 "$out/bench_stdout.txt" > "$out/bench_headline.json"; cp gpurun_out/bench_detail.json "$out/bench_detail.json" 2>/dev/null
    grep "^bench.py \[" "$out/bench.err" | tail -20
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes linear"
    run_chain() { tag=$1; shift; env "$@" timeout 200 $B $NR --detail "$out/chain_$tag.json" > /dev/null 2>> "$out/err.txt"; }
    run_chain q0 CHZ_OWN_QUEUES=0
    run_chain q1 CHZ_OWN_QUEUES=1
    run_chain q2 CHZ_OWN_QUEUES=2
    run_chain linu4 CHZ_LIB=$PWD/ka9q-radio_amd/libchz_hip_linu4.so
    run_chain linu8 CHZ_LIB=$PWD/ka9q-radio_amd/libchz_hip_linu8.so
    run_chain q1b CHZ_OWN_QUEUES=1
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/chain_*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), [(x.get("mode"), round(x.get("pipelined_ms_per_block", 0), 3), x.get("pcm_mismatches"), {k: round(v, 3) for k, v in (x.get("ns_per_channel") or {}).items()}) if "error" not in x else x for x in (j.get("next_rows") or [])], "headline us/step", round(j["ms_per_step"] * 1e3, 2))
PY
    wc -c "$out/bench_headline.json"; head -c 1500 "$out/bench_headline.json"
    ;;
  r5d)       # round 5, fourth call: demod_lin_lanes at 4 wavefronts per SIMD (A/B build), then the whole GPU suite on the tree with own hardware queues
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes linear"
    LW4=$PWD/ka9q-radio_amd/libchz_hip_linw4.so
    for rep in 1 2; do
      $B $NR --detail "$out/chain_default_$rep.json" > /dev/null 2>> "$out/err.txt"
      CHZ_LIB=$LW4 $B $NR --detail "$out/chain_linw4_$rep.json" > /dev/null 2>> "$out/err.txt"
    done
    CHZ_OWN_QUEUES=0 $B $NR --detail "$out/chain_plain_streams.json" > /dev/null 2>> "$out/err.txt"
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/chain_*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), [(x.get("mode"), round(x.get("pipelined_ms_per_block", 0), 3), x.get("pcm_mismatches"), {k: round(v, 3) for k, v in (x.get("ns_per_channel") or {}).items()}) if "error" not in x else x for x in (j.get("next_rows") or [])], "headline us/step", round(j["ms_per_step"] * 1e3, 2))
PY
    timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > "$out/gpu_suite.txt" 2>&1; echo "suite rc=$?" >> "$out/rc.txt"
    tail -6 "$out/gpu_suite.txt"
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; tail -1 "$out/smoke.txt"
    ;;
  r5e)       # round 5, last call: the driver's command on the final tree + its profiles
    $B > "$out/bench_stdout.txt" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -n 1 "$out/bench_stdout.txt" > "$out/bench_headline.json"; cp gpurun_out/bench_detail.json "$out/bench_detail.json" 2>/dev/null
    wc -c "$out/bench_headline.json"; cat "$out/bench_headline.json"
    SKIP_PMC=0 timeout 1500 bash scripts/gpu_profile.sh r05 > "$out/profile.txt" 2>&1
    ;;
  r5c)       # round 5, third call: is the CU-mask gain the partition or the hardware-queue assignment?  + the batched passes, fixed build
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes linear"
    run_chain() { tag=$1; shift; env "$@" $B $NR --detail "$out/chain_$tag.json" > "$out/chain_$tag.head" 2>> "$out/err.txt"; }
    run_chain base1 X=1
    run_chain q8_1 GPU_MAX_HW_QUEUES=8
    run_chain cus256_1 CHZ_TAIL_CUS=256
    run_chain cus96_1 CHZ_TAIL_CUS=96
    run_chain cus128_1 CHZ_TAIL_CUS=128
    run_chain base2 X=1
    run_chain q8_2 GPU_MAX_HW_QUEUES=8
    run_chain cus256_2 CHZ_TAIL_CUS=256
    run_chain cus96_2 CHZ_TAIL_CUS=96
    run_chain cus96q8 CHZ_TAIL_CUS=96 GPU_MAX_HW_QUEUES=8
    LIBB=$PWD/ka9q-radio_amd/libchz_hip_batch.so
    for rep in 1 2; do
      BENCH_NO_NOTCH=1 $B --quick --detail "$out/fwd_base_$rep.json" > /dev/null 2>> "$out/err.txt"
      BENCH_NO_NOTCH=1 GPU_MAX_HW_QUEUES=8 $B --quick --detail "$out/fwd_baseq8_$rep.json" > /dev/null 2>> "$out/err.txt"
      BENCH_NO_NOTCH=1 CHZ_LIB=$LIBB $B --quick --detail "$out/fwd_batchlib_n0_$rep.json" > /dev/null 2>> "$out/err.txt"
      BENCH_NO_NOTCH=1 CHZ_LIB=$LIBB CHZ_FWD_BATCH_N=2 $B --quick --detail "$out/fwd_batch2_$rep.json" > /dev/null 2>> "$out/err.txt"
      BENCH_NO_NOTCH=1 CHZ_LIB=$LIBB CHZ_FWD_BATCH_N=4 $B --quick --detail "$out/fwd_batch4_$rep.json" > /dev/null 2>> "$out/err.txt"
    done
    for n in 2 4; do CHZ_LIB=$LIBB CHZ_FWD_BATCH_N=$n timeout 300 python scripts/batch_check.py >> "$out/batch_parity.txt" 2>&1; done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/chain_*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), [(x.get("mode"), round(x.get("pipelined_ms_per_block", 0), 3), x.get("pcm_mismatches"), {k: round(v, 3) for k, v in (x.get("ns_per_channel") or {}).items()}) if "error" not in x else x for x in (j.get("next_rows") or [])], "headline us/step", round(j["ms_per_step"] * 1e3, 2))
for f in sorted(glob.glob(out + "/fwd_*.json")):
    j = json.load(open(f)); r = j["roofline"]
    print(os.path.basename(f), "ms_per_step", round(j["ms_per_step"] * 1e3, 2), "fwd_pipelined_us", round(r["pipelined"]["forward_us_per_block"], 2), "frac", round(r["pipelined"]["frac"], 4))
PY
    cat "$out/batch_parity.txt"
    ;;
  r5b)       # round 5, second call: profiles of the shipped kernels (folded notch), the CU-mask and batched-pass experiments, the packed PCM store
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes linear"
    for cus in 0 32 64 96; do      # (0 = unset: plain streams)
      if [ $cus = 0 ]; then $B $NR --detail "$out/chain_cus0.json" > "$out/chain_cus0.head" 2>> "$out/err.txt"
      else CHZ_TAIL_CUS=$cus $B $NR --detail "$out/chain_cus$cus.json" > "$out/chain_cus$cus.head" 2>> "$out/err.txt"; fi
    done
    $B --no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline --next-rows-modes pll,fm --detail "$out/chain_pllfm.json" > "$out/chain_pllfm.head" 2>> "$out/err.txt"
    LIBB=$PWD/ka9q-radio_amd/libchz_hip_batch.so
    for rep in 1 2; do
      BENCH_NO_NOTCH=1 $B --quick --detail "$out/fwd_base_$rep.json" > /dev/null 2>> "$out/err.txt"
      BENCH_NO_NOTCH=1 CHZ_LIB=$LIBB $B --quick --detail "$out/fwd_batchlib_n0_$rep.json" > /dev/null 2>> "$out/err.txt"
      BENCH_NO_NOTCH=1 CHZ_LIB=$LIBB CHZ_FWD_BATCH_N=2 $B --quick --detail "$out/fwd_batch2_$rep.json" > /dev/null 2>> "$out/err.txt"
      BENCH_NO_NOTCH=1 CHZ_LIB=$LIBB CHZ_FWD_BATCH_N=4 $B --quick --detail "$out/fwd_batch4_$rep.json" > /dev/null 2>> "$out/err.txt"
    done
    for n in 2 4; do CHZ_LIB=$LIBB CHZ_FWD_BATCH_N=$n timeout 300 python scripts/batch_check.py >> "$out/batch_parity.txt" 2>&1; done
    python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/chain_*.json")):
    j = json.load(open(f))
    print(os.path.basename(f), [(x.get("mode"), round(x.get("pipelined_ms_per_block", 0), 3), x.get("pcm_mismatches"), {k: round(v, 3) for k, v in (x.get("ns_per_channel") or {}).items()}) if "error" not in x else x for x in (j.get("next_rows") or [])])
for f in sorted(glob.glob(out + "/fwd_*.json")):
    j = json.load(open(f)); r = j["roofline"]
    print(os.path.basename(f), "ms_per_step", round(j["ms_per_step"] * 1e3, 2), "fwd_pipelined_us", round(r["pipelined"]["forward_us_per_block"], 2), "frac", round(r["pipelined"]["frac"], 4))
PY
    timeout 600 python -m pytest tests/test_dropin.py tests/test_gpu_scale.py -m gpu -q --timeout 400 -k "clique or two_shards or end_to_end" > "$out/newtests.txt" 2>&1; echo "newtests rc=$?" >> "$out/rc.txt"
    tail -3 "$out/newtests.txt"
    SKIP_PMC=0 timeout 1500 bash scripts/gpu_profile.sh r05 > "$out/profile.txt" 2>&1
    tail -3 "$out/batch_parity.txt"
    ;;
  r5a)       # round 5, first call: the drop-in's cold start + sharding on hardware, the compact bench line, the C_rt search
    timeout 900 python -m pytest tests/test_dropin.py -m gpu -q --timeout 600 > "$out/dropin.txt" 2>&1; echo "dropin rc=$?" >> "$out/rc.txt"
    tail -5 "$out/dropin.txt"
    $B > "$out/bench_stdout.txt" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/rc.txt"
    tail -n 1 "$out/bench_stdout.txt" > "$out/bench_headline.json"; cp gpurun_out/bench_detail.json "$out/bench_detail.json" 2>/dev/null
    wc -c "$out/bench_headline.json"; tail -c 400 "$out/bench.err"
    timeout 1200 python -m pytest tests/test_bench_contract.py -m gpu -q --timeout 900 -k "contract_keys or crt_search or quick" > "$out/contract.txt" 2>&1; echo "contract rc=$?" >> "$out/rc.txt"
    tail -5 "$out/contract.txt"
    ;;
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
  xcd)       # round 4 experiment: XCD-affine fwd_cols -> fwd_rows hand-over (build variants libchz_hip_xcd{1,2}.so), timing + L2 counters
    X1="CHZ_LIB=$PWD/ka9q-radio_amd/libchz_hip_xcd1.so CHZ_NOTCH_FOLD=0"      # affine placement + plain hand-over stores
    X2="CHZ_LIB=$PWD/ka9q-radio_amd/libchz_hip_xcd2.so CHZ_NOTCH_FOLD=0"      # affine placement, write-through stores kept
    env $X1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "forward_matches_oracle and (2592000 or 1296000)" > "$out/parity_affine.txt" 2>&1; echo "parity rc=$?" >> "$out/rc.txt"
    for rep in 1 2; do
      CHZ_NOTCH_FOLD=0 $B --quick > "$out/base_$rep.json" 2>> "$out/err.txt"
      env $X1 $B --quick > "$out/affine_plain_$rep.json" 2>> "$out/err.txt"
      env $X2 $B --quick > "$out/affine_wt_$rep.json" 2>> "$out/err.txt"
    done
    for lanes in 1 2; do
      CHZ_NOTCH_FOLD=0 CHZ_STREAMS=$lanes $B --quick > "$out/base_lanes$lanes.json" 2>> "$out/err.txt"
      env $X1 CHZ_STREAMS=$lanes $B --quick > "$out/affine_plain_lanes$lanes.json" 2>> "$out/err.txt"
    done
    R=$PWD; cd /tmp
    for v in base affine; do
      E="CHZ_NOTCH_FOLD=0 CHZ_NOTCH_ORDER=event"; [ $v = affine ] && E="$X1 CHZ_NOTCH_ORDER=event"
      for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
        n=$(echo $c | cut -d' ' -f1)
        env $E timeout 240 rocprofv3 --pmc $c -f csv -d $R/$out/pmc_${v}_$n -o $n -- python $R/bench.py --steps 160 --warmup 16 --min-seconds 0.05 --quick > $R/$out/pmc_${v}_$n.log 2>&1
      done
    done
    cd $R
    python scripts/rocprof_summary.py $out/pmc_base_TCC_HIT_sum $out/pmc_affine_TCC_HIT_sum $out/pmc_base_FETCH_SIZE $out/pmc_affine_FETCH_SIZE $out/pmc_base_WRITE_SIZE $out/pmc_affine_WRITE_SIZE > "$out/pmc_summary.txt" 2>&1
    rm -rf $out/pmc_*/ 2>/dev/null
    ;;
  agc)       # round 4: L2 hand-over microbenchmark; the AGC's first look in chan_ifft's epilogue, A/B at 1.5 M channels; demodulator tests
    ./scripts/micro/xcd_l2_handover.bin > "$out/xcd_l2_handover.txt" 2>&1
    NR="--no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline"
    $B $NR > "$out/next_rows_peak.json" 2> "$out/err.txt"
    CHZ_AGC_PEAK=0 $B $NR > "$out/next_rows_nopeak.json" 2>> "$out/err.txt"
    timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_pipeline.py tests/test_golden.py -m gpu -x -q --timeout 600 -k "demod or scale or 70001 or golden or coherent or fm" > "$out/demod_tests.txt" 2>&1; echo "tests rc=$?" >> "$out/rc.txt"
    ;;
  profile)   # the round's profile: kernel trace (4 streams and 1), PMC passes, summaries for profiles/
    GRAFT_REPO_ROOT=$PWD bash scripts/gpu_profile.sh r04 > "$out/profile.txt" 2>&1; echo "profile rc=$?" >> "$out/rc.txt"
    ;;
  cpu)       # headline + the CPU baseline leg only
    $B --no-crt --no-next-rows --no-dropin --no-dropin-paced --no-crt-pcie > "$out/bench_cpu.json" 2> "$out/err.txt"; echo "bench rc=$?" >> "$out/rc.txt"
    ;;
  chain)     # the 8f chain legs only (1.5 M channels, verified) + the large-bank and demodulator tests
    $B --no-crt --no-dropin --no-dropin-paced --no-crt-pcie --no-cpu-baseline > "$out/next_rows.json" 2> "$out/err.txt"; echo "bench rc=$?" >> "$out/rc.txt"
    timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_pipeline.py tests/test_golden.py -m gpu -x -q --timeout 600 -k "demod or scale or 70001 or golden or coherent or fm" > "$out/demod_tests.txt" 2>&1; echo "tests rc=$?" >> "$out/rc.txt"
    ;;
  chainpmc)  # SQ counters for the 8f chain's kernels (tuned chan_ifft, noise_est, demod_lin_lanes / pll_lanes / demod_fm_lanes), one --pmc set per pass and mode.
             # round 6: scripts/chain_profile.py runs ONE mode's chain on 300,000 channels and nothing else (bench.py's leg sits behind the headline loops,
             # tens of thousands of launches that a counter pass serialises: round 5's passes did not finish in 140 s); instruction counts per channel do not depend on the bank size
    R=$PWD; cd /tmp
    run() { n=$1; m=$2; shift; shift; CHZ_NOTCH_ORDER=event timeout 200 rocprofv3 --pmc "$@" -f csv -d $R/$out/$n -o $n -- python $R/scripts/chain_profile.py $m 300000 > $R/$out/$n.log 2>&1; echo "$n rc=$?" >> $R/$out/rc.txt; }
    for m in linear pll fm; do
      run sq1_$m $m SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
      run sq2_$m $m SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU
    done
    cd $R
    for m in linear pll fm; do echo "## mode $m (300,000 channels)"; python scripts/rocprof_summary.py $out/sq1_$m $out/sq2_$m; done > "$out/chain_pmc.txt" 2>&1
    rm -rf $out/sq1_* $out/sq2_*
    cat $out/rc.txt; grep -c . $out/chain_pmc.txt
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
