#!/usr/bin/env python3
"""bench.py -- free-running throughput of the MI355X overlap-save channelizer.

A "step" is one 20 ms block of the hot path: the shared forward transform of the overlapped
input window plus gather x response + backward transform of every channel, inputs already
resident in HBM (an 8-block sig_gen stream in the device ring, replayed cyclically).

Workloads (BASELINE.json configs; --config picks one, the default follows --gpus):
  1  sig_gen complex 2.4 MS/s, ONE IQ-mode 12 kHz channel (P = 300): the reference's own CPU-runnable plumbing case, here on one GPU
  2  sig_gen real  64.8 MS/s, 256 x 12 kHz NBFM channels (P = 300), one GPU
  3  sig_gen real 129.6 MS/s, 1024 mixed usb/cw/iq 12 kHz channels (P = 300), one GPU   [default at --gpus 1]
  4  sig_gen real 129.6 MS/s, 1024 x 24 kHz channels (P = 600) PER GPU, sharded by frequency; rank 0 owns the
     front end and the forward transform, the block spectrum travels over xGMI through RCCL (called from
     libchz_hip.so: chz_run_blocks_sharded) on the slot's own HIP stream                 [default at --gpus > 1]
  5  one independent 129.6 MS/s front end per GPU (sig_gen seed = rank + 1), 1024 mixed channels each,
     no RCCL in the data path ("replicas only")

Timing: after W warm-up steps, ONE region of exactly K steps is timed between device synchronisations
(barrier across ranks on both sides) and reported as drained_k_step_region; it also sizes `reps`.  The
headline is the steady state: `reps` back-to-back repetitions of the K-step loop (reps*K blocks, no drain in
between) form one timed region, bracketed the same way; `regions` of them cover --min-seconds and the MEDIAN
(max over ranks each) gives ms_per_step and value.  So the figure does not depend on how small K is.

metric: channels sustained at the input rate = channel-blocks per second / 50 blocks/s
(real-time-equivalent channels of the benchmarked configuration).  The c_rt object is the literal
"simultaneous channels in real time": one huge bank, every single block inside its 20 ms slot.

Prints ONE JSON line on rank 0 (the last line of stdout).
"""
import argparse
import ctypes
import json
import os
import socket
import statistics
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCKTIME = 0.02
RING_BLOCKS = 8
HBM_PEAK_GBS = 8000.0                   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:35
COPY_RATE_GBS = 6290.0                  # the chip's measured achievable copy rate (same guide)
STREAM_COPY_GBS = 4880.0                # measured here: device-to-device copies of 24-48 GB, i.e. far beyond the Infinity Cache
                                        # (scripts/hbm_stream_probe.py, profiles/r02_hbm_stream.json): what HBM streams, reads + writes mixed

# the 129.6 MS/s geometry is also what tests and scripts import from here
FS = 129.6e6
L = int(round(FS * BLOCKTIME))          # 2,592,000
M = L // 4 + 1                          # 648,001  (overlap 5, src/radio.c:582-586)
N = L + M - 1                           # 3,240,000
BINS = N // 2 + 1
FWD_BYTES = 4 * N + 8 * BINS            # 25,920,008  


def geometry(fs, real=True):
    l = int(round(fs * BLOCKTIME))
    m = l // 4 + 1
    n = l + m - 1
    return l, m, n, (n // 2 + 1 if real else n)


def fwd_bytes(n, real=True):
    if not real:
        return 8 * n + 8 * n            # complex window in, full spectrum out
    return 4 * n + 8 * (n // 2 + 1)     # read the real window once, write the half spectrum once


def chan_bytes(P, olen):
    return 8 * P + 8 * P + 8 * olen     # 6,720 (P=300) / 13,440 (P=600)


KINDS = [(50 / 12000, 3000 / 12000), (-200 / 12000, 200 / 12000), (-5000 / 12000, 5000 / 12000)]   # usb / cw / iq


def channel_plan_config2(nch, fs=64.8e6):
    """256 x 12 kHz NBFM: f_i = 10 MHz + i*12.5 kHz, +-5 kHz."""
    hz = fs / geometry(fs)[2]
    return [(int(round((10e6 + i * 12.5e3) / hz)), -5000 / 12000, 5000 / 12000) for i in range(nch)]


def channel_plan_config3(nch):
    """1024 mixed channels: thirds usb / cw / iq, f_i = 1 MHz + i*60 kHz + (i mod 40) Hz."""
    hz_per_bin = FS / N
    plan = []
    for i in range(nch):
        f = 1e6 + (i % 1040) * 60e3 + (i % 40)
        plan.append((int(round(f / hz_per_bin)),) + KINDS[i % 3])
    return plan


def channel_plan_config4(nch, rank):
    """8192 x 24 kHz channels, f_i = 0.5 MHz + i*7.8 kHz, nch per GPU (rank r owns [r*nch, (r+1)*nch)), +-10 kHz."""
    hz_per_bin = FS / N
    plan = []
    for j in range(nch):
        i = rank * nch + j
        f = 0.5e6 + i * 7.8e3
        plan.append((int(round(f / hz_per_bin)), -10000 / 24000, 10000 / 24000))
    return plan


def workload_for(config, rank, world, nch):
    """Everything that defines what one rank runs."""
    real = True
    if config == 1:
        # BASELINE config 1: sig_gen complex 2.4 MS/s, ONE IQ-mode channel (the reference's own CPU-runnable plumbing case)
        fs, P, olen, real = 2.4e6, 300, 240, False
        nch = nch or 1
        hz = fs / geometry(fs, False)[2]
        plan = [(int(round((100e3 + i * 15e3) / hz)), -5000 / 12000, 5000 / 12000) for i in range(nch)]
        name = "config1: sig_gen complex 2.4 MS/s, %d IQ-mode 12 kHz channel%s (P=300)" % (nch, "" if nch == 1 else "s")
        seed = 1
    elif config == 2:
        fs, P, olen = 64.8e6, 300, 240
        nch = nch or 256
        plan = channel_plan_config2(nch, fs)
        name = "config2: sig_gen real 64.8 MS/s, %d x 12 kHz NBFM channels (P=300)" % nch
        seed = 1
    elif config == 3:
        fs, P, olen = FS, 300, 240
        nch = nch or 1024
        plan = channel_plan_config3(nch)
        name = "config3: sig_gen real 129.6 MS/s, %d mixed usb/cw/iq 12 kHz channels (P=300)" % nch
        seed = 1
    elif config == 4:
        fs, P, olen = FS, 600, 480
        nch = nch or 1024
        plan = channel_plan_config4(nch, rank)
        name = ("config4: sig_gen real 129.6 MS/s, %d x 24 kHz channels (P=600), %d per GPU sharded by frequency over %d MI355X"
                % (nch * world, nch, world))
        seed = 1
    elif config == 5:
        fs, P, olen = FS, 300, 240
        nch = nch or 1024
        plan = channel_plan_config3(nch)
        name = ("config5: %d independent 129.6 MS/s sig_gen front ends (seed = rank+1), %d mixed 12 kHz channels each, "
                "no RCCL in the data path" % (world, nch))
        seed = rank + 1
    else:
        raise SystemExit("--config must be 1, 2, 3, 4 or 5")
    l, m, n, bins = geometry(fs, real)
    return dict(config=config, fs=fs, L=l, M=m, N=n, bins=bins, P=P, olen=olen, nch=nch, plan=plan, name=name, seed=seed, real=real)


def siggen_ring(oracle_lib, fs=FS, seed=1, l=None, real=True):
    """8 blocks of the deterministic sig_gen stream (CW carrier 10.00002 MHz -- 100.02 kHz for the complex 2.4 MS/s case --,
    -20 dBFS, noise -40 dBFS).  The generator is test infrastructure (oracle/); it only produces INPUT, outside the timed region."""
    l = l or geometry(fs)[0]
    carrier = 10.00002e6 if fs > 2.1e7 else 100.02e3
    g = oracle_lib.SigGen(carrier / fs, 10 ** (-20 / 20), 10 ** (-40 / 20),
                          oracle_lib.scale_ad(real, 1), real, seed=seed)
    return g.generate(RING_BLOCKS * l)


def kernel_sources_sha16():
    """fingerprint of the forward kernels' sources (the same one scripts/rocprof_summary.py writes into the committed profiles)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("chz_kernels.h", "regfft.h", "chz_plan.h", "chz_launch.h"):
        h.update(open(os.path.join(ROOT, "ka9q-radio_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def profile_sha(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get("kernel_sources_sha16")
    except Exception:
        return None


def pmc_traffic_bytes():
    """HBM-side bytes per block of the forward kernels from the committed rocprofv3 PMC passes
    (profiles/pmc_forward.json, written by scripts/rocprof_summary.py --json from separate FETCH_SIZE and
    WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes).  None if no profile has been committed."""
    path = os.path.join(ROOT, "profiles", "pmc_forward.json")
    try:
        return json.load(open(path))["forward_traffic_bytes_per_block"]
    except Exception:
        return None


def rocprof_kernel_us():
    """Average durations of the forward kernels from the committed rocprofv3 --kernel-trace --stats run of this same
    command with one HIP stream (profiles/rocprof_kernels.json, written by scripts/rocprof_summary.py --kernels-json):
    what the profiler says next to what the HIP events of this run say.  None if no profile has been committed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "rocprof_kernels.json")))
        return d
    except Exception:
        return None


def cpu_baseline(oracle_lib, ring, wl, seconds=12.0, crt_blocks=500):
    """Reference filter.c (oracle/_ref, FFT butterflies from the project's float32 provider, NOT FFTW)
    timed on this host's cores: 1 forward-FFT worker thread + a pool of channel threads, radiod style."""
    if not oracle_lib.have_ref():
        return None
    R = oracle_lib.ref()
    R.oracle_fft_set_precision(1)          # float32 arithmetic for a fair CPU timing
    cores = os.cpu_count() or 1
    pool = max(1, min(cores - 1, 16))
    ring = np.ascontiguousarray(ring, np.float32 if wl.get("real", True) else np.complex64)
    plan, P, olen = wl["plan"], wl["P"], wl["olen"]
    sarr = np.array([p[0] for p in plan], np.int32)

    def run(workers, budget, internal=1):
        m = oracle_lib.RefMaster(wl["L"], wl["M"], oracle_lib.REAL if wl.get("real", True) else oracle_lib.COMPLEX, worker_threads=workers, internal_threads=internal)
        chans = []
        for shift, low, high in plan:
            c = m.channel(olen, oracle_lib.COMPLEX)
            c.set_filter(low, high, 11.0)
            chans.append(c)
        harr = (ctypes.c_void_p * len(chans))(*[c.h for c in chans])
        # calibrate with 2 blocks, then run a bounded sample
        t = R.refchz_bench(m.h, harr, sarr.ctypes.data, len(chans), ring.ctypes.data, RING_BLOCKS, 2, pool)
        nblk = int(max(3, min(200, budget / max(t / 2, 1e-3))))
        t = R.refchz_bench(m.h, harr, sarr.ctypes.data, len(chans), ring.ctypes.data, RING_BLOCKS, nblk, pool)
        mn, mx, avg = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
        R.refchz_fft_times(ctypes.byref(mn), ctypes.byref(mx), ctypes.byref(avg))
        m.close()
        return nblk, t / nblk, avg.value / 1e6

    # how far is the portable provider from a tuned library?  pocketfft (scipy.fft, float32, SIMD) on the same window
    calib = None
    try:
        import scipy.fft as sfft
        win = np.ascontiguousarray(ring[:wl["N"]], np.float32)
        calib = {}
        for workers in (1, 2):
            sfft.rfft(win, workers=workers)
            t0 = time.perf_counter(); reps = 0
            while time.perf_counter() - t0 < 1.0:
                sfft.rfft(win, workers=workers); reps += 1
            calib["pocketfft_f32_rfft_ms_workers_%d" % workers] = (time.perf_counter() - t0) / reps * 1e3
    except Exception as ex:
        calib = {"error": str(ex)[:120]}
    nblk, per_block, fft_ms = run(1, seconds * 0.6)
    nblk2, per_block2, fft_ms2 = run(2, seconds * 0.3)      # fft-threads = 2, the reference's advice for this rate (docs/ka9q-radio.md:232)
    nblk4, per_block4, fft_ms4 = run(1, seconds * 0.2, internal=4)   # fft-internal-threads = 4: one worker, the transform itself on 4 threads (src/filter.c:131-133)

    # microseconds per channel-block on ONE core: an inline master (N_worker_threads = 0: the transform runs on the caller, src/filter.c:562-600),
    # then execute_filter_output() of every channel on this thread, timed as a whole
    us_per_chan = None
    try:
        m0 = oracle_lib.RefMaster(wl["L"], wl["M"], oracle_lib.REAL if wl.get("real", True) else oracle_lib.COMPLEX, worker_threads=0)
        cs = []
        for shift, low, high in plan[:256]:
            c = m0.channel(olen, oracle_lib.COMPLEX); c.set_filter(low, high, 11.0); cs.append((c, shift))
        m0.write(ring[:wl["L"]])
        obuf = np.zeros(olen, np.complex64)
        call, optr = R.refchz_chan_execute, obuf.ctypes.data
        hs = [(c.h, int(sh)) for c, sh in cs]
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 0.5:
            for h, sh in hs:
                call(h, sh, optr)                      # (~1 us of ctypes overhead per call rides along)
            reps += 1
        us_per_chan = (time.perf_counter() - t0) / (reps * len(cs)) * 1e6
        m0.close()
    except Exception as ex:
        us_per_chan = None

    # C_rt(CPU): "simultaneous channels sustained in real time" as radiod experiences it -- the front end hands a block over every 20 ms of
    # wall clock and never waits (oracle/ref_driver.c:refchz_bench_blocks, pace_us = 20000); a channel count is sustained if over
    # `crt_blocks` consecutive blocks NO channel is ever lapped (block_drops == 0, src/filter.c:686-701) and the last block completes
    # within ND block times of its arrival (the backlog did not grow).  The GPU's c_rt is the stricter statement (every block done
    # inside its own 20 ms); for a CPU whose forward transform alone takes longer than a block, pipelining over FFT workers is how it
    # keeps up at all, so the reference's own criterion is the fair one.  Ascending ladder around the free-running estimate.
    crt_cpu = None
    if hasattr(R, "refchz_bench_blocks") and wl.get("real", True):
        REPEATS = 3
        per_rep = max(60, crt_blocks // REPEATS)

        def probe(m, chans, n, workers):
            while len(chans) < n:
                shift, low, high = plan[len(chans) % len(plan)]
                c = m.channel(olen, oracle_lib.COMPLEX); c.set_filter(low, high, 11.0); chans.append(c)
            harr = (ctypes.c_void_p * n)(*[c.h for c in chans[:n]])
            sh = np.array([plan[i % len(plan)][0] for i in range(n)], np.int32)
            reps, hiccups = [], []

            def one():
                st = (ctypes.c_double * 4)()
                if hasattr(R, "refchz_reset_drops"):
                    R.refchz_reset_drops(harr, n)
                R.refchz_bench_blocks(m.h, harr, sh.ctypes.data, n, ring.ctypes.data, RING_BLOCKS, per_rep + 8, pool, 8, st, int(BLOCKTIME * 1e6))
                return {"block_drops": int(st[3]), "mean_completion_interval_ms": st[1], "worst_completion_interval_ms": st[0], "worst_latency_ms": st[2]}
            for _ in range(REPEATS):
                r = one()
                # ONE stall of the shared host (> 4 block times: every channel still waiting is lapped once) is not overload: the backlog did not
                # grow (mean completion interval inside 1 %) and only a few per cent of the channel-blocks were lapped.  Round 5 / 6: exactly
                # that made the figure flip between two rungs from run to run (65,536 channels need 6.7 of the quota's 16 CPUs).  Such a
                # repeat is run again, at most twice per rung; a rung that laps again is not sustained.  Overload looks different (131,072:
                # a third of all channel-blocks lapped, mean interval 22 ms) and fails at once.
                while (r["block_drops"] > 0 and r["block_drops"] < 0.05 * n * per_rep and r["mean_completion_interval_ms"] <= 1.01 * BLOCKTIME * 1e3 and len(hiccups) < 2):
                    hiccups.append(r)
                    r = one()
                reps.append(r)
                if r["block_drops"] > 0:
                    break                                 # lapped: not sustained, the other repeats would say nothing new
            ok = len(reps) == REPEATS and all(r["block_drops"] == 0 and r["mean_completion_interval_ms"] <= 1.01 * BLOCKTIME * 1e3 for r in reps)
            return {"channels": n, "repeats": reps, "repeats_run_again_after_a_host_stall": hiccups or None, "fft_workers": workers,
                    "block_drops": sum(r["block_drops"] for r in reps), "worst_latency_ms": max(r["worst_latency_ms"] for r in reps), "sustained": bool(ok)}
        # x4 ladder from 4096 (down to 1024 / 256 if that already fails), then ONE bisection step between the last sustained and the
        # first failed count; a rung = 3 repeats of crt_blocks/3 paced blocks, sustained only if NO repeat lapped a channel
        m = oracle_lib.RefMaster(wl["L"], wl["M"], oracle_lib.REAL, worker_threads=2)
        chans, probes, best, failed = [], [], None, None
        n = 4096
        while n <= 262144:
            pr = probe(m, chans, n, 2)
            probes.append(pr)
            if not pr["sustained"]:
                failed = pr
                break
            best = pr
            n *= 4
        if best is None:
            for n in (1024, 256):
                pr = probe(m, chans, n, 2)
                probes.append(pr)
                if pr["sustained"]:
                    best = pr
                    break
                failed = pr
        if best and failed and failed["channels"] > best["channels"]:
            mid = int((best["channels"] * failed["channels"]) ** 0.5) // 64 * 64
            pr = probe(m, chans, mid, 2)
            probes.append(pr)
            if pr["sustained"]:
                best = pr
            else:
                failed = pr
        m.close()
        crt_cpu = {"channels": best["channels"] if best else 0, "sustained": bool(best), "limit_above_ladder": bool(best and not failed),
                   "first_failed_channels": failed["channels"] if failed else None,
                   "repeats": REPEATS, "blocks_per_repeat": per_rep, "fft_workers": 2, "cores": 2 + pool,
                   "worst_latency_ms": best["worst_latency_ms"] if best else None, "probes": probes,
                   "definition": "front end paced at one block per 20 ms of wall clock, never waiting (as an A/D); a channel count is sustained if in EACH of %d "
                                 "repeats of %d blocks no channel was lapped (block_drops = 0, src/filter.c:686-701) and the mean completion interval stayed "
                                 "within 1 %% of 20 ms (no growing backlog) -- a repeat that lapped < 5 %% of its channel-blocks with the mean interval intact (ONE stall of the "
                                 "shared host) is run again, at most twice per rung; the worst arrival-to-last-channel latency is reported, not judged (scheduler noise "
                                 "of a shared host decided round 4's figure); 2 FFT worker threads (docs/ka9q-radio.md:232) + a POOL of %d channel threads each "
                                 "looping over a static channel subset (SURVEY 8d; radiod itself runs one thread per channel and stops at Nchannels = 2000, "
                                 "src/radio.h:356); ladder x4 from 4096 + one bisection step.  NOT the GPU's criterion: c_rt (GPU) demands every single block complete "
                                 "inside its own 20 ms; this one lets the FFT workers pipeline blocks and ignores the worst latency -- the reference's own notion of "
                                 "keeping up (no lapped slave), the only one a CPU whose forward transform takes most of a block time can meet" % (REPEATS, per_rep, pool)}
    R.oracle_fft_set_precision(0)
    out = {
        "value": len(plan) * BLOCKTIME / per_block, "unit": "channels",
        "cores": 1 + pool, "kind": "reference",
        "sample": "%d blocks of the same workload (%d channels P=%d), reference filter.c with the project's "
                  "float32 FFT provider (AVX2 four-step, one thread per transform; FFTW3 is not installed on this image), 1 FFT worker + %d channel threads"
                  % (nblk, len(plan), P, pool),
        "ms_per_block": per_block * 1e3, "fwd_fft_ms_avg": fft_ms, "host_cores": cores,
        "two_fft_workers": {"value": len(plan) * BLOCKTIME / per_block2, "cores": 2 + pool, "blocks": nblk2,
                            "ms_per_block": per_block2 * 1e3, "fwd_fft_ms_avg": fft_ms2},
        "four_fft_internal_threads": {"value": len(plan) * BLOCKTIME / per_block4, "cores": 4 + pool, "blocks": nblk4,
                                      "ms_per_block": per_block4 * 1e3, "fwd_fft_ms_avg": fft_ms4},
        # real time is a statement about EVERY block (c_rt_cpu below); these two say whether the MEAN block time of each leg is inside 20 ms
        "real_time": bool(crt_cpu["sustained"]) if crt_cpu else bool(min(per_block, per_block2) <= BLOCKTIME),
        "mean_block_inside_20ms": {"one_fft_worker": bool(per_block <= BLOCKTIME), "two_fft_workers": bool(per_block2 <= BLOCKTIME)},
        "c_rt_cpu": crt_cpu, "us_per_channel_block": us_per_chan, "fwd_fft_ms": fft_ms,
        # what the channel side could carry if nothing but its own arithmetic limited it: the pool's cores (after the forward transform's
        # two), each doing one channel-block per us_per_channel_block, for 20 ms
        "throughput_bound_channels": int(pool * BLOCKTIME * 1e6 / us_per_chan) if us_per_chan else None,
        # the forward transform is what bounds the CPU path at this rate; how the provider compares with a tuned library on this host
        # (round 6: the provider's long transform runs 8 sub-transforms side by side in AVX2 lanes -- round 5's scalar one was 1.9 x slower than pocketfft)
        "fft_calibration": dict(calib or {}, portable_provider_fwd_fft_ms=fft_ms,
                                note="forward N=%d real transform, float32: the project's provider inside reference filter.c vs "
                                     "scipy's pocketfft (SIMD) on this host; FFTW3 itself is not installed" % wl["N"]),
    }
    if calib and "pocketfft_f32_rfft_ms_workers_1" in calib and calib["pocketfft_f32_rfft_ms_workers_1"] < fft_ms:
        pf = calib["pocketfft_f32_rfft_ms_workers_1"]
        out["with_tuned_fft_estimate"] = {
            "value": len(plan) * BLOCKTIME / max(per_block - (fft_ms - pf) * 1e-3, pf * 1e-3),
            "note": "the measured block time with the provider's forward-transform time replaced by pocketfft's (1 worker)"}
    return out


def crt_leg(pkg, eng, wl, counts, blocks, run_one, shared=False, verify=64, agree=None):
    """C_rt: one bank of channels of the workload's kind tiled from its plan, I/O resident in HBM;
    every block is run to completion ON ITS OWN (run_one(job) -> ms) and must take <= 20 ms.
    counts = [..]  an explicit ascending ladder probed inside ONE allocated bank (stops at the first failing rung)
    counts = None  search: two short calibration runs give the line mean_ms(n); the first full rung sits just under the count at
                   which the MEAN block time would reach 20 ms, then 0.5 M steps up while sustained (or down until sustained: a slower
                   box still reports a sustained count), then one bisection step.  Reported beside it: that mean-crossing count, which
                   does not depend on one late block (box-independent to about 1 %).
    agree(x) -> max over ranks (multi-rank runs: every rank must take the same decisions, the blocks contain collectives)."""
    P, olen, tile = wl["P"], wl["olen"], 3072
    solo = agree is None
    agree = agree or (lambda x: x)
    top = 22_000_000 if P == 300 else 11_000_000          # the allocated bank (responses + 4 output images: 222 GB of the 288): the search must not be capped by it
    if counts:
        counts = sorted(set(int(c) - int(c) % tile for c in counts))
        nmax = counts[-1]
    else:
        nmax = top - top % tile
    # every rank takes the same way out: a rank that fails here (the bank does not fit, an upload error) must not leave the others waiting in
    # the next collective -- the outcome of the set-up is agreed on (max over ranks of a failure flag) before anybody goes on or raises
    setup_error = None
    try:
        bank = eng.bank(P, olen, nmax, shared_rows=3 if shared else 0)
        if wl["config"] == 4:
            base = channel_plan_config3(tile)
            plan = [(sh, -10000 / 24000, 10000 / 24000) for sh, _, _ in base]
        else:
            plan = channel_plan_config3(tile)
        resp = np.stack([pkg.filterapi.design_response(P, olen, wl["N"], True, lo, hi, 11.0) for _, lo, hi in plan[:3]])
        resp = np.ascontiguousarray(np.tile(resp, (tile // 3, 1)))
        shifts = np.array([p[0] for p in plan], np.int32)
        if shared:                                   # the three filters of the mix, ONE copy each; every channel names its row
            bank.set_row_responses(0, resp[:3])
            rows = (np.arange(tile) % 3).astype(np.int32)
        for c0 in range(0, nmax, tile):
            if shared:
                bank.set_rows(c0, rows)
            else:
                bank.set_responses(c0, resp)
            bank.set_shifts(c0, shifts + (c0 // tile) % 7)
    except Exception as ex:
        setup_error = ex
    if agree(1.0 if setup_error is not None else 0.0) > 0:
        raise RuntimeError("c_rt: the bank of %d channels could not be set up on %s: %s" % (nmax, "this rank" if setup_error is not None else "another rank", setup_error))
    probes, calib, best = [], [], None
    state = {"job": 0}

    def rung(nch, nblk, full=True):
        """8 untimed blocks, then up to nblk timed ones; a solo run leaves the rung at the first block outside its slot"""
        bank.set_active(nch)
        for j in range(8):
            run_one(state["job"]); state["job"] += 1
        worst = tot = 0.0
        late = 0
        for j in range(nblk):
            ms = run_one(state["job"]); state["job"] += 1          # forward (root) [+ exchange] + the small bank + this bank, then a device sync
            worst = max(worst, ms); tot += ms
            late += ms > BLOCKTIME * 1e3
            if full and solo and late and j >= 20:
                break                            # one block outside the slot: the rung is lost, no need to sit through the rest
        n_run = j + 1
        worst, mean = agree(worst), agree(tot / n_run)
        pr = {"channels": nch + wl["nch"], "bank_channels": nch, "blocks": n_run, "worst_block_ms": worst, "mean_block_ms": mean, "blocks_over_20ms": int(late),
              "sustained": bool(full and worst <= BLOCKTIME * 1e3 and n_run == nblk)}
        if full and verify and pr["sustained"]:
            # the rung's LAST block (still in its slot): `verify` channels sampled across the active range -- first, last, workgroup
            # edges, random -- against the oracle's channel on the device's own spectrum of that block (tests/scale_check.py)
            import scale_check as sc
            chans = sc.sample_channels(nch, verify, seed=nch)
            v = sc.check_plain(eng, bank, (state["job"] - 1) % 4, chans, lambda c: int(shifts[c % tile]) + (c // tile) % 7,
                               lambda c: resp[c % tile])
            pr.update(verified_channels=v["verified_channels"], max_rel_err=v["max_rel_err"], highest_channel_checked=v["highest_channel_checked"])
            if agree(1.0 if v["failed"] else 0.0) > 0:               # (agreed on: every rank leaves the leg together)
                raise RuntimeError("c_rt rung of %d channels: outputs of channels %s differ from the oracle (max rel err %.3g)" % (nch, v["failed"][:8], v["max_rel_err"]))
        (probes if full else calib).append(pr)
        return pr

    def grid(n, g):
        n = int(n) // g * g
        return max(tile, min(nmax, n - n % tile))

    if counts:
        for nch in counts:
            pr = rung(nch, blocks)
            if pr["sustained"]:
                best = pr
            else:
                break
        search = "explicit ascending ladder inside one bank, %d blocks per rung" % blocks
    else:
        step = 500_000 if P == 300 else 250_000
        a = rung(grid(0.45 * nmax, step), 24, full=False)
        b = rung(grid(0.85 * nmax, step), 24, full=False)
        slope = (b["mean_block_ms"] - a["mean_block_ms"]) / (b["channels"] - a["channels"])
        icpt = a["mean_block_ms"] - slope * a["channels"]
        est = (BLOCKTIME * 1e3 - icpt) / slope if slope > 0 else nmax
        n = grid(0.975 * est, step)
        pr = rung(n, blocks)
        lo = hi = None
        if pr["sustained"]:
            lo = best = pr
            while n + step <= nmax:
                n += step
                pr = rung(n, blocks)
                if not pr["sustained"]:
                    hi = pr
                    break
                lo = best = pr
        else:
            hi = pr
            floor = max(tile, int(0.5 * est))
            while n - step >= floor:
                n -= step
                pr = rung(n, blocks)
                if pr["sustained"]:
                    lo = best = pr
                    break
                hi = pr
        if lo and hi:                                                        # one bisection step, if the grid has a count in between
            mid = grid((lo["bank_channels"] + hi["bank_channels"]) // 2 + step // 4, step // 2)
            if lo["bank_channels"] < mid < hi["bank_channels"]:
                pr = rung(mid, blocks)
                if pr["sustained"]:
                    best = pr
        search = ("two 24-block calibration runs -> mean_ms(n); first rung at 0.975 x the count whose MEAN block would take 20 ms, then %d-channel steps "
                  "up while sustained / down until sustained, then one bisection step; %d blocks per rung, a rung is left at its first late block" % (step, blocks))
    bank.set_active(0)
    bank.destroy()                               # 170+ GB: the next leg needs the room
    # where the MEAN block time crosses 20 ms: least squares over every run of this leg that has at least 20 blocks
    pts = [(p["channels"], p["mean_block_ms"]) for p in calib + probes if p["blocks"] >= 20]
    crossing = None
    if len(set(x for x, _ in pts)) >= 2:
        xs, ys = np.array([x for x, _ in pts], float), np.array([y for _, y in pts], float)
        k, c = np.polyfit(xs, ys, 1)
        if k > 0:
            crossing = int((BLOCKTIME * 1e3 - c) / k)
    rep = best or min(probes, key=lambda p: p["channels"])
    total_ch, mean, worst = rep["channels"], rep["mean_block_ms"], rep["worst_block_ms"]
    chan_alg = total_ch * chan_bytes(P, olen)
    dram = total_ch * ((0 if shared else 8 * P) + 8 * olen)   # responses in + outputs out; the gathered master bins (and shared rows) are cache hits
    return {"channels": total_ch if best else 0, "P": P, "blocks": rep["blocks"], "worst_block_ms": worst, "mean_block_ms": mean,
            "sustained": bool(best is not None), "smallest_probed_channels": None if best else total_ch,
            "mean_crossing_channels": crossing, "rungs": len(probes),
            "verified_channels": rep.get("verified_channels", 0), "max_rel_err": rep.get("max_rel_err"),
            "highest_channel_checked": rep.get("highest_channel_checked"),
            "verification": "after the timed blocks of every sustained rung: sampled channels of the rung's last block (first, last, workgroup edges, random) "
                            "against the oracle's execute_filter_output on the device's own block spectrum; err_rms <= 1e-5 rms + float32 floor",
            "probes": probes, "calibration": calib, "search": search,
            "algorithmic_GBps": (fwd_bytes(wl["N"]) + chan_alg) / (mean * 1e-3) / 1e9,
            "dram_side_GBps": dram / (mean * 1e-3) / 1e9, "dram_side_frac_of_hbm_peak": dram / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "dram_side_frac_of_measured_stream_copy": dram / (mean * 1e-3) / 1e9 / STREAM_COPY_GBS,
            "responses": "3 rows shared by all channels (chz_bank_create_shared)" if shared else "one row per channel",
            "note": "every block individually <= 20 ms; I/O resident in HBM (responses %.0f GB, 4 output images %.0f GB for the allocated bank); the "
                    "algorithmic figure counts the gathered master bins, which the caches serve -- the DRAM-side figure is "
                    "responses + outputs only" % ((0 if shared else nmax * P * 8) / 1e9, 4 * nmax * olen * 8 / 1e9)}


_COMB = {}


def comb_ring(wl, seed=1):
    """8 blocks of input for the legs whose results are VERIFIED: white noise (0.05 rms, as before) plus one CW carrier next to every
    channel of the config-3 raster (bin 25000 + 1500 m + 3: the tiled banks listen 0..7 bins around it), all periodic over the ring
    so the cyclic replay has no seam.  A carrier in every channel gives the coherent-mode PLLs something to lock to: on noise alone a
    PLL is chaotic at its phase detector's wrap and no two arithmetic libraries walk the same trajectory."""
    key = (wl["L"], seed)
    if key not in _COMB:
        n = RING_BLOCKS * wl["L"]
        rng = np.random.default_rng(seed)
        X = np.zeros(n // 2 + 1, np.complex128)
        per_bin = n / wl["N"]
        for m in range(1040):
            k = int(round((25000 + 1500 * m + 3) * per_bin))
            if k < n // 2:
                X[k] = 0.004 * (1 + m % 3) * (n / 2) * np.exp(1j * rng.uniform(0, 2 * np.pi))
        x = np.fft.irfft(X, n) + 0.05 * rng.standard_normal(n)
        _COMB[key] = x.astype(np.float32)
    return _COMB[key]


def verify_chain(pkg, eng, bank, wl, nch, tile, shifts, resp, one, job0, nver, k, step, pcm_host=None):
    """After a leg's timed blocks: restart the demodulators of `k` sampled channels (first, last, workgroup edges, random) at block
    `job0`, run `nver` more blocks at the leg's full scale and default dispatch, and compare every stage of every sampled channel with
    the oracle (tests/scale_check.py): chan_ifft + downconvert() tail, estimate_noise(), demodulator + PCM.  step(job) runs one block
    and returns the host PCM image of that block (uint8 [nch][stride]) or None for device reads."""
    import oracle_lib as ol
    import scale_check as sc
    chans = sc.sample_channels(nch, k, seed=nch + 1)
    off = pkg.engine.DemodParams(channels=0)
    for c in chans:                                            # a demodulator switched off and on again starts from its initial state
        bank.set_demod(job0, c, [off], BLOCKTIME)
        bank.set_demod(job0, c, [one], BLOCKTIME)
    lp = ol.LinParams(*[getattr(one, f) for f, _ in ol.LinParams._fields_])
    sh = [int(shifts[c % tile]) + (c // tile) % 7 for c in chans]
    chk = sc.ChainChecker(wl["L"], wl["M"], wl["fs"], 12000.0, wl["P"], wl["olen"], chans, sh, [3.3] * len(chans), lambda c: resp[c % tile],
                          [lp] * len(chans), pre_blocks=job0, strict_pll=True)
    for j in range(job0, job0 + nver):
        host = step(j)
        eng.sync()
        out, power, noise, pcm, status = sc.read_sampled(bank, j % 4, chans, pcm_host=host)
        chk.block(eng.spectrum(j % 4), out, power, noise, pcm, status)
    r = chk.result()
    if r["failed"] or r["pcm_mismatches"] or r["status_mismatches"]:
        raise RuntimeError("verification failed: %s" % json.dumps(r))
    return r


def crt_pcie_leg(pkg, wl, nch, blocks, demod, dev_index, pipelined=False):
    """C_rt WITH the host link in the loop (never `value`): every block takes its L new samples from pinned host memory
    (H2D) and returns every channel's result to pinned host memory (D2H) before it counts as done, and must take <= 20 ms:
      demod=False  the channel's olen complex baseband samples (what execute_filter_output hands a channel thread)
      demod=True   fine tuning + noise estimate + linear demodulator on the device; mono S16BE PCM + one status byte go back
    Bytes are the bytes actually shipped.
    pipelined=False  every block on its own: H2D -> kernels -> D2H -> host sync, and the whole round trip must fit 20 ms
    pipelined=True   a double-buffered host loop: block j is handed to the device, THEN the host waits for block j-1's results
                     (two pinned output buffers in turn); the time between two consecutive completions must stay <= 20 ms, a
                     block's results are in host memory at most two block times after its samples were (reported as latency)"""
    lib = pkg.engine.lib()
    C = ctypes
    lib.chz_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.chz_bank_read_pcm_flags_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.chz_bank_pcm_wait.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.chz_slot_sync.argtypes = [C.c_void_p, C.c_int]
    Lw, P, olen, tile = wl["L"], wl["P"], wl["olen"], 3072
    nch -= nch % tile
    eng = pkg.engine.Engine(wl["L"], wl["M"], pkg.engine.REAL, device=dev_index, ring_blocks=RING_BLOCKS)
    try:
        per_ch = (2 * olen + 1) if demod else 8 * olen
        hin = C.c_void_p()
        houts, hfls = [C.c_void_p(), C.c_void_p()], [C.c_void_p(), C.c_void_p()]
        assert lib.chz_host_alloc(C.byref(hin), 4 * Lw) == 0
        for k in range(2 if pipelined else 1):
            assert lib.chz_host_alloc(C.byref(houts[k]), (2 * olen if demod else 8 * olen) * nch) == 0
            assert lib.chz_host_alloc(C.byref(hfls[k]), nch) == 0
        xin = np.ctypeslib.as_array(C.cast(hin, C.POINTER(C.c_float)), shape=(Lw,))
        xin[:] = comb_ring(wl)[:Lw]                              # one block of noise + carriers, handed over again for every block
        bank = eng.bank(P, olen, nch)
        plan = channel_plan_config3(tile)
        resp = np.stack([pkg.filterapi.design_response(P, olen, wl["N"], True, lo, hi, 11.0) for _, lo, hi in plan[:3]])
        resp = np.ascontiguousarray(np.tile(resp, (tile // 3, 1)))
        shifts = np.array([p[0] for p in plan], np.int32)
        for c0 in range(0, nch, tile):
            bank.set_responses(c0, resp); bank.set_shifts(c0, shifts + (c0 // tile) % 7)
        eng.set_notches([0], 0.01)
        if demod:
            for c0 in range(0, nch, tile):
                bank.set_tuning(0, c0, shifts + (c0 // tile) % 7, np.full(tile, -3.3 / 12000.0))
            bank.enable_noise(wl["fs"])
            bank.set_pcm_stride(2 * olen)                 # mono S16: 480 B per channel and block, contiguous
            v = lambda db: 10 ** (db / 20.0)                # radiod's defaults for a linear mode (src/modes.c:40-60,224-246)
            one = pkg.engine.DemodParams(channels=1, env=0, agc=1, encoding=pkg.engine.PCM_S16BE, snr_squelch=0, squelch_tail=1, tuned=1, kind=0,
                                         samprate=12000.0, headroom=v(-15.0), threshold=v(-15.0), recovery_rate=v(20.0), hangtime=1.1, dc_alpha=0.0,
                                         bandwidth=2950.0, shift=0.0, squelch_open=10 ** 0.8, squelch_close=10 ** 0.7, gain=v(50.0))
            for c0 in range(0, nch, 65536):
                bank.set_demod(0, c0, [one] * min(65536, nch - c0), BLOCKTIME)
        bank.set_active(nch)
        worst = tot = 0.0
        lat_worst = 0.0
        times = []
        if not pipelined:
            for j in range(blocks + 8):
                t0 = time.perf_counter()
                assert lib.chz_input_write(eng._h, hin, Lw) == 0              # H2D of the block's new samples (pinned source)
                assert lib.chz_step(eng._h, j) == 0
                if demod:
                    assert lib.chz_bank_read_pcm_flags_async(eng._h, bank.id, j % 4, 0, nch, houts[0], hfls[0]) == 0
                else:
                    assert lib.chz_bank_read_async(eng._h, bank.id, j % 4, 0, nch, houts[0]) == 0
                eng.sync()
                dt = (time.perf_counter() - t0) * 1e3
                if j >= 8:
                    worst = max(worst, dt); tot += dt; times.append(dt)
        else:
            issued = {}
            last_done = None
            for j in range(blocks + 9):
                if j < blocks + 8:
                    issued[j] = time.perf_counter()
                    assert lib.chz_input_write(eng._h, hin, Lw) == 0
                    assert lib.chz_step(eng._h, j) == 0
                    if demod:
                        assert lib.chz_bank_read_pcm_flags_async(eng._h, bank.id, j % 4, 0, nch, houts[j % 2], hfls[j % 2]) == 0
                    else:
                        assert lib.chz_bank_read_async(eng._h, bank.id, j % 4, 0, nch, houts[j % 2]) == 0
                if j >= 1:                                                    # block j-1's results: the other host buffer
                    if demod:
                        assert lib.chz_bank_pcm_wait(eng._h, bank.id, (j - 1) % 4) == 0
                    else:
                        assert lib.chz_slot_sync(eng._h, (j - 1) % 4) == 0
                    now = time.perf_counter()
                    if j - 1 >= 8:
                        period = (now - last_done) * 1e3
                        worst = max(worst, period); tot += period; times.append(period)
                        lat_worst = max(lat_worst, (now - issued[j - 1]) * 1e3)
                    last_done = now
            eng.sync()
        # ---- verification, after the timed loop: what came back over the link against the oracle
        import scale_check as sc
        jlast = blocks + 7
        if not demod:
            host = np.ctypeslib.as_array(C.cast(houts[jlast % 2 if pipelined else 0], C.POINTER(C.c_float)), shape=(nch, 2 * olen)).view(np.complex64)
            v = sc.check_plain(eng, bank, jlast % 4, sc.sample_channels(nch, 64, seed=nch), lambda c: int(shifts[c % tile]) + (c // tile) % 7,
                               lambda c: resp[c % tile], out_host=host)
            if v["failed"]:
                raise RuntimeError("c_rt_pcie: baseband of channels %s differs from the oracle" % v["failed"][:8])
            ver = {"verified_channels": v["verified_channels"], "max_rel_err": v["max_rel_err"], "highest_channel_checked": v["highest_channel_checked"]}
        else:
            def step(j):
                assert lib.chz_input_write(eng._h, hin, Lw) == 0
                assert lib.chz_step(eng._h, j) == 0
                assert lib.chz_bank_read_pcm_flags_async(eng._h, bank.id, j % 4, 0, nch, houts[0], hfls[0]) == 0
                assert lib.chz_bank_pcm_wait(eng._h, bank.id, j % 4) == 0
                return np.ctypeslib.as_array(C.cast(houts[0], C.POINTER(C.c_ubyte)), shape=(nch, 2 * olen))
            j0 = (jlast + 1 + 7) // 8 * 8
            for j in range(jlast + 1, j0):                     # keep block numbers and the input ring in step up to the restart block
                step(j)
            ver = verify_chain(pkg, eng, bank, wl, nch, tile, shifts, resp, one, j0, 6, 64, step)
        mean = tot / blocks
        return {"channels": nch, "blocks": blocks, "worst_block_ms": worst, "mean_block_ms": mean, "sustained": bool(worst <= BLOCKTIME * 1e3),
                # (the host side of this leg is one Python thread on a shared box: how many blocks, if any, fell outside the slot, and the tail)
                "blocks_over_20ms": int(sum(t > BLOCKTIME * 1e3 for t in times)), "p99_block_ms": float(np.percentile(times, 99)) if times else None,
                "verified_channels": ver["verified_channels"], "max_rel_err": ver["max_rel_err"], "pcm_mismatches": ver.get("pcm_mismatches"),
                "verification": ver,
                "returns": "mono S16BE PCM + 1 status byte per channel (tuning, noise estimate, linear demodulator on the device)" if demod
                           else "olen complex float32 baseband samples per channel",
                "h2d_bytes_per_block": 4 * Lw, "d2h_bytes_per_block": per_ch * nch, "d2h_bytes_per_channel": per_ch,
                "d2h_GBps": per_ch * nch / (mean * 1e-3) / 1e9, "h2d_GBps_equiv": 4 * Lw / (mean * 1e-3) / 1e9,
                "loop": ("double-buffered: block j is handed to the device, then the host waits for block j-1 (two pinned output buffers); "
                         "worst/mean_block_ms = time between two consecutive completions" if pipelined else
                         "per block: H2D samples -> forward + channels [+ demodulators] -> D2H results -> host sync (no overlap between blocks)"),
                "worst_latency_ms": lat_worst if pipelined else worst}
    finally:
        eng.close()


def next_rows_leg(pkg, wl, nch, dev_index, mode="linear"):
    """SURVEY 8f's rows at scale, I/O resident in HBM: one bank of `nch` channels with fine tuning, the device-side estimate_noise()
    and a demodulator + PCM packer behind every channel; time per block with 4 blocks in flight, and each stage's own kernel time
    (instrumented run, HIP events around every launch) per channel.  Never `value`: the callers' loops either side of the path."""
    tile = 3072
    nch -= nch % tile
    P, olen = wl["P"], wl["olen"]
    eng = pkg.engine.Engine(wl["L"], wl["M"], pkg.engine.REAL, device=dev_index, ring_blocks=RING_BLOCKS)
    try:
        x = comb_ring(wl)
        eng.write(x[:RING_BLOCKS * wl["L"] - (wl["M"] - 1)]); eng.write(x[RING_BLOCKS * wl["L"] - (wl["M"] - 1):])
        bank = eng.bank(P, olen, nch)
        plan = channel_plan_config3(tile)
        resp = np.stack([pkg.filterapi.design_response(P, olen, wl["N"], True, lo, hi, 11.0) for _, lo, hi in plan[:3]])
        resp = np.ascontiguousarray(np.tile(resp, (tile // 3, 1)))
        shifts = np.array([p[0] for p in plan], np.int32)
        for c0 in range(0, nch, tile):
            bank.set_responses(c0, resp)
            bank.set_tuning(0, c0, shifts + (c0 // tile) % 7, np.full(tile, -3.3 / 12000.0))
        bank.enable_noise(wl["fs"])
        bank.set_pcm_stride(2 * olen)
        v = lambda db: 10 ** (db / 20.0)                    # radiod's defaults for a linear mode (src/modes.c:40-60,224-246)
        if mode == "fm":
            # NBFM with radiod's default SNR estimator for an open squelch (amplitude variance + fm_snr(), src/fm.c:110-129); thresholds that hold it open on noise
            one = pkg.engine.DemodParams(channels=1, env=0, agc=0, encoding=pkg.engine.PCM_S16BE, snr_squelch=0, squelch_tail=1, tuned=1, kind=1,
                                         samprate=12000.0, headroom=v(-15.0), threshold=0.0, recovery_rate=0.0, hangtime=0.0, dc_alpha=0.0,
                                         bandwidth=6000.0, shift=0.0, squelch_open=-2.0, squelch_close=-3.0, gain=1.0,
                                         deemph_rate=float(-np.expm1(-1.0 / (530.5e-6 * 12000.0))), deemph_gain=v(12.0))
        else:
            one = pkg.engine.DemodParams(channels=1, env=0, agc=1, encoding=pkg.engine.PCM_S16BE, snr_squelch=0, squelch_tail=1, tuned=1, kind=0,
                                         samprate=12000.0, headroom=v(-15.0), threshold=v(-15.0), recovery_rate=v(20.0), hangtime=1.1, dc_alpha=0.0,
                                         bandwidth=2950.0, shift=0.0, squelch_open=10 ** 0.8, squelch_close=10 ** 0.7, gain=v(50.0),
                                         pll_enable=1 if mode == "pll" else 0, pll_loop_bw=100.0 if mode == "pll" else 0.0)
        for c0 in range(0, nch, 65536):
            bank.set_demod(0, c0, [one] * min(65536, nch - c0), BLOCKTIME)
        bank.set_active(nch)
        eng.set_notches([0], 0.01)
        eng.run_blocks(0, 8)
        t = eng.run_blocks(8, 16)
        it = eng.run_blocks(0, 16, instrument=True)
        per = lambda ms, n: (ms / n * 1e6 / nch) if n else None
        # ---- verification, after the timed runs: sampled channels of the full-size bank, stage by stage against the oracle

        def step(j):
            eng.step(j)
            return None
        ver = verify_chain(pkg, eng, bank, wl, nch, tile, shifts, resp, one, 64, 6, 64, step)
        return {"channels": nch, "mode": mode, "P": P, "olen": olen,
                "verified_channels": ver["verified_channels"], "max_rel_err": ver["max_rel_err"], "pcm_mismatches": ver["pcm_mismatches"],
                "verification": ver,
                "what": "fine tuning (downconvert() tail) inside chan_ifft, estimate_noise() on the device, %s demodulator + S16 PCM behind every channel; I/O in HBM" % mode,
                "pipelined_ms_per_block": t.total_ms / 16, "fits_20ms": bool(t.total_ms / 16 <= BLOCKTIME * 1e3),
                "ns_per_channel": {"chan_ifft_with_tuning_and_power": per(it.chan_ms, it.chan_n), "noise_est": per(it.notch_ms, it.notch_n),
                                   "demodulator_and_pcm": per(it.demod_ms, it.demod_n)}}
    finally:
        eng.close()


def dropin_leg(wl, ring_host, nthreads, nblocks, env, label, paced_us=0, verify=0):
    """The same workload THROUGH ka9q-radio's filter.h (libka9q_filter_hip.so), driven radiod-style from C by tests/c/dropin_harness.c:
    a front-end thread copying samples into the host ring and calling write_rfilter(), one pthread per channel looping
    execute_filter_output() (src/radio.c:1460), a SPECTRUM block clock.  PCIe is in the loop (H2D samples, D2H outputs and --
    unless switched off -- the block spectrum).  Free-running: how fast blocks pass through the boundary."""
    import struct
    import subprocess
    import tempfile
    libdir = os.path.join(ROOT, "ka9q-radio_amd")
    plan = wl["plan"]
    if nthreads > len(plan):
        plan = (plan * ((nthreads + len(plan) - 1) // len(plan)))[:nthreads]
    plan = plan[:nthreads]
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "harness")
        subprocess.run(["gcc", "-O2", "-std=gnu11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "dropin_harness.c"), "-o", exe,
                        "-L", libdir, "-lka9q_filter_hip", "-lchz_hip", "-Wl,-rpath," + libdir, "-lpthread", "-lm"], check=True)
        open(os.path.join(tmp, "cfg.txt"), "w").write("%d %d %d %d %d %d %d\n" % (wl["L"], wl["M"], 2 if wl.get("real", True) else 1, wl["olen"], len(plan), nblocks, 65536))   # enum filtertype: 2 = REAL, 1 = COMPLEX
        with open(os.path.join(tmp, "plan.bin"), "wb") as f:
            for shift, lo, hi in plan:
                f.write(struct.pack("iiiiddddd", shift, shift, 10 ** 9, 10 ** 9, lo, hi, 11.0, lo, hi))
        np.ascontiguousarray(ring_host, np.float32 if wl.get("real", True) else np.complex64).tofile(os.path.join(tmp, "in.bin"))
        e = dict(os.environ, HARNESS_INPUT_BLOCKS=str(RING_BLOCKS), HARNESS_KEEP="0", KA9Q_HIP_PROFILE="1")
        if paced_us:
            e["HARNESS_PACED_US"] = str(int(paced_us))
        e.update(env)
        def cgroup_cpu():
            # CFS bandwidth control of the container this runs in: a quota makes the kernel stop EVERY thread of the cgroup for the rest of
            # the period once the quota is used up -- with 1000+ threads waking at once, the tens-of-ms stragglers of these legs
            out = {}
            for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
                try:
                    for ln in open(path):
                        k, _, v = ln.partition(" ")
                        out[k] = int(v)
                    break
                except Exception:
                    continue
            try:
                out["cpu.max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
            except Exception:
                pass
            return out
        cg0 = cgroup_cpu()
        t0 = time.perf_counter()
        # (a leg that hangs must cost this run minutes, not its line: 500 paced blocks take 10 s + the start of 2000 threads)
        # BENCH_DROPIN_WRAPPER="rocprofv3 --hip-trace -d DIR --": the harness under a tracer (scripts/gpu/call.sh r6c: which HIP call blocks in block 1?)
        import shlex
        r = subprocess.run(shlex.split(os.environ.get("BENCH_DROPIN_WRAPPER", "")) + [exe, tmp], capture_output=True, text=True,
                           timeout=max(180, int(3 * nblocks * BLOCKTIME) + 120), env=e)
        wall = time.perf_counter() - t0
        cg1 = cgroup_cpu()
        if r.returncode != 0:
            return {"label": label, "error": r.stderr[-300:]}
        meta = open(os.path.join(tmp, "meta.txt")).read().split()
        m = dict(zip(meta[::2], meta[1::2]))
        lat = None
        if os.path.exists(os.path.join(tmp, "latency.bin")):
            lat = np.fromfile(os.path.join(tmp, "latency.bin"), np.int64).reshape(-1, 2)
        dropped = np.fromfile(os.path.join(tmp, "dropped.bin"), np.uint8) if os.path.exists(os.path.join(tmp, "dropped.bin")) else None
        ver = None
        if verify:
            # what the channel threads were handed for the LAST block against the oracle's execute_filter_output on the host-visible
            # spectrum of that block (master->fdomain[], from the first device), sampled channels incl. both ends of every device's share
            import oracle_lib as ol
            res = np.fromfile(os.path.join(tmp, "out.bin"), np.complex64).reshape(-1, len(plan), wl["olen"])[-1]
            spec = np.fromfile(os.path.join(tmp, "spec.bin"), np.complex64).astype(np.complex128)
            rng = np.random.default_rng(len(plan))
            pick = sorted(set([0, len(plan) - 1] + [k for k in range(1023, len(plan) - 1, 1024)] + [k for k in range(1024, len(plan), 1024)] +
                              [int(v) for v in rng.choice(len(plan), min(verify, len(plan)), replace=False)]))
            bad, worst = [], 0.0
            peak = float(np.abs(spec).max())
            for i in pick:
                sh, lo, hi = plan[i]
                resp = ol.set_filter(wl["P"], wl["olen"], wl["N"], wl.get("real", True), lo, hi, 11.0)
                want = ol.channel(spec, ol.REAL if wl.get("real", True) else ol.COMPLEX, wl["P"], wl["olen"], sh, resp)
                err = float(np.sqrt(np.mean(np.abs(res[i] - want) ** 2))); rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
                worst = max(worst, err / max(rms, 1e-30))
                if not err <= 1e-5 * rms + 2e-8 * peak * float(np.linalg.norm(resp)):
                    bad.append(i)
            ver = {"verified_channels": len(pick), "mismatched_channels": len(bad), "max_rel_err": worst, "first_mismatches": bad[:8]}
    prof = [ln for ln in r.stderr.splitlines() if ln.startswith("filter_hip profile:")]
    pv = {}
    if prof:
        for kv in prof[-1].split()[2:]:
            k, _, v = kv.partition("=")
            try:
                pv[k] = float(v)
            except ValueError:
                pass
    first_blocks = [ln for ln in r.stderr.splitlines() if ln.startswith("filter_hip first blocks:")]
    el = float(m["elapsed_s"])
    paced = None
    if paced_us and lat is not None:
        # per block: from write_rfilter() returning for the block's last samples to the LAST channel thread leaving execute_filter_output() with it
        served = lat[8:]                                    # the first blocks carry one-time set-up (first launches, first touch of every buffer)
        ok = served[(served[:, 0] >= 0) & (served[:, 1] == len(plan))][:, 0] / 1e6
        first8 = lat[:8][lat[:8, 0] >= 0][:, 0] / 1e6
        paced = {"block_period_ms": paced_us / 1e3,
                 "latency_ms": {"p50": float(np.percentile(ok, 50)) if ok.size else None, "p99": float(np.percentile(ok, 99)) if ok.size else None,
                                "max": float(ok.max()) if ok.size else None, "max_first_8_blocks": float(first8.max()) if first8.size else None},
                 # flat copies for the headline; block 0 is the first block any slave ever fetched -- it must be an ordinary block
                 "p50_ms": float(np.percentile(ok, 50)) if ok.size else None, "p99_ms": float(np.percentile(ok, 99)) if ok.size else None,
                 "max_ms": float(ok.max()) if ok.size else None, "max_first_8_blocks_ms": float(first8.max()) if first8.size else None,
                 "block0_ms": float(lat[0, 0] / 1e6) if lat[0, 0] >= 0 else None, "block0_slaves_served": int(lat[0, 1]),
                 "drops_first_8_blocks": int(dropped.reshape(-1, len(plan))[:8].sum()) if dropped is not None and dropped.size == nblocks * len(plan) else None,
                 "blocks_served_to_every_channel": int(ok.size), "blocks_measured": int(served.shape[0]),
                 "blocks_later_than_one_period": int((ok > paced_us / 1e3).sum()),
                 "slowest_blocks": [{"block": int(i), "latency_ms": float(lat[i, 0] / 1e6), "slaves_served": int(lat[i, 1])}
                                    for i in np.argsort(-lat[:, 0])[:4]],
                 "block_drops": int(m["drops"]), "skipped_blocks": int(m.get("skipped", 0)),
                 "front_end_worst_wakeup_lateness_ms": int(m.get("fe_late_worst_ns", 0)) / 1e6,
                 "front_end_longest_write_rfilter_ms": int(m.get("fe_call_worst_ns", 0)) / 1e6,
                 "gpu_busy_pct": 100.0 * (int(m["avg_block_ns"]) / 1e3) / paced_us,
                 "definition": "front end on its own wall clock (absolute deadlines, one 20 ms block per 20 ms, chunks as they would arrive from the A/D), never "
                               "waits for a channel; latency = last sample of the block handed to write_rfilter -> the last of the channel threads has the block; "
                               "gpu_busy = mean enqueue-to-callback time of a block (H2D + kernels + D2H) / period"}
    dev_counts = [int(v) for v in m.get("dev_counts", "-").split(":")] if m.get("dev_counts", "-") != "-" else None
    return {"label": label, "threads": len(plan), "blocks": nblocks, "env": env, "paced": paced,
            "devices": int(m.get("devices", 1)), "slaves_per_device": dev_counts,
            "verified_channels": ver["verified_channels"] if ver else None, "mismatched_channels": ver["mismatched_channels"] if ver else None,
            "verification": ver,
            "ms_per_block": el / nblocks * 1e3, "realtime_margin": BLOCKTIME / (el / nblocks),
            "worst_block_gap_ms": int(m["worst_gap_ns"]) / 1e6, "mean_block_gap_ms": int(m["mean_gap_ns"]) / 1e6,
            "drops": int(m["drops"]),
            "device_block_us_avg": int(m["avg_block_ns"]) / 1e3,
            "device_block_us_max_incl_first_blocks": int(m["max_block_ns"]) / 1e3,    # Max_fft_time: the first blocks carry one-time set-up (code objects, notch upload, first launches)
            "device_block_us_max_after_8_blocks": pv.get("dev_block_max_us_after_8"),
            "front_end_us_per_block": {"copying_samples_into_the_ring": int(m["fe_copy_ns"]) / 1e3 / nblocks,
                                       "inside_write_rfilter": int(m["fe_call_ns"]) / 1e3 / nblocks,
                                       "waiting_for_the_slowest_channel": int(m["fe_wait_ns"]) / 1e3 / nblocks},
            "host_profile": {"execute_filter_input_us_per_block": pv.get("input_us"), "of_which_waiting_for_the_device_us": pv.get("input_wait_us"),
                             "callback_to_slave_has_its_block_us_mean": pv.get("consume_mean_us"), "callback_to_slave_has_its_block_us_worst": pv.get("consume_worst_us"),
                             "callback_to_slave_has_its_block_us_worst_after_8_blocks": pv.get("consume_worst_after_8_us"),
                             "staged_hits": pv.get("hits"), "misses": pv.get("misses")},
            "host_cgroup": {"cpu.max": cg1.get("cpu.max"), "periods_throttled_during_the_leg": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                            "throttled_ms_during_the_leg": (cg1.get("throttled_usec", cg1.get("throttled_time", 0)) - cg0.get("throttled_usec", cg0.get("throttled_time", 0))) / 1e3,
                            # CPU seconds the container used per second of the leg (the quota is cpu.max's first number / its second)
                            "cpus_used_mean": round((cg1.get("usage_usec", 0) - cg0.get("usage_usec", 0)) / 1e6 / wall, 2) if wall > 0 and "usage_usec" in cg1 else None},
            "first_blocks_profile": first_blocks[-1][len("filter_hip first blocks:"):].strip() if first_blocks else None,
            "first_8_blocks_latency_ms": [round(float(v) / 1e6, 3) for v in lat[:8, 0]] if lat is not None else None,
            "process_wall_s": wall}



HEADLINE_MAX_BYTES = 6000               # the driver parses the LAST stdout line; round 4's 23 KB line came back unparsed


def _clean(x, sig=6):
    """JSON-strict copy: numpy scalars -> python, non-finite floats -> None, floats rounded to `sig` significant digits."""
    if isinstance(x, dict):
        return {str(k): _clean(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v, sig) for v in x]
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x)) if sig else x
    return x


def _pick(d, *keys):
    return {k: d[k] for k in keys if d is not None and k in d} if d else None


def headline(out, detail_path):
    """The ONE line the driver parses: the contract keys, roofline, cpu_baseline and one-number summaries of the other legs.
    Everything else (probes, ladders, definitions, notes) is in the detail file."""
    roof, cpu, crt = out.get("roofline"), out.get("cpu_baseline"), out.get("c_rt")
    h = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data")}
    h["config"] = _pick(out["config"], "workload", "baseline_config", "channels_total", "P", "olen", "N", "L", "M", "plan", "lanes", "value_is")
    h.update(_pick(out, "reps", "regions", "ms_per_step_min", "ms_per_step_max", "drained_k_step_region_ms_per_step", "blocks_per_s",
                   "step_algorithmic_GBps", "exchange_mode"))
    if out.get("legs"):
        h["legs"] = {k: _pick(v, "ms_per_step", "value") for k, v in out["legs"].items()}
    if roof:
        r = _pick(roof, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "rocprof_frac", "algorithmic_bytes_per_block",
                  "forward_us_per_block", "launches_timed", "kernels_us")
        r["kernel"] = "fwd_first_real + fwd_cols + fwd_rows" if "fwd_cols" in (roof.get("kernels_us") or {}) else "fwd_first_real + fwd_rows"
        r["pipelined"] = _pick(roof.get("pipelined"), "frac", "forward_us_per_block", "lanes")
        r["streamed"] = _pick(roof.get("streamed"), "frac", "forward_us_per_block", "input_ring_MB", "error")
        m = roof.get("profiles_match_this_tree") or {}
        r["profiles_match_this_tree"] = None if m.get("pmc_forward.json") is None else bool(m.get("pmc_forward.json") and m.get("rocprof_kernels.json"))
        if roof.get("batched"):
            r["batched"] = _pick(roof["batched"], "blocks_per_launch", "forward_us_per_block", "frac")
        h["roofline"] = r
    else:
        h["roofline"] = None
    if cpu:
        c = _pick(cpu, "value", "unit", "cores", "kind", "ms_per_block", "fwd_fft_ms", "us_per_channel_block", "real_time", "host_cores",
                  "throughput_bound_channels")
        c["sample"] = cpu["sample"][:160]
        cc = cpu.get("c_rt_cpu")
        c["c_rt_cpu"] = _pick(cc, "channels", "sustained", "repeats", "blocks_per_repeat", "first_failed_channels")
        h["cpu_baseline"] = c
    else:
        h["cpu_baseline"] = None
    if crt:
        h["c_rt"] = _pick(crt, "channels", "sustained", "worst_block_ms", "mean_block_ms", "blocks", "verified_channels", "max_rel_err",
                          "mean_crossing_channels", "rungs", "dram_side_GBps", "dram_side_frac_of_hbm_peak", "gpus", "error")
    else:
        h["c_rt"] = None

    def many(name, f):
        v = out.get(name)
        h[name] = None if v is None else [({"error": x["error"][:120]} if "error" in x else f(x)) for x in v]
    many("dropin", lambda x: _pick(x, "threads", "ms_per_block", "drops"))
    many("dropin_paced", lambda x: dict(_pick(x, "threads", "drops"),
                                        **{k: (x.get("paced") or {}).get(k) for k in ("drops_first_8_blocks", "block0_ms", "max_first_8_blocks_ms", "p50_ms", "p99_ms", "max_ms")}))
    many("dropin_sharded", lambda x: dict(_pick(x, "threads", "devices", "devices_distinct", "slaves_per_device", "drops", "verified_channels", "mismatched_channels"),
                                          **{k: (x.get("paced") or {}).get(k) for k in ("drops_first_8_blocks", "block0_ms", "p99_ms", "max_ms")}))
    many("c_rt_pcie", lambda x: _pick(x, "channels", "sustained", "worst_block_ms", "d2h_bytes_per_channel", "pcm_mismatches"))
    many("next_rows", lambda x: dict(_pick(x, "mode", "channels", "pcm_mismatches", "verified_channels"), ms_per_block=x.get("pipelined_ms_per_block")))
    h.update(_pick(out, "leg_seconds", "rccl_ranks", "quick", "leg_timeouts", "headline_from", "exchange_errors"))
    h["detail"] = detail_path
    h = _clean(h, 5)
    line = json.dumps(h, allow_nan=False, separators=(",", ":"))
    if len(line) > HEADLINE_MAX_BYTES:                      # never hand the driver a line it cannot parse: shed the optional summaries
        for k in ("legs", "leg_seconds", "dropin", "c_rt_pcie", "next_rows", "dropin_sharded", "dropin_paced"):
            h.pop(k, None)
            line = json.dumps(h, allow_nan=False, separators=(",", ":"))
            if len(line) <= HEADLINE_MAX_BYTES:
                break
    return line


def self_spawn(args):
    """--gpus N > 1 without a launcher: become `torch.distributed.run` with one rank per GPU."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; refusing to run fewer ranks than asked" % (args.gpus, have))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, default=0, help="BASELINE config 1, 2, 3, 4 or 5 (default: 3 at --gpus 1, 4 otherwise)")
    ap.add_argument("--exchange", default=os.environ.get("BENCH_EXCHANGE", "auto"),
                    help="config 4, how the block reaches the ranks: auto | subband | broadcast (spectrum over RCCL) | samples (the block's new samples "
                         "over RCCL, every rank transforms) | replicate (no collective)")
    ap.add_argument("--channels", type=int, default=0, help="channels per GPU (default: the config's)")
    ap.add_argument("--plan", default="", help="forward plan override, e.g. 144x100x225")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="keep repeating the K-step region until this much is measured")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay one hipGraph per ring cycle instead of eager launches (N=1)")
    ap.add_argument("--no-crt", action="store_true", help="skip the C_rt leg (one large bank, every block inside 20 ms)")
    ap.add_argument("--crt-channels", type=int, default=0, help="ONE rung of this many channels instead of the search")
    ap.add_argument("--crt-blocks", type=int, default=500)
    ap.add_argument("--no-next-rows", action="store_true", help="skip the SURVEY 8f chain leg (tuning + noise estimate + demodulator behind 1.5 M channels)")
    ap.add_argument("--next-rows-channels", type=int, default=1_500_000)
    ap.add_argument("--next-rows-modes", default="linear,pll,fm", help="which demodulators the 8f chain leg runs (profiling passes take one)")
    ap.add_argument("--crt-ladder", default="", help="comma-separated channel counts (millions) of an explicit C_rt ladder; default: a search around the count whose mean block takes 20 ms")
    ap.add_argument("--no-dropin", action="store_true", help="skip the legs through the filter.h drop-in")
    ap.add_argument("--dropin-blocks", type=int, default=500)
    ap.add_argument("--no-dropin-paced", action="store_true", help="skip the wall-clock-paced legs through the filter.h drop-in (500 blocks = 10 s each)")
    ap.add_argument("--no-crt-pcie", action="store_true", help="skip the C_rt probes with the host link in the loop")
    ap.add_argument("--crt-pcie-blocks", type=int, default=500)
    ap.add_argument("--crt-shared", type=int, default=0,
                    help="also run the C_rt leg with this many channels SHARING their response rows (0 = skip; config 3, 1 GPU)")
    ap.add_argument("--detail", default=os.environ.get("BENCH_DETAIL", ""), help="where the full detail object goes (default gpurun_out/bench_detail.json); "
                    "stdout's last line is the compact headline (< 6 KB)")
    ap.add_argument("--quick", action="store_true",
                    help="iteration mode: the headline and the roofline object only (no C_rt ladder, PCIe probes, 8f chain, drop-in or CPU legs): "
                         "< 30 s of GPU time.  The driver's command (no --quick) keeps the full line")
    args = ap.parse_args()
    if args.quick:
        args.no_crt = args.no_next_rows = args.no_dropin = args.no_dropin_paced = args.no_crt_pcie = args.no_cpu_baseline = True
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("bad --gpus/--steps/--warmup")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    emit_box = {}                         # the function that prints the line, once it exists (defined below, after the engine is up)
    # ---- per-leg watchdog.  Every leg announces itself through progress(); a leg that is still running when its budget is spent is reported in
    # the line ("leg_timeouts") and the run ENDS there with rc 0 and whatever the finished legs measured: a thread, not a signal (the main thread
    # may be blocked inside a HIP call, and a blocked ctypes / torch call does not run Python signal handlers).  Budgets: several times what the
    # leg takes on a healthy box (profiles/r05_bench_detail.json leg_seconds), BENCH_LEG_BUDGET_SCALE scales them, BENCH_LEG_BUDGET_S replaces them.
    LEG_BUDGET_S = {"startup": 420, "headline": 120, "exchange": 90, "roofline": 120, "c_rt": 180, "cpu_baseline": 150, "dropin": 150, "dropin_paced": 150,
                    "dropin_sharded": 90, "c_rt_pcie": 200, "next_rows": 240}
    wd = {"leg": None, "t0": 0.0, "budget": 0.0}

    def progress(what):
        if rank == 0:
            print("bench.py [%6.1f s] %s" % (time.perf_counter() - t_start, what), file=sys.stderr, flush=True)
        key = what.split()[0].rstrip(":")
        key = {"dropin": "dropin"}.get(key, key)
        budget = float(os.environ.get("BENCH_LEG_BUDGET_S", 0)) or LEG_BUDGET_S.get(key, 240) * float(os.environ.get("BENCH_LEG_BUDGET_SCALE", "1"))
        # the other ranks give rank 0 the time to print before the launcher sees anybody exit
        wd["t0"], wd["budget"], wd["leg"] = time.perf_counter(), budget + (0 if rank == 0 else 20), what
        if os.environ.get("BENCH_TEST_HANG_LEG") and what.startswith(os.environ["BENCH_TEST_HANG_LEG"]):
            while True:                              # (tests: a leg that never comes back)
                time.sleep(0.2)

    def watchdog():
        while True:
            time.sleep(0.5)
            leg = wd["leg"]
            if leg is None or time.perf_counter() - wd["t0"] <= wd["budget"]:
                continue
            print("bench.py: leg '%s' has been running for %.0f s (budget %.0f s): giving up on it -- printing the line with what was measured and ending the run"
                  % (leg, time.perf_counter() - wd["t0"], wd["budget"]), file=sys.stderr, flush=True)
            try:
                import faulthandler
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)      # where every thread of this process sits
            except Exception:
                pass
            try:
                if emit_box.get("fn"):
                    emit_box["fn"]([leg])
                elif rank == 0:                      # nothing has been measured yet (the runtime / the engine never came up): still ONE strict line, rc 0
                    print(json.dumps({"metric": "channels sustained @129.6 MS/s input (real-time-equivalent: channel-blocks/s / 50)", "value": None, "unit": "channels",
                                      "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                                      "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "not reached"}, "roofline": None, "cpu_baseline": None,
                                      "leg_timeouts": [leg], "headline_from": None}))
            finally:
                sys.stdout.flush(); sys.stderr.flush()
                os._exit(0)
    threading.Thread(target=watchdog, daemon=True, name="bench-watchdog").start()
    progress("startup")                   # importing torch on a fresh box, the process group, the engine and its bank, the input ring
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")     # gloo: several ranks on ONE GPU (control-flow checks only)
    if backend == "nccl" and torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: GPU %d is not visible (%d devices)" % (rank, local_rank, torch.cuda.device_count()))
    dev_index = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(dev_index)
    # BENCH_FORCE_DIST=1: run the multi-rank code path (process group, RCCL exchange) with ONE rank on a 1-GPU box
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=rank, world_size=world)      # "nccl" IS RCCL on ROCm; control plane only

    config = args.config or (4 if use_dist else 3)
    wl = workload_for(config, rank, world, args.channels)
    exch = None
    if config == 4:
        exch = args.exchange
        if exch == "auto" and not use_dist:
            exch = "replicate"                                  # one rank: nothing to exchange (BENCH_FORCE_DIST=1 keeps the collective)
        if exch not in ("auto", "subband", "broadcast", "samples", "replicate"):
            raise SystemExit("--exchange must be auto, subband, broadcast, samples or replicate")
        if backend != "nccl" and exch != "replicate":
            raise SystemExit("the RCCL exchange needs one GPU per rank (backend nccl)")

    pkg = ge.load()
    real = wl["real"]
    eng = pkg.engine.Engine(wl["L"], wl["M"], pkg.engine.REAL if real else pkg.engine.COMPLEX, device=dev_index, plan=args.plan, ring_blocks=RING_BLOCKS)

    # ---- inputs resident in HBM before anything is timed
    ring_host = siggen_ring(oracle_lib, wl["fs"], wl["seed"], wl["L"], real)
    # the device ring starts with the write position M-1 ahead (zeros before time 0); fill it to the brim
    eng.write(ring_host[:RING_BLOCKS * wl["L"] - (wl["M"] - 1)])
    eng.write(ring_host[RING_BLOCKS * wl["L"] - (wl["M"] - 1):])       # wraps: ring now holds the cyclic 8-block stream
    nch, P, olen, plan = wl["nch"], wl["P"], wl["olen"], wl["plan"]
    bank = eng.bank(P, olen, nch)
    resp = np.stack([pkg.filterapi.design_response(P, olen, wl["N"], real, lo, hi, 11.0) for _, lo, hi in plan])
    bank.set_responses(0, resp)
    bank.set_shifts(0, np.array([p[0] for p in plan], np.int32))
    bank.set_active(nch)
    if os.environ.get("BENCH_NO_NOTCH") != "1":             # (A/B knob for scripts; the bench line always has the notch)
        eng.set_notches([0], 0.01)                           # DC notch is always present (src/radio.c:601-620)

    # ---- the RCCL communicator lives behind the engine's C ABI; torch.distributed only ships the id and keeps time
    comm = None
    rows = None
    mode = None
    if config == 4 and exch != "replicate":
        uid = [pkg.engine.comm_unique_id() if rank == 0 else None]
        if use_dist:
            dist.broadcast_object_list(uid, src=0)
        comm = pkg.engine.Comm(rank, world, uid[0], device=dev_index)
        na, pitch, _off = eng.spec_layout
        nrows = (wl["bins"] + na - 1) // na
        mine = pkg.sharding.needed_rows([p[0] for p in plan], P, wl["bins"], na)
        all_rows = [mine]
        if use_dist:
            all_rows = [None] * world
            dist.all_gather_object(all_rows, mine)
        mode = exch
        if exch == "auto":
            mode = pkg.sharding.plan_exchange(all_rows, nrows) if world > 1 else "broadcast"
        rows = ([r[0] for r in all_rows], [r[1] for r in all_rows])

    def barrier():
        if use_dist:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def timed(run_k, job, nblocks):
        barrier()
        t0 = time.perf_counter()
        last = run_k(job, nblocks)                            # returns after this rank's streams have drained
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        return max_over_ranks(el), last

    def measure(run_k):
        """W warm-up steps; ONE drained K-step region (the literal contract); then regions of `reps` back-to-back
        repetitions of the K-step loop (reps*K blocks issued without draining in between, each region bracketed by
        barrier + synchronize, max over ranks), enough of them to cover --min-seconds.  The median region gives the
        steady-state ms_per_step.  Returns (seconds per K steps, [seconds per K steps of every region], last timing
        object, reps, seconds of the single drained K-step region)."""
        job = 0
        if args.warmup:
            run_k(job, args.warmup); job += args.warmup
        single, _ = timed(run_k, job, args.steps); job += args.steps
        # same decisions on every rank: they are made from the reduced time
        reps = int(min(20000, max(1, np.ceil(0.1 / max(single, 1e-6)))))          # one region ~ 0.1 s
        nreg = int(min(50, max(5, np.ceil(args.min_seconds / 0.1))))
        times, last = [], None
        for _ in range(nreg):
            el, last = timed(run_k, job, reps * args.steps)
            times.append(el / reps)
            job += reps * args.steps
        return statistics.median(times), times, last, reps, single

    legs = {}
    leg_seconds = {}                      # wall seconds of every leg of this run (rank 0): the time budget of the bench line
    # ---- everything the line is built from exists from here on (None until its leg has run), so that the line can be printed at ANY moment:
    # by the normal end of the run, or by the watchdog when a leg overruns its time budget (round 5: one bench run with every stream
    # CU-masked never came back and cost the whole record -- a hung leg must cost its own numbers, not the line)
    roof = cpu = crt = crt_shared = dropin = dropin_paced = dropin_sharded = crt_pcie = next_rows = None
    rccl_ranks, ranks_info, eng_lanes, eng_plan = 0, None, eng.lanes, eng.plan
    main_leg = intended_leg = None
    exchange_errors = {}
    emitted = threading.Lock()

    def emit(leg_timeouts):
        if not emitted.acquire(blocking=False):
            return
        elapsed, times, timing, reps, single = legs[main_leg] if main_leg in legs else (None, None, None, None, None)
        if rank == 0:
            def leg_obj(name, leg):
                el, ts, tm, rp, sg = leg
                return {"ms_per_step": el * 1e3 / args.steps, "value": nch * world * BLOCKTIME / (el / args.steps), "reps": rp, "regions": len(ts),
                        "ms_per_step_min": min(ts) * 1e3 / args.steps, "ms_per_step_max": max(ts) * 1e3 / args.steps,
                        "drained_k_step_region_ms_per_step": sg * 1e3 / args.steps,
                        "gpu_event_ms_per_step": tm.total_ms / tm.blocks, "host_enqueue_ms_per_step": tm.enqueue_ms / tm.blocks}
            have = elapsed is not None
            ms_per_step = elapsed * 1e3 / args.steps if have else None
            total_ch = nch * world
            value = total_ch * BLOCKTIME / (elapsed / args.steps) if have else None
            fwd_copies = world if (config == 5 or main_leg == "replicate") else 1
            step_bytes = fwd_copies * fwd_bytes(wl["N"], wl["real"]) + total_ch * chan_bytes(P, olen)
            exchange_desc = None
            if config == 4:
                exchange_desc = {
                    "subband": "RCCL grouped ncclSend/ncclRecv of the spectrum rows each rank's channels read (chz_spectrum_exchange_rows), on the slot's HIP stream",
                    "broadcast": "RCCL ncclBroadcast of the whole spectrum slot (chz_spectrum_broadcast), on the slot's HIP stream",
                    "samples": "RCCL ncclBroadcast of the block's L new samples into every rank's input ring on the communicator's stream; every rank runs the forward transform itself",
                    "replicate": "none: every rank transforms its own HBM-resident copy of the samples",
                }[main_leg] + "; %d rank(s)" % world
            elif config == 5:
                exchange_desc = "none (replicas only): %d independent front ends" % world
            out = {
                "metric": "channels sustained @%.1f MS/s input (real-time-equivalent: channel-blocks/s / 50)" % (wl["fs"] / 1e6),
                "value": value, "unit": "channels", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": wl["name"] + (", 1 MI355X" if world == 1 else ""), "baseline_config": config,
                           "channels_total": total_ch, "P": P, "olen": olen, "N": wl["N"], "L": wl["L"], "M": wl["M"],
                           "launch": ("hipGraph replays of >= %s blocks" % os.environ.get("CHZ_GRAPH_BLOCKS", "32") if graph else "eager") + ", %d HIP streams, notch recurrence ordered by %s"
                                     % (eng_lanes, "HIP events" if os.environ.get("CHZ_NOTCH_ORDER") == "event" else "device ticket"),
                           "plan": eng_plan, "lanes": eng_lanes,
                           "value_is": "free-running channel-blocks/s / 50 of this configuration, inputs resident in HBM; the literal simultaneous-channel count is c_rt.channels"},
                "timing": "steady state: median over `regions` timed regions, each = `reps` back-to-back repetitions of the K-step loop "
                          "(reps*K blocks, barrier+sync on both sides, max over ranks); drained_k_step_region = ONE K-step region on its own, "
                          "pipeline fill and drain included",
                "reps": reps, "regions": len(times) if have else None, "ms_per_step_min": min(times) * 1e3 / args.steps if have else None,
                "ms_per_step_max": max(times) * 1e3 / args.steps if have else None,
                "drained_k_step_region_ms_per_step": single * 1e3 / args.steps if have else None,
                "exchange": exchange_desc, "exchange_mode": (main_leg if config == 4 else None),
                "legs": {k: leg_obj(k, v) for k, v in legs.items() if k != main_leg} or None,
                "blocks_per_s": args.steps / elapsed if have else None, "realtime_margin": BLOCKTIME / (elapsed / args.steps) if have else None,
                "step_algorithmic_GBps": step_bytes / (elapsed / args.steps) / 1e9 if have else None,
                "gpu_event_ms_per_step": timing.total_ms / timing.blocks if have else None,
                "host_enqueue_ms_per_step": timing.enqueue_ms / timing.blocks if have else None,
                # legs that overran their time budget (the watchdog printed this line and ended the run there); headline_from says which
                # leg `value` comes from when the intended one never finished
                "leg_timeouts": list(leg_timeouts) or None, "headline_from": main_leg, "intended_headline_leg": intended_leg,
                "exchange_errors": exchange_errors or None,
                "roofline": roof, "cpu_baseline": cpu, "c_rt": crt, "c_rt_shared_responses": crt_shared,
                "dropin": dropin, "dropin_paced": dropin_paced, "dropin_sharded": dropin_sharded, "c_rt_pcie": crt_pcie, "next_rows": next_rows,
                "leg_seconds": {k: round(v, 1) for k, v in leg_seconds.items()},
                "rccl_ranks": rccl_ranks, "ranks": ranks_info, "quick": bool(args.quick),
            }
        if rank == 0:
            # RCCL prints a banner through C stdio: flush it first so the JSON is the LAST line of stdout
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            # the detail (every probe, ladder, definition and note) goes to a side file; stdout's LAST line is the compact strict-JSON headline
            detail_path = args.detail or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
            try:
                os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
                with open(detail_path, "w") as f:
                    json.dump(_clean(out, 0), f, allow_nan=False)
                    f.write("\n")
                shown = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT + os.sep) else detail_path
            except OSError as ex:
                print("bench.py: could not write the detail file %s: %s" % (detail_path, ex), file=sys.stderr)
                shown = None
            sys.stdout.write(headline(out, shown) + "\n")
            sys.stdout.flush()

    t_leg = time.perf_counter()
    graph = bool(args.graph) and not use_dist
    emit_box["fn"] = emit
    progress("headline")
    intended_leg = None
    if comm is not None:
        def run_sharded(m):
            return lambda job0, k: eng.run_blocks_sharded(comm, job0, k, root=0, rows=(rows if m == "subband" else None), samples=(m == "samples"))
        # the leg without a collective runs FIRST: if an exchange leg never comes back (the first real multi-GPU run: collectives of one communicator
        # on four lane streams, a notch ticket spinning beside RCCL's kernels -- DESIGN.md section 6), the line still carries a measured value
        # (headline_from = "replicate", leg_timeouts says which exchange hung) instead of nothing
        intended_leg = mode
        legs["replicate"] = measure(lambda job0, k: eng.run_blocks(job0, k))
        main_leg = "replicate"
        # an exchange that FAILS (a HIP / RCCL error out of the engine) is reported and the next hand-over becomes the headline: the
        # ladder is the intended one, then samples (one stream, no collective on the lane streams), then whichever is left
        ladder = [mode] + [m for m in ("samples", "broadcast", "subband") if m != mode]
        if not (world > 1 or os.environ.get("BENCH_ALL_EXCHANGES") == "1"):
            ladder = ladder[:1]
        for m in ladder:
            progress("exchange " + m)
            try:
                legs[m] = measure(run_sharded(m))
                if main_leg == "replicate":
                    main_leg = m
            except Exception as ex:                                  # every rank sees the same failure or the next barrier times the leg out
                exchange_errors[m] = str(ex)[:300]
    else:
        intended_leg = "replicate" if config == 4 else "local"
        legs[intended_leg] = measure(lambda job0, k: eng.run_blocks(job0, k, graph=graph))
        main_leg = intended_leg
    elapsed, times, timing, reps, single = legs[main_leg]
    leg_seconds["headline_and_exchange_legs"] = time.perf_counter() - t_leg
    t_leg = time.perf_counter()

    progress("roofline: per-kernel times")
    # ---- per-kernel durations with HIP events on the launch stream (eager, instrumented; >= 200 launches whatever K is)
    roof = None
    if rank == 0:
        nk = max(200, min(args.steps, 1000))
        eng.run_blocks(0, nk, graph=False, instrument=True)          # discarded: clocks settle after the load change
        it = eng.run_blocks(0, nk, graph=False, instrument=True)
        kern = {}
        for name, ms, n in (("fwd_first_real", it.first_ms, it.first_n), ("fwd_cols", it.cols_ms, it.cols_n),
                            ("fwd_rows", it.rows_ms, it.rows_n), ("notch_fix", it.fix_ms, it.fix_n), ("chan_ifft", it.chan_ms, it.chan_n)):
            if n:
                kern[name] = ms / n * 1e3       # microseconds per launch
        # the same three kernels as the engine really runs them: blocks of 4 HIP streams in flight together (no channel bank)
        bank.set_active(0)
        eng.run_blocks(0, 400)
        fo = eng.run_blocks(400, 2000)
        bank.set_active(nch)
        fwd_pipe_us = fo.total_ms / fo.blocks * 1e3
        # ... and with the INPUT genuinely streamed from HBM: the 8-block ring above (83 MB) lives in the 256 MiB Infinity Cache, so the figures
        # above are fabric rates; a second engine with a 48-block ring (498 MB of samples at 129.6 MS/s, read once per 48 blocks) cannot keep it there
        streamed = None
        if real and config in (3, 4, 5) and os.environ.get("BENCH_NO_STREAMED") != "1":
            try:
                sb = 48
                e2 = pkg.engine.Engine(wl["L"], wl["M"], pkg.engine.REAL, device=dev_index, plan=args.plan, ring_blocks=sb)
                try:
                    for k in range(sb // RING_BLOCKS):
                        e2.write(ring_host if k else ring_host[:RING_BLOCKS * wl["L"] - (wl["M"] - 1)])
                    e2.write(ring_host[RING_BLOCKS * wl["L"] - (wl["M"] - 1):])
                    if os.environ.get("BENCH_NO_NOTCH") != "1":
                        e2.set_notches([0], 0.01)
                    e2.run_blocks(0, 480)
                    f2 = e2.run_blocks(480, 1920)
                    us = f2.total_ms / f2.blocks * 1e3
                    streamed = {"ring_blocks": sb, "input_ring_MB": sb * wl["L"] * 4 / 1e6, "blocks_timed": f2.blocks, "forward_us_per_block": us,
                                "achieved": fwd_bytes(wl["N"], real) / (us * 1e-6) / 1e9, "frac": fwd_bytes(wl["N"], real) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "note": "forward transform alone, four blocks in flight, the samples read from a ring larger than the Infinity Cache"}
                finally:
                    e2.close()
            except Exception as ex:
                streamed = {"error": str(ex)[:300]}
        Ra = eng.axes[0] // 2 + 1 if real else eng.axes[0]
        inner_bytes = Ra * eng.axes[1] * eng.axes[2] * 8
        own = {"fwd_first_real": (4 if real else 8) * wl["N"] + inner_bytes, "fwd_cols": 2 * inner_bytes,
               "fwd_rows": inner_bytes + 8 * wl["bins"], "chan_ifft": nch * chan_bytes(P, olen)}
        if eng.axes[1] <= 1:
            own.pop("fwd_cols")                      # two-axis plan: no middle pass
        fwd_us = sum(kern.get(k, 0.0) for k in ("fwd_first_real", "fwd_cols", "fwd_rows"))
        fb = fwd_bytes(wl["N"], real)
        achieved = fb / (fwd_us * 1e-6) / 1e9 if fwd_us else 0.0
        passes = 3 if eng.axes[1] > 1 else 2
        traffic = pmc_traffic_bytes() if config in (3, 4, 5) else None
        rp = rocprof_kernel_us() if config in (3, 4, 5) else None
        roof = {
            "bound": "hbm", "kernel": ("forward transform = fwd_first_real + fwd_cols + fwd_rows (one launch each per block)" if real else
                                       "forward transform of a COMPLEX master: first axis through fwd_cols (reported under fwd_first_real) + fwd_rows%s"
                                       % (" + fwd_cols" if passes == 3 else "")),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": ("committed profile, not this run: profiles/pmc_forward.json (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate "
                               "passes over this same command, FETCH_SIZE doubled as the gfx950 guide prescribes)") if traffic else None,
            # do the committed profiles belong to the kernels this run executes?  (same fingerprint of the kernel sources)
            "profiles_match_this_tree": {"pmc_forward.json": profile_sha("pmc_forward.json") == kernel_sources_sha16() if traffic else None,
                                         "rocprof_kernels.json": profile_sha("rocprof_kernels.json") == kernel_sources_sha16() if rp else None,
                                         "kernel_sources_sha16": kernel_sources_sha16()},
            "rocprof": rp,
            "rocprof_frac": (fb / (rp["forward_us_1_stream"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp and rp.get("forward_us_1_stream") else None,
            "frac_of_measured_copy_rate": achieved / COPY_RATE_GBS,
            "structural_cap": "%d HBM passes: at most 1/%d of peak on algorithmic bytes" % (passes, passes),
            "algorithmic_bytes_per_block": fb, "forward_us_per_block": fwd_us, "launches_timed": nk,
            "kernels_us": kern,
            # one pass moves 13 MB in and 13 MB out -- about what the memory system must have in flight to run at full rate --
            # so a pass on its own is one generation of wavefronts: read burst, butterflies, write burst, nothing overlapping.
            # In the engine, passes of 4 blocks overlap; this is the forward transform alone measured that way (2000 blocks).
            "pipelined": {"forward_us_per_block": fwd_pipe_us, "achieved": fb / (fwd_pipe_us * 1e-6) / 1e9,
                          "frac": fb / (fwd_pipe_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "lanes": eng.lanes, "blocks_timed": fo.blocks,
                          "traffic_GBps": (traffic / (fwd_pipe_us * 1e-6) / 1e9) if traffic else None,
                          "measured_copy_rate_GBps": COPY_RATE_GBS},
            "streamed": streamed,
            "kernels_own_GBps": {k: own[k] / (kern[k] * 1e-6) / 1e9 for k in own if k in kern and kern[k] > 0},
        }

    leg_seconds["roofline_kernels"] = time.perf_counter() - t_leg
    t_leg = time.perf_counter()
    progress("c_rt")
    # ---- C_rt leg: one bank of millions of channels of the workload's kind; every block on its own <= 20 ms
    crt = None
    if not args.no_crt and config in (3, 4):
        if args.crt_ladder:
            crt_n = [int(float(x) * 1e6) for x in args.crt_ladder.split(",")]
        elif args.crt_channels:
            crt_n = [args.crt_channels]
        else:
            crt_n = None                                     # search (crt_leg): never "nothing sustained" on a slower box
        if comm is not None:
            # the big bank's channels span the whole spectrum on every rank: whole-slot broadcast, whatever the headline leg moved
            def run_one(job):
                return eng.run_blocks_sharded(comm, job, 1, root=0, rows=None).total_ms
        else:
            def run_one(job):
                return eng.run_blocks(job, 1).total_ms
        try:
            mine_crt = crt_leg(pkg, eng, wl, crt_n, args.crt_blocks, run_one, agree=(max_over_ranks if use_dist else None))
        except Exception as ex:      # e.g. not enough free HBM: report, do not fail the bench line
            mine_crt = {"error": str(ex)[:600]}
        every = [mine_crt]
        if use_dist:
            every = [None] * world
            dist.all_gather_object(every, mine_crt)          # reached by every rank, whatever happened above
        if rank == 0:
            bad = [c for c in every if "error" in c]
            if bad:
                crt = bad[0]
            else:
                crt = dict(every[0])
                crt["channels"] = sum(c["channels"] for c in every)
                crt["mean_crossing_channels"] = sum(c.get("mean_crossing_channels") or 0 for c in every) or None
                crt["worst_block_ms"] = max(c["worst_block_ms"] for c in every)
                crt["mean_block_ms"] = max(c["mean_block_ms"] for c in every)
                crt["sustained"] = all(c["sustained"] for c in every)
                crt["algorithmic_GBps"] = sum(c["algorithmic_GBps"] for c in every)
                crt["dram_side_GBps"] = sum(c["dram_side_GBps"] for c in every)
                crt["dram_side_frac_of_hbm_peak"] = crt["dram_side_GBps"] / (HBM_PEAK_GBS * len(every))
                crt["dram_side_frac_of_measured_stream_copy"] = crt["dram_side_GBps"] / (STREAM_COPY_GBS * len(every))
                crt["gpus"] = len(every)
                crt["exchange"] = "broadcast" if comm is not None else main_leg

    leg_seconds["c_rt"] = time.perf_counter() - t_leg
    t_leg = time.perf_counter()
    crt_shared = None
    if args.crt_shared and rank == 0 and world == 1 and config == 3:
        # the same leg with the channels SHARING the three response rows of the mix: what the card carries when channels of one mode
        # use one filter (not the headline: the reference gives every channel its own copy)
        try:
            crt_shared = crt_leg(pkg, eng, wl, [args.crt_shared], args.crt_blocks, lambda job: eng.run_blocks(job, 1).total_ms, shared=True)
        except Exception as ex:
            crt_shared = {"error": str(ex)[:600]}

    progress("cpu_baseline")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t_leg = time.perf_counter()
        cpu = cpu_baseline(oracle_lib, ring_host, wl, crt_blocks=args.crt_blocks)
        leg_seconds["cpu_baseline"] = time.perf_counter() - t_leg

    rccl_ranks = comm.world if comm is not None else 0
    if comm is not None:
        comm.close()
    eng_lanes, eng_plan = eng.lanes, eng.plan
    eng.close()
    ranks_info = [{"rank": rank, "device": dev_index, "device_name": torch.cuda.get_device_name(dev_index)}]
    if use_dist:
        allr = [None] * world
        dist.all_gather_object(allr, ranks_info[0])
        ranks_info = allr

    progress("dropin (free-running)")
    # ---- the boundary itself: the same workload through filter.h (C harness, one pthread per channel), PCIe in the loop
    dropin = None
    if rank == 0 and world == 1 and not args.no_dropin and config in (1, 2, 3):
        t_leg = time.perf_counter()
        dropin = []
        fs_arg = "%.1f" % wl["fs"]
        for label, nthr, env in [
                ("filter.h as an unmodified radiod uses it (every block's spectrum copied to fdomain[] for the host's estimate_noise)", nch, {}),
                ("radiod with the noise estimate taken from the device (INTEGRATION.md section 1 patch): no spectrum copy", nch,
                 {"KA9Q_HIP_FDOMAIN": "0", "KA9Q_HIP_NOISE_SAMPRATE": fs_arg}),
                ("2000 channel threads (Nchannels, src/radio.h:356), noise estimate from the device", 2000,
                 {"KA9Q_HIP_FDOMAIN": "0", "KA9Q_HIP_NOISE_SAMPRATE": fs_arg})][:1 if config == 1 else 3]:
            try:
                dropin.append(dropin_leg(wl, ring_host, nthr, args.dropin_blocks, env, label))
            except Exception as ex:
                dropin.append({"label": label, "error": str(ex)[:600]})
        leg_seconds["dropin"] = time.perf_counter() - t_leg
    progress("dropin_paced")
    # ---- the boundary the way radiod runs it: the front end on its own 20 ms clock, channel threads blocking in execute_filter_output
    dropin_paced = None
    if rank == 0 and world == 1 and not args.no_dropin_paced and config in (2, 3):
        t_leg = time.perf_counter()
        dropin_paced = []
        fs_arg = "%.1f" % wl["fs"]
        for label, nthr, env in [
                ("unmodified radiod at real time: one pthread per channel, every block's spectrum copied back (default environment)", nch, {}),
                ("2000 channel threads at real time, noise estimate from the device (KA9Q_HIP_FDOMAIN=0)", 2000,
                 {"KA9Q_HIP_FDOMAIN": "0", "KA9Q_HIP_NOISE_SAMPRATE": fs_arg})]:
            try:
                res = dropin_leg(wl, ring_host, nthr, args.dropin_blocks, env, label, paced_us=BLOCKTIME * 1e6)
                cg = res.get("host_cgroup") or {}
                if cg.get("periods_throttled_during_the_leg", 0) > 0 and (res.get("drops") or (res.get("paced") or {}).get("blocks_later_than_one_period")):
                    # the CONTAINER was stopped by its CPU quota while the leg ran (every thread of it, the front end included) and blocks
                    # were late or dropped: that attempt says what a stalled host does, not what the boundary does -- it stays in the line,
                    # and the leg runs once more
                    again = dropin_leg(wl, ring_host, nthr, args.dropin_blocks, env, label, paced_us=BLOCKTIME * 1e6)
                    again["first_attempt_while_the_container_was_throttled"] = {
                        "drops": res.get("drops"), "latency_ms": (res.get("paced") or {}).get("latency_ms"),
                        "blocks_later_than_one_period": (res.get("paced") or {}).get("blocks_later_than_one_period"),
                        "front_end_worst_wakeup_lateness_ms": (res.get("paced") or {}).get("front_end_worst_wakeup_lateness_ms"), "host_cgroup": cg}
                    res = again
                dropin_paced.append(res)
            except Exception as ex:
                dropin_paced.append({"label": label, "error": str(ex)[:600]})
        leg_seconds["dropin_paced"] = time.perf_counter() - t_leg
    progress("dropin_sharded")
    # ---- channels sharded over the node's GPUs BEHIND filter.h (KA9Q_HIP_DEVICES): config 4's shape, 1024 x 24 kHz channels per device,
    # one master, one pthread per channel, the front end at wall-clock pace.  On a one-GPU run the two shards share the device ("0,0").
    dropin_sharded = None
    if rank == 0 and not args.no_dropin and not args.no_dropin_paced and config in (3, 4):
        t_leg = time.perf_counter()
        ndev = world if world > 1 else 2
        devs = ",".join(str(i) for i in range(world)) if world > 1 else "%d,%d" % (dev_index, dev_index)
        wl4 = workload_for(4, 0, 1, 1024 * ndev)
        try:
            res = dropin_leg(wl4, ring_host, 1024 * ndev, min(args.dropin_blocks, 250), {"KA9Q_HIP_DEVICES": devs}, "config 4's shape behind filter.h: %d x 24 kHz "
                             "channels (P=600), one master, slaves sharded over KA9Q_HIP_DEVICES=%s (1024 per device), one pthread per channel, front end at "
                             "wall-clock pace, spectrum copied back from the first device" % (1024 * ndev, devs), paced_us=BLOCKTIME * 1e6, verify=24)
            res["devices_distinct"] = bool(world > 1)
            res["hardware"] = "measured on %d MI355X" % world if world > 1 else "two shards on ONE MI355X (the box has one): the multi-device path itself is unmeasured on hardware"
            dropin_sharded = [res]
        except Exception as ex:
            dropin_sharded = [{"error": str(ex)[:600]}]
        leg_seconds["dropin_sharded"] = time.perf_counter() - t_leg
    progress("c_rt_pcie")
    crt_pcie = None
    if rank == 0 and world == 1 and not args.no_crt_pcie and config == 3:
        t_leg = time.perf_counter()
        crt_pcie = []
        nb = int(os.environ.get("BENCH_PCIE_PIPELINED_CHANNELS", "1428480"))     # (scripts: the double-buffered probe at another size)
        probes = ((460_800, False, False), (1_320_960, True, False), (nb, True, True))
        if os.environ.get("BENCH_PCIE_PROBES"):              # (scripts: only some of the probes, by index)
            probes = tuple(probes[int(i)] for i in os.environ["BENCH_PCIE_PROBES"].split(","))
        for n, dm, pipe in probes:
            progress("c_rt_pcie %d channels demod=%s pipelined=%s" % (n, dm, pipe))
            try:
                crt_pcie.append(crt_pcie_leg(pkg, wl, n, args.crt_pcie_blocks, dm, dev_index, pipe))
            except Exception as ex:
                crt_pcie.append({"channels": n, "error": str(ex)[:600]})
        leg_seconds["c_rt_pcie"] = time.perf_counter() - t_leg
    progress("next_rows")
    next_rows = None
    if rank == 0 and world == 1 and not args.no_next_rows and config == 3:
        t_leg = time.perf_counter()
        next_rows = []
        for mode in [m for m in ("linear", "pll", "fm") if m in args.next_rows_modes.split(",")]:
            progress("next_rows " + mode)
            try:
                next_rows.append(next_rows_leg(pkg, wl, args.next_rows_channels, dev_index, mode))
            except Exception as ex:
                next_rows.append({"mode": mode, "error": str(ex)[:600]})
        leg_seconds["next_rows"] = time.perf_counter() - t_leg
    wd["leg"] = None                          # nothing left to watch
    if use_dist:
        dist.destroy_process_group()
    emit(())


if __name__ == "__main__":
    main()
