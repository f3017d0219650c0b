#!/usr/bin/env python3
"""bench.py -- free-running throughput of the MI355X overlap-save channelizer.

A "step" is one 20 ms block of the hot path: the shared forward transform of the
N = 3,240,000-sample window (129.6 MS/s real input) plus the gather x response +
backward transform of every channel, inputs already resident in HBM (an 8-block
sig_gen stream pre-generated into the device ring and replayed cyclically).

  N = 1   BASELINE config 3: 1024 mixed usb/cw/iq channels, 12 kHz (P = 300), one GPU.
  N > 1   BASELINE config 4: rank 0 owns the front end and transforms; the block
          spectrum (12.96 MB) is RCCL-broadcast; every rank runs its own 1024 x 24 kHz
          channels (P = 600) -> weak scaling in channels.

metric: channels sustained at 129.6 MS/s = channel-blocks per second / 50 blocks/s
(real-time-equivalent channels: how many channels of this configuration the measured
block rate could serve in real time).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FS = 129.6e6
BLOCKTIME = 0.02
L = int(round(FS * BLOCKTIME))          # 2,592,000
M = L // 4 + 1                          # 648,001  (overlap 5, src/radio.c:582-586)
N = L + M - 1                           # 3,240,000
BINS = N // 2 + 1
RING_BLOCKS = 8
HBM_PEAK_GBS = 8000.0                   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:35
FWD_BYTES = 4 * N + 8 * BINS            # 25,920,008  


def pmc_traffic_bytes():
    """HBM-side bytes per block of the three forward kernels from the committed rocprofv3 PMC passes
    (profiles/pmc_forward.json, written by scripts/rocprof_summary.py --json from separate FETCH_SIZE and
    WRITE_SIZE passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on
    gfx950).  None if no profile has been committed for this plan."""
    path = os.path.join(ROOT, "profiles", "pmc_forward.json")
    try:
        return json.load(open(path))["forward_traffic_bytes_per_block"]
    except Exception:
        return None


def chan_bytes(P, olen):
    return 8 * P + 8 * P + 8 * olen     # 6,720 (P=300) / 13,440 (P=600)


def channel_plan_config3(nch):
    """1024 mixed channels: thirds usb / cw / iq, f_i = 1 MHz + i*60 kHz + (i mod 40) Hz."""
    kinds = [(50 / 12000, 3000 / 12000), (-200 / 12000, 200 / 12000), (-5000 / 12000, 5000 / 12000)]
    hz_per_bin = FS / N
    plan = []
    for i in range(nch):
        f = 1e6 + (i % 1040) * 60e3 + (i % 40)
        plan.append((int(round(f / hz_per_bin)),) + kinds[i % 3])
    return plan


def channel_plan_config4(nch, rank):
    """8192 x 24 kHz channels, f_i = 0.5 MHz + i*7.8 kHz, 1024 per GPU, +-10 kHz."""
    hz_per_bin = FS / N
    plan = []
    for j in range(nch):
        i = rank * nch + j
        f = 0.5e6 + i * 7.8e3
        plan.append((int(round(f / hz_per_bin)), -10000 / 24000, 10000 / 24000))
    return plan


def channel_plan_sharded(nch, rank, world):
    """nch*world mixed channels on an even raster over 1..62.4 MHz; rank r owns channels [r*nch, (r+1)*nch)."""
    kinds = [(50 / 12000, 3000 / 12000), (-200 / 12000, 200 / 12000), (-5000 / 12000, 5000 / 12000)]
    hz_per_bin = FS / N
    step = 61.4e6 / (nch * world)
    plan = []
    for j in range(nch):
        i = rank * nch + j
        f = 1e6 + i * step + (i % 40)
        plan.append((int(round(f / hz_per_bin)),) + kinds[i % 3])
    return plan


def siggen_ring(oracle_lib, seed=1):
    """8 blocks of the deterministic sig_gen stream (CW carrier 10.00002 MHz, -20 dBFS, noise -40 dBFS).
    The generator is test infrastructure (oracle/); it only produces INPUT, outside the timed region."""
    g = oracle_lib.SigGen(10.00002e6 / FS, 10 ** (-20 / 20), 10 ** (-40 / 20),
                          oracle_lib.scale_ad(True, 1), True, seed=seed)
    return g.generate(RING_BLOCKS * L)


def cpu_baseline(oracle_lib, ring, plan, P, olen, seconds=12.0):
    """Reference filter.c (oracle/_ref, FFT butterflies from the project's float32 provider, NOT FFTW)
    timed on this host's cores: 1 forward-FFT worker thread + a pool of channel threads, radiod style."""
    if not oracle_lib.have_ref():
        return None
    R = oracle_lib.ref()
    R.oracle_fft_set_precision(1)          # float32 arithmetic for a fair CPU timing
    cores = os.cpu_count() or 1
    pool = max(1, min(cores - 1, 16))
    ring = np.ascontiguousarray(ring, np.float32)
    sarr = np.array([p[0] for p in plan], np.int32)

    def run(workers, budget):
        m = oracle_lib.RefMaster(L, M, oracle_lib.REAL, worker_threads=workers)
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

    nblk, per_block, fft_ms = run(1, seconds)
    nblk2, per_block2, fft_ms2 = run(2, seconds / 2)        # fft-threads = 2, the reference's advice for this rate (docs/ka9q-radio.md:232)
    R.oracle_fft_set_precision(0)
    return {
        "value": len(plan) * BLOCKTIME / per_block, "unit": "channels",
        "cores": 1 + pool, "kind": "reference",
        "sample": "%d blocks of the same workload (%d channels P=%d), reference filter.c with the project's "
                  "portable float32 FFT provider (FFTW3 is not installed on this image), 1 FFT worker + %d channel threads"
                  % (nblk, len(plan), P, pool),
        "ms_per_block": per_block * 1e3, "fwd_fft_ms_avg": fft_ms, "host_cores": cores,
        "two_fft_workers": {"value": len(plan) * BLOCKTIME / per_block2, "cores": 2 + pool, "blocks": nblk2,
                            "ms_per_block": per_block2 * 1e3, "fwd_fft_ms_avg": fft_ms2},
        "real_time": bool(min(per_block, per_block2) <= BLOCKTIME),
    }


def crt_leg(pkg, eng, nch, blocks=60):
    P, olen, tile = 300, 240, 3072
    nch -= nch % tile
    bank = eng.bank(P, olen, nch)
    plan = channel_plan_config3(tile)
    resp = np.stack([pkg.filterapi.design_response(P, olen, N, True, lo, hi, 11.0) for _, lo, hi in plan[:3]])
    resp = np.ascontiguousarray(np.tile(resp, (tile // 3, 1)))
    shifts = np.array([p[0] for p in plan], np.int32)
    for c0 in range(0, nch, tile):
        bank.set_responses(c0, resp)
        bank.set_shifts(c0, shifts + (c0 // tile) % 7)
    bank.set_active(nch)
    eng.run_blocks(0, 4)
    worst = tot = 0.0
    for j in range(blocks):
        t = eng.run_blocks(4 + j, 1)                 # forward + the 1024-channel bank + this bank, then a device sync
        worst = max(worst, t.total_ms); tot += t.total_ms
    mean = tot / blocks
    bank.set_active(0)
    alg = FWD_BYTES + (nch + 1024) * chan_bytes(P, olen)
    return {"channels": nch + 1024, "blocks": blocks, "worst_block_ms": worst, "mean_block_ms": mean, "sustained": worst <= BLOCKTIME * 1e3,
            "algorithmic_GBps": alg / (mean * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "every block individually <= 20 ms; I/O resident in HBM (responses %.0f GB, 4 output images %.0f GB); "
                    "bisected C_rt over 500 blocks: profiles/r01_crt.json" % (nch * P * 8 / 1e9, 4 * nch * olen * 8 / 1e9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    ap.add_argument("--plan", default="", help="forward plan override, e.g. 144x100x225")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay one hipGraph per ring cycle instead of eager launches")
    ap.add_argument("--eager", action="store_true", help="(default) eager launches; kept for compatibility")
    ap.add_argument("--no-crt", action="store_true", help="skip the C_rt leg (one large bank, every block inside 20 ms)")
    ap.add_argument("--crt-channels", type=int, default=17_000_000, help="channels of the C_rt leg's bank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 exercises the multi-GPU code path (RCCL broadcast, torch-owned spectrum slots)
    # with a single rank, so it can be smoke-tested on a 1-GPU box
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # "nccl" is RCCL on ROCm.  BENCH_DIST_BACKEND=gloo exists only to exercise the multi-rank control flow with several
        # ranks on ONE GPU (RCCL refuses two ranks per device); the replicate mode has no data-path collective anyway.
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world)

    # N > 1: channels shard across ranks.  How the shared spectrum gets to every rank:
    #   replicate (default)  every rank runs the forward transform on its own HBM-resident copy of the samples
    #   subband | broadcast  rank 0 transforms, the needed spectrum rows / the whole slot travel over xGMI (RCCL)
    #   auto                 subband or broadcast, whichever moves fewer bytes (sharding.plan_exchange)
    exch = os.environ.get("BENCH_EXCHANGE", "replicate") if use_dist else None
    pkg = ge.load()
    eng = pkg.engine.Engine(L, M, pkg.engine.REAL, device=local_rank, plan=args.plan, ring_blocks=RING_BLOCKS)

    nch = args.channels
    P, olen = 300, 240
    if not use_dist:
        workload = "config3: sig_gen real 129.6 MS/s, %d mixed usb/cw/iq 12 kHz channels (P=300), 1 MI355X" % nch
        plan = channel_plan_config3(nch)
    else:
        # weak scaling: every GPU runs the SAME per-GPU workload as N=1 (1024 mixed 12 kHz channels); the
        # node's channels cover 1..62.4 MHz and rank r owns a contiguous slice of that raster (config 4's
        # architecture: rank 0 owns the front end, the spectrum travels over xGMI)
        if exch == "replicate":
            how = ("the sample stream is resident in every GPU's HBM (the bench's input rule; in service the host feeds each GPU over its "
                   "own PCIe link) and each GPU transforms it itself (about 0.1 % of a 20 ms block): no data-path collective")
        else:
            how = "rank 0 owns the front end and the forward transform, the spectrum travels over xGMI via RCCL (%s)" % exch
        workload = ("config3 per GPU x %d: sig_gen real 129.6 MS/s, %d mixed usb/cw/iq 12 kHz channels (P=300) sharded by "
                    "frequency over %d MI355X; %s" % (world, nch * world, world, how))
        plan = channel_plan_sharded(nch, rank, world)

    # ---- inputs resident in HBM before anything is timed
    ring_host = siggen_ring(oracle_lib)
    # the device ring starts with the write position M-1 ahead (zeros before time 0); fill it to the brim
    eng.write(ring_host[:RING_BLOCKS * L - (M - 1)])
    eng.write(ring_host[RING_BLOCKS * L - (M - 1):])       # wraps: ring now holds the cyclic 8-block stream
    bank = eng.bank(P, olen, nch)
    resp = np.stack([pkg.filterapi.design_response(P, olen, N, True, lo, hi, 11.0) for _, lo, hi in plan])
    bank.set_responses(0, resp)
    bank.set_shifts(0, np.array([p[0] for p in plan], np.int32))
    bank.set_active(nch)
    eng.set_notches([0], 0.01)                               # DC notch is always present (src/radio.c:601-620)

    def barrier():
        if use_dist:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    if not use_dist or exch == "replicate":
        graph = bool(args.graph) and not use_dist
        eng.run_blocks(0, args.warmup, graph=graph)
        barrier()
        t0 = time.perf_counter()
        timing = eng.run_blocks(args.warmup, args.steps, graph=graph)   # returns after the stream drained
        barrier()
        elapsed = time.perf_counter() - t0
        gpu_ms = timing.total_ms
        exchange_mode = "none (forward transform replicated on every rank)" if use_dist else None
    else:
        # spectrum slots are torch tensors so RCCL can move them; each block's exchange is enqueued on the
        # engine's own per-slot HIP stream (wrapped as a torch ExternalStream), so block j's exchange and
        # channels overlap block j+1's forward transform on another stream
        slots = [torch.zeros(2 * eng.spec_elems, dtype=torch.float32, device="cuda") for _ in range(4)]
        for i, t in enumerate(slots):
            eng.attach_spectrum(i, t.data_ptr())
        streams = [torch.cuda.ExternalStream(eng.slot_stream(i)) for i in range(4)]
        na, pitch, _off = eng.spec_layout
        nrows = (BINS + na - 1) // na
        mine = pkg.sharding.needed_rows([p[0] for p in plan], P, BINS, na)
        all_rows = [None] * world
        dist.all_gather_object(all_rows, mine)
        mode = pkg.sharding.plan_exchange(all_rows, nrows) if exch == "auto" else exch
        if mode not in ("broadcast", "subband"):
            raise SystemExit("BENCH_EXCHANGE must be replicate, auto, subband or broadcast")

        def exchange(j):
            s = j % 4
            with torch.cuda.stream(streams[s]):
                if mode == "broadcast":
                    dist.broadcast(slots[s], src=0, async_op=True).wait()   # stream-level wait, host does not block
                elif mode == "subband":
                    if rank == 0:
                        ops = [dist.P2POp(dist.isend, slots[s][2 * pitch * lo:2 * pitch * hi], r)
                               for r, (lo, hi) in enumerate(all_rows) if r != 0 and hi > lo]
                    else:
                        lo, hi = mine
                        ops = [dist.P2POp(dist.irecv, slots[s][2 * pitch * lo:2 * pitch * hi], 0)] if hi > lo else []
                    for w in (dist.batch_isend_irecv(ops) if ops else []):
                        w.wait()

        def run(job0, n):
            for j in range(job0, job0 + n):
                if rank == 0:
                    eng.forward(j)
                exchange(j)
                bank.execute(j % 4)

        run(0, args.warmup)
        barrier()
        t0 = time.perf_counter()
        run(args.warmup, args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        gpu_ms = elapsed * 1e3
        exchange_mode = mode

    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel durations with HIP events on the launch stream (eager, instrumented)
    roof = None
    if rank == 0:
        eng_t = eng
        # one kernel at a time on one stream; the first pass after the multi-stream run is discarded
        # (the clocks need a few ms to settle after the load change: scripts/data_probe.py)
        eng_t.run_blocks(0, min(args.steps, 200), graph=False, instrument=True)
        it = eng_t.run_blocks(0, min(args.steps, 200), graph=False, instrument=True)
        kern = {}
        for name, ms, n in (("fwd_first_real", it.first_ms, it.first_n), ("fwd_cols", it.cols_ms, it.cols_n),
                            ("fwd_rows", it.rows_ms, it.rows_n), ("chan_ifft", it.chan_ms, it.chan_n)):
            if n:
                kern[name] = ms / n * 1e3       # microseconds per launch
        Ra = eng.axes[0] // 2 + 1
        inner_bytes = Ra * eng.axes[1] * eng.axes[2] * 8
        own = {"fwd_first_real": 4 * N + inner_bytes, "fwd_cols": 2 * inner_bytes,
               "fwd_rows": inner_bytes + 8 * BINS, "chan_ifft": nch * chan_bytes(P, olen)}
        fwd_us = sum(kern.get(k, 0.0) for k in ("fwd_first_real", "fwd_cols", "fwd_rows"))
        achieved = FWD_BYTES / (fwd_us * 1e-6) / 1e9 if fwd_us else 0.0
        roof = {
            "bound": "hbm", "kernel": "forward transform = fwd_first_real + fwd_cols + fwd_rows (one launch each per block)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic_bytes(),
            "frac_of_measured_copy_rate": achieved / 6290.0,        # 6.29 TB/s: the chip's achievable copy rate (MI355X_MICROARCH.md)
            "structural_cap": "three HBM passes: at most 1/3 of peak on algorithmic bytes",
            "algorithmic_bytes_per_block": FWD_BYTES, "forward_us_per_block": fwd_us,
            "kernels_us": kern,
            "kernels_own_GBps": {k: own[k] / (kern[k] * 1e-6) / 1e9 for k in own if k in kern and kern[k] > 0},
        }

    # ---- C_rt leg: ONE MI355X, one bank of millions of 12 kHz channels tiled from the same
    # config-3 plan, inputs and outputs resident in HBM; every block is run to completion on its own and must take
    # <= 20 ms (the literal "channels sustained in real time"; bisected value: profiles/r01_crt.json)
    # With N > 1 (replicated forward) every rank carries its own bank of that size: the node's figure is the sum, and it
    # holds only if the slowest block of the slowest rank stays inside 20 ms.
    crt = None
    if not args.no_crt and (not use_dist or exch == "replicate"):
        try:
            mine_crt = crt_leg(pkg, eng, args.crt_channels)
        except Exception as ex:      # e.g. not enough free HBM: report, do not fail the bench line
            mine_crt = {"error": str(ex)[:200]}
        if use_dist:
            every = [None] * world
            dist.all_gather_object(every, mine_crt)          # reached by every rank, whatever happened above
        else:
            every = [mine_crt]
        if rank == 0:
            bad = [c for c in every if "error" in c]
            if bad:
                crt = bad[0]
            else:
                crt = dict(every[0])
                crt["channels"] = sum(c["channels"] for c in every)
                crt["worst_block_ms"] = max(c["worst_block_ms"] for c in every)
                crt["mean_block_ms"] = max(c["mean_block_ms"] for c in every)
                crt["sustained"] = all(c["sustained"] for c in every)
                crt["algorithmic_GBps"] = sum(c["algorithmic_GBps"] for c in every)
                crt["frac_of_hbm_peak"] = crt["algorithmic_GBps"] / (HBM_PEAK_GBS * len(every))
                crt["gpus"] = len(every)

    cpu = None
    if rank == 0 and not use_dist and not args.no_cpu_baseline:
        cpu = cpu_baseline(oracle_lib, ring_host, plan, P, olen)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_ch = nch * world
        value = total_ch * BLOCKTIME / (elapsed / args.steps)
        replicated = use_dist and exch == "replicate"
        step_bytes = (world if replicated else 1) * FWD_BYTES + total_ch * chan_bytes(P, olen) + (0 if (replicated or not use_dist) else (world - 1) * 8 * BINS)
        out = {
            "metric": "channels sustained @129.6 MS/s input (real-time-equivalent: channel-blocks/s / 50)",
            "value": value, "unit": "channels", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "channels_total": total_ch, "P": P, "olen": olen, "N": N, "L": L, "M": M,
                       "launch": ("hipGraph(8 blocks)" if (args.graph and not use_dist) else "eager") + ", %d HIP streams" % eng.lanes,
                       "plan": eng.plan},
            "exchange": (exchange_mode if use_dist else None),
            "blocks_per_s": args.steps / elapsed, "realtime_margin": BLOCKTIME / (elapsed / args.steps),
            "step_algorithmic_GBps": step_bytes / (elapsed / args.steps) / 1e9,
            "gpu_event_ms_per_step": gpu_ms / args.steps,
            "host_enqueue_ms_per_step": (timing.enqueue_ms / args.steps) if (not use_dist or exch == "replicate") else None,
            "roofline": roof, "cpu_baseline": cpu, "c_rt": crt,
        }
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a banner through C stdio: flush it first so the JSON is the LAST line of stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
