#!/usr/bin/env python3
"""Round 6: the engine's stream set-up against NULL-STREAM traffic of the host.  CU-masked streams are blocking streams (ordered with the
legacy null stream: scripts/micro/masked_stream_blocking.hip); round 5's bench run with every lane masked deadlocked on exactly that.  Here a
thread hammers the null stream through torch (the default stream of torch IS the legacy null stream) while the main thread runs, twice over,
what bench.py runs in one process: free-running blocks on the four lanes with the ticketed notch, a retune and a response swap in between,
and the 8f chain of all three modes (demodulator stream + PCM copy stream) on 70,000 channels with its verification.  Must come back.
usage: CHZ_OWN_QUEUES=1|2|4 python scripts/null_stream_soak.py   (prints one JSON line)"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
import __graft_entry__ as ge

pkg = ge.load()
torch.cuda.set_device(0)
stop = False
ops = [0]


def hammer():
    x = torch.ones(1 << 20, device="cuda")
    while not stop:
        x.add_(1.0)                      # a kernel on the legacy null stream
        y = x[:1024].cpu()               # a synchronous copy on it
        ops[0] += 1
        time.sleep(0.0005)


th = threading.Thread(target=hammer, daemon=True)
th.start()
t0 = time.perf_counter()
wl = bench.workload_for(3, 0, 1, 1024)
out = {"own_queues": os.environ.get("CHZ_OWN_QUEUES", "(default)"), "rounds": []}
for rnd in range(2):
    eng = pkg.engine.Engine(wl["L"], wl["M"], pkg.engine.REAL, ring_blocks=8)
    x = (np.random.default_rng(rnd).standard_normal(8 * wl["L"]) * 0.05).astype(np.float32)
    eng.write(x[:8 * wl["L"] - (wl["M"] - 1)]); eng.write(x[8 * wl["L"] - (wl["M"] - 1):])
    bank = eng.bank(wl["P"], wl["olen"], 1024)
    resp = np.stack([pkg.filterapi.design_response(wl["P"], wl["olen"], wl["N"], True, lo, hi, 11.0) for _, lo, hi in wl["plan"]])
    shifts = np.array([p[0] for p in wl["plan"]], np.int32)
    bank.set_responses(0, resp); bank.set_shifts(0, shifts); bank.set_active(1024)
    eng.set_notches([0], 0.01)
    job = 0
    for it in range(6):
        t = eng.run_blocks(job, 1500); job += 1500
        bank.set_shifts(100, shifts[100:150] + it)            # staged descriptor refresh (desc_push) on the lanes
        bank.set_responses(10, resp[10:20])                   # response swap over the upload stream
    eng.close()
    legs = {}
    for mode in ("linear", "pll", "fm"):
        r = bench.next_rows_leg(pkg, wl, 70000, 0, mode)
        assert r["pcm_mismatches"] == 0, r
        legs[mode] = round(r["pipelined_ms_per_block"], 3)
    out["rounds"].append({"free_running_us_per_block": round(t.total_ms / t.blocks * 1e3, 2), "chain_ms_per_block": legs})
stop = True
th.join()
out["null_stream_ops"] = ops[0]
out["seconds"] = round(time.perf_counter() - t0, 1)
print(json.dumps(out))
