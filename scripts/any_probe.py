#!/usr/bin/env python3
"""Cost of the one-workgroup-per-channel kernel (chan_any) on config 3's master (129.6 MS/s real, N = 3,240,000): channel sizes the
register-tiled menu does not hold.  P = 9600 is the WFM channel of src/wfm.c:37-39 (384 kHz); prints us per launch and per channel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load()
L = 2592000; M = L // 4 + 1
rng = np.random.default_rng(0)
x = (rng.standard_normal(8 * L) * 0.05).astype(np.float32)
res = {}
for P, olen, nch in ((9600, 7680, 256), (9600, 7680, 2048), (1000, 800, 4096), (2700, 2160, 2048), (300, 240, 4096)):
    eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
    eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
    b = eng.bank(P, olen, nch)
    resp = np.ones((min(nch, 64), P), np.complex64) / P
    for c0 in range(0, nch, resp.shape[0]):
        b.set_responses(c0, resp[:min(resp.shape[0], nch - c0)])
    b.set_shifts(0, (25000 + np.arange(nch) * 700).astype(np.int32)); b.set_active(nch)
    eng.run_blocks(0, 40)
    t = eng.run_blocks(40, 200)
    it = eng.run_blocks(0, 100, instrument=True)
    key = "P=%d nch=%d" % (P, nch)
    us = it.chan_ms / it.chan_n * 1e3
    res[key] = {"chan_us": us, "ns_per_channel": us * 1e3 / nch, "pipelined_us_per_block": t.total_ms / 200 * 1e3,
                "channels_at_20ms": int(nch * 20000.0 / max(t.total_ms / 200 * 1e3, 1e-9)),
                "out_GBps": nch * olen * 8 / (us * 1e-6) / 1e9}
    print(key, json.dumps(res[key]), flush=True)
    eng.close()
print(json.dumps(res))
