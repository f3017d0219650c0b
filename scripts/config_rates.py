#!/usr/bin/env python3
"""Free-running block rate and algorithmic GB/s of the BASELINE configurations that fit one GPU:
config 2 (64.8 MS/s real, 256 x 12 kHz), config 3 (129.6 MS/s, 1024 mixed 12 kHz), and one GPU's share of config 4
(129.6 MS/s, 1024 x 24 kHz channels, P = 600).  Inputs resident in HBM, 4 HIP streams, eager launches."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load()
rng = np.random.default_rng(0)
out = {}
for name, fs, nch, P, olen in (("config2", 64.8e6, 256, 300, 240), ("config3", 129.6e6, 1024, 300, 240), ("config4_per_gpu", 129.6e6, 1024, 600, 480)):
    L = int(round(fs * 0.02)); M = L // 4 + 1; N = L + M - 1
    eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
    x = (rng.standard_normal(8 * L) * 0.05).astype(np.float32)
    eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
    b = eng.bank(P, olen, nch)
    hz = fs / N
    if name == "config2":
        plan = [(int(round((10e6 + i * 12.5e3) / hz)), -5000 / 12000, 5000 / 12000) for i in range(nch)]
    elif name == "config3":
        plan = bench.channel_plan_config3(nch)
    else:
        plan = bench.channel_plan_config4(nch, 0)
    b.set_responses(0, np.stack([pkg.filterapi.design_response(P, olen, N, True, lo, hi, 11.0) for _, lo, hi in plan]))
    b.set_shifts(0, np.array([p[0] for p in plan], np.int32)); b.set_active(nch)
    eng.set_notches([0], 0.01)
    eng.run_blocks(0, 200)
    t = eng.run_blocks(200, 2000)
    per = t.total_ms / 2000 * 1e-3
    fwd = 4 * N + 8 * (N // 2 + 1)
    alg = fwd + nch * (16 * P + 8 * olen)
    out[name] = {"us_per_block": per * 1e6, "blocks_per_s": 1 / per, "realtime_margin": 0.02 / per, "algorithmic_bytes_per_block": alg,
                 "algorithmic_GBps": alg / per / 1e9, "frac_of_8TBps": alg / per / 8e12, "plan": eng.plan}
    print(name, json.dumps(out[name]))
    eng.close()
print(json.dumps(out))
