#!/usr/bin/env python3
"""Solo kernel times (HIP-event dispatch timestamps, one kernel at a time on one stream) and the pipelined block time
of config 3 -- the quick A/B loop for kernel work.  usage: kernel_times.py [label]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load()
wl = bench.workload_for(int(os.environ.get("KT_CONFIG", "3")), 0, 1, 0)
eng = pkg.engine.Engine(wl["L"], wl["M"], pkg.engine.REAL, ring_blocks=8, plan=os.environ.get("KT_PLAN", ""))
rng = np.random.default_rng(0)
x = (rng.standard_normal(8 * wl["L"]) * 0.05).astype(np.float32)
eng.write(x[:8 * wl["L"] - (wl["M"] - 1)]); eng.write(x[8 * wl["L"] - (wl["M"] - 1):])
b = eng.bank(wl["P"], wl["olen"], wl["nch"])
plan = wl["plan"]
b.set_responses(0, np.stack([pkg.filterapi.design_response(wl["P"], wl["olen"], wl["N"], True, lo, hi, 11.0) for _, lo, hi in plan]))
b.set_shifts(0, np.array([p[0] for p in plan], np.int32)); b.set_active(wl["nch"])
eng.set_notches([0], 0.01)
eng.run_blocks(0, 300, instrument=True)
it = eng.run_blocks(0, 400, instrument=True)
k = {"first": it.first_ms / max(it.first_n, 1) * 1e3, "cols": it.cols_ms / max(it.cols_n, 1) * 1e3, "rows": it.rows_ms / max(it.rows_n, 1) * 1e3,
     "fix": it.fix_ms / max(it.fix_n, 1) * 1e3, "chan": it.chan_ms / max(it.chan_n, 1) * 1e3}
eng.run_blocks(0, 400)
t = eng.run_blocks(400, 4000)
fwd = k["first"] + k["cols"] + k["rows"]
print(json.dumps({"label": sys.argv[1] if len(sys.argv) > 1 else "", "kernels_us": {a: round(v, 2) for a, v in k.items()}, "fwd_us": round(fwd, 2),
                  "frac": round(bench.fwd_bytes(wl["N"]) / fwd / 1e3 / 8000, 4), "pipelined_us_per_block": round(t.total_ms / 4000 * 1e3, 2), "plan": eng.plan}))
eng.close()
