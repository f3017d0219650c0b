#!/usr/bin/env python3
"""Soak: config 3 free-running for a million blocks with channels retuning and responses being swapped all the way (no drains),
the notch recurrence ticketed across 4 streams and 2 issuing threads.  At the end the bank must produce, for the block it has
just run, exactly what a fresh engine with the same final settings produces for that block (bit for bit: same kernels, same
spectrum -- the DC notch only touches bin 0, which no channel here reads)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import bench
pkg = ge.load()
L, M, N = bench.L, bench.M, bench.N
P, olen, nch = 300, 240, 1024
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(11)
x = (rng.standard_normal(8 * L) * 0.05).astype(np.float32)

def make():
    e = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
    e.write(x[:8 * L - (M - 1)]); e.write(x[8 * L - (M - 1):])
    b = e.bank(P, olen, nch)
    return e, b

plan = bench.channel_plan_config3(nch)
resp = np.stack([pkg.filterapi.design_response(P, olen, N, True, lo, hi, 11.0) for _, lo, hi in plan]).astype(np.complex64)
shifts = np.array([p[0] for p in plan], np.int32)
eng, bank = make()
bank.set_responses(0, resp); bank.set_shifts(0, shifts); bank.set_active(nch)
eng.set_notches([0], 0.01)
job = 0
t0 = time.time()
for it in range(iters):
    t = eng.run_blocks(job, 2000); job += 2000
    c0 = int(rng.integers(0, nch - 64))
    shifts[c0:c0 + 50] += rng.integers(-40, 40, 50).astype(np.int32)
    bank.set_shifts(c0, shifts[c0:c0 + 50])                     # takes effect with the next block of each slot, no drain
    r0 = int(rng.integers(0, nch - 16))
    lo = float(rng.uniform(-0.3, 0.0)); hi = lo + float(rng.uniform(0.05, 0.3))
    newr = pkg.filterapi.design_response(P, olen, N, True, lo, hi, 9.0).astype(np.complex64)
    resp[r0:r0 + 10] = newr
    bank.set_responses(r0, resp[r0:r0 + 10])                     # spare rows + fences, no drain
    eng.check()
eng.run_blocks(job, 8); job += 8
last = (job - 1) & 0xFFFFFFFF
got = bank.read_slot(last % 4)
el = time.time() - t0
eng.close()
e2, b2 = make()
b2.set_responses(0, resp); b2.set_shifts(0, shifts); b2.set_active(nch)
e2.set_notches([0], 0.01)
e2.run_blocks(last - 7, 8)        # same ring phase: the ring holds a cyclic 8-block stream
want = b2.read_slot(last % 4)
e2.close()
same = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
print(json.dumps({"blocks": job, "retunes": iters * 50, "response_swaps": iters * 10, "seconds": el, "us_per_block": el / job * 1e6,
                  "final_block_bit_identical_to_fresh_engine": same, "max_abs_diff": float(np.abs(got - want).max())}))
sys.exit(0 if same else 1)
