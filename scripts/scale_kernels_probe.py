#!/usr/bin/env python3
"""Per-kernel times of the full device chain (tuned chan_ifft, noise_est, demodulator) at 1.5 M channels on config 3's master:
which stage sets the PCIe-inclusive channel count once the link is no longer the limit."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
import bench
import oracle_lib as ol
pkg = ge.load()
L, M, N = bench.L, bench.M, bench.N
P, olen = 300, 240
cap = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 1_500_000
cap -= cap % 3072
eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=bench.RING_BLOCKS)
x = (np.random.default_rng(1).standard_normal(8 * L) * 0.05).astype(np.float32)
eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
bank = eng.bank(P, olen, cap)
tile = 3072
plan = bench.channel_plan_config3(tile)
resp = np.stack([pkg.filterapi.design_response(P, olen, N, True, lo, hi, 11.0) for _, lo, hi in plan[:3]])
resp = np.ascontiguousarray(np.tile(resp, (tile // 3, 1)))
shifts = np.array([p[0] for p in plan], np.int32)
for c0 in range(0, cap, tile):
    bank.set_responses(c0, resp)
    bank.set_tuning(0, c0, shifts + (c0 // tile) % 7, np.full(tile, -3.3 / 12000.0))
bank.enable_noise(129.6e6)
bank.set_pcm_stride(2 * olen)
mode = sys.argv[2] if len(sys.argv) > 2 else "linear"       # "pll": every channel in a coherent mode (carrier PLL, src/linear.c:83-153)
if mode.startswith("fm"):       # "fm", "fmtone" (PL-tone squelch, src/fm.c:264-311), "fmpll" (PLL demodulator, :176-203); squelch held open
    # "fmvar": the amplitude-variance SNR estimator of an open NBFM squelch (src/fm.c:110-129), radiod's default, instead of the bb_power / N0 one
    q = ol.fm_params(samprate=12000.0, bandwidth=6000.0, snr_squelch=(mode != "fmvar"), squelch_open=-2.0, squelch_close=-3.0,
                     tone_freq=(100.0 if mode == "fmtone" else 0.0), pll=(mode == "fmpll"))
else:
    q = ol.lin_params(pll=(mode == "pll"))
one = pkg.engine.DemodParams(*[getattr(q, f) for f, _ in ol.LinParams._fields_])
for c0 in range(0, cap, 65536):
    bank.set_demod(0, c0, [one] * min(65536, cap - c0), 0.02)
bank.set_active(cap)
eng.set_notches([0], 0.01)
eng.run_blocks(0, 8)
t = eng.run_blocks(8, 16)
it = eng.run_blocks(0, 16, instrument=True)
print(json.dumps({"channels": cap, "mode": mode, "pipelined_ms_per_block": t.total_ms / 16,
                  "chan_ms": it.chan_ms / it.chan_n, "noise_ms": it.notch_ms / it.notch_n if it.notch_n else None,
                  "demod_ms": it.demod_ms / it.demod_n if it.demod_n else None,
                  "ns_per_channel": {"chan": it.chan_ms / it.chan_n * 1e6 / cap, "noise": (it.notch_ms / it.notch_n * 1e6 / cap) if it.notch_n else None,
                                     "demod": (it.demod_ms / it.demod_n * 1e6 / cap) if it.demod_n else None}}))
eng.close()
