#!/usr/bin/env python3
"""Cost of the SURVEY 8(f) rows on config 3 (129.6 MS/s real, 1024 x P=300): the fine-tuning epilogue in
chan_ifft (rank 1) and raw int16 input converted in fwd_first_real (rank 3), and estimate_noise() on the device (rank 2), each against the plain path.
Prints pipelined us/block and per-kernel us (HIP dispatch timestamps, one kernel at a time)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load()
L = 2592000; M = L // 4 + 1
rng = np.random.default_rng(0)
x16 = np.clip(rng.normal(0, 6000, 8 * L), -32768, 32767).astype(np.int16)
scale = np.float32(1.0 / 32768)
xf = x16.astype(np.float32) * scale
nch = 1024
shifts = (25000 + np.arange(nch) * 1500).astype(np.int32)
freq = rng.uniform(-20, 20, nch) / 12000.0
resp = np.ones((nch, 300), np.complex64) / 300
res = {}
CASES = ((False, False, False, False), (True, False, False, False), (False, True, False, False), (False, False, True, False),
         (True, True, True, False), (True, False, True, True), (True, True, True, True),
         (True, False, True, "pll"), (True, False, True, "fm"), (True, False, True, "fm_pll"), (True, False, True, "fm_tone"))
if os.environ.get("PROBE_ONLY_DEMOD") == "1":
    CASES = tuple(c for c in CASES if c[3])
for tuned, i16, noise, demod in CASES:
    if True:
        eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
        if i16:
            eng.write_i16(x16[:8 * L - (M - 1)], scale); eng.write_i16(x16[8 * L - (M - 1):], scale)
        else:
            eng.write(xf[:8 * L - (M - 1)]); eng.write(xf[8 * L - (M - 1):])
        b = eng.bank(300, 240, nch)
        b.set_responses(0, resp); b.set_active(nch)
        if tuned:
            b.set_tuning(0, 0, shifts, freq)
        else:
            b.set_shifts(0, shifts)
        if noise:
            b.enable_noise(129.6e6)
        if demod:      # rank 4: usb-like mono S16BE demodulators with AGC behind every channel
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            import oracle_lib as ol
            q = {True: ol.lin_params, "pll": lambda: ol.lin_params(pll=True),
                 # (squelch thresholds next to zero: the input is noise, and a closed squelch would skip the demodulator proper)
                 "fm": lambda: ol.fm_params(samprate=12000.0, bandwidth=8000.0, squelch_open=1e-9, squelch_close=1e-10),
                 "fm_pll": lambda: ol.fm_params(samprate=12000.0, bandwidth=8000.0, pll=True, squelch_open=1e-9, squelch_close=1e-10),
                 "fm_tone": lambda: ol.fm_params(samprate=12000.0, bandwidth=8000.0, tone_freq=100.0, squelch_open=1e-9, squelch_close=1e-10)}[demod]()
            b.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(q, f) for f, _ in ol.LinParams._fields_])] * nch, 0.02)
        eng.set_notches([0], 0.01)
        eng.run_blocks(0, 160)
        t = eng.run_blocks(160, 1600)
        eng.run_blocks(0, 200, instrument=True)
        it = eng.run_blocks(0, 200, instrument=True)
        key = "tuned=%d int16=%d noise=%d demod=%s" % (tuned, i16, noise, demod)
        res[key] = {"us_per_block": t.total_ms / 1600 * 1e3, "first_us": it.first_ms / it.first_n * 1e3, "cols_us": it.cols_ms / it.cols_n * 1e3,
                    "rows_us": it.rows_ms / it.rows_n * 1e3, "chan_us": it.chan_ms / it.chan_n * 1e3,
                    "noise_us": (it.notch_ms / it.notch_n * 1e3) if it.notch_n else None,
                    "demod_us": (it.demod_ms / it.demod_n * 1e3) if it.demod_n else None}
        print(key, json.dumps(res[key]))
        eng.close()
print(json.dumps(res))
