#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libka9q_ref.so =
/root/reference/src/filter.c, window.c, misc.c, osc.c, gauss.c compiled unmodified, FFT
butterflies from the project's float64 DFT shim).  Run in the build container only
(/root/reference must exist):   python tests/golden/make_golden.py

Each file holds: the deterministic sig_gen input, the per-block master spectrum (after the DC
notch), every channel's response as set_filter built it, the per-block per-channel output of
execute_filter_output, and the parameters.  The reference ships no vectors of its own.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def make(name, L, M, in_type, olen, fs, carrier_hz, chans, nblocks, notch_bins):
    isreal = in_type == ol.REAL
    gen = ol.RefSigGen(carrier_hz / fs, 10 ** (-20 / 20), 10 ** (-40 / 20), ol.scale_ad(isreal, 1), isreal, seed=1)
    m = ol.RefMaster(L, M, in_type)
    m.set_notches(notch_bins, 0.01)
    cs = []
    for shift, low, high, beta in chans:
        c = m.channel(olen, ol.COMPLEX)
        assert c.set_filter(low, high, beta) == 0
        cs.append(c)
    x = gen.generate(nblocks * L)
    specs, outs = [], []
    for b in range(nblocks):
        assert m.write(x[b * L:(b + 1) * L]) == 1
        specs.append(m.spectrum())
        outs.append(np.stack([c.execute(ch[0]) for c, ch in zip(cs, chans)]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        L=L, M=M, in_type=in_type, olen=olen, P=cs[0].points, fs=fs, carrier_hz=carrier_hz,
                        notch_bins=np.array(notch_bins, np.int32), notch_alpha=0.01,
                        chans=np.array(chans, np.float64), x=x, spectrum=np.stack(specs),
                        response=np.stack([c.response() for c in cs]), output=np.stack(outs))
    m.close()
    print(name, "written")


def make_downconvert(name):
    """Tail of downconvert() (src/radio.c:1476-1520) around the reference's OWN oscillator (src/osc.c, src/sincospi.c
    compiled unmodified; the dozen glue statements are restated in oracle/ref_driver.c): a tuning history with shift
    changes, remainder changes and a sweep, applied to seeded channel samples."""
    L, M, fs_out, olen = 11520, 2881, 12000.0, 240
    rng = np.random.default_rng(2026)
    history = [(2500, 13.7, 0.0)] * 2 + [(2501, 13.7, 0.0)] * 2 + [(2501, -7.25, 0.5)] * 70 + [(-3123, 3.0, 0.0)] * 2 + \
              [(-3123, 0.0, 0.0)] + [(1234, 19.99, -3.0)] * 3
    d = ol.Downconv(L, M, fs_out, "ref")
    xs, ys, pw = [], [], []
    for sh, rem, dr in history:
        x = (rng.standard_normal(olen) + 1j * rng.standard_normal(olen)).astype(np.complex64)
        y, p = d.block(x, sh, rem, dr)
        xs.append(x); ys.append(y); pw.append(p)
    keep = list(range(6)) + list(range(68, len(history)))          # the sweep's middle only advances the oscillator
    np.savez_compressed(os.path.join(HERE, name + ".npz"), L=L, M=M, fs_out=fs_out, olen=olen,
                        history=np.array(history, np.float64), keep=np.array(keep, np.int32),
                        x=np.stack(xs), y=np.stack(ys)[keep], power=np.array(pw)[keep])
    print(name, "written")


if __name__ == "__main__":
    if not ol.have_ref():
        raise SystemExit("oracle/_ref/libka9q_ref.so missing: run `make -C oracle` where /root/reference exists")
    make_downconvert("downconvert_tail")
    # scaled-down RX888 geometry (real input, N = 14400, 40 Hz bins), P = 300: usb / cw / iq / inverted / edge channels
    make("real_n14400_p300", 11520, 2881, ol.REAL, 240, 576e3, 100020.0,
         [(2500, 50 / 12000, 3000 / 12000, 11.0), (2501, -200 / 12000, 200 / 12000, 11.0), (2500, -5000 / 12000, 5000 / 12000, 11.0),
          (-2500, -5000 / 12000, 5000 / 12000, 11.0), (60, -0.4, 0.4, 6.0), (7150, -0.4, 0.4, 11.0), (5000, 0.1, 0.1, 3.0)],
         3, [125, 0])
    # the same master with 24 kHz channels, P = 600
    make("real_n14400_p600", 11520, 2881, ol.REAL, 480, 576e3, 100020.0,
         [(2500, -10000 / 24000, 10000 / 24000, 11.0), (-6000, -8000 / 24000, 8000 / 24000, 11.0), (7000, 0.0, 0.45, 11.0)],
         2, [0])
    # complex master (config-1 geometry scaled: N = 14400 complex), channels through DC and both band edges
    make("complex_n14400_p300", 11520, 2881, ol.COMPLEX, 240, 576e3, 100020.0 + 0.0,
         [(2500, -5000 / 12000, 5000 / 12000, 11.0), (-2500, -5000 / 12000, 5000 / 12000, 11.0), (0, -0.4, 0.4, 11.0),
          (7200, -0.4, 0.4, 11.0), (-7200, -0.4, 0.4, 11.0), (7100, -0.4, 0.4, 11.0), (14000, -0.4, 0.4, 11.0)],
         2, [0])
