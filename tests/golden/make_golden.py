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


def make_next_rows(name):
    """SURVEY 8(f) ranks 2 and 3 from the reference's OWN radio.c / rx888.c (oracle/ref_radio_wrap.c, ref_rx888_wrap.c):
    estimate_noise() (src/radio.c:1783-1866) on a seeded REAL-master spectrum for a set of shifts and slave sizes, and
    convert_avx2() (src/rx888.c:694-751) on seeded int16 samples with and without the LTC2208 de-randomiser."""
    r = np.random.default_rng(808)
    bins, fs = 16201, 1.296e6
    spec = ((r.standard_normal(bins) + 1j * r.standard_normal(bins)) * 3e-3).astype(np.complex64)
    idx = r.integers(0, bins, 40)
    spec[idx] += ((r.standard_normal(40) + 1j * r.standard_normal(40)) * 0.5).astype(np.complex64)
    shifts = np.array([0, 1, -1, 300, 500, 700, -700, 5000, -5000, bins - 1, -(bins - 1), bins - 400, bins - 600, bins + 50] +
                      [int(x) for x in r.integers(-bins, bins, 18)], np.int32)
    s_bins = np.array([300, 600, 1200], np.int32)
    n0 = np.array([[ol.ref_estimate_noise(spec, ol.REAL, int(sb), int(sh), fs) for sh in shifts] for sb in s_bins])
    x = r.integers(-32768, 32768, 4096, dtype=np.int64).astype(np.int16)
    x[:8] = [32767, -32768, 32766, -32766, -32767, 0, 1, -1]
    scale = np.float32(ol.scale_ad(True, 1) * 1.2345)
    conv = {}
    for rnd in (0, 1):
        out = ol.ref_convert_i16(x, scale, bool(rnd), avx2=True)
        if out is None:
            raise SystemExit("this host has no AVX2: convert_avx2 cannot be run")
        conv["conv%d" % rnd], conv["energy%d" % rnd], conv["clips%d" % rnd] = out[0], np.uint64(out[1]), out[2]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), spectrum=spec, samprate=fs, shifts=shifts, s_bins=s_bins, n0=n0,
                        x16=x, scale=scale, **conv)
    print(name, "written")


def make_demod_rows(name):
    """SURVEY 8(f) rank 4 from the reference's OWN linear.c / fm.c (+ osc.c, iir.c, misc.c; oracle/ref_linear_wrap.c,
    ref_fm_wrap.c): demod_linear() and demod_fm() run block after block on seeded baseband -- plain, envelope, coherent (PLL,
    squaring PLL), FM with threshold extension, FM through the PLL demodulator, FM behind a PL-tone squelch."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    import test_oracle_vs_reference as T
    out = {}
    def smooth(est):
        n0s = np.zeros(len(est)); s = np.nan
        for b in range(len(est)):
            s = est[b] if np.isnan(s) else s + 0.10 * (est[b] - s)
            n0s[b] = s
        return n0s
    # (round 4) lin_pllhold / lin_pllsqhold: the carrier is there from the first block (a wide loop left alone with noise can run off
    # before it arrives) and STAYS: 150 blocks (3 s) of a loop that holds lock, every one of them compared strictly -- the 50-block
    # window of the other two coherent cases ends where their carrier does
    lin = [("lin_usb", dict(), False), ("lin_am", dict(env=True, dc_alpha=0.002, encoding=ol.PCM_S16LE), False),
           ("lin_pll", dict(pll=True), True), ("lin_pllsq", dict(pll=True, square=True, pll_bw=20.0, channels=2, encoding=ol.PCM_F32LE), True),
           ("lin_pllhold", dict(pll=True), "hold"), ("lin_pllsqhold", dict(pll=True, square=True, pll_bw=20.0, channels=2, encoding=ol.PCM_F32LE), "hold")]
    for key, kw, coh in lin:
        nblk, N = (150, 240) if coh == "hold" else (60, 240) if coh else (40, 240)
        r = np.random.default_rng(3)
        if coh == "hold":
            bb, power = T._coherent_case(r, nblk, N, kw.get("square", False), last=10 ** 9, first=0)
        elif coh:                                # the 90-block case of the pin test (seed 3 locks), its first 60 blocks
            bb, power = T._coherent_case(r, 90, N, kw.get("square", False))
            bb, power = bb[:nblk], power[:nblk]
        else:
            bb, power = T._demod_case(r, nblk, N)
        est = np.full(nblk, 2 * 4e-4 ** 2 / 12000.0) if coh else 1e-8 * (1 + 0.3 * r.standard_normal(nblk)) / 12000.0
        p = ol.lin_params(**kw)
        pll = np.zeros((nblk, 5))
        pcm, frame, mute, pw, gain = ol.ref_linear_run(p, bb, power, smooth(est), 0.02, pll_out=pll)
        out.update({key + "_kw": np.array(repr(kw)), key + "_bb": bb, key + "_bbpower": power, key + "_est": est, key + "_pcm": pcm,
                    key + "_frame": frame, key + "_mute": mute, key + "_opower": pw, key + "_gain": gain, key + "_pll": pll})
    fm = [("fm_plain", dict(), 0.0, 36, 26), ("fm_thr", dict(threshold_extend=True, encoding=ol.PCM_F32LE), 0.0, 36, 26),
          ("fm_pll", dict(pll=True), 0.0, 36, 26), ("fm_tone", dict(tone_freq=100.0), 100.0, 56, 46)]
    for key, kw, sent, nblk, last in fm:
        N, fs = 480, 24000.0
        r = np.random.default_rng(41)
        bb, power = T._fm_case(r, nblk, N, fs, tone=sent, last=last)
        est = (2 * 2e-3 ** 2 / fs) * (1 + 0.1 * r.standard_normal(nblk))
        p = ol.fm_params(**kw)
        ref = ol.ref_fm_run(p, bb, power, smooth(est), 0.02)
        out.update({key + "_kw": np.array(repr(kw)), key + "_bb": bb, key + "_bbpower": power, key + "_est": est})
        out.update({key + "_" + k: v for k, v in ref.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written")


if __name__ == "__main__":
    if not ol.have_ref():
        raise SystemExit("oracle/_ref/libka9q_ref.so missing: run `make -C oracle` where /root/reference exists")
    if not (ol.have_ref_radio() and ol.have_ref_rx888()):
        raise SystemExit("oracle/_ref/libka9q_ref_radio.so / _rx888.so missing: run `make -C oracle` where /root/reference exists")
    if ol.have_ref_linear() and ol.have_ref_fm():
        make_demod_rows("demod_rows")
    make_next_rows("next_rows")
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
