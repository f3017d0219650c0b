"""Pin the CPU restatement (oracle/chz_oracle.c) against the reference ITSELF.

oracle/_ref/libka9q_ref.so is the reference's own src/filter.c, window.c, misc.c,
osc.c, gauss.c compiled unmodified (oracle/Makefile).  The reference ships no
tests or golden vectors for this path, so these comparisons
are what pins the oracle.  CPU only.
"""
import ctypes as C
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")

rng = np.random.default_rng(12345)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 12, 60, 75, 300, 600, 1200, 7 * 11 * 13, 1024, 32400])
def test_dft_provider_matches_pocketfft(oracle_built, n):
    # independent cross-check of the FFT restatement (oracle/dft.c) against numpy's pocketfft
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    got = ol.forward(x, ol.COMPLEX, f64=True)
    want = np.fft.fft(x.astype(np.complex128))
    assert rel(got, want) < 1e-13
    if n >= 2:
        xr = rng.standard_normal(n).astype(np.float32)
        got = ol.forward(xr, ol.REAL, f64=True)
        want = np.fft.rfft(xr.astype(np.float64))
        assert rel(got, want) < 1e-13


def test_i0_and_kaiser_match_reference(oracle_built):
    R = ol.ref()
    for z in [0.0, 0.5, 3.0, 11.0, 20.0, 35.0]:
        assert ol.oracle().chzo_i0(z) == pytest.approx(R.refchz_i0(z), rel=1e-15)
    for M, beta in [(61, 11.0), (121, 11.0), (2, 3.0), (41, 0.0), (240, 7.5)]:
        a = np.zeros(M, np.float32); b = np.zeros(M, np.float32)
        ol.oracle().chzo_make_kaiser(a.ctypes.data, M, beta)
        R.refchz_make_kaiserf(b.ctypes.data, M, beta)
        np.testing.assert_array_equal(a, b)


CASES = [  # (L, M, in_type, olen, out_type)
    (25920, 6481, ol.REAL, 240, ol.COMPLEX),     # scaled-down RX888: N=32400, P=300
    (25920, 6481, ol.REAL, 480, ol.COMPLEX),     # P=600
    (4800, 1201, ol.COMPLEX, 240, ol.COMPLEX),   # scaled-down config 1: N=6000, P=300
    (25920, 6481, ol.REAL, 240, ol.REAL),        # real->real
    (4800, 1201, ol.COMPLEX, 240, ol.REAL),      # complex->real ("untested" upstream)
]


@pytest.mark.parametrize("L,M,in_type,olen,out_type", CASES)
def test_set_filter_matches_reference(oracle_built, L, M, in_type, olen, out_type):
    m = ol.RefMaster(L, M, in_type)
    try:
        c = m.channel(olen, out_type)
        P = c.points
        for low, high, beta in [(-5000 / 12000, 5000 / 12000, 11.0), (50 / 12000, 3000 / 12000, 11.0),
                                (-200 / 12000, 200 / 12000, 3.0), (0.3, -0.7, 6.0), (0.1, 0.1, 11.0)]:
            assert c.set_filter(low, high, beta) == 0
            want = c.response()
            got = ol.set_filter(P, olen, m.N, in_type == ol.REAL, low, high, beta, out_type)
            scale = np.abs(want).max()
            assert np.abs(got - want).max() <= 2e-7 * scale
    finally:
        m.close()


@pytest.mark.parametrize("L,M,in_type,olen,out_type", CASES)
def test_gather_and_channel_match_reference(oracle_built, L, M, in_type, olen, out_type):
    m = ol.RefMaster(L, M, in_type)
    st = ol.Stream(L, M, in_type)
    try:
        c = m.channel(olen, out_type)
        P, sb = c.points, c.bins
        c.set_filter(-0.4, 0.4, 11.0)
        resp = c.response()
        B = m.bins
        shifts = [0, 1, -1, 7, 1000, -1000, P // 2, -(P // 2), B - 1, -(B - 1), B - P // 2, B // 2,
                  B // 2 + 3, -(B // 2) - 3, B + 5, -(B + 5), m.N // 2 - 1, -(m.N // 2 - 1)]
        shifts += list(rng.integers(-B - P, B + P, 12))
        for blk in range(2):
            if in_type == ol.REAL:
                x = rng.standard_normal(L).astype(np.float32)
            else:
                x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
            assert m.write(x) == 1
            spec_ref = m.spectrum()
            spec = st.push(x)
            assert rel(spec, spec_ref) < 1e-7           # same DFT provider, float32-rounded once
            for s in shifts:
                out_ref = c.execute(s)
                fd_ref = c.fdomain()
                fd = ol.gather(spec_ref, in_type, sb, s, resp, out_type)
                # the gather is one float32 complex multiply per bin: identical up to
                # the reference's -funsafe-math/-fcx-limited-range reassociation
                assert np.abs(fd - fd_ref).max() <= 1e-6 * max(np.abs(fd_ref).max(), 1e-30), s
                np.testing.assert_array_equal(fd == 0, fd_ref == 0)
                out = ol.channel(spec_ref, in_type, P, olen, s, resp, out_type)
                assert np.abs(out - out_ref).max() <= 2e-6 * max(np.abs(out_ref).max(), 1e-30), s
    finally:
        m.close()


def test_isb_unpack_matches_reference(oracle_built):
    L, M = 25920, 6481
    m = ol.RefMaster(L, M, ol.REAL)
    try:
        c = m.channel(240, ol.COMPLEX)
        c.set_filter(-0.4, 0.4, 11.0)
        c.set_isb(True)
        x = rng.standard_normal(L).astype(np.float32)
        m.write(x)
        spec = m.spectrum()
        for s in [3000, -3000]:
            out_ref = c.execute(s)
            out = ol.channel(spec, ol.REAL, 300, 240, s, c.response(), isb=True)
            assert np.abs(out - out_ref).max() <= 2e-6 * np.abs(out_ref).max()
    finally:
        m.close()


def test_notch_matches_reference(oracle_built):
    L, M = 25920, 6481
    m = ol.RefMaster(L, M, ol.REAL)
    st = ol.Stream(L, M, ol.REAL)
    try:
        bins = [125, 4000, 0]
        m.set_notches(bins, 0.01)
        state = np.zeros(2 * len(bins), np.float64)
        for blk in range(4):
            x = (rng.standard_normal(L) + 0.3).astype(np.float32)   # DC offset to give the notch work
            m.write(x)
            want = m.spectrum()
            got = st.push(x)
            ol.notch(state, bins, 0.01, got)
            assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
            assert np.abs(got[0] - want[0]) <= 1e-6 * abs(want[0]) + 1e-3
    finally:
        m.close()


@pytest.mark.parametrize("isreal", [True, False])
def test_siggen_stream_matches_reference(oracle_built, isreal):
    scale = ol.scale_ad(isreal=isreal, bitspersample=1)
    assert scale == pytest.approx(10 ** (3 / 20) if isreal else 1.0)
    f = 10.00002e6 / 129.6e6
    a = ol.SigGen(f, 0.1, 0.01, scale, isreal, seed=1)
    b = ol.RefSigGen(f, 0.1, 0.01, scale, isreal, seed=1)
    n = 40000       # crosses the 16384-step renormalisation twice
    for _ in range(2):
        x, y = a.generate(n), b.generate(n)
        assert np.abs(x - y).max() <= 1e-7
        # noise term must be bit-identical (integer RNG): nearly all samples equal exactly
        assert np.mean(x == y) > 0.99


def test_compute_tuning():
    N = 3240000
    r, shift, rem = ol.compute_tuning(N, 129.6e6, 10.00002e6)
    assert (r, shift) == (0, 250001) or (r, shift) == (0, 250000)
    assert abs(rem) <= 20.0
    r, shift, rem = ol.compute_tuning(N, 129.6e6, 70e6)
    assert r == -1


def test_first_block_is_preceded_by_zeros_and_overlap_carries(oracle_built):
    # window k = stream[kL-(M-1), kL+L) with zeros before time 0 (src/filter.c:244,259)
    L, M = 2400, 601
    N = L + M - 1
    m = ol.RefMaster(L, M, ol.REAL)
    try:
        x = rng.standard_normal(3 * L).astype(np.float32)
        full = np.concatenate([np.zeros(M - 1, np.float32), x])
        for k in range(3):
            m.write(x[k * L:(k + 1) * L])
            want = np.fft.rfft(full[k * L:k * L + N].astype(np.float64))
            assert rel(m.spectrum(), want) < 1e-6
    finally:
        m.close()


def test_downconvert_tail_matches_reference_oscillator(oracle_built):
    """Restated downconvert() tail (oracle/chz_oracle.c) against the same statements wrapped around the
    reference's OWN set_osc/step_osc/cispi (src/osc.c, src/sincospi.c compiled unmodified) over shift
    changes, remainder changes, sweeps and several oscillator renormalisations.  The phasors agree to
    double rounding (the reference is built with -funsafe-math-optimizations), so the float32 products are
    identical except for a rare last-bit flip."""
    L, M, fs, olen = 2592000, 648001, 12000.0, 240
    rng = np.random.default_rng(5)
    a, b = ol.Downconv(L, M, fs, "oracle"), ol.Downconv(L, M, fs, "ref")
    history = [(25000, 13.7, 0.0)] * 3 + [(25001, 13.7, 0.0)] * 2 + [(25001, -7.25, 0.5)] * 150 + \
              [(-31234, 3.0, 0.0)] * 3 + [(-31234, 0.0, 0.0)] * 2 + [(12345, 19.99, -3.0)] * 80
    flips = 0
    for sh, rem, dr in history:
        x = (rng.standard_normal(olen) + 1j * rng.standard_normal(olen)).astype(np.complex64)
        ya, pa = a.block(x, sh, rem, dr)
        yb, pb = b.block(x, sh, rem, dr)
        d = ya - yb
        ulp = np.spacing(np.maximum(np.abs(yb.real), np.abs(yb.imag)).astype(np.float32))
        assert (np.maximum(np.abs(d.real), np.abs(d.imag)) <= ulp).all()
        flips += int((ya != yb).sum())
        assert abs(pa - pb) <= 1e-9 * pb
    assert flips <= 1e-3 * olen * len(history)
    # a shift that is a multiple of V with zero remainder leaves the samples untouched apart from the
    # constant start-up phase; |y| is preserved in every case
    c = ol.Downconv(L, M, fs, "oracle")
    x = (rng.standard_normal(olen) + 1j * rng.standard_normal(olen)).astype(np.complex64)
    y0, _ = c.block(x, 25000, 0.0, 0.0)
    y1, _ = c.block(x, 25000, 0.0, 0.0)
    assert np.allclose(np.abs(y0), np.abs(x), rtol=1e-6) and np.array_equal(y0, y1)


def test_beam_gather_matches_reference(oracle_built):
    """slave->beam (src/filter.c:756-775) on a COMPLEX master: the restatement against the reference's own branch, for
    antenna selections and a genuine two-antenna combination, shifts through DC, the band edges and the +Nyquist seam."""
    L, M, olen = 11520, 2881, 240
    rng = np.random.default_rng(8)
    m = ol.RefMaster(L, M, ol.COMPLEX)
    st = ol.Stream(L, M, ol.COMPLEX)
    N = L + M - 1
    P = olen * N // L
    weights = [(1.0, 0.0), (0.0, 1.0), (0.7 + 0.2j, -0.3 + 0.6j)]
    shifts = [0, 1, -1, 150, -150, 3000, -3000, N // 2 - 100, -(N // 2) + 100, N // 2 - 10, N // 2 + 40]
    chans = []
    for (iw, qw) in weights:
        for sh in shifts:                       # a slave runs once per block: one slave per (weights, shift)
            c = m.channel(olen, ol.COMPLEX)
            assert c.set_filter(-0.4, 0.35, 7.0) == 0
            ab = c.set_beam(iw, qw)
            assert np.allclose(ab, ol.beam_weights(iw, qw), rtol=0, atol=1e-15)
            chans.append((c, ab, sh))
    for blk in range(2):
        x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
        assert m.write(x) == 1
        spec = m.spectrum()
        assert rel(st.push(x), spec) <= 1e-7
        for c, (alpha, beta), sh in chans:
            resp = c.response()
            got = c.execute(sh)
            fd = np.zeros(P, np.complex64); c.lib.refchz_chan_fdomain(c.h, fd.ctypes.data)
            want_fd = np.zeros(P, np.complex64)
            ol.oracle().chzo_gather_beam(spec.ctypes.data, spec.size, P, sh, resp.ctypes.data,
                                         alpha.real, alpha.imag, beta.real, beta.imag, want_fd.ctypes.data)
            # the reference's scratch vector was cleared once at set_beam (ref_driver.c); the walk of a fixed shift
            # writes the same bins every block, so unwritten bins stay zero on both sides
            assert np.abs(fd - want_fd).max() <= 2e-7 * max(np.abs(want_fd).max(), 1e-30), (blk, sh)
            want = ol.channel_beam(spec, P, olen, sh, resp, alpha, beta)
            assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1e-30), (blk, sh)
    m.close()


# ------------------------------------------------------------------------------------------------
# the restatements are pinned to the reference's OWN radio.c / rx888.c
# (oracle/ref_radio_wrap.c, oracle/ref_rx888_wrap.c include the files unmodified and export their
# static functions; everything else is discarded at link time)
# ------------------------------------------------------------------------------------------------
def _noisy_spectrum(r, bins, nsig=40):
    s = ((r.standard_normal(bins) + 1j * r.standard_normal(bins)) * 3e-3).astype(np.complex64)
    idx = r.integers(0, bins, nsig)
    s[idx] += ((r.standard_normal(nsig) + 1j * r.standard_normal(nsig)) * 0.5).astype(np.complex64)   # carriers
    return s


@pytest.mark.skipif(not ol.have_ref_radio(), reason="oracle/_ref/libka9q_ref_radio.so not built (needs /root/reference)")
def test_quantile_and_quickselect_match_reference_radio_c(oracle_built):
    # src/radio.c:1724-1775; the restatement finds the same order statistics (it never copies the array order)
    R = ol.ref_radio()
    r = np.random.default_rng(5)
    for n in (1, 2, 3, 10, 999, 1000, 1001, 1200):
        a = r.exponential(1.0, n)
        for p in (0.0, 0.1, 0.5, 0.999, 1.0):
            b = a.copy()
            got = R.refradio_quantile(b.ctypes.data, n, p)
            pos = p * (n - 1); i = int(np.floor(pos)); fr = pos - i
            srt = np.sort(a)
            want = srt[i] if fr == 0 else srt[i] + fr * (srt[i + 1] - srt[i])
            assert got == pytest.approx(want, rel=1e-15)
        k = int(r.integers(0, n))
        b = a.copy()
        assert R.refradio_quickselect(b.ctypes.data, n, k) == np.sort(a)[k]


@pytest.mark.skipif(not ol.have_ref_radio(), reason="oracle/_ref/libka9q_ref_radio.so not built (needs /root/reference)")
@pytest.mark.parametrize("in_type,bins", [(ol.REAL, 16201), (ol.REAL, 1620001 // 100), (ol.COMPLEX, 14400)])
def test_estimate_noise_matches_reference_radio_c(oracle_built, in_type, bins):
    # estimate_noise() (src/radio.c:1783-1866) run from the reference's own radio.c on the same spectrum: window placement
    # incl. the clamps at DC / Nyquist, inverted (negative-shift) channels, slave bins below and above Min_noise_bins.
    # The reference sums the qualifying energies in the order quickselect left them; the restatement sums in window order:
    # the two agree to the last few bits of a double.
    r = np.random.default_rng(bins + in_type)
    spec = _noisy_spectrum(r, bins)
    fs = 1.296e6
    if in_type == ol.REAL:
        shifts = [0, 1, -1, 300, 499, 500, 501, 700, -700, 5000, -5000, bins - 1, -(bins - 1), bins - 400, bins - 600,
                  bins - 501, bins + 50] + [int(x) for x in r.integers(-bins, bins, 40)]
    else:
        # A complex master's window that reaches the +Nyquist seam is left partly UNINITIALISED by the reference
        # (src/radio.c:1826-1836 breaks out of the fill loop and then reads all nbins entries of a VLA): those shifts
        # have no defined answer and are not compared.  Windows through DC (the wrap) are.
        half = bins // 2
        shifts = [0, 1, -1, 200, -200, 600, -600, 2500, -2500, half - 800, -(half - 800)] + \
                 [int(x) for x in r.integers(-(half - 800), half - 800, 40)]
    for s_bins in (300, 600, 1000, 1200):
        for sh in shifts:
            if in_type == ol.COMPLEX:
                nb = max(s_bins, 1000)
                start = (sh - nb // 2) % bins
                if start < half <= start + nb or start + nb >= bins + half:
                    continue
            want = ol.ref_estimate_noise(spec, in_type, s_bins, sh, fs)
            got = ol.estimate_noise(spec, in_type, s_bins, sh, fs)
            assert got == pytest.approx(want, rel=1e-12, abs=0.0), (s_bins, sh)
            assert want > 0


@pytest.mark.skipif(not ol.have_ref_rx888(), reason="oracle/_ref/libka9q_ref_rx888.so not built (needs /root/reference)")
@pytest.mark.parametrize("randomize", [False, True])
def test_convert_matches_reference_rx888_c(oracle_built, randomize):
    # src/rx888.c:694-767 run from the reference's own rx888.c.  convert_avx2 is what an x86-64 radiod executes;
    # the restatement must equal it bit for bit (samples, energy sum, clip count), de-randomiser included.
    r = np.random.default_rng(77 + randomize)
    x = r.integers(-32768, 32768, 64 * 1024, dtype=np.int64).astype(np.int16)
    x[:8] = [32767, -32768, 32766, -32766, -32767, 0, 1, -1]
    scale = ol.scale_ad(True, 1) * 1.2345
    got, en, clips = ol.convert_i16(x, scale, randomize)
    av = ol.ref_convert_i16(x, scale, randomize, avx2=True)
    if av is not None:
        assert np.array_equal(av[0].view(np.uint32), got.view(np.uint32)) and av[1] == en and av[2] == clips
    port = ol.ref_convert_i16(x, scale, randomize, avx2=False)
    if not randomize:
        assert np.array_equal(port[0].view(np.uint32), got.view(np.uint32)) and port[1] == en and port[2] == clips
    else:
        # the portable fallback's `x ^= (x << 15) >> 14` is evaluated in int and is NOT the LTC2208 de-randomiser its
        # comment and the AVX2 routine describe (if bit 0 is set, flip bits 1..15): the two reference routines disagree
        # with each other here, and the restatement (and the kernel) follow the AVX2 one
        assert av is None or not np.array_equal(port[0], av[0])


# ------------------------------------------------------------------------------------------------
# the linear demodulator.  The reference's OWN demod_linear() (src/linear.c, included unmodified by
# oracle/ref_linear_wrap.c, oscillator = its osc.c, PCM packed by its import.h) is run block after block on seeded
# baseband; the restatement must reproduce its frames.
# ------------------------------------------------------------------------------------------------
def _demod_case(r, nblk, N, bursts=True):
    """baseband with level changes that drive every AGC branch: quiet noise, a strong carrier fading in, a loud burst"""
    t = np.arange(nblk * N)
    x = (r.standard_normal(nblk * N) + 1j * r.standard_normal(nblk * N)) * 1e-4
    x += 0.02 * np.exp(2j * np.pi * 0.031 * t) * np.clip((t - 3 * N) / (2.0 * N), 0, 1)          # carrier fades in over blocks 3..5
    if bursts:
        x[9 * N + 50:9 * N + 120] += 0.9 * np.exp(2j * np.pi * 0.11 * t[9 * N + 50:9 * N + 120])  # loud 6 ms burst: peak-level branch
        x[14 * N:] *= 0.01                                                                        # signal drops: hang, then recovery
    bb = x.astype(np.complex64).reshape(nblk, N)
    power = np.array([np.mean(np.abs(b.astype(np.complex128)) ** 2) for b in bb])
    return bb, power


@pytest.mark.skipif(not ol.have_ref_linear(), reason="oracle/_ref/libka9q_ref_linear.so not built (needs /root/reference)")
@pytest.mark.parametrize("kw", [
    dict(),                                                          # usb-like: mono, AGC, S16BE
    dict(channels=2, encoding=ol.PCM_F32LE),                         # iq: stereo float
    dict(env=True, dc_alpha=0.002, encoding=ol.PCM_S16LE),           # am: envelope + carrier removal
    dict(channels=2, env=True, dc_alpha=0.01, encoding=ol.PCM_F32BE),
    dict(agc=False, gain_db=30.0, shift=500.0),                      # cw: fixed gain, post-detection shift oscillator
    dict(snr_squelch=True, squelch_tail=2, bandwidth=2950.0),        # SNR squelch closing and re-opening
    dict(tuned=False),
    dict(encoding=ol.PCM_MULAW), dict(channels=2, encoding=ol.PCM_ALAW),   # G.711 companding (src/rtp.c:459-533 via export_mulaw / _alaw)
])
def test_linear_demodulator_matches_reference_linear_c(oracle_built, kw):
    r = np.random.default_rng(len(kw) * 7 + 3)
    nblk, N, bt = 40, 240, 0.02
    bb, power = _demod_case(r, nblk, N)
    n0_est = 1e-8 * (1 + 0.3 * r.standard_normal(nblk)) / 2950.0 * 2950.0 / 12000.0
    if kw.get("snr_squelch"):
        power = power.copy(); power[20:28] = 1e-12                   # SNR collapses for 8 blocks, then comes back
    p = ol.lin_params(**kw)
    n0s = np.zeros(nblk); s = np.nan
    for b in range(nblk):                                            # src/radio.c:1466-1473, Power_alpha = 0.10
        s = n0_est[b] if np.isnan(s) else s + 0.10 * (n0_est[b] - s)
        n0s[b] = s
    pcm_r, frame_r, mute_r, pow_r, gain_r = ol.ref_linear_run(p, bb, power, n0s, bt)
    d = ol.LinDemod(p)
    seen = set()
    for b in range(nblk):
        pcm, st = d.block(bb[b], power[b], n0_est[b], bt)
        assert st.frame == frame_r[b] and st.mute == mute_r[b], (b, st.frame, frame_r[b], st.mute, mute_r[b])
        # envelope modes go through cabsf(), which the reference's -funsafe-math-optimizations build inlines as
        # sqrtf(re*re + im*im) (<= 1 float ulp from the library's hypotf the restatement calls); everything else is the same
        # double arithmetic in the same order
        tol = 2e-7 if kw.get("env") else 1e-12
        assert st.gain == pytest.approx(gain_r[b], rel=1e-12)
        assert st.output_power == pytest.approx(pow_r[b], rel=tol, abs=1e-300)
        seen.add((st.frame, st.mute))
        if st.frame == ol.FRAME_DATA:
            if p.encoding in (ol.PCM_MULAW, ol.PCM_ALAW):
                assert np.mean(pcm != pcm_r[b]) < 0.01                              # a code flips only where a sample sits on a step
            elif p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                a = pcm.view(">i2" if p.encoding == ol.PCM_S16BE else "<i2").astype(np.int32)
                w = pcm_r[b].view(">i2" if p.encoding == ol.PCM_S16BE else "<i2").astype(np.int32)
                assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0          # the reference is built with -funsafe-math...
            else:
                a = pcm.view(">f4" if p.encoding == ol.PCM_F32BE else "<f4").astype(np.float64)
                w = pcm_r[b].view(">f4" if p.encoding == ol.PCM_F32BE else "<f4").astype(np.float64)
                assert np.abs(a - w).max() <= 2e-7 * max(np.abs(w).max(), 1e-30)
    if kw.get("snr_squelch"):
        assert (ol.FRAME_SILENCE, 0) in seen and (ol.FRAME_SILENCE, 1) in seen and (ol.FRAME_DATA, 0) in seen


@pytest.mark.skipif(not ol.have_ref_linear(), reason="oracle/_ref/libka9q_ref_linear.so not built (needs /root/reference)")
def test_g711_companding_matches_reference_rtp_c(oracle_built):
    # float_to_mulaw / float_to_alaw of the reference's own rtp.c (through export_mulaw / export_alaw in the linear wrapper's
    # send_output) against the restatement, for every 16-bit level and beyond the clip points: bit-exact
    import ctypes as C
    O = ol.oracle()
    O.chzo_float_to_mulaw.argtypes = [C.c_float]; O.chzo_float_to_mulaw.restype = C.c_ubyte
    O.chzo_float_to_alaw.argtypes = [C.c_float]; O.chzo_float_to_alaw.restype = C.c_ubyte
    lv = np.concatenate([np.arange(-32768, 32769) / 32768.0, [1.5, -1.5, 3e-5, -3e-5, 0.999999, -0.999999]]).astype(np.float32)
    N = 240
    lv = np.concatenate([lv, np.zeros((-len(lv)) % N, np.float32)])
    bb = (lv + 0j).astype(np.complex64).reshape(-1, N)              # real part out at unit gain: I channel, no AGC
    for enc, fn in ((ol.PCM_MULAW, O.chzo_float_to_mulaw), (ol.PCM_ALAW, O.chzo_float_to_alaw)):
        p = ol.lin_params(agc=False, gain_db=0.0, encoding=enc)
        pcm_r, frame_r, _, _, _ = ol.ref_linear_run(p, bb, np.ones(len(bb)), np.full(len(bb), 1e-12), 0.02)
        assert (frame_r == ol.FRAME_DATA).all()
        want = pcm_r.reshape(-1)[:len(lv)]
        got = np.array([fn(float(v)) for v in lv], np.uint8)
        assert np.array_equal(got, want)
    assert O.chzo_float_to_mulaw(0.0) == 0xFF and O.chzo_float_to_alaw(0.0) == 0x55       # the idle codes


def test_pll_oscillator_matches_reference_osc_c(oracle_built):
    # nco() of the reference's own osc.c (inside oracle/_ref/libka9q_ref.so) against the restatement, over the whole phase circle
    import ctypes as C
    R = ol.ref(); O = ol.oracle()
    for L in (R.nco, O.chzo_nco):
        L.argtypes = [C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]; L.restype = None
    r = np.random.default_rng(5)
    acc = np.concatenate([r.integers(0, 2 ** 32, 5000, dtype=np.uint64), np.array([0, 1, 2 ** 20 - 1, 2 ** 20, 2 ** 30 - 1, 2 ** 30, 2 ** 31, 2 ** 32 - 1], np.uint64)])
    for a in acc:
        s1, c1, s2, c2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        R.nco(int(a), C.byref(s1), C.byref(c1)); O.chzo_nco(int(a), C.byref(s2), C.byref(c2))
        assert abs(s1.value - s2.value) <= 3e-16 and abs(c1.value - c2.value) <= 3e-16      # -ffp-contract / -funsafe-math in the reference build
        th = 2 * np.pi * int(a) / 2.0 ** 32
        assert abs(s2.value - np.sin(th)) < 2e-9 and abs(c2.value - np.cos(th)) < 2e-9      # what the table + Taylor step is good for


def _coherent_case(r, nblk, N, square, last=50, first=4):
    """A carrier 30 Hz off tune (5 Hz for the squaring loop, which false-locks near fs/4 when it is left to run on noise with a
    wide bandwidth) that comes up at block 4 and goes away at block 50; AM (or, for the squaring loop, BPSK) on it"""
    t = np.arange(nblk * N)
    fs = 12000.0
    if square:
        mod = np.sign(np.sin(2 * np.pi * 31.25 * t / fs + 0.3))                 # phase reversals: only a squaring loop locks
    else:
        mod = 1 + 0.5 * np.sin(2 * np.pi * 400 * t / fs)
    lvl = np.where((t >= first * N) & (t < last * N), 0.02, 0.0)
    x = lvl * mod * np.exp(2j * np.pi * ((5.0 if square else 30.0) * t / fs) + 0.7j)
    x = x + (r.standard_normal(nblk * N) + 1j * r.standard_normal(nblk * N)) * 4e-4
    bb = x.astype(np.complex64).reshape(nblk, N)
    power = np.array([np.mean(np.abs(b.astype(np.complex128)) ** 2) for b in bb])
    return bb, power


@pytest.mark.skipif(not ol.have_ref_linear(), reason="oracle/_ref/libka9q_ref_linear.so not built (needs /root/reference)")
@pytest.mark.parametrize("kw", [
    dict(pll=True),                                                  # coherent AM: carrier tracking, I channel out
    dict(pll=True, square=True, pll_bw=20.0, channels=2, encoding=ol.PCM_F32LE),  # squaring loop on a BPSK-like signal, I/Q out
    dict(pll=True, env=True, dc_alpha=0.002, pll_bw=50.0, squelch_tail=0, encoding=ol.PCM_S16LE),
])
def test_linear_pll_matches_reference_linear_c(oracle_built, kw):
    # the PLL branch of demod_linear() (src/linear.c:83-153) incl. the lock detector and the squelch it drives, the loop
    # itself from the reference's osc.c
    r = np.random.default_rng(3)             # (on some noise the wide loop runs off to a false lock near fs/4 before the carrier comes up)
    nblk, N, bt = 90, 240, 0.02
    bb, power = _coherent_case(r, nblk, N, kw.get("square", False))
    n0_est = np.full(nblk, 2 * 4e-4 ** 2 / 12000.0)
    p = ol.lin_params(**kw)
    n0s = np.zeros(nblk); s = np.nan
    for b in range(nblk):
        s = n0_est[b] if np.isnan(s) else s + 0.10 * (n0_est[b] - s)
        n0s[b] = s
    pll_r = np.zeros((nblk, 5))
    pcm_r, frame_r, mute_r, pow_r, gain_r = ol.ref_linear_run(p, bb, power, n0s, bt, pll_out=pll_r)
    d = ol.LinDemod(p)
    locked = 0
    seen = set()
    for b in range(nblk):
        pcm, st = d.block(bb[b], power[b], n0_est[b], bt)
        assert st.frame == frame_r[b] and st.mute == mute_r[b], (b, st.frame, frame_r[b], st.mute, mute_r[b])
        assert st.pll_lock == int(pll_r[b, 1]) and st.pll_rotations == int(pll_r[b, 3]), b
        assert st.pll_snr == pytest.approx(pll_r[b, 0], rel=1e-6, abs=1e-9)
        dphi = (st.pll_cphase - pll_r[b, 2] + np.pi) % (2 * np.pi) - np.pi
        assert abs(dphi) < 1e-6                                      # the VCO phase word: a truncation of a double, 2^-32 cycle per flip
        assert st.foffset == pytest.approx(pll_r[b, 4], rel=1e-6, abs=1e-6)
        assert st.gain == pytest.approx(gain_r[b], rel=1e-7)
        assert st.output_power == pytest.approx(pow_r[b], rel=1e-6, abs=1e-300)
        locked += st.pll_lock
        seen.add((st.frame, st.mute))
        if st.frame == ol.FRAME_DATA:
            if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                a, w = pcm.view(dt).astype(np.int32), pcm_r[b].view(dt).astype(np.int32)
                assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0
            else:
                dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
                a, w = pcm.view(dt).astype(np.float64), pcm_r[b].view(dt).astype(np.float64)
                assert np.abs(a - w).max() <= 2e-6 * max(np.abs(w).max(), 1e-30)
    assert 10 < locked < nblk - 10                                   # it locked on the carrier and let go after it went away
    assert (ol.FRAME_DATA, 0) in seen and (ol.FRAME_SILENCE, 1) in seen
    # the loop really tracked the 30 Hz offset while locked
    assert any(abs(pll_r[b, 4] - 30.0) < 1.0 for b in range(30, 48)) or kw.get("square")


def _fm_case(r, nblk, N, fs, tone=0.0, last=26):
    """NBFM baseband: a tone-modulated carrier with a frequency offset that comes up out of the noise, stays, and fades"""
    t = np.arange(nblk * N)
    dev, fmod, foff = 3000.0, 1000.0, 350.0
    phase = 2 * np.pi * (foff * t / fs) - (dev / fmod) * np.cos(2 * np.pi * fmod * t / fs)
    if tone:
        phase = phase - (600.0 / tone) * np.cos(2 * np.pi * tone * t / fs)      # a PL tone with 600 Hz of deviation
    level = np.full(nblk * N, 0.05)
    level[:6 * N] = 0.0; level[last * N:] = 0.0                                   # carrier present in blocks 6..last-1
    level[(last - 4) * N:last * N] = 0.05 * np.linspace(1, 0.02, 4 * N)           # fading out: the squelch tail sequence
    x = level * np.exp(1j * phase) + (r.standard_normal(nblk * N) + 1j * r.standard_normal(nblk * N)) * 2e-3
    bb = x.astype(np.complex64).reshape(nblk, N)
    power = np.array([np.mean(np.abs(b.astype(np.complex128)) ** 2) for b in bb])
    return bb, power


@pytest.mark.skipif(not ol.have_ref_fm(), reason="oracle/_ref/libka9q_ref_fm.so not built (needs /root/reference)")
@pytest.mark.parametrize("kw", [dict(), dict(threshold_extend=True, encoding=ol.PCM_F32LE), dict(deemph_tc=0, encoding=ol.PCM_S16LE),
                                dict(snr_squelch=True, squelch_tail=3, encoding=ol.PCM_F32BE)])
def test_fm_demodulator_matches_reference_fm_c(oracle_built, kw):
    # demod_fm() (src/fm.c:19-345; the PLL and PL-tone branches have their own test below) run from the reference's own fm.c / misc.c / iir.c, block after block
    r = np.random.default_rng(len(kw) + 40)
    nblk, N, fs, bt = 36, 480, 24000.0, 0.02
    bb, power = _fm_case(r, nblk, N, fs)
    p = ol.fm_params(**kw)
    n0_est = (2 * 2e-3 ** 2 / fs) * (1 + 0.1 * r.standard_normal(nblk))         # noise density of the test signal, jittered
    n0s = np.zeros(nblk); s = np.nan
    for b in range(nblk):
        s = n0_est[b] if np.isnan(s) else s + 0.10 * (n0_est[b] - s)
        n0s[b] = s
    ref = ol.ref_fm_run(p, bb, power, n0s, bt)
    d = ol.FmDemod(p)
    seen = set()
    for b in range(nblk):
        pcm, st = d.block(bb[b], power[b], n0_est[b], bt)
        assert st.frame == ref["frame"][b] and st.mute == ref["mute"][b], (b, st.frame, st.mute, ref["frame"][b], ref["mute"][b])
        assert st.snr == pytest.approx(ref["snr"][b], rel=1e-6, abs=1e-12)     # cabsf under -funsafe-math: an ulp of a float per sample
        seen.add((st.frame, st.mute))
        if st.frame == ol.FRAME_DATA:
            assert st.output_power == pytest.approx(ref["power"][b], rel=1e-6)
            assert st.gain == pytest.approx(ref["gain"][b], rel=1e-14)
            assert st.foffset == pytest.approx(ref["foffset"][b], rel=1e-6, abs=1e-6)
            assert st.pdeviation == pytest.approx(ref["pdev"][b], rel=1e-5, abs=1e-3)
            if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                a, w = pcm.view(dt).astype(np.int32), ref["pcm"][b].view(dt).astype(np.int32)
                assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0
            else:
                dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
                a, w = pcm.view(dt).astype(np.float64), ref["pcm"][b].view(dt).astype(np.float64)
                assert np.abs(a - w).max() <= 2e-6 * max(np.abs(w).max(), 1e-30)
    assert (ol.FRAME_DATA, 0) in seen and (ol.FRAME_SILENCE, 0) in seen and (ol.FRAME_SILENCE, 1) in seen


@pytest.mark.skipif(not ol.have_ref_fm(), reason="oracle/_ref/libka9q_ref_fm.so not built (needs /root/reference)")
@pytest.mark.parametrize("kw,tone_sent", [
    (dict(pll=True, encoding=ol.PCM_F32LE), 0.0),                     # PLL demodulator (src/fm.c:176-203)
    (dict(pll=True, threshold_extend=True), 0.0),
    (dict(tone_freq=100.0), 100.0),                                   # PL tone squelch: the tone is there -> opens after 0.24 s
    (dict(tone_freq=100.0, deemph_tc=0, encoding=ol.PCM_S16LE), 0.0), # ... and is not: stays muted
    (dict(tone_freq=123.0, pll=True), 100.0),                         # the wrong tone
])
def test_fm_pll_and_tone_squelch_match_reference_fm_c(oracle_built, kw, tone_sent):
    r = np.random.default_rng(len(kw) + 60 + int(tone_sent))
    nblk, N, fs, bt = 72, 480, 24000.0, 0.02
    bb, power = _fm_case(r, nblk, N, fs, tone=tone_sent, last=60)    # the tone detector integrates 12 blocks at a time
    bb = bb.copy(); power = power.copy()
    p = ol.fm_params(**kw)
    n0_est = (2 * 2e-3 ** 2 / fs) * (1 + 0.1 * r.standard_normal(nblk))
    n0s = np.zeros(nblk); s = np.nan
    for b in range(nblk):
        s = n0_est[b] if np.isnan(s) else s + 0.10 * (n0_est[b] - s)
        n0s[b] = s
    ref = ol.ref_fm_run(p, bb, power, n0s, bt)
    d = ol.FmDemod(p)
    data = 0
    for b in range(nblk):
        pcm, st = d.block(bb[b], power[b], n0_est[b], bt)
        assert st.frame == ref["frame"][b] and st.mute == ref["mute"][b], (b, st.frame, st.mute, ref["frame"][b], ref["mute"][b])
        assert st.snr == pytest.approx(ref["snr"][b], rel=1e-6, abs=1e-12)
        assert st.tone_deviation == pytest.approx(ref["tonedev"][b], rel=1e-5, abs=1e-6)
        if st.frame == ol.FRAME_DATA:
            data += 1
            assert st.output_power == pytest.approx(ref["power"][b], rel=2e-6)
            assert st.foffset == pytest.approx(ref["foffset"][b], rel=1e-5, abs=1e-5)
            assert st.pdeviation == pytest.approx(ref["pdev"][b], rel=1e-5, abs=1e-3)
            if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                a, w = pcm.view(dt).astype(np.int32), ref["pcm"][b].view(dt).astype(np.int32)
                assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0
            else:
                dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
                a, w = pcm.view(dt).astype(np.float64), ref["pcm"][b].view(dt).astype(np.float64)
                assert np.abs(a - w).max() <= 4e-6 * max(np.abs(w).max(), 1e-30)
    if p.tone_freq and tone_sent != p.tone_freq:
        assert data == 0                                             # muted throughout
    else:
        assert data >= 5
    if p.tone_freq == tone_sent and tone_sent:
        assert max(ref["tonedev"]) == pytest.approx(600.0, rel=0.1)  # the detector measures the tone's deviation


# ------------------------------------------------------------------------------------------------
# randomised sweeps: the restated demodulators against the reference's own code over parameter combinations nobody picked by hand
# ------------------------------------------------------------------------------------------------
def _cmp_pcm(p, got, want, tol_f):
    if p.encoding in (ol.PCM_MULAW, ol.PCM_ALAW, ol.PCM_F16LE, ol.PCM_F16BE):
        return np.mean(got != want) == 0
    if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
        dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
        a, w = got.view(dt).astype(np.int32), want.view(dt).astype(np.int32)
        return np.abs(a - w).max() <= 1 and np.mean(a != w) == 0
    dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
    a, w = got.view(dt).astype(np.float64), want.view(dt).astype(np.float64)
    return np.abs(a - w).max() <= tol_f * max(np.abs(w).max(), 1e-30)


@pytest.mark.skipif(not ol.have_ref_linear(), reason="oracle/_ref/libka9q_ref_linear.so not built (needs /root/reference)")
def test_linear_demodulator_random_parameter_sweep(oracle_built):
    rng = np.random.default_rng(20260926)
    encs = [ol.PCM_S16BE, ol.PCM_S16LE, ol.PCM_F32LE, ol.PCM_F32BE, ol.PCM_MULAW, ol.PCM_ALAW]
    for case in range(30):
        kw = dict(channels=int(rng.integers(1, 3)), env=bool(rng.integers(0, 2)), agc=bool(rng.integers(0, 4) > 0),
                  encoding=encs[int(rng.integers(0, len(encs)))], snr_squelch=bool(rng.integers(0, 3) == 0), squelch_tail=int(rng.integers(0, 4)),
                  tuned=bool(rng.integers(0, 8) > 0), headroom_db=float(rng.uniform(-25, -5)), threshold_db=float(rng.uniform(-25, -5)),
                  recovery_db_per_s=float(rng.uniform(5, 40)), hangtime=float(rng.uniform(0.0, 0.3)),
                  dc_alpha=float(rng.choice([0.0, 0.002, 0.02])), bandwidth=float(rng.uniform(500, 6000)),
                  shift=float(rng.choice([0.0, 0.0, 700.0, -431.5])), gain_db=float(rng.uniform(20, 70)))
        nblk, N = 24, int(rng.choice([240, 160, 480]))
        kw["samprate"] = N * 50.0
        bb, power = _demod_case(np.random.default_rng(1000 + case), nblk, N, bursts=bool(rng.integers(0, 2))) if nblk >= 16 else (None, None)
        if kw["snr_squelch"]:
            power = power.copy(); power[10:15] = 1e-12
        n0_est = 1e-8 * (1 + 0.3 * rng.standard_normal(nblk)) / kw["samprate"]
        p = ol.lin_params(**kw)
        n0s = np.zeros(nblk); s = np.nan
        for b in range(nblk):
            s = n0_est[b] if np.isnan(s) else s + 0.10 * (n0_est[b] - s)
            n0s[b] = s
        pcm_r, frame_r, mute_r, pow_r, gain_r = ol.ref_linear_run(p, bb, power, n0s, 0.02)
        d = ol.LinDemod(p)
        for b in range(nblk):
            pcm, st = d.block(bb[b], power[b], n0_est[b], 0.02)
            assert st.frame == frame_r[b] and st.mute == mute_r[b], (case, kw, b)
            assert st.gain == pytest.approx(gain_r[b], rel=1e-9), (case, b)
            assert st.output_power == pytest.approx(pow_r[b], rel=3e-7, abs=1e-300), (case, b)
            if st.frame == ol.FRAME_DATA:
                # (envelope detection goes through cabsf -- an ulp of a float between the two builds -- and carrier removal then subtracts
                # nearly all of it: what is left carries that ulp at a relative size of envelope / residue)
                assert _cmp_pcm(p, pcm, pcm_r[b][:pcm.size], 1e-4 if (kw["env"] and kw["dc_alpha"]) else 3e-7), (case, kw, b)


@pytest.mark.skipif(not ol.have_ref_fm(), reason="oracle/_ref/libka9q_ref_fm.so not built (needs /root/reference)")
def test_fm_demodulator_random_parameter_sweep(oracle_built):
    rng = np.random.default_rng(7)
    encs = [ol.PCM_S16BE, ol.PCM_S16LE, ol.PCM_F32LE, ol.PCM_F32BE, ol.PCM_MULAW, ol.PCM_ALAW]
    for case in range(20):
        tone = float(rng.choice([0.0, 0.0, 100.0, 88.5]))
        kw = dict(encoding=encs[int(rng.integers(0, len(encs)))], snr_squelch=bool(rng.integers(0, 3) == 0), squelch_tail=int(rng.integers(0, 4)),
                  threshold_extend=bool(rng.integers(0, 2)), deemph_tc=float(rng.choice([0.0, 530.5e-6, 75e-6])),
                  headroom_db=float(rng.uniform(-20, -6)), pll=bool(rng.integers(0, 4) == 0), tone_freq=tone,
                  squelch_open=float(rng.uniform(3.0, 8.0)))
        kw["squelch_close"] = kw["squelch_open"] * float(rng.uniform(0.5, 0.9))
        nblk, N, fs = 52, 480, 24000.0
        sent = tone if rng.integers(0, 3) else 0.0
        bb, power = _fm_case(np.random.default_rng(500 + case), nblk, N, fs, tone=sent, last=44)
        p = ol.fm_params(**kw)
        n0_est = (2 * 2e-3 ** 2 / fs) * (1 + 0.1 * rng.standard_normal(nblk))
        n0s = np.zeros(nblk); s = np.nan
        for b in range(nblk):
            s = n0_est[b] if np.isnan(s) else s + 0.10 * (n0_est[b] - s)
            n0s[b] = s
        ref = ol.ref_fm_run(p, bb, power, n0s, 0.02)
        d = ol.FmDemod(p)
        for b in range(nblk):
            pcm, st = d.block(bb[b], power[b], n0_est[b], 0.02)
            assert st.frame == ref["frame"][b] and st.mute == ref["mute"][b], (case, kw, b)
            assert st.snr == pytest.approx(ref["snr"][b], rel=1e-6, abs=1e-12)
            assert st.tone_deviation == pytest.approx(ref["tonedev"][b], rel=1e-5, abs=1e-6)
            if st.frame == ol.FRAME_DATA:
                assert st.output_power == pytest.approx(ref["power"][b], rel=3e-6)
                assert _cmp_pcm(p, pcm, ref["pcm"][b][:pcm.size], 6e-6), (case, kw, b)


def test_f16_packing_equals_the_references_import_h():
    """F16LE / F16BE PCM (src/audio.c:135-139): the restated float -> binary16 conversion against the reference's own
    export_f16_le / export_f16_be (src/import.h:140-157,207-212; built with clang, which has _Float16 where this image's gcc does
    not), bit for bit: every binary16 value and its neighbours' midpoints (all the rounding ties), 4 M random bit patterns,
    audio-range samples, subnormals, overflow, infinities; and the bytes import back to the values they stand for."""
    ref = ol.ref_f16()
    if ref is None:
        pytest.skip("oracle/_ref/libka9q_ref_f16.so absent (no clang with _Float16)")
    rng = np.random.default_rng(16)
    halfs = np.arange(0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)          # every finite non-negative binary16
    mids = (halfs[:-1].astype(np.float64) + halfs[1:].astype(np.float64)) / 2                  # exact ties between neighbours
    ties = np.concatenate([mids.astype(np.float32), np.nextafter(mids.astype(np.float32), np.float32(0)), np.nextafter(mids.astype(np.float32), np.float32(1e9))])
    rnd = rng.integers(0, 2 ** 32, 4_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    rnd = rnd[np.isfinite(rnd)]
    audio = (rng.standard_normal(1_000_000) * 0.2).astype(np.float32)
    special = np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 70000.0, -70000.0, np.inf, -np.inf, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -26,
                        6.1035156e-05, 6.0975552e-05, 1e-8, -1e-8], np.float32)
    x = np.concatenate([halfs, -halfs, ties, -ties, rnd, audio, special])
    for be in (0, 1):
        want = np.zeros(x.size, np.uint16)
        ref.ref_export_f16(want.ctypes.data, x.ctypes.data, x.size, be)
        got = np.zeros(x.size, np.uint16)
        ol.oracle().chzo_pcm_pack.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        ol.oracle().chzo_pcm_pack(ol.PCM_F16BE if be else ol.PCM_F16LE, x.ctypes.data, x.size, got.ctypes.data)
        assert np.array_equal(got, want), (be, x[got != want][:5], got[got != want][:5], want[got != want][:5])
    back = np.zeros(halfs.size, np.float32)
    bits = np.arange(0x7c00, dtype=np.uint16)
    ref.ref_import_f16(back.ctypes.data, bits.ctypes.data, bits.size, 0)
    assert np.array_equal(back, halfs)


@pytest.mark.parametrize("n", [3240000, 134400, 2 * 65536 * 3])
def test_float32_timing_provider_long_real_transform(n):
    """oracle/dft.c's float32 path for long real transforms -- the CPU BASELINE's forward FFT, not the parity oracle: the four-step
    transform with 8 sub-transforms side by side in AVX2 lanes (oracle/dft_batch.h) and the work split over fft-internal-threads
    (odft_set_threads <- fftwf_plan_with_nthreads).  Against numpy's float64 rfft; the result does not depend on the thread count.
    n = 3,240,000 is config 3's window (n1 = 1296, n2 = 1250: tail groups of 2 columns), 134,400 has a radix-7 level."""
    import ctypes as C
    lib = ol.oracle()
    lib.odft_create.restype = C.c_void_p; lib.odft_create.argtypes = [C.c_int, C.c_int]
    lib.odft_set_threads.argtypes = [C.c_void_p, C.c_int]; lib.odft_warm.argtypes = [C.c_void_p, C.c_int]
    lib.odft_r2c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; lib.odft_destroy.argtypes = [C.c_void_p]
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    want = np.fft.rfft(x.astype(np.float64))
    outs = []
    for threads in (1, 3):
        p = lib.odft_create(n, 1)
        lib.odft_set_threads(p, threads); lib.odft_warm(p, 1)
        out = np.zeros(n + 2, np.float32)
        lib.odft_r2c(p, x.ctypes.data, out.ctypes.data)
        lib.odft_destroy(p)
        got = out.view(np.complex64)
        assert np.linalg.norm(got - want) <= 5e-7 * np.linalg.norm(want)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])
