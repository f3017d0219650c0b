"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Everything here needs a real MI355X (`-m gpu`).  The HIP path is compared with
  * oracle/chz_oracle.c carried in float64 (the "exact" answer), and
  * oracle/_ref = the reference's own filter.c (prebuilt .so travels with the repo)
on identical seeded / sig_gen inputs.  Tolerances (float32 path):
per-channel relative L2 <= 1e-5 and max-abs <= 1e-4 x rms of the oracle block; the
forward spectrum itself is held to relative L2 <= 1e-6.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_pkg

pytestmark = pytest.mark.gpu

REL_L2 = 1e-5          # per-channel relative L2
MAXABS_RMS = 1e-4      # per-channel max-abs, in units of the oracle block's rms
SPEC_REL = 1e-6        # forward spectrum relative L2
# float32 dynamic-range floor.  A float32 transform's rounding error in a bin is not
# proportional to what that bin holds: it is set by the strongest line of the window (the
# partial sums of the sig_gen carrier are large in every butterfly layer that still contains
# it), so a weak narrow channel far from the carrier carries an ABSOLUTE error.  Measured on
# config 2/3 (scripts/diag_accuracy.py): worst per-bin error in a channel passband
# 2.3e-9 x max|X|, for this HIP path and for the reference's filter.c on a float32 CPU FFT
# alike.  Allowed: FLOOR x max|X| per bin, i.e. a third of a float32 ulp of the strongest
# line, propagated through the response as ||H||_2.
FLOOR = 2e-8


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    if p.engine.lib().chz_device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests cannot run (there is no CPU fallback)")
    ol.build()
    return p


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def noise_floor(spec64, resp):
    """FLOOR x max|X| x ||H||_2 : absolute rms error allowance from float32 forward-transform noise."""
    return FLOOR * float(np.abs(spec64).max()) * float(np.linalg.norm(resp))


def check_channel(got, want, floor=0.0):
    """err_rms <= REL_L2 * want_rms + floor  and  max|err| <= MAXABS_RMS * want_rms + 6 floor."""
    want = np.asarray(want)
    rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
    if rms == 0:
        assert np.abs(got).max() == 0
        return 0.0
    err = float(np.sqrt(np.mean(np.abs(got - want) ** 2)))
    assert err <= REL_L2 * rms + floor, "rms error %g vs channel rms %g (floor %g)" % (err, rms, floor)
    assert np.abs(got - want).max() <= MAXABS_RMS * rms + 6 * floor
    return err / rms


# ------------------------------------------------------------------------------
# forward transform
# ------------------------------------------------------------------------------
@pytest.mark.parametrize("L,M,in_type,plan", [
    (25920, 6481, ol.REAL, ""),               # N = 32400
    (25920, 6481, ol.REAL, "81x400"),         # two-axis plan
    (25920, 6481, ol.REAL, "225x144"),        # odd first axis
    (11520, 2881, ol.REAL, "16x25x36"),       # N = 14400
    (48000, 12001, ol.COMPLEX, ""),           # config 1: N = 60000 complex
    (11520, 2881, ol.COMPLEX, "16x25x36"),
    (1296000, 324001, ol.REAL, ""),           # config 2: N = 1,620,000
    (2592000, 648001, ol.REAL, ""),           # config 3: N = 3,240,000
    (400000, 100001, ol.REAL, ""),            # Airspy R2, rof500000 (docs/FFTW3.md:52,60)
    (18240, 18241, ol.COMPLEX, ""),           # Airspy HF+, cof36480 = 2^7*3*5*19 (docs/FFTW3.md:52,62): needs the radix-19 butterfly
    (2600000, 650001, ol.REAL, ""),           # RX888 at its 130 MS/s ceiling: N = 3,250,000 = 2^4*5^6*13 (axes of 130 = 10x13 and 200 = 10x20)
    (2500000, 625001, ol.REAL, ""),           # 125 MS/s: N = 3,125,000 = 2^3*5^8
])
def test_forward_matches_oracle(pkg, L, M, in_type, plan):
    rng = np.random.default_rng(L + in_type)
    eng = pkg.engine.Engine(L, M, in_type, plan=plan)
    st = ol.Stream(L, M, in_type)
    try:
        for job in range(3):
            if in_type == ol.REAL:
                x = rng.standard_normal(L).astype(np.float32)
            else:
                x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
            eng.write(x)
            eng.forward(job)
            got = eng.spectrum(job % 4)
            want = st.push(x, f64=True)
            assert rel(got, want) <= SPEC_REL, eng.plan
            # element-wise: no bin may be off by more than a few float32 ulps of the spectrum scale
            assert np.abs(got - want).max() <= 2e-5 * np.sqrt(np.mean(np.abs(want) ** 2)) * np.sqrt(np.log2(eng.N))
    finally:
        eng.close()


def test_forward_ring_wraps_over_many_blocks(pkg):
    # 11 blocks through an 8-block device ring: every window straddles the wrap at some point
    L, M = 25920, 6481
    rng = np.random.default_rng(5)
    eng = pkg.engine.Engine(L, M, ol.REAL)
    st = ol.Stream(L, M, ol.REAL)
    try:
        for job in range(11):
            x = rng.standard_normal(L).astype(np.float32)
            eng.write(x)
            eng.forward(job)
            assert rel(eng.spectrum(job % 4), st.push(x, f64=True)) <= SPEC_REL, job
    finally:
        eng.close()


def test_forward_linearity_full_size(pkg):
    # size-independent property at BASELINE's full size: F(a x + b y) = a F(x) + b F(y)
    L, M = 2592000, 648001
    rng = np.random.default_rng(9)
    x = rng.standard_normal(L).astype(np.float32)
    y = rng.standard_normal(L).astype(np.float32)
    z = (0.5 * x - 0.25 * y).astype(np.float32)     # exact in float32
    specs = []
    for v in (x, y, z):
        eng = pkg.engine.Engine(L, M, ol.REAL)
        eng.write(v); eng.forward(0)
        specs.append(eng.spectrum(0))
        eng.close()
    assert rel(specs[2], 0.5 * specs[0] - 0.25 * specs[1]) <= 2e-6
    # Parseval on the first window (M-1 zeros then x): sum|x|^2 = (|X0|^2 + |X_N/2|^2 + 2 sum|X_k|^2)/N
    N = L + M - 1
    X = specs[0].astype(np.complex128)
    e_f = (abs(X[0]) ** 2 + abs(X[-1]) ** 2 + 2 * np.sum(np.abs(X[1:-1]) ** 2)) / N
    e_t = float(np.sum(x.astype(np.float64) ** 2))
    assert abs(e_f - e_t) <= 1e-5 * e_t


def test_notch_state_carries_across_blocks(pkg):
    L, M = 25920, 6481
    rng = np.random.default_rng(11)
    eng = pkg.engine.Engine(L, M, ol.REAL)
    st = ol.Stream(L, M, ol.REAL)
    bins = [125, 4000, 0]
    eng.set_notches(bins, 0.01)
    state = np.zeros(2 * len(bins))
    try:
        for job in range(6):
            x = (rng.standard_normal(L) + 0.3).astype(np.float32)
            eng.write(x); eng.forward(job)
            want = st.push(x)
            ol.notch(state, bins, 0.01, want)
            got = eng.spectrum(job % 4)
            for b in bins:
                assert abs(got[b] - want[b]) <= 2e-6 * abs(want[b]) + 2e-3, (job, b)
            assert rel(got, want) <= SPEC_REL
    finally:
        eng.close()


# ------------------------------------------------------------------------------
# channels
# ------------------------------------------------------------------------------
def _mixed_plan(n, fs_in, N, rng):
    """Config-3 style channel plan: thirds usb / cw / iq (share/presets.conf), P = 300."""
    kinds = [(50 / 12000, 3000 / 12000), (-200 / 12000, 200 / 12000), (-5000 / 12000, 5000 / 12000)]
    plan = []
    for i in range(n):
        if 1e6 + n * 60e3 < fs_in / 2:       # SURVEY section 8d, config 3 raster
            f = 1e6 + i * 60e3 + (i % 40)
        else:                                # same idea scaled to a smaller master
            f = 0.02 * fs_in + i * (0.46 * fs_in / n) + (i % 40)
        _, shift, _ = ol.compute_tuning(N, fs_in, f)
        low, high = kinds[i % 3]
        plan.append((shift, low, high))
    return plan


@pytest.mark.parametrize("P,olen", [(300, 240), (600, 480), (200, 160), (400, 320), (1200, 960),
                                    (150, 120), (160, 128), (320, 256), (480, 384), (800, 640), (960, 768), (1920, 1536)])
def test_channel_sizes_random_spectrum(pkg, P, olen):
    # every compiled backward-transform size, random responses, edge + random shifts, both signs
    L, M = 25920, 6481
    fa = pkg.filterapi
    rng = np.random.default_rng(P)
    master = fa.create_filter_input(L, M, fa.REAL)
    st = ol.Stream(L, M, ol.REAL)
    B = master.bins
    shifts = [0, 1, -1, 5000, -5000, P // 2, -(P // 2), B - 1, -(B - 1), B - P // 2, B + 10, -(B + 10),
              B + P, -(B + P)] + [int(s) for s in rng.integers(-B - P, B + P, 34)]
    slaves = []
    for _ in shifts:
        s = fa.create_filter_output(master, olen, fa.COMPLEX)
        assert s is not None and s.points == P
        fa.set_response(s, (rng.standard_normal(P) + 1j * rng.standard_normal(P)).astype(np.complex64))
        slaves.append(s)
    try:
        for blk in range(2):
            x = rng.standard_normal(L).astype(np.float32)
            assert fa.write_rfilter(master, x) == 1
            spec64 = st.push(x, f64=True)
            for s, sh in zip(slaves, shifts):
                assert fa.execute_filter_output(s, sh) == 0
                check_channel(s.output, ol.channel(spec64, ol.REAL, P, olen, sh, s.response))
    finally:
        fa.delete_filter_input(master)


@pytest.mark.parametrize("P,olen", [(250, 200), (1000, 800), (2700, 2160), (9600, 7680), (5500, 4400), (10080, 8064)])   # the last two: 220 k, 403.2 k (factors 11, 7)
def test_channel_sizes_outside_the_menu(pkg, P, olen):
    # sizes without a register-tiled kernel run through chan_any (one workgroup per channel, Stockham stages in LDS);
    # 9600 is the WFM channel of src/wfm.c:37-39 (384 kHz x 20 ms x overlap 5/4)
    L, M = 25920, 6481
    fa = pkg.filterapi
    rng = np.random.default_rng(P)
    master = fa.create_filter_input(L, M, fa.REAL)
    st = ol.Stream(L, M, ol.REAL)
    B = master.bins
    shifts = [0, -1, 5000, P // 2, -(P // 2), B - 1, B - P // 2, -(B + 10)] + [int(s) for s in rng.integers(-B - P, B + P, 4 if P > 4000 else 16)]
    slaves = []
    for _ in shifts:
        s = fa.create_filter_output(master, olen, fa.COMPLEX)
        assert s is not None and s.points == P
        fa.set_response(s, (rng.standard_normal(P) + 1j * rng.standard_normal(P)).astype(np.complex64))
        slaves.append(s)
    try:
        for blk in range(2):
            x = rng.standard_normal(L).astype(np.float32)
            assert fa.write_rfilter(master, x) == 1
            spec64 = st.push(x, f64=True)
            for s, sh in zip(slaves, shifts):
                assert fa.execute_filter_output(s, sh) == 0
                check_channel(s.output, ol.channel(spec64, ol.REAL, P, olen, sh, s.response))
    finally:
        fa.delete_filter_input(master)


@pytest.mark.parametrize("P,olen", [(85, 68), (115, 92), (2495, 1996), (9995, 7996)])   # factors 17, 23, 499, 1999; the last one beyond the LDS (M = 32768)
def test_channel_sizes_with_large_prime_factors(pkg, P, olen):
    # round 3: FFTW plans any size (src/filter.c:101-163); a channel size with a prime factor above 13 now runs through Bluestein's
    # chirp-z identity inside chan_any, through the filter.h mirror, against the restatement (whose transform is the DFT by definition)
    L, M = 25920, 6481
    fa = pkg.filterapi
    rng = np.random.default_rng(P)
    master = fa.create_filter_input(L, M, fa.REAL)
    st = ol.Stream(L, M, ol.REAL)
    B = master.bins
    shifts = [0, -1, 5000, P // 2, -(P // 2), B - 1] + [int(s) for s in rng.integers(-B - P, B + P, 4)]
    slaves = []
    for _ in shifts:
        s = fa.create_filter_output(master, olen, fa.COMPLEX)
        assert s is not None and s.points == P
        fa.set_response(s, (rng.standard_normal(P) + 1j * rng.standard_normal(P)).astype(np.complex64))
        slaves.append(s)
    try:
        for blk in range(2):
            x = rng.standard_normal(L).astype(np.float32)
            assert fa.write_rfilter(master, x) == 1
            spec64 = st.push(x, f64=True)
            for s, sh in zip(slaves, shifts):
                assert fa.execute_filter_output(s, sh) == 0
                check_channel(s.output, ol.channel(spec64, ol.REAL, P, olen, sh, s.response))
    finally:
        fa.delete_filter_input(master)


@pytest.mark.parametrize("L,M,kind", [(1800, 219, "real"), (5400, 1352, "real"), (1000, 10, "complex"), (4000, 201, "complex"), (25920, 6482, "real")])
def test_master_lengths_outside_the_compiled_axes(pkg, L, M, kind):
    # FFTW plans every N (src/filter.c:222-231): N = 2018 = 2 x 1009, 6751 (prime), 1009 (prime), 4200 (smooth, but its factor 7 is in
    # none of the compiled axis lengths), 32401 = 3 x 10800 + 1 -- the engine runs them as Bluestein's chirp-z over the next planned
    # complex length.  Spectrum against the restatement's float64 DFT, channels through it,
    # blocks in sequence (the ring wraps), a notch on top.
    in_type = ol.REAL if kind == "real" else ol.COMPLEX
    N = L + M - 1
    rng = np.random.default_rng(N)
    eng = pkg.engine.Engine(L, M, in_type, ring_blocks=8)
    try:
        assert "chirp-z" in eng.plan
        mk = (lambda: rng.standard_normal(L).astype(np.float32)) if kind == "real" else (lambda: (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64))
        st = ol.Stream(L, M, in_type)
        B = eng.bins
        # the smallest channel this master can have: P / N = olen / L in integers
        g = int(np.gcd(N, L))
        olen, P = L // g, N // g
        while P < 8:
            olen, P = 2 * olen, 2 * P
        if P > 8000:
            P = None                                                  # (N and L coprime: the only "channel" is the whole band)
        for job in range(11):
            x = mk()
            eng.write(x); eng.forward(job)
            want = st.push(x, f64=True)
            got = eng.spectrum(job % 4)
            assert got.shape == want.shape
            assert rel(got, want) <= 3 * SPEC_REL, (job, rel(got, want))
        if P:
            nch = 6
            bank = eng.bank(P, olen, nch)
            resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64) / P
            shifts = np.array([0, 7, -9, B // 3, -(B // 3), B // 2 - 1], np.int32) if kind == "complex" else np.array([0, 7, -9, B // 3, -(B // 3), B - 1], np.int32)
            bank.set_responses(0, resp); bank.set_shifts(0, shifts); bank.set_active(nch)
            x = mk()
            eng.write(x); eng.step(11)
            spec64 = st.push(x, f64=True)
            out = bank.read_slot(11 % 4)
            for c in range(nch):
                check_channel(out[c], ol.channel(spec64, in_type, P, olen, int(shifts[c]), resp[c]), floor=0.0)
    finally:
        eng.close()


def test_channel_sizes_beyond_the_lds(pkg):
    # 768 kHz and 1.536 MHz channels of the full-rate master (P = 19200, 38400; share/*.conf has both): chan_any with its two
    # transform buffers in global scratch.  The restatement's gather + float64 inverse DFT on the DEVICE's own spectrum.
    L, M = 2592000, 648001
    rng = np.random.default_rng(5)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    try:
        eng.write((rng.standard_normal(L) * 0.05).astype(np.float32))
        eng.forward(0)
        spec = eng.spectrum(0).astype(np.complex128)
        B = spec.shape[0]
        # [r6] ... and beyond round 5's 65536-point limit: a 3.072 MHz and a 9.6 MHz channel, one whose size has a prime factor above 13 (chirp-z over 2^18 points), and a million-point one (40 MHz wide)
        for P, olen in ((19200, 15360), (38400, 30720), (76800, 61440), (240000, 192000), (85000, 68000), (1000000, 800000)):
            shifts = np.array([250000, -(B - P // 4), B - 9000], np.int32)
            bank = eng.bank(P, olen, len(shifts))
            resp = (rng.standard_normal((len(shifts), P)) + 1j * rng.standard_normal((len(shifts), P))).astype(np.complex64) / P
            bank.set_responses(0, resp); bank.set_shifts(0, shifts); bank.set_active(len(shifts))
            bank.execute(0)
            eng.sync()
            got = bank.read_slot(0)
            for i, sh in enumerate(shifts):
                want = ol.channel(spec, ol.REAL, P, olen, int(sh), resp[i])
                assert np.linalg.norm(got[i] - want) <= 3e-6 * np.linalg.norm(want), (P, i)
    finally:
        eng.close()


def test_complex_master_channels(pkg):
    # config 1 geometry (2.4 MS/s complex, N = 60000): wrap through DC, both Nyquist seams
    L, M = 48000, 12001
    fa = pkg.filterapi
    rng = np.random.default_rng(21)
    master = fa.create_filter_input(L, M, fa.COMPLEX)
    st = ol.Stream(L, M, ol.COMPLEX)
    B = master.bins
    shifts = [0, 2500, -2500, B // 2, -(B // 2), B // 2 - 100, -(B // 2) + 100, B // 2 + 100, B - 1, -(B - 1),
              B, -B, B + 200] + [int(s) for s in rng.integers(-B, B, 30)]
    slaves = []
    for _ in shifts:
        s = fa.create_filter_output(master, 240, fa.COMPLEX)
        assert fa.set_filter(s, -5000 / 12000, 5000 / 12000, 11.0) == 0
        slaves.append(s)
    try:
        for blk in range(2):
            x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
            assert fa.write_cfilter(master, x) == 1
            spec64 = st.push(x, f64=True)
            for s, sh in zip(slaves, shifts):
                assert fa.execute_filter_output(s, sh) == 0
                check_channel(s.output, ol.channel(spec64, ol.COMPLEX, 300, 240, sh, s.response))
    finally:
        fa.delete_filter_input(master)


def test_bin_centred_carrier_known_answer(pkg):
    # SURVEY section 8c known answer (1): a cos(2 pi k0 n / N), channel at shift k0, symmetric filter
    #  -> every output sample = a / sqrt(2); blocks rotate by 2 pi (k0 mod 5) 4/5.
    L, M = 25920, 6481
    N = L + M - 1
    fa = pkg.filterapi
    master = fa.create_filter_input(L, M, fa.REAL)
    s = fa.create_filter_output(master, 240, fa.COMPLEX)
    fa.set_filter(s, -5000 / 12000, 5000 / 12000, 11.0)
    k0, a = 2501, 0.1
    n = np.arange(4 * L)
    x = (a * np.cos(2 * np.pi * k0 * n / N)).astype(np.float32)
    outs = []
    try:
        for blk in range(4):
            fa.write_rfilter(master, x[blk * L:(blk + 1) * L])
            fa.execute_filter_output(s, k0)
            outs.append(s.output.copy())
        for blk in range(1, 4):          # block 0 still contains the start-up transient
            assert np.allclose(np.abs(outs[blk]), a / np.sqrt(2), rtol=2e-5)
            step = outs[blk][0] / outs[blk - 1][0] if blk > 1 else None
            if step is not None:
                assert np.angle(step) == pytest.approx(np.angle(np.exp(2j * np.pi * (k0 % 5) * 4 / 5)), abs=1e-4)
    finally:
        fa.delete_filter_input(master)


def _siggen_run(pkg, L, M, fs, nch, P, olen, nblocks, ref_check, carrier_hz=10.00002e6):
    """sig_gen input, config-style channel plan; compare with float64 oracle (and the reference)."""
    fa = pkg.filterapi
    N = L + M - 1
    rng = np.random.default_rng(77)
    master = fa.create_filter_input(L, M, fa.REAL)
    fa.set_notches(master, [0], 0.01)
    plan = _mixed_plan(nch, fs, N, rng)
    slaves = []
    for shift, low, high in plan:
        s = fa.create_filter_output(master, olen, fa.COMPLEX)
        assert fa.set_filter(s, low, high, 11.0) == 0
        slaves.append(s)
    gen = ol.SigGen(carrier_hz / fs, 10 ** (-20 / 20), 10 ** (-40 / 20), ol.scale_ad(True, 1), True, seed=1)
    st = ol.Stream(L, M, ol.REAL)
    state = np.zeros(2)
    refm = refc = None
    if ref_check and ol.have_ref():
        refm = ol.RefMaster(L, M, ol.REAL)
        refm.set_notches([0], 0.01)
        refc = []
        for (shift, low, high) in plan[:ref_check]:
            c = refm.channel(olen, ol.COMPLEX)
            c.set_filter(low, high, 11.0)
            refc.append(c)
    worst = 0.0
    try:
        for blk in range(nblocks):
            x = gen.generate(L)
            assert fa.write_rfilter(master, x) == 1
            spec64 = st.push(x, f64=True)
            dc = spec64[:1].astype(np.complex64)
            ol.notch(state, [0], 0.01, dc)
            spec64[0] = dc[0]
            if refm is not None:
                refm.write(x)
            for i, (s, (shift, low, high)) in enumerate(zip(slaves, plan)):
                assert fa.execute_filter_output(s, shift) == 0
                fl = noise_floor(spec64, s.response)
                worst = max(worst, check_channel(s.output, ol.channel(spec64, ol.REAL, P, olen, shift, s.response), fl))
                if refc is not None and i < len(refc):
                    # against the reference's own filter.c output (float32 buffers, float64 FFT shim)
                    want = refc[i].execute(shift)
                    check_channel(s.output, want, fl)
                    assert np.abs(refc[i].response() - s.response).max() <= 2e-7 * np.abs(s.response).max()
    finally:
        fa.delete_filter_input(master)
        if refm is not None:
            refm.close()
    return worst


def test_siggen_scaled_down_vs_reference(pkg):
    # 1.296 MS/s real (RX888 geometry / 100): N = 32400, 48 channels, every one checked against the
    # reference's own filter.c as well as the float64 oracle
    _siggen_run(pkg, 25920, 6481, 1.296e6, 48, 300, 240, 4, ref_check=48, carrier_hz=100020.0)


def test_siggen_config2_halfrate_256_channels(pkg):
    # BASELINE config 2: 64.8 MS/s real, 256 x 12 kHz channels (P = 300); round 6: EVERY channel also against the reference's own filter.c
    _siggen_run(pkg, 1296000, 324001, 64.8e6, 256, 300, 240, 2, ref_check=256)


def test_siggen_config3_fullrate_1024_channels(pkg):
    # BASELINE config 3: 129.6 MS/s real, 1024 mixed usb/cw/iq channels (P = 300); round 6: EVERY channel also against the reference's own
    # filter.c (rounds 2-5 compared the first 8: the reference side costs 0.2 s per block, not minutes)
    _siggen_run(pkg, 2592000, 648001, 129.6e6, 1024, 300, 240, 2, ref_check=1024)


def test_config4_style_p600(pkg):
    # BASELINE config 4 per-GPU share: 1024 x 24 kHz channels (P = 600) at 129.6 MS/s
    L, M, fs = 2592000, 648001, 129.6e6
    fa = pkg.filterapi
    N = L + M - 1
    master = fa.create_filter_input(L, M, fa.REAL)
    gen = ol.SigGen(10.00002e6 / fs, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    st = ol.Stream(L, M, ol.REAL)
    slaves, shifts = [], []
    for i in range(1024):
        s = fa.create_filter_output(master, 480, fa.COMPLEX)
        fa.set_filter(s, -10000 / 24000, 10000 / 24000, 11.0)
        slaves.append(s)
        shifts.append(ol.compute_tuning(N, fs, 0.5e6 + i * 7.8e3)[1])
    try:
        x = gen.generate(L)
        fa.write_rfilter(master, x)
        spec64 = st.push(x, f64=True)
        for s, sh in zip(slaves, shifts):
            fa.execute_filter_output(s, sh)
        for i in list(range(0, 1024, 37)) + [1023]:
            check_channel(slaves[i].output, ol.channel(spec64, ol.REAL, 600, 480, shifts[i], slaves[i].response),
                          noise_floor(spec64, slaves[i].response))
    finally:
        fa.delete_filter_input(master)


# ------------------------------------------------------------------------------
# filter.h semantics of the host mirror
# ------------------------------------------------------------------------------
def test_api_errors_and_drop_semantics(pkg):
    fa = pkg.filterapi
    assert fa.create_filter_input(25920, 6481, fa.SPECTRUM) is None        # src/filter.c:228-230
    assert fa.create_filter_input(1, 1, fa.REAL) is None                   # bins < 2 (:198-199)
    master = fa.create_filter_input(25920, 6481, fa.REAL)
    assert fa.create_filter_input(25920, 6481, fa.REAL, master=master) is master   # idempotent (:191-192)
    assert fa.create_filter_output(master, 250, fa.COMPLEX) is None        # 250*32400 % 25920 != 0 (:313-316)
    assert fa.create_filter_output(None, 240, fa.COMPLEX) is None
    s = fa.create_filter_output(master, 240, fa.COMPLEX)
    spec = fa.create_filter_output(master, 0, fa.SPECTRUM)
    assert spec is not None
    assert fa.set_filter(None, 0.1, 0.2, 11.0) == -1                        # (:969-970)
    assert fa.set_filter(s, float("nan"), 0.2, 11.0) == -1
    assert fa.execute_filter_output(None, 0) == -1
    x = np.random.default_rng(1).standard_normal(25920).astype(np.float32)
    assert fa.write_rfilter(master, x[:1000]) == 0                           # no full block yet
    assert fa.write_rfilter(master, x[1000:]) == 1
    # response not set yet: returns 0 without touching the output (:715-718)
    assert fa.execute_filter_output(s, 100) == 0 and s.output is None
    assert fa.set_filter(s, -0.4, 0.4, 11.0) == 0
    assert fa.execute_filter_output(spec, 0) == 0 and spec.next_jobnum == 1   # block clock only
    # fall ND blocks behind -> zeros + block_drops (:690-701)
    for _ in range(5):
        fa.write_rfilter(master, x)
    assert fa.execute_filter_output(s, 100) == 0
    assert s.block_drops == 1 and not s.output.any()
    assert fa.write_rfilter(master, np.zeros(25920 * 9, np.float32)) == -1   # overrun guard (:1117-1118)
    fa.delete_filter_output(s)
    fa.delete_filter_input(master)
    assert master.init is False


def test_retune_and_new_filter_between_blocks(pkg):
    # a shift change or set_filter between blocks must take effect on the next execute
    L, M = 25920, 6481
    fa = pkg.filterapi
    rng = np.random.default_rng(31)
    master = fa.create_filter_input(L, M, fa.REAL)
    st = ol.Stream(L, M, ol.REAL)
    a = fa.create_filter_output(master, 240, fa.COMPLEX)
    b = fa.create_filter_output(master, 240, fa.COMPLEX)
    fa.set_filter(a, -0.4, 0.4, 11.0); fa.set_filter(b, 0.01, 0.25, 11.0)
    try:
        for blk, (sa, sb) in enumerate([(1000, 2000), (1000, 2000), (1500, 2000), (1500, -2000)]):
            x = rng.standard_normal(L).astype(np.float32)
            fa.write_rfilter(master, x)
            spec64 = st.push(x, f64=True)
            if blk == 3:
                fa.set_filter(a, -0.1, 0.1, 5.0)
            fa.execute_filter_output(a, sa); fa.execute_filter_output(b, sb)
            check_channel(a.output, ol.channel(spec64, ol.REAL, 300, 240, sa, a.response))
            check_channel(b.output, ol.channel(spec64, ol.REAL, 300, 240, sb, b.response))
    finally:
        fa.delete_filter_input(master)


def test_notch_recurrence_is_ordered_across_streams(pkg):
    # 4 HIP streams run consecutive blocks concurrently; the notch state must still be the sequential
    # recurrence of src/filter.c:464-474 (ordered by the device-side ticket, not by stream events)
    L, M = 25920, 6481
    nblk = 37
    rng = np.random.default_rng(51)
    ring = (rng.standard_normal(8 * L) + 0.4).astype(np.float32)
    bins = [125, 4000, 16000, 0]
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    assert eng.lanes >= 1
    eng.set_notches(bins, 0.05)
    eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
    eng.run_blocks(0, nblk, graph=False)
    got = eng.spectrum((nblk - 1) % 4)
    eng.close()
    # oracle: the same cyclic stream, block by block (the first window starts with the ring's last M-1 samples)
    st = ol.Stream(L, M, ol.REAL)
    st.push(ring[7 * L:8 * L])                     # primes the history with the samples before block 0
    state = np.zeros(2 * len(bins))
    for j in range(nblk):
        want = st.push(ring[(j % 8) * L:(j % 8 + 1) * L])
        ol.notch(state, bins, 0.05, want)
    for b in bins:
        assert abs(got[b] - want[b]) <= 5e-6 * abs(want[b]) + 5e-3, b
    assert rel(got, want) <= SPEC_REL


@pytest.mark.parametrize("in_type,L,M,plan", [(ol.REAL, 25920, 6481, ""), (ol.REAL, 11520, 2881, "16x25x36"), (ol.REAL, 25920, 6481, "225x144"),
                                              (ol.COMPLEX, 11520, 2881, "16x25x36"), (ol.COMPLEX, 48000, 12001, "")])
def test_notch_folded_into_the_last_pass(pkg, monkeypatch, in_type, L, M, plan):
    """Round 4: lists of up to 8 entries are applied INSIDE fwd_rows by the thread that stores the listed bin (no notch_fix launch);
    the host names that thread by restating the pass's index arithmetic (notch_owner, chz_launch.h).  Bins served by direct and by
    conjugate-mirrored stores, the first and last bin, a bin named twice, two- and three-axis plans, REAL and COMPLEX masters:
    identical to the separate kernel (CHZ_NOTCH_FOLD=0), equal to the reference's recurrence, and really folded (no fix launches)."""
    N = L + M - 1
    B = N // 2 + 1 if in_type == ol.REAL else N
    # (a thread of the pass remembers ONE listed bin: DC together with the last bin can land on one thread -- such a list keeps the
    #  notch_fix kernel, test_notch_list_that_cannot_be_folded_keeps_the_kernel)
    bins = [125, B - 40, B // 2 + 17, 7, 125, B - 33, 1, 0]
    rng = np.random.default_rng(L + in_type)
    nblk = 22
    if in_type == ol.REAL:
        ring = (rng.standard_normal(8 * L) + 0.4).astype(np.float32)
    else:
        ring = (rng.standard_normal(8 * L) + 1j * rng.standard_normal(8 * L) + 0.4).astype(np.complex64)
    got = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("CHZ_NOTCH_FOLD", fold)
        eng = pkg.engine.Engine(L, M, in_type, plan=plan, ring_blocks=8)
        try:
            eng.set_notches(bins, 0.05)
            eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
            eng.run_blocks(0, nblk)
            got[fold] = eng.spectrum((nblk - 1) % 4)
            it = eng.run_blocks(nblk, 4, instrument=True)
            assert it.rows_n == 4 and it.fix_n == (0 if fold == "1" else 4), (fold, it.fix_n)
            eng.check()
        finally:
            eng.close()
    assert np.array_equal(got["1"], got["0"])
    st = ol.Stream(L, M, in_type)
    st.push(ring[7 * L:8 * L])
    state = np.zeros(2 * len(bins))
    for j in range(nblk):
        want = st.push(ring[(j % 8) * L:(j % 8 + 1) * L])
        ol.notch(state, bins, 0.05, want)
    for b in bins:
        assert abs(got["1"][b] - want[b]) <= 5e-6 * abs(want[b]) + 5e-3, b
    assert rel(got["1"], want) <= SPEC_REL


def test_notch_list_that_cannot_be_folded_keeps_the_kernel(pkg):
    # DC and the Nyquist bin of N = 32400 = 135 x 240 are stored by the same thread of fwd_rows (ka = 0, k1 = 0): the list stays with notch_fix
    L, M = 25920, 6481
    B = (L + M - 1) // 2 + 1
    bins = [B - 1, 0]
    rng = np.random.default_rng(9)
    ring = (rng.standard_normal(8 * L) + 0.4).astype(np.float32)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    try:
        eng.set_notches(bins, 0.05)
        eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
        eng.run_blocks(0, 9)
        got = eng.spectrum(0)
        it = eng.run_blocks(9, 4, instrument=True)
        assert it.fix_n == 4
    finally:
        eng.close()
    st = ol.Stream(L, M, ol.REAL)
    st.push(ring[7 * L:8 * L])
    state = np.zeros(2 * len(bins))
    for j in range(9):
        want = st.push(ring[(j % 8) * L:(j % 8 + 1) * L])
        ol.notch(state, bins, 0.05, want)
    for b in bins:
        assert abs(got[b] - want[b]) <= 5e-6 * abs(want[b]) + 5e-3, b


def test_run_blocks_graph_equals_eager(pkg):
    # the hipGraph replay of a ring cycle must produce exactly what eager launches produce
    L, M = 25920, 6481
    rng = np.random.default_rng(41)
    outs = []
    for graph in (False, True):
        eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        bank = eng.bank(300, 240, 16)
        bank.set_responses(0, (np.random.default_rng(3).standard_normal((16, 300)) + 0j).astype(np.complex64))
        bank.set_shifts(0, np.arange(16) * 700 + 200)
        bank.set_active(16)
        x = np.random.default_rng(4).standard_normal(8 * L).astype(np.float32)
        eng.write(x[:8 * L - (M - 1)])     # fill the ring exactly once (write position starts at M-1)
        t = eng.run_blocks(0, 16, graph=graph)
        assert t.blocks == 16 and t.total_ms > 0
        outs.append((eng.spectrum(3), bank.read(0, 16)))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_graph_replay_keeps_the_notch_ticket(pkg):
    # round 3: captured notch kernels take their ticket relative to a device word, so graph replays, eager runs and
    # graph replays again continue ONE recurrence (src/filter.c:464-474), block after block, in any mix
    L, M = 25920, 6481
    rng = np.random.default_rng(52)
    ring = (rng.standard_normal(8 * L) + 0.3).astype(np.float32)
    bins = [125, 16000, 0]                          # (the reference's list ends with its DC entry, src/filter.c:464-474)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    if eng.lanes < 2:
        eng.close(); pytest.skip("one lane: nothing to order")
    eng.set_notches(bins, 0.05)
    eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
    plan = [(True, 40), (False, 13), (True, 19), (True, 64), (False, 3)]      # (graph, blocks): replays of 32 or 8 blocks + eager tails
    job = 0
    for graph, n in plan:
        eng.run_blocks(job, n, graph=graph)
        job += n
    got = eng.spectrum((job - 1) % 4)
    eng.close()
    st = ol.Stream(L, M, ol.REAL)
    st.push(ring[7 * L:8 * L])
    state = np.zeros(2 * len(bins))
    for j in range(job):
        want = st.push(ring[(j % 8) * L:(j % 8 + 1) * L])
        ol.notch(state, bins, 0.05, want)
    for b in bins:
        assert abs(got[b] - want[b]) <= 5e-6 * abs(want[b]) + 5e-3, b
    assert rel(got, want) <= SPEC_REL


# ------------------------------------------------------------------------------
# the tail of downconvert() fused into the channel kernel
# ------------------------------------------------------------------------------
def _ulp_close(got, want):
    ulp = np.spacing(np.maximum(np.abs(want.real), np.abs(want.imag)).astype(np.float32))
    d = got - want
    worst = float((np.maximum(np.abs(d.real), np.abs(d.imag)) / ulp).max())
    return worst, float((got == want).mean())


@pytest.mark.parametrize("P,olen,nch", [(300, 240, 24), (1000, 800, 12)])      # a menu size; a size served by chan_any
def test_tuned_bank_follows_downconvert(pkg, P, olen, nch):
    # per-channel fine oscillator, block phase correction, shift-change kick and bb_power
    # (src/radio.c:1476-1520) against the restated tail AND the reference's own osc.c, block by block
    L, M, fs_in = 25920, 6481, 1.296e6
    fs_out = olen / 0.02
    N = L + M - 1
    rng = np.random.default_rng(71)
    f_hz = 50e3 + rng.uniform(0, 500e3, nch)
    f_hz[0] = 40.0 * 2500                      # exactly on a bin whose shift is a multiple of V: no rotation at all
    f_hz[1] = 40.0 * 2501                      # on a bin, shift % V = 1: block phase correction only
    sweep = np.zeros(nch); sweep[5] = 35.0; sweep[6] = -120.0     # Hz/s
    resp = np.stack([ol.set_filter(P, olen, N, True, -0.35, 0.35, 9.0)] * nch).astype(np.complex64)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    tuned = eng.bank(P, olen, nch); plain = eng.bank(P, olen, nch)
    for b in (tuned, plain):
        b.set_responses(0, resp); b.set_active(nch)
    st = ol.Stream(L, M, ol.REAL)
    dco = [ol.Downconv(L, M, fs_out, "oracle") for _ in range(nch)]
    dcr = [ol.Downconv(L, M, fs_out, "ref") for _ in range(nch)] if ol.have_ref() else None
    dce = [ol.Downconv(L, M, fs_out, "oracle") for _ in range(nch)]       # end-to-end chain on the float64 oracle
    job0 = 0xFFFFFFF8                          # ring-aligned (multiple of 8); the block counter wraps during the run
    shifts = np.zeros(nch, np.int32); rems = np.zeros(nch)
    try:
        for blk in range(11):
            job = (job0 + blk) & 0xFFFFFFFF
            if blk in (0, 3, 5, 9):               # retune: everything at 0, a few channels later
                sel = range(nch) if blk == 0 else ([2, 3, 6] if blk == 3 else ([3, 9] if blk == 5 else [1, 2]))
                for ch in sel:
                    if blk:
                        f_hz[ch] += rng.uniform(-3e3, 3e3)
                    _, sh, rem = ol.compute_tuning(N, fs_in, f_hz[ch])
                    shifts[ch], rems[ch] = sh, rem
                sel = np.array(list(sel))
                lo, hi = int(sel.min()), int(sel.max()) + 1
                tuned.set_tuning(job, lo, shifts[lo:hi], -rems[lo:hi] / fs_out, sweep[lo:hi] / fs_out ** 2)
                plain.set_shifts(0, shifts)
            x = rng.standard_normal(L).astype(np.float32)
            eng.write(x)
            spec64 = st.push(x, f64=True)
            eng.step(job)
            got = tuned.read_slot(job % 4); raw = plain.read_slot(job % 4)
            pw = tuned.read_power(job % 4)
            for ch in range(nch):
                # (1) the rotation itself, applied to the GPU's own un-rotated samples: float-exact
                want, wpw = dco[ch].block(raw[ch], shifts[ch], rems[ch], sweep[ch])
                worst, same = _ulp_close(got[ch], want)
                # never more than an ulp of a float; bit-identical for nearly all samples (a swept oscillator stepped 800 times a block
                # by the reference collects more last-bit differences against the closed form than a fixed one)
                assert worst <= 1.0 and same >= (0.85 if sweep[ch] else 0.97), (blk, ch, worst, same)
                assert abs(pw[ch] - wpw) <= 1e-6 * wpw
                if dcr is not None:
                    want_r, rpw = dcr[ch].block(raw[ch], shifts[ch], rems[ch], sweep[ch])
                    assert _ulp_close(want_r, want)[0] <= 1.0 and abs(rpw - wpw) <= 1e-9 * wpw
                # (2) end to end against the float64 oracle chain
                ideal = ol.channel(spec64, ol.REAL, P, olen, int(shifts[ch]), resp[ch])
                ideal, ipw = dce[ch].block(ideal, shifts[ch], rems[ch], sweep[ch])
                check_channel(got[ch], ideal, noise_floor(spec64, resp[ch]))
                assert abs(pw[ch] - ipw) <= 1e-4 * ipw
            if blk == 1:
                # shift % V == 0 and zero remainder: a constant phasor (the start-up kick of :1494), no rotation
                ph = got[0] * np.conj(raw[0])
                assert np.allclose(ph / np.abs(ph), (ph / np.abs(ph))[np.argmax(np.abs(ph))], atol=1e-3)
    finally:
        eng.close()


def test_tuned_bank_pipelined_over_streams(pkg):
    # the rotation is a closed form in the block number: many blocks in flight on 4 streams must give
    # exactly what block-at-a-time stepping gives
    L, M, fs_out, P, olen = 25920, 6481, 12000.0, 300, 240
    nch, nblk = 64, 23
    rng = np.random.default_rng(72)
    shifts = rng.integers(300, 16000, nch).astype(np.int32)
    freq = rng.uniform(-20, 20, nch) / fs_out
    rate = np.where(np.arange(nch) % 7 == 0, 50.0 / fs_out ** 2, 0.0)
    resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64) / P
    ring = rng.standard_normal(8 * L).astype(np.float32)
    outs = []
    for pipelined in (False, True):
        eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        b = eng.bank(P, olen, nch); b.set_responses(0, resp); b.set_active(nch)
        b.set_tuning(5, 0, shifts, freq, rate)
        eng.write(ring[:8 * L - (M - 1)])
        if pipelined:
            eng.run_blocks(5, nblk, graph=False)
            with pytest.raises(pkg.engine.ChzError):
                eng.run_blocks(5, 8, graph=True)        # graph replay would freeze the block number
        else:
            for j in range(nblk):
                eng.step(5 + j)
        eng.sync()
        outs.append([(b.read_slot(s), b.read_power(s)) for s in range(4)])
        eng.close()
    for (a, pa), (c, pc) in zip(*outs):
        assert np.array_equal(a, c) and np.array_equal(pa, pc)
        assert np.abs(a).max() > 0


# ------------------------------------------------------------------------------
# raw int16 A/D samples converted where the first pass loads them
# ------------------------------------------------------------------------------
# (25920, 6483): N = 32402 = 2 x 17 x 953 runs as chirp-z -- round 4: raw A/D samples are converted where blue_pre loads them
@pytest.mark.parametrize("L,M,randomize", [(25920, 6481, False), (1296000, 324001, True), (2592000, 648001, False), (25920, 6483, True)])
def test_int16_input_matches_convert_then_float_path(pkg, L, M, randomize):
    rng = np.random.default_rng(L)
    scale = np.float32(ol.scale_ad(True, 16, 0.0, 0.0, 0.0))
    e16 = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    ef = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    st = ol.Stream(L, M, ol.REAL)
    try:
        for blk in range(3):
            x16 = np.clip(rng.normal(0, 9000, L), -32768, 32767).astype(np.int16)
            x16[rng.integers(0, L, 17)] = 32767
            xf, energy, clips = ol.convert_i16(x16, scale, randomize)
            # written in the odd-sized chunks a USB callback delivers
            for a in range(0, L, 16384 * 3 + 2):
                e16.write_i16(x16[a:a + 16384 * 3 + 2], scale, randomize)
            ef.write(xf)
            e16.forward(blk); ef.forward(blk)
            got, ref = e16.spectrum(blk % 4), ef.spectrum(blk % 4)
            assert np.array_equal(got, ref)               # same arithmetic, conversion is exact
            assert e16.input_stats(blk % 4) == (energy, clips)
            want = st.push(xf, f64=True)
            assert rel(got, want) <= SPEC_REL
        with pytest.raises(pkg.engine.ChzError):
            e16.write(np.zeros(16, np.float32))           # one engine, one sample format
    finally:
        e16.close(); ef.close()


# ------------------------------------------------------------------------------
# estimate_noise() on the device
# ------------------------------------------------------------------------------
@pytest.mark.parametrize("L,M,P,olen,nch", [(25920, 6481, 300, 240, 40), (2592000, 648001, 300, 240, 1024), (1296000, 324001, 600, 480, 128)])
def test_noise_estimate_matches_radio_c(pkg, L, M, P, olen, nch):
    N = L + M - 1
    B = N // 2 + 1
    fs = 50.0 * L
    rng = np.random.default_rng(P + nch)
    x = rng.standard_normal(L).astype(np.float32)
    t = np.arange(L)
    for f in rng.uniform(0.02, 0.45, 6):
        x += (30 * np.cos(2 * np.pi * f * t)).astype(np.float32)          # carriers inside some of the windows
    shifts = rng.integers(-(B - 2), B - 1, nch).astype(np.int32)
    shifts[:6] = [0, 3, B - 1, -(B - 1), 500, -499]
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    try:
        b = eng.bank(P, olen, nch)
        b.set_responses(0, np.ones((nch, P), np.complex64) / P); b.set_shifts(0, shifts); b.set_active(nch)
        with pytest.raises(pkg.engine.ChzError):
            b.read_noise(0)
        b.enable_noise(fs)
        st = ol.Stream(L, M, ol.REAL)
        for job in range(2):
            eng.write(x); eng.step(job)
            spec64 = st.push(x, f64=True)
        got = b.read_noise(1)
        spec = eng.spectrum(1)
        want = np.array([ol.estimate_noise(spec, ol.REAL, P, int(s), fs) for s in shifts])
        assert np.all(want > 0) and np.allclose(got, want, rtol=1e-12, atol=0)      # same spectrum: same estimate
        ideal = np.array([ol.estimate_noise(spec64.astype(np.complex64), ol.REAL, P, int(s), fs) for s in shifts])
        assert np.allclose(got, ideal, rtol=2e-4)                                    # float32 transform noise only
        b.enable_noise(0.0)
        eng.step(2)
        assert np.array_equal(b.read_noise(1), got)                                  # switched off: nothing new written
    finally:
        eng.close()


def test_noise_estimate_with_a_stale_binade_guess(pkg, monkeypatch):
    # round 4: noise_est starts from the binade the channel's quantile fell into last time and verifies it with the counts it needs
    # anyway.  A guess is never a result: a level that jumps by 40 dB up, 100 dB down and back between blocks (every guess wrong), a
    # level that stays (every guess right) and the guess switched off give radio.c's estimate on the device's own spectrum, bit for bit
    # the same in all three
    L, M, P, olen, nch = 25920, 6481, 300, 240, 48
    B = (L + M - 1) // 2 + 1
    fs = 50.0 * L
    rng = np.random.default_rng(5)
    x = rng.standard_normal(L).astype(np.float32)
    shifts = np.array([0, 3, B - 1, -(B - 1), 500, -499] + [900 + 263 * i for i in range(nch - 6)], np.int32)
    gains = [1.0, 1.0, 100.0, 1e-3, 1e-3, 1.0, 1.0 + 2 ** -20, 7.0]
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CHZ_NOISE_HINT", mode)
        eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        try:
            b = eng.bank(P, olen, nch)
            b.set_responses(0, np.ones((nch, P), np.complex64) / P); b.set_shifts(0, shifts); b.set_active(nch)
            b.enable_noise(fs)
            rows = []
            for job, g in enumerate(gains):
                eng.write((x * np.float32(g)).astype(np.float32)); eng.step(job)
                n0 = b.read_noise(job % 4).copy()
                if mode == "1":
                    spec = eng.spectrum(job % 4)
                    want = np.array([ol.estimate_noise(spec, ol.REAL, P, int(s), fs) for s in shifts])
                    assert np.all(want > 0) and np.allclose(n0, want, rtol=1e-12, atol=0), job
                rows.append(n0)
            got[mode] = np.stack(rows)
        finally:
            eng.close()
    np.testing.assert_array_equal(got["1"], got["0"])


@pytest.mark.parametrize("P,olen", [(300, 240), (1200, 960)])      # windows of 1000 bins (16 values per lane) and of 1200 (32 per lane)
@pytest.mark.parametrize("master", ["real", "complex"])
def test_noise_windows_from_the_energy_image(pkg, monkeypatch, master, P, olen):
    # large banks take |X|^2 once per bin (spec_energy) and their noise windows read that image instead of the spectrum: the same
    # estimate bit for bit (forced on here for a small bank) -- windows clamped at both ends of a REAL master, an inverted channel,
    # windows that wrap or stop at the seam of a COMPLEX master -- and radio.c's arithmetic on the device's own spectrum
    L, M, nch = 25920, 6481, 40
    in_type = ol.REAL if master == "real" else ol.COMPLEX
    N = L + M - 1
    B = N // 2 + 1 if master == "real" else N
    fs = 50.0 * L
    rng = np.random.default_rng(21)
    x = rng.standard_normal(L if master == "real" else 2 * L).astype(np.float32)
    if master == "real":
        shifts = np.array([3, 140, 499, 501, B - 100, B - 20, -700] + [900 + 371 * i for i in range(nch - 7)], np.int32)
    else:
        shifts = np.array([0, 3, -3, 499, -501, N // 2 - 100, N // 2 + 100 - N, -(N // 2) + 5] + [(-1) ** i * (700 + 371 * i) for i in range(nch - 8)], np.int32)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CHZ_NOISE_ENERGY", mode)
        eng = pkg.engine.Engine(L, M, in_type, ring_blocks=8)
        try:
            b = eng.bank(P, olen, nch)
            b.set_responses(0, np.ones((nch, P), np.complex64) / P); b.set_shifts(0, shifts); b.set_active(nch)
            b.enable_noise(fs)
            for job in range(2):
                eng.write(x); eng.step(job)
            got[mode] = b.read_noise(1).copy()
            spec = eng.spectrum(1)
        finally:
            eng.close()
    np.testing.assert_array_equal(got["0"], got["1"])
    want = np.array([ol.estimate_noise(spec, in_type, P, int(s), fs) for s in shifts])
    assert np.allclose(got["1"], want, rtol=1e-12, atol=0) and np.any(want > 0)


def test_staged_output_path_is_bit_identical(pkg, monkeypatch):
    # large launches send the output rows through LDS as full-line stores (chz_kernels.h: p.stage); forced on here for a
    # small bank, incl. a channel count that is not a multiple of the channels per wavefront and range launches at odd offsets
    L, M = 25920, 6481
    rng = np.random.default_rng(91)
    x = rng.standard_normal(L).astype(np.float32)
    outs = {}
    for stage in ("0", "1"):
        monkeypatch.setenv("CHZ_CHAN_STAGE", stage)
        eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        res = []
        for P, olen, nch in ((300, 240, 1024), (300, 240, 7), (600, 480, 65), (200, 160, 10), (1200, 960, 5)):
            b = eng.bank(P, olen, nch)
            r2 = np.random.default_rng(P + nch)
            b.set_responses(0, (r2.standard_normal((nch, P)) + 1j * r2.standard_normal((nch, P))).astype(np.complex64) / P)
            b.set_shifts(0, r2.integers(-16000, 16000, nch)); b.set_active(nch)
        eng.write(x); eng.step(0)
        for b in eng.banks:
            res.append(b.read_slot(0))
        b = eng.banks[0]
        b.execute_range(0, 5, 11) if hasattr(b, "execute_range") else None
        eng.sync()
        res.append(b.read_slot(0))
        outs[stage] = res
        eng.close()
    for a, c in zip(outs["0"], outs["1"]):
        assert np.array_equal(a, c) and np.abs(a).max() > 0


# ------------------------------------------------------------------------------
# REAL-output slaves (src/filter.c:372-395, gather :794-809, c2r :914)
# ------------------------------------------------------------------------------
@pytest.mark.parametrize("in_type,L,M,P,olen", [(2, 7680, 1921, 1200, 960), (2, 25920, 6481, 300, 240), (1, 11520, 2881, 600, 480),
                                                (2, 2592000, 648001, 1200, 960),
                                                (2, 7680, 7681, 1920, 960),        # wfm's composite master as src/wfm.c:50-76 builds it (M = L + 1)
                                                (2, 25920, 6481, 1000, 800), (1, 11520, 2881, 2700, 2160)])   # outside the menu: chan_any
def test_real_output_slaves(pkg, in_type, L, M, P, olen):
    fa = pkg.filterapi
    rng = np.random.default_rng(P + L)
    master = fa.create_filter_input(L, M, in_type)
    N = L + M - 1
    B = N // 2 + 1 if in_type == ol.REAL else N
    st = ol.Stream(L, M, in_type)
    plan = [(0, 0.0, 0.31), (17, 0.02, 0.4), (-5, 0.1, 0.1), (B - P // 4, 0.0, 0.45), (B // 3, 0.2, 0.05), (-(B // 3), 0.0, 0.5)]
    slaves = []
    try:
        for sh, lo, hi in plan:
            s = fa.create_filter_output(master, olen, fa.REAL)
            assert s is not None and s.bins == P // 2 + 1 and s.points == P
            assert fa.set_filter(s, lo, hi, 5.0) == 0
            ref = ol.set_filter(P, olen, N, in_type == ol.REAL, lo, hi, 5.0, out_type=ol.REAL)
            assert np.abs(s.response - ref).max() <= 3e-7 * np.abs(ref).max()
            slaves.append(s)
        cplx = fa.create_filter_output(master, olen, fa.COMPLEX)        # a COMPLEX slave of the same size shares the master
        fa.set_filter(cplx, -0.3, 0.3, 5.0)
        for blk in range(3):
            if in_type == ol.REAL:
                x = rng.standard_normal(L).astype(np.float32); fa.write_rfilter(master, x)
            else:
                x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64); fa.write_cfilter(master, x)
            spec = st.push(x)
            for s, (sh, lo, hi) in zip(slaves, plan):
                assert fa.execute_filter_output(s, sh) == 0
                assert s.output.dtype == np.float32 and s.output.shape == (olen,)
                want = ol.channel(spec, in_type, P, olen, sh, s.response, out_type=ol.REAL)
                nrm = float(np.linalg.norm(want))
                if nrm == 0:
                    assert not s.output.any()
                else:
                    assert np.linalg.norm(s.output - want) <= 1e-5 * nrm + 2e-8 * float(np.abs(spec).max()) * float(np.linalg.norm(s.response)) * np.sqrt(olen)
            assert fa.execute_filter_output(cplx, 40) == 0
            check_channel(cplx.output, ol.channel(spec.astype(np.complex128), in_type, P, olen, 40, cplx.response))
    finally:
        fa.delete_filter_input(master)


def test_isb_slaves(pkg):
    # slave->isb set by the caller after create (src/radio.c:1586): LSB/USB unpacked to I/Q (src/filter.c:895-909)
    L, M, P, olen = 25920, 6481, 300, 240
    fa = pkg.filterapi
    rng = np.random.default_rng(77)
    master = fa.create_filter_input(L, M, fa.REAL)
    st = ol.Stream(L, M, ol.REAL)
    slaves = [fa.create_filter_output(master, olen, fa.COMPLEX) for _ in range(5)]
    shifts = [1200, -3000, 40, 9000, 1200]
    try:
        for s in slaves:
            fa.set_filter(s, -0.45, 0.45, 9.0)
        for blk in range(4):
            x = rng.standard_normal(L).astype(np.float32)
            fa.write_rfilter(master, x)
            spec64 = st.push(x, f64=True)
            spec32 = spec64.astype(np.complex64)
            for i, (s, sh) in enumerate(zip(slaves, shifts)):
                s.isb = (i != 4) and (blk != 2 or i != 0)          # flags flip between blocks; one channel stays plain
                assert fa.execute_filter_output(s, sh) == 0
                want = ol.channel(spec32, ol.REAL, P, olen, sh, s.response, isb=bool(s.isb))
                assert np.linalg.norm(s.output - want) <= 1e-5 * np.linalg.norm(want) + noise_floor(spec64, s.response) * np.sqrt(olen) * 2, (blk, i)
            assert not np.array_equal(slaves[1].output, slaves[4].output)
    finally:
        fa.delete_filter_input(master)


def test_beam_slaves(pkg):
    # slave->beam with set_filter_weights (src/filter.c:756-775, :922-929; radio.c:938-940) on a COMPLEX master
    L, M, P, olen = 11520, 2881, 300, 240
    fa = pkg.filterapi
    rng = np.random.default_rng(78)
    master = fa.create_filter_input(L, M, fa.COMPLEX)
    st = ol.Stream(L, M, ol.COMPLEX)
    N = L + M - 1
    weights = [(1.0, 0.0), (0.0, 1.0), (0.7 + 0.2j, -0.3 + 0.6j), None]
    shifts = [0, 2000, -(N // 2) + 60, 350]
    slaves = [fa.create_filter_output(master, olen, fa.COMPLEX) for _ in weights]
    try:
        for s, w in zip(slaves, weights):
            fa.set_filter(s, -0.4, 0.4, 9.0)
            if w is not None:
                s.beam = True
                assert fa.set_filter_weights(s, *w) == 0
        for blk in range(3):
            x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
            fa.write_cfilter(master, x)
            spec = st.push(x)
            if blk == 2:
                fa.set_filter_weights(slaves[0], 0.25, -0.5j)          # weights change mid-stream
            for s, w, sh in zip(slaves, weights, shifts):
                assert fa.execute_filter_output(s, sh) == 0
                if w is None:
                    want = ol.channel(spec, ol.COMPLEX, P, olen, sh, s.response)
                else:
                    want = ol.channel_beam(spec, P, olen, sh, s.response, s.alpha, s.beta)
                assert np.linalg.norm(s.output - want) <= 1e-5 * np.linalg.norm(want), (blk, sh)
    finally:
        fa.delete_filter_input(master)
