"""Golden vectors generated from the reference itself (tests/golden/make_golden.py).

CPU: the oracle restatement reproduces them.  GPU: the HIP path reproduces them.
"""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_pkg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz")) if os.path.basename(f).split("_")[0] in ("real", "complex"))
DC_FILE = os.path.join(GOLDEN, "downconvert_tail.npz")
NEXT_FILE = os.path.join(GOLDEN, "next_rows.npz")


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_golden_files_present():
    assert len(FILES) >= 3


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(oracle_built, path):
    g = np.load(path)
    L, M, in_type, olen, P = int(g["L"]), int(g["M"]), int(g["in_type"]), int(g["olen"]), int(g["P"])
    isreal = in_type == ol.REAL
    # input: the restated sig_gen law reproduces the reference's stream
    gen = ol.SigGen(float(g["carrier_hz"]) / float(g["fs"]), 10 ** (-20 / 20), 10 ** (-40 / 20), ol.scale_ad(isreal, 1), isreal, seed=1)
    x = gen.generate(g["x"].shape[0])
    assert np.abs(x - g["x"]).max() <= 1e-7
    st = ol.Stream(L, M, in_type)
    state = np.zeros(2 * len(g["notch_bins"]))
    N = L + M - 1
    for b in range(g["spectrum"].shape[0]):
        spec = st.push(g["x"][b * L:(b + 1) * L])
        ol.notch(state, g["notch_bins"], float(g["notch_alpha"]), spec)
        assert rel(spec, g["spectrum"][b]) <= 1e-7
        for i, (shift, low, high, beta) in enumerate(g["chans"]):
            resp = ol.set_filter(P, olen, N, isreal, low, high, beta)
            assert np.abs(resp - g["response"][i]).max() <= 2e-7 * np.abs(g["response"][i]).max()
            out = ol.channel(g["spectrum"][b], in_type, P, olen, int(shift), g["response"][i])
            assert np.abs(out - g["output"][b, i]).max() <= 2e-6 * max(np.abs(g["output"][b, i]).max(), 1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_reproduces_golden(path):
    pkg = load_pkg()
    fa = pkg.filterapi
    g = np.load(path)
    L, M, in_type, olen, P = int(g["L"]), int(g["M"]), int(g["in_type"]), int(g["olen"]), int(g["P"])
    master = fa.create_filter_input(L, M, in_type)
    fa.set_notches(master, list(g["notch_bins"]), float(g["notch_alpha"]))
    slaves = []
    for (shift, low, high, beta) in g["chans"]:
        s = fa.create_filter_output(master, olen, fa.COMPLEX)
        assert s.points == P and fa.set_filter(s, low, high, beta) == 0
        slaves.append(s)
    write = fa.write_rfilter if in_type == fa.REAL else fa.write_cfilter
    try:
        for b in range(g["spectrum"].shape[0]):
            assert write(master, g["x"][b * L:(b + 1) * L]) == 1
            spec = master._engine.spectrum(b % 4)
            assert rel(spec, g["spectrum"][b]) <= 1e-6
            peak = float(np.abs(g["spectrum"][b]).max())
            for i, (s, ch) in enumerate(zip(slaves, g["chans"])):
                assert np.abs(s.response - g["response"][i]).max() <= 3e-7 * np.abs(g["response"][i]).max()
                assert fa.execute_filter_output(s, int(ch[0])) == 0
                want = g["output"][b, i]
                err = float(np.sqrt(np.mean(np.abs(s.output - want) ** 2)))
                rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
                assert err <= 1e-5 * rms + 2e-8 * peak * float(np.linalg.norm(g["response"][i])), (b, i)
    finally:
        fa.delete_filter_input(master)


def _ulps(got, want):
    ulp = np.spacing(np.maximum(np.abs(want.real), np.abs(want.imag)).astype(np.float32))
    d = got - want
    return float((np.maximum(np.abs(d.real), np.abs(d.imag)) / ulp).max()), float((got == want).mean())


def test_oracle_reproduces_downconvert_golden(oracle_built):
    """The restated downconvert() tail against vectors produced with the reference's own osc.c / cispi."""
    g = np.load(DC_FILE)
    d = ol.Downconv(int(g["L"]), int(g["M"]), float(g["fs_out"]), "oracle")
    keep = {int(k): i for i, k in enumerate(g["keep"])}
    for b, (sh, rem, dr) in enumerate(g["history"]):
        y, p = d.block(g["x"][b], int(sh), rem, dr)
        if b in keep:
            worst, same = _ulps(y, g["y"][keep[b]])
            assert worst <= 1.0 and same >= 0.99
            assert abs(p - g["power"][keep[b]]) <= 1e-9 * g["power"][keep[b]]


@pytest.mark.gpu
def test_hip_reproduces_downconvert_golden():
    """chan_ifft's epilogue on the same history: the channel samples are produced by the kernel itself (identity
    response on a crafted spectrum is not needed -- the un-rotated twin bank supplies them), the rotation must land
    within one float ulp of what the reference's oscillator gives for those samples."""
    pkg = load_pkg()
    g = np.load(DC_FILE)
    L, M, fs_out, olen = int(g["L"]), int(g["M"]), float(g["fs_out"]), int(g["olen"])
    P = olen * (L + M - 1) // L
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    have_ref = ol.have_ref()
    try:
        tuned = eng.bank(P, olen, 1); plain = eng.bank(P, olen, 1)
        resp = np.ones((1, P), np.complex64) / P
        for b in (tuned, plain):
            b.set_responses(0, resp); b.set_active(1)
        d_or = ol.Downconv(L, M, fs_out, "oracle")
        d_ref = ol.Downconv(L, M, fs_out, "ref") if have_ref else None
        rng = np.random.default_rng(7)
        last = None
        for blk, (sh, rem, dr) in enumerate(g["history"]):
            cur = (int(sh), float(rem), float(dr))
            if last is None or cur[0] != last[0] or cur[1] != last[1]:      # set_osc runs only then (src/radio.c:1479)
                tuned.set_tuning(blk, 0, [cur[0]], [-cur[1] / fs_out], [cur[2] / fs_out ** 2])
                plain.set_shifts(0, [cur[0]])
                last = cur
            eng.write(rng.standard_normal(L).astype(np.float32))
            eng.step(blk)
            got = tuned.read_slot(blk % 4)[0]; raw = plain.read_slot(blk % 4)[0]
            want, _ = d_or.block(raw, *last)
            worst, same = _ulps(got, want)
            # During a sweep the reference multiplies phasor_step by phasor_step_step every sample (src/osc.c:64-68): its
            # rounding errors add up quadratically in the step count (~5e-9 rad after 10^4 samples), while the kernel
            # evaluates the phase in closed form.  Both stay far inside one float ulp of the product; the share of
            # samples whose last bit differs grows slowly along the sweep.
            assert worst <= 1.0 and same >= (0.97 if last[2] == 0.0 else 0.85), (blk, worst, same)
            if d_ref is not None:
                wr, _ = d_ref.block(raw, *last)
                assert _ulps(wr, want)[0] <= 1.0
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------------
# vectors produced by the reference's own radio.c / rx888.c
# ------------------------------------------------------------------------------------------------
def test_oracle_reproduces_noise_and_conversion_golden(oracle_built):
    g = np.load(NEXT_FILE)
    for i, sb in enumerate(g["s_bins"]):
        got = np.array([ol.estimate_noise(g["spectrum"], ol.REAL, int(sb), int(sh), float(g["samprate"])) for sh in g["shifts"]])
        assert np.allclose(got, g["n0"][i], rtol=1e-12, atol=0)
    for rnd in (0, 1):
        out, en, clips = ol.convert_i16(g["x16"], float(g["scale"]), bool(rnd))
        assert np.array_equal(out.view(np.uint32), g["conv%d" % rnd].view(np.uint32))
        assert en == int(g["energy%d" % rnd]) and clips == int(g["clips%d" % rnd])


@pytest.mark.gpu
def test_hip_reproduces_noise_and_conversion_golden():
    """noise_est on the golden spectrum (written straight into a spectrum slot) against the reference's estimate_noise();
    the int16 input path against the reference's convert_avx2(): the spectrum of the raw samples must be bit-identical
    to the spectrum of the reference-converted floats, and the energy / clip statistics must match."""
    import ctypes as C
    pkg = load_pkg()
    g = np.load(NEXT_FILE)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L, M = 25920, 6481                                  # REAL master with exactly the golden spectrum's 16201 bins
    fs = float(g["samprate"])
    for sb in g["s_bins"]:
        P = int(sb); olen = P * 4 // 5
        eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        try:
            assert eng.bins == g["spectrum"].shape[0]
            na, pitch, off = eng.spec_layout
            k = np.arange(eng.bins)
            dev = np.zeros(eng.spec_elems, np.complex64)
            dev[(k // na) * pitch + off + k % na] = g["spectrum"]
            assert hip.hipMemcpy(eng.spectrum_ptr(1), dev.ctypes.data, dev.nbytes, 1) == 0
            nch = len(g["shifts"])
            b = eng.bank(P, olen, nch)
            b.set_responses(0, np.ones((nch, P), np.complex64) / P); b.set_shifts(0, g["shifts"]); b.set_active(nch)
            b.enable_noise(fs)
            b.execute(1)
            got = b.read_noise(1)
            want = g["n0"][list(g["s_bins"]).index(sb)]
            assert np.allclose(got, want, rtol=1e-12, atol=0)
        finally:
            eng.close()
    # rank 3: raw int16 in, converted on load
    n = g["x16"].shape[0]
    for rnd in (0, 1):
        e16 = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        ef = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        try:
            x16 = np.tile(g["x16"], L // n + 1)[:L]
            xf = np.tile(g["conv%d" % rnd], L // n + 1)[:L]
            e16.write_i16(x16, float(g["scale"]), bool(rnd)); ef.write(xf)
            e16.forward(0); ef.forward(0)
            assert np.array_equal(e16.spectrum(0).view(np.uint32), ef.spectrum(0).view(np.uint32))
            en, clips = e16.input_stats(0)
            reps, rest = divmod(L, n)
            _, en_r, cl_r = ol.convert_i16(g["x16"][:rest], float(g["scale"]), bool(rnd)) if rest else (None, 0, 0)
            assert en == reps * int(g["energy%d" % rnd]) + en_r and clips == reps * int(g["clips%d" % rnd]) + cl_r
        finally:
            e16.close(); ef.close()


# ------------------------------------------------------------------------------------------------
# frames produced by the reference's own demod_linear() / demod_fm() (linear.c, fm.c, osc.c, iir.c, misc.c)
# ------------------------------------------------------------------------------------------------
DEMOD_FILE = os.path.join(GOLDEN, "demod_rows.npz")
LIN_KEYS = ["lin_usb", "lin_am", "lin_pll", "lin_pllsq", "lin_pllhold", "lin_pllsqhold"]


def _strict_blocks(key):
    """how long a coherent case is compared strictly: while its carrier is there (round 4: the *hold cases keep it for all 150 blocks)"""
    return 10 ** 9 if key.endswith("hold") else 50
FM_KEYS = ["fm_plain", "fm_thr", "fm_pll", "fm_tone"]


def _golden_params(g, key):
    kw = eval(str(g[key + "_kw"]), {"__builtins__": {}}, {"dict": dict})       # a literal dict written by make_golden.py
    return ol.lin_params(**kw) if key.startswith("lin") else ol.fm_params(**kw)


def _pcm_close(p, got, want, nsamp, tol_f):
    if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
        dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
        a, w = got[:2 * nsamp].view(dt).astype(np.int32), want[:2 * nsamp].view(dt).astype(np.int32)
        return np.abs(a - w).max() <= 1 and np.mean(a != w) == 0
    dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
    a, w = got[:4 * nsamp].view(dt).astype(np.float64), want[:4 * nsamp].view(dt).astype(np.float64)
    return np.abs(a - w).max() <= tol_f * max(np.abs(w).max(), 1e-30)


@pytest.mark.parametrize("key", LIN_KEYS + FM_KEYS)
def test_oracle_reproduces_demodulator_golden(oracle_built, key):
    g = np.load(DEMOD_FILE)
    p = _golden_params(g, key)
    bb, power, est = g[key + "_bb"], g[key + "_bbpower"], g[key + "_est"]
    d = ol.LinDemod(p) if key.startswith("lin") else ol.FmDemod(p)
    ndata = 0
    for b in range(bb.shape[0]):
        pcm, st = d.block(bb[b], power[b], est[b], 0.02)
        assert st.frame == g[key + "_frame"][b] and st.mute == g[key + "_mute"][b], (key, b)
        if key.startswith("lin"):
            assert st.gain == pytest.approx(g[key + "_gain"][b], rel=1e-7)
            assert st.output_power == pytest.approx(g[key + "_opower"][b], rel=1e-6, abs=1e-300)
            if p.pll_enable:
                snr, lock, cph, rot, foff = g[key + "_pll"][b]
                assert st.pll_lock == int(lock) and st.pll_rotations == int(rot)
                assert st.pll_snr == pytest.approx(snr, rel=1e-6, abs=1e-9) and st.foffset == pytest.approx(foff, rel=1e-6, abs=1e-6)
        else:
            assert st.snr == pytest.approx(g[key + "_snr"][b], rel=1e-6, abs=1e-12)
            assert st.tone_deviation == pytest.approx(g[key + "_tonedev"][b], rel=1e-5, abs=1e-6)
        if st.frame == ol.FRAME_DATA:
            ndata += 1
            assert _pcm_close(p, pcm, g[key + "_pcm"][b], bb.shape[1] * p.channels, 4e-6), (key, b)
    assert ndata >= 5


@pytest.mark.gpu
def test_hip_reproduces_demodulator_golden():
    """The golden baseband written straight into a bank's output image (with its bb_power and noise estimates), then ONLY the
    demodulator stage run on the device, block after block: frames, statistics and PCM against what the reference's own
    demod_linear() / demod_fm() produced from the same blocks."""
    import ctypes as C
    pkg = load_pkg()
    g = np.load(DEMOD_FILE)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib = pkg.engine.lib()
    for keys, P, olen in ((LIN_KEYS, 300, 240), (FM_KEYS, 600, 480)):
        eng = pkg.engine.Engine(25920, 6481, ol.REAL, ring_blocks=8)
        try:
            nch = len(keys)
            params = [_golden_params(g, k) for k in keys]
            bank = eng.bank(P, olen, nch)
            bank.set_responses(0, np.ones((nch, P), np.complex64) / P)
            bank.set_tuning(0, 0, np.full(nch, 2500, np.int32), np.zeros(nch))
            bank.set_active(nch); bank.enable_noise(1.296e6); bank.set_pcm_stride(8 * olen)
            bank.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in params], 0.02)
            nblk = max(g[k + "_bb"].shape[0] for k in keys)
            ndata = np.zeros(nch, int)
            for b in range(nblk):
                slot = b % 4
                x = np.zeros((nch, olen), np.complex64); pw = np.zeros(nch); ne = np.zeros(nch)
                for i, k in enumerate(keys):
                    bb = g[k + "_bb"]
                    if b < bb.shape[0]:
                        x[i] = bb[b]; pw[i] = g[k + "_bbpower"][b]; ne[i] = g[k + "_est"][b]
                    else:
                        pw[i] = 1e-30; ne[i] = g[k + "_est"][-1]
                bank.inject(slot, x, pw, ne)
                bank.demod_only(b)
                pcm, status = bank.read_pcm(slot)
                for i, k in enumerate(keys):
                    if b >= g[k + "_bb"].shape[0]:
                        continue
                    p, got = params[i], status[i]
                    assert got.frame == g[k + "_frame"][b] and got.mute == g[k + "_mute"][b], (k, b)
                    if k.startswith("lin"):
                        assert got.gain == pytest.approx(g[k + "_gain"][b], rel=1e-6)
                        assert got.output_power == pytest.approx(g[k + "_opower"][b], rel=1e-5, abs=1e-300)
                        if p.pll_enable and b < _strict_blocks(k):
                            snr, lock, cph, rot, foff = g[k + "_pll"][b]
                            assert got.pll_lock == int(lock) and got.pll_rotations == int(rot), (k, b)
                            assert got.pll_snr == pytest.approx(snr, rel=1e-5, abs=1e-9) and got.foffset == pytest.approx(foff, rel=1e-6, abs=1e-5)
                    else:
                        assert got.snr == pytest.approx(g[k + "_snr"][b], rel=1e-5, abs=1e-9)
                        assert got.tone_deviation == pytest.approx(g[k + "_tonedev"][b], rel=1e-5, abs=1e-5)
                    if got.frame == ol.FRAME_DATA and (b < _strict_blocks(k) or not p.pll_enable):
                        ndata[i] += 1
                        assert _pcm_close(p, pcm[i], g[k + "_pcm"][b], olen * p.channels, 8e-6), (k, b)
            assert (ndata >= 5).all()
        finally:
            eng.close()
