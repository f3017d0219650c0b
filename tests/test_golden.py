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
