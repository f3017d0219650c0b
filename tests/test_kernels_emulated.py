"""Run the UNMODIFIED HIP kernel sources on the CPU through tests/hipemu (a fiber-based
wavefront/workgroup emulator, test infrastructure only) and compare with the oracle.

This checks butterfly wiring, twiddle tables, LDS indexing, ring wrap, the Hermitian
split and the mirrored store before the kernels see a GPU; the GPU parity tests
(-m gpu) remain the real gate.  The emulator is never part of the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
CSRC = os.path.join(ROOT, "ka9q-radio_amd", "csrc")


@pytest.fixture(scope="module")
def emu(oracle_built):
    so = os.path.join(EMU_DIR, "libchz_emu.so")
    srcs = [os.path.join(EMU_DIR, "emu_kernels.cpp"), os.path.join(EMU_DIR, "hip", "hip_runtime.h")] + \
           [os.path.join(CSRC, f) for f in ("chz_kernels.h", "chz_launch.h", "chz_plan.h", "regfft.h", "chz_finetune.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", EMU_DIR, "-I", CSRC,
                        srcs[0], "-o", so], check=True)
    lib = C.CDLL(so)
    lib.emu_forward.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_double]
    lib.emu_channels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.emu_forward_i16.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int, C.c_char_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emu_channels_real.argtypes = lib.emu_channels.argtypes
    lib.emu_channels_beam.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emu_channels_isb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.emu_noise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.emu_fine_create.restype = C.c_void_p
    lib.emu_fine_create.argtypes = [C.c_int]
    lib.emu_fine_delete.argtypes = [C.c_void_p]
    lib.emu_fine_retune.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.emu_channels_tuned.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_uint, C.c_void_p]
    lib.emu_demod.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_double, C.c_void_p]
    lib.emu_demod_ext_init.argtypes = [C.c_void_p, C.c_int]
    lib.emu_demod_tone_consts.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    lib.emu_demod_ext_size.restype = C.c_int
    lib.emu_demod_sizes.argtypes = [C.c_void_p]
    lib.emu_mini.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("N,in_type,spec,start", [
    (14400, ol.REAL, b"16x25x36", 0), (14400, ol.REAL, b"", 602), (32400, ol.REAL, b"", 0),
    (32400, ol.REAL, b"81x400", 0), (32400, ol.REAL, b"225x144", 100), (14400, ol.REAL, b"36x400", 15002),
    (14400, ol.COMPLEX, b"", 0), (14400, ol.COMPLEX, b"16x25x36", 3000), (60000, ol.COMPLEX, b"", 0),
    (36480, ol.COMPLEX, b"", 0), (36480, ol.COMPLEX, b"152x240", 20000),
    (162000, ol.REAL, b"", 0), (64800, ol.REAL, b"72x25x36", 65000),
    (86400, ol.REAL, b"135x640"[:0] + b"45x16x120", 0), (162000, ol.REAL, b"81x2000"[:0] + b"45x25x144", 1024), (57600, ol.REAL, b"25x16x144", 0),
    (26000, ol.COMPLEX, b"130x200", 8), (52000, ol.REAL, b"130x400", 0), (40000, ol.REAL, b"200x200", 34),   # axes 130 = 10x13, 200 = 10x20
])
def test_forward_kernels(emu, N, in_type, spec, start):
    rng = np.random.default_rng(N + start)
    per = 1 if in_type == ol.REAL else 2
    ring_len = (N + 1000) * per
    ring = rng.standard_normal(ring_len).astype(np.float32)
    win = ring[(start + np.arange(N * per)) % ring_len]
    bins = N // 2 + 1 if in_type == ol.REAL else N
    out = np.zeros(bins, np.complex64)
    desc = C.create_string_buffer(256)
    assert emu.emu_forward(ring.ctypes.data, ring_len, start, N, in_type, spec, out.ctypes.data, desc, 256, None, None, 0, 0.0) == 0
    want = ol.forward(win if in_type == ol.REAL else win.view(np.complex64), in_type, f64=True)
    assert rel(out, want) < 5e-7, desc.value


def test_division_free_storage_index_is_exact(emu):
    """chan_ifft turns a master bin into its storage index with a multiply-high by ceil(2^32 / na): checked against the plain
    k + (k / na) * (pitch - na) + off for EVERY bin of the layouts the planner produces (config 3: 135 / 144 / 4 over 1,620,001
    bins), of every first-axis length in the menu with the largest master it can carry, and of the natural order."""
    emu.emu_spec_index_check.restype = C.c_long
    emu.emu_spec_index_check.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long]
    assert emu.emu_spec_index_check(135, 144, 4, 1620001) == -1
    assert emu.emu_spec_index_check(75, 80, 2, 810001) == -1
    for na in _menu("CHZ_FWD_MENU"):
        pitch = (na + 15) // 16 * 16 + 16
        assert emu.emu_spec_index_check(na, pitch, 5, 6_000_000) == -1, na
    assert emu.emu_spec_index_check(5_200_001, 5_200_001, 0, 5_200_001) == -1      # natural order: quotient times zero


def _menu(name):
    import re
    src = open(os.path.join(CSRC, "chz_plan.h")).read()
    body = re.search(r"#define %s\(X\)(.*?)\n(?://|\n)" % name, src, re.S).group(1)
    return sorted({int(a) * int(b) for a, b in re.findall(r"X\((\d+),\s*(\d+)\)", body)})


def test_forward_kernels_random_plans(emu):
    """Property test over the planner's whole menu: random two- and three-axis products of compiled axis lengths (N up to 70,000),
    REAL and COMPLEX input, a random even window start in a ring that wraps -- the automatic plan against the float64 oracle."""
    axes = _menu("CHZ_FWD_MENU")
    assert 130 in axes and 200 in axes and 152 in axes
    rng = np.random.default_rng(2024)
    done = 0
    while done < 40:
        k = int(rng.integers(2, 4))
        pick = [int(a) for a in rng.choice(axes, k)]
        N = int(np.prod(pick))
        if N > 70000 or N < 64:
            continue
        in_type = ol.REAL if rng.integers(0, 2) else ol.COMPLEX
        if in_type == ol.REAL and N % 2:
            continue
        per = 1 if in_type == ol.REAL else 2
        ring_len = (N + 2 * int(rng.integers(10, 600))) * per
        start = 2 * int(rng.integers(0, ring_len // 2))
        ring = rng.standard_normal(ring_len).astype(np.float32)
        win = ring[(start + np.arange(N * per)) % ring_len]
        bins = N // 2 + 1 if in_type == ol.REAL else N
        out = np.zeros(bins, np.complex64)
        desc = C.create_string_buffer(256)
        r = emu.emu_forward(ring.ctypes.data, ring_len, start, N, in_type, b"", out.ctypes.data, desc, 256, None, None, 0, 0.0)
        assert r == 0, (N, in_type, pick)                    # a product of menu axes always has a plan
        want = ol.forward(win if in_type == ol.REAL else win.view(np.complex64), in_type, f64=True)
        assert rel(out, want) < 5e-7, (N, in_type, desc.value)
        done += 1


def test_generic_channel_sizes_random(emu):
    """Property test for chan_any: random sizes with prime factors up to 13 (outside the register-tiled menu), random master type,
    shifts in and out of range, COMPLEX / ISB / REAL output -- against the restatement."""
    menu = set(_menu("CHZ_CHAN_MENU"))
    rng = np.random.default_rng(77)
    primes = [2, 2, 2, 3, 3, 5, 5, 7, 11, 13]
    done = 0
    while done < 30:
        P = int(np.prod(rng.choice(primes, int(rng.integers(3, 8)))))
        if P in menu or P < 16 or P > 6000:
            continue
        mode = ("plain", "isb", "real")[int(rng.integers(0, 3))]
        if mode == "real" and P % 2:
            continue
        olen = max(1, int(P * rng.uniform(0.3, 1.0)))
        in_type, B = [(ol.REAL, 4801), (ol.COMPLEX, 6000), (ol.COMPLEX, 6001)][int(rng.integers(0, 3))]
        spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
        shifts = [0, -(P // 2), B - 1] + [int(x) for x in rng.integers(-B - P, B + P, 3)]
        nch = len(shifts)
        resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64)
        sh = np.array(shifts, np.int32)
        if mode == "real":
            out = np.zeros((nch, olen), np.float32)
            assert emu.emu_channels_real(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, 0, 0, 0) == 0
        elif mode == "isb":
            flags = (np.arange(nch) % 2).astype(np.uint8)
            out = np.zeros((nch, olen), np.complex64)
            assert emu.emu_channels_isb(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, flags.ctypes.data, out.ctypes.data) == 0
        else:
            out = np.zeros((nch, olen), np.complex64)
            assert emu.emu_channels(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, 0, 0, 0) == 0
        for i, sft in enumerate(shifts):
            kw = dict(out_type=ol.REAL) if mode == "real" else dict(isb=bool(flags[i])) if mode == "isb" else {}
            want = ol.channel(spec, in_type, P, olen, sft, resp[i], **kw)
            if np.linalg.norm(want) == 0:
                assert not out[i].any()
            else:
                assert rel(out[i], want) < 2e-6, (P, olen, mode, in_type, sft)
        done += 1


@pytest.mark.parametrize("P,olen", [(17, 9), (34, 20), (289, 200), (323, 150), (1999, 1200), (4099, 4000), (2 * 4099, 5000), (23 * 29 * 4, 2000)])
def test_channel_sizes_with_large_prime_factors(emu, P, olen):
    """round 3: a channel size with a prime factor above 13 (FFTW plans any size, src/filter.c:101-163) runs through Bluestein's
    chirp-z identity inside chan_any -- primes, squares of primes, sizes beyond the LDS (M = 16384: global scratch), COMPLEX / ISB /
    REAL output -- against the restatement, whose P-point transform is the DFT by definition."""
    rng = np.random.default_rng(P)
    for in_type, B in ((ol.REAL, 4801), (ol.COMPLEX, 9000)):
        spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
        shifts = [0, -(P // 2), 700, -1300] if P < B else [0, 100]
        nch = len(shifts)
        resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64)
        sh = np.array(shifts, np.int32)
        modes = ["plain", "isb"] + (["real"] if P % 2 == 0 else [])
        for mode in modes:
            if mode == "real":
                out = np.zeros((nch, olen), np.float32)
                assert emu.emu_channels_real(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, 0, 0, 0) == 0
            elif mode == "isb":
                flags = (np.arange(nch) % 2).astype(np.uint8)
                out = np.zeros((nch, olen), np.complex64)
                assert emu.emu_channels_isb(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, flags.ctypes.data, out.ctypes.data) == 0
            else:
                out = np.zeros((nch, olen), np.complex64)
                assert emu.emu_channels(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, 0, 0, 0) == 0
            for i, sft in enumerate(shifts):
                kw = dict(out_type=ol.REAL) if mode == "real" else dict(isb=bool(flags[i])) if mode == "isb" else {}
                want = ol.channel(spec, in_type, P, olen, sft, resp[i], **kw)
                if np.linalg.norm(want) == 0:
                    assert not out[i].any()
                else:
                    assert rel(out[i], want) < 4e-6, (P, olen, mode, in_type, sft, rel(out[i], want))


@pytest.mark.parametrize("stage", [0, 1])
@pytest.mark.parametrize("in_type,B", [(ol.REAL, 16201), (ol.COMPLEX, 6000), (ol.COMPLEX, 6001)])
@pytest.mark.parametrize("P,olen", [(300, 240), (600, 480), (200, 160), (400, 320), (1200, 960), (1920, 1536), (150, 120),
                                    (20, 16), (30, 24), (160, 128), (320, 256), (480, 384), (800, 640), (960, 768)])
def test_channel_kernel(emu, in_type, B, P, olen, stage, monkeypatch):
    monkeypatch.setenv("CHZ_CHAN_STAGE", str(stage))     # output rows straight from the lanes / staged through LDS
    rng = np.random.default_rng(B * 7 + P)
    spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
    h = (B + 1) // 2
    shifts = [0, 1, -1, P // 2, -(P // 2), B - 1, -(B - 1), B - P // 2, B + 10, -(B + 10), B + P, -(B + P), h, -h,
              h - P // 2, h + P // 2 - 1] + [int(s) for s in rng.integers(-B - P, B + P, 12)]
    nch = len(shifts)
    resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64)
    sh = np.array(shifts, np.int32)
    out = np.zeros((nch, olen), np.complex64)
    lay = (0, 0, 0) if P % 2 else (135, 144, 4)      # natural order, or the padded layout of a 135-point first axis
    assert emu.emu_channels(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, *lay) == 0
    s64 = spec.astype(np.complex128)
    for i, s in enumerate(shifts):
        want = ol.channel(s64, in_type, P, olen, s, resp[i])
        nrm = np.linalg.norm(want)
        if nrm == 0:
            assert not out[i].any()
        else:
            assert rel(out[i], want) < 1e-6, (s, i)


@pytest.mark.parametrize("N,in_type,spec", [(32400, ol.REAL, b""), (14400, ol.REAL, b"36x400"), (14400, ol.COMPLEX, b"")])
def test_fused_notch(emu, N, in_type, spec):
    # notches ride in the last forward kernel: direct bins, mirrored bins, DC, Nyquist; state persists
    rng = np.random.default_rng(N)
    per = 1 if in_type == ol.REAL else 2
    bins_n = N // 2 + 1 if in_type == ol.REAL else N
    nb = np.array([17, 4001, 5000, 123, bins_n - 1, bins_n // 2 + 7, 0], np.int32)
    st_a, st_b = np.zeros(2 * len(nb)), np.zeros(2 * len(nb))
    desc = C.create_string_buffer(256)
    for blk in range(3):
        ring = (rng.standard_normal(N * per) + 0.25).astype(np.float32)
        out = np.zeros(bins_n, np.complex64)
        assert emu.emu_forward(ring.ctypes.data, N * per, 0, N, in_type, spec, out.ctypes.data, desc, 256,
                               nb.ctypes.data, st_a.ctypes.data, len(nb), 0.01) == 0
        want = ol.forward(ring if in_type == ol.REAL else ring.view(np.complex64), in_type)
        ol.notch(st_b, nb, 0.01, want)
        assert rel(out, want) < 5e-7
        for b in nb:
            assert abs(out[b] - want[b]) <= 3e-6 * abs(want[b]) + 2e-3
    np.testing.assert_allclose(st_a, st_b, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("L,M,olen", [(11520, 2881, 240), (11520, 5761, 200), (11520, 11521, 150)])   # overlap factors V = 5 (the real configurations), 3, 2
def test_tuned_channel_epilogue_follows_downconvert(emu, L, M, olen):
    """Channel kernel + fine-tuning epilogue + host re-basing (chz_finetune.h) against the oracle's
    execute_filter_output followed by the restated downconvert() tail, over a tuning history with shift
    changes, remainder changes, a frequency sweep, and a wrapped job counter (round 4: the block phase
    correction's residues come from the host, 2^32 mod V included: V = 3 and 5 do not divide 2^32)."""
    N, P = L + M - 1, 300
    fs = olen / (L / 576000.0)
    assert P == olen * N // L
    V = 1 + L // (M - 1)
    B = N // 2 + 1
    nch = 5
    rng = np.random.default_rng(17)
    resp = np.stack([ol.set_filter(P, olen, N, True, -0.3, 0.3, 9.0) for _ in range(nch)]).astype(np.complex64)
    # per channel tuning plans: (first block, shift, remainder Hz, doppler rate Hz/s)
    plans = {
        0: [(0, 1000, 3.25, 0.0)],
        1: [(0, 1001, -17.5, 0.0), (3, 1002, -17.5, 0.0), (5, 1002, 4.0, 0.0)],
        2: [(0, -2003, 0.0, 0.0), (4, -2004, 11.0, 0.0)],
        3: [(0, 3004, 5.5, 40.0), (6, 3004, 6.5, -25.0)],     # set_osc runs only when shift/remainder change (:1479)
        4: [(0, 7, 19.999, 0.0), (2, 8, -19.999, 0.0), (7, 6, 1e-3, 0.0)],
    }
    job0 = 0xFFFFFFFD                                   # the unsigned block counter wraps inside the run
    fine = emu.emu_fine_create(nch)
    dcs = [ol.Downconv(L, M, fs, "oracle") for _ in range(nch)]
    cur = [None] * nch
    worst_ulp, same, total = 0.0, 0, 0
    for blk in range(9):
        job = (job0 + blk) & 0xFFFFFFFF
        for ch in range(nch):
            for (b0, sh, rem, dr) in plans[ch]:
                if b0 == blk:
                    cur[ch] = (sh, rem, dr)
                    emu.emu_fine_retune(fine, ch, job, olen, V, sh, -rem / fs, dr / (fs * fs))
        spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
        shifts = np.array([c[0] for c in cur], np.int32)
        got = np.zeros((nch, olen), np.complex64)
        power = np.zeros(nch)
        assert emu.emu_channels_tuned(spec.ctypes.data, B, ol.REAL, P, olen, nch, resp.ctypes.data, shifts.ctypes.data,
                                      got.ctypes.data, fine, V, job, power.ctypes.data) == 0
        for ch in range(nch):
            plain = np.zeros(olen, np.complex64)
            assert emu.emu_channels(spec.ctypes.data, B, ol.REAL, P, olen, 1, resp[ch].ctypes.data, shifts[ch:ch + 1].ctypes.data,
                                    plain.ctypes.data, 0, 0, 0) == 0
            want, pw = dcs[ch].block(plain, *cur[ch])
            ulp = np.spacing(np.maximum(np.abs(want.real), np.abs(want.imag)).astype(np.float32))
            d = got[ch] - want
            worst_ulp = max(worst_ulp, float((np.maximum(np.abs(d.real), np.abs(d.imag)) / ulp).max()))
            same += int((got[ch] == want).sum()); total += olen
            assert abs(power[ch] - pw) <= 1e-6 * pw
    emu.emu_fine_delete(fine)
    assert worst_ulp <= 1.0 and same >= 0.98 * total, (worst_ulp, same, total)


@pytest.mark.parametrize("N,M,spec,start,randomize", [(14400, 2881, b"", 0, False), (14400, 2881, b"16x25x36", 15002, True),
                                                       (32400, 6481, b"", 100, True)])
def test_int16_input_is_converted_on_load(emu, N, M, spec, start, randomize):
    """First pass fed with raw int16 A/D samples: identical spectrum to converting with the restated
    rx888.c convert() first, and the energy / clip statistics cover exactly the L new samples."""
    rng = np.random.default_rng(N + start)
    ring_len = N + 1000
    ring16 = rng.integers(-32768, 32768, ring_len).astype(np.int16)
    ring16[rng.integers(0, ring_len, 40)] = 32767
    ring16[rng.integers(0, ring_len, 40)] = -32768
    scale = np.float32(3.7e-5)
    win16 = ring16[(start + np.arange(N)) % ring_len]
    conv, _, _ = ol.convert_i16(win16, scale, randomize)
    _, energy, clips = ol.convert_i16(win16[M - 1:], scale, randomize)
    ringf, _, _ = ol.convert_i16(ring16, scale, randomize)
    bins = N // 2 + 1
    out = np.zeros(bins, np.complex64); ref = np.zeros(bins, np.complex64)
    en = C.c_ulonglong(0); cl = C.c_uint(0)
    assert emu.emu_forward_i16(ring16.ctypes.data, scale, int(randomize), M - 1, ring_len, start, N, spec, out.ctypes.data,
                               C.byref(en), C.byref(cl)) == 0
    assert emu.emu_forward(ringf.ctypes.data, ring_len, start, N, ol.REAL, spec, ref.ctypes.data, None, 0, None, None, 0, 0.0) == 0
    assert np.array_equal(out, ref)                       # conversion on load is bit-exact
    assert (en.value, cl.value) == (energy, clips)
    assert rel(out, ol.forward(conv, ol.REAL, f64=True)) < 5e-7


@pytest.mark.parametrize("in_type,B,s_bins,lay", [(ol.REAL, 16201, 300, (0, 0, 0)), (ol.REAL, 16201, 1200, (135, 144, 4)),
                                                  (ol.REAL, 7201, 600, (75, 80, 2)), (ol.COMPLEX, 14400, 300, (0, 0, 0))])
def test_noise_estimate_kernel(emu, in_type, B, s_bins, lay):
    """estimate_noise() (src/radio.c:1783-1866) as a kernel: window placement incl. the clamps at DC and
    Nyquist, inverted spectra, the quantile and the thresholded mean -- against the restatement."""
    rng = np.random.default_rng(B + s_bins)
    spec = ((rng.standard_normal(B) + 1j * rng.standard_normal(B)) * 2.5).astype(np.complex64)
    for c in rng.integers(0, B, 12):
        spec[c:c + 30] *= 40.0                                   # carriers the estimator must step over
    if in_type == ol.REAL:
        shifts = np.array([0, 17, 499, 500, 501, 4000, -4000, B - 700, B - 1, -(B - 1), 2500, -2501], np.int32)
    else:
        shifts = np.array([0, 600, -600, 3000, -3000, B // 2 - 200, -(B // 2) + 100, 6000], np.int32)
    n0 = np.zeros(shifts.size)
    assert emu.emu_noise(spec.ctypes.data, B, in_type, s_bins, shifts.size, shifts.ctypes.data, 1.296e6, n0.ctypes.data, *lay) == 0
    want = np.array([ol.estimate_noise(spec, in_type, s_bins, int(s), 1.296e6) for s in shifts])
    assert np.all(want > 0)
    assert np.allclose(n0, want, rtol=1e-12, atol=0)


@pytest.mark.skipif(os.environ.get("CHZ_TEST_TWSHUFFLE") != "1", reason="70 s for a rejected build variant (make twshuffle, profiles/r03_twiddle_ab.jsonl): set CHZ_TEST_TWSHUFFLE=1")
def test_forward_with_lane_generated_twiddles(emu, tmp_path):
    """-DCHZ_TW_SHUFFLE=1: the first pass's column factors generated across each row of 16 lanes (base factor broadcast, powers
    handed on in four DPP steps) instead of read from the table -- the north star's "wavefront-shuffle twiddles", kept as a
    measured-and-rejected build variant.  Same spectrum within the stated tolerance (a few more roundings per factor)."""
    so = str(tmp_path / "libchz_emu_twshuffle.so")
    subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-DCHZ_TW_SHUFFLE=1", "-I", EMU_DIR, "-I", CSRC,
                    os.path.join(EMU_DIR, "emu_kernels.cpp"), "-o", so], check=True)
    alt = C.CDLL(so)
    alt.emu_forward.argtypes = emu.emu_forward.argtypes
    N = 21600                                  # 135 x 160: 80 packed columns, five tiles of 16
    rng = np.random.default_rng(5)
    ring = rng.standard_normal(N).astype(np.float32)
    want = ol.forward(ring, ol.REAL, f64=True)
    errs = []
    for lib in (emu, alt):
        spec = np.zeros(N // 2 + 1, np.complex64)
        desc = C.create_string_buffer(128)
        assert lib.emu_forward(ring.ctypes.data, N, 0, N, ol.REAL, b"135x160", spec.ctypes.data, desc, 128, None, None, 0, 0.0) == 0
        errs.append(rel(spec, want))
    assert errs[0] < 5e-7 and errs[1] < 1e-6 and errs[1] != errs[0], (errs, desc.value)


@pytest.mark.parametrize("kind", ["one_binade", "all_equal", "mostly_zero", "two_values", "wide_range", "quantile_at_a_binade_edge"])
@pytest.mark.parametrize("s_bins", [300, 1500])
def test_noise_estimate_selection_corner_cases(emu, kind, s_bins):
    """The rank selection of round 3 (exponent bits over the whole window, mantissa bits over the values of that one binade
    compacted through LDS, the full-width loop when the binade holds more than 256 values) on windows that stress each branch:
    every energy in one binade (the fallback), all equal, mostly exact zeros, two distinct values, 60 orders of magnitude, and the
    next order statistic lying in the NEXT binade."""
    B = 16201
    rng = np.random.default_rng(len(kind) * 7 + s_bins)
    if kind == "one_binade":
        mag = np.sqrt(rng.uniform(1.0, 1.999, B))
    elif kind == "all_equal":
        mag = np.full(B, 3.0)
    elif kind == "mostly_zero":
        mag = np.where(rng.random(B) < 0.85, 0.0, rng.uniform(0.5, 2.0, B))
    elif kind == "two_values":
        mag = np.where(rng.random(B) < 0.5, 1.0, 4.0)
    elif kind == "wide_range":
        mag = 10.0 ** rng.uniform(-15, 15, B)
    else:
        # about a tenth of the window just below 2.0 in energy, the rest just above: the two order statistics straddle the edge
        mag = np.sqrt(np.where(rng.random(B) < 0.1, rng.uniform(1.5, 1.9999, B), rng.uniform(2.0, 2.5, B)))
    spec = (mag * np.exp(2j * np.pi * rng.random(B))).astype(np.complex64)
    shifts = np.array([0, 499, 700, 4000, -4000, 9000, B - 800, -(B - 1), 2500, 12000], np.int32)
    n0 = np.zeros(shifts.size)
    assert emu.emu_noise(spec.ctypes.data, B, ol.REAL, s_bins, shifts.size, shifts.ctypes.data, 1.296e6, n0.ctypes.data, 135, 144, 4) == 0
    want = np.array([ol.estimate_noise(spec, ol.REAL, s_bins, int(s), 1.296e6) for s in shifts])
    assert np.allclose(n0, want, rtol=1e-12, atol=0), (n0, want)


@pytest.mark.parametrize("in_type,B", [(ol.REAL, 16201), (ol.COMPLEX, 12000), (ol.COMPLEX, 12001)])
@pytest.mark.parametrize("P,olen,mode", [(64, 48, "plain"), (250, 200, "plain"), (720, 576, "isb"), (1000, 800, "real"), (2048, 1024, "plain"),
                                         (2700, 2160, "isb"), (9600, 7680, "plain"), (9600, 7680, "real"), (4096, 2048, "real"),
                                         (5500, 4400, "plain"), (6930, 5544, "isb"), (4032, 3200, "real"),       # 220 k / 277.2 k / 161.28 k: factors 7, 11
                                         (19200, 15360, "plain"), (25200, 20160, "real")])                       # 768 k / 1008 k: beyond the LDS
def test_generic_size_channel_kernel(emu, in_type, B, P, olen, mode):
    """chan_any: any 2-3-5-smooth P the register-tiled menu does not hold (wfm's 384 kHz channel is P = 9600), one workgroup
    per channel, Stockham stages in LDS -- COMPLEX output, ISB unpacking and REAL output against the restatement."""
    src = open(os.path.join(CSRC, "chz_plan.h")).read()
    import re
    menu = re.search(r"#define CHZ_CHAN_MENU\(X\)(.*?)\n\n", src, re.S).group(1)
    assert P not in {int(a) * int(b) for a, b in re.findall(r"X\((\d+),\s*(\d+)\)", menu)}      # really the generic path
    rng = np.random.default_rng(B + 3 * P)
    spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
    shifts = [0, -1, P // 2, -(P // 2) - 3, B - 1, B - P // 2, -(B + 10), (B + 1) // 2] + ([int(x) for x in rng.integers(-B, B, 4)] if P < 4000 else [])
    if P >= 9600 and in_type != ol.REAL:
        shifts = shifts[:3]                                   # the float64 DFT by definition behind the oracle is O(P^2)
    if P > 10240:
        if in_type != ol.REAL:
            pytest.skip("one master type is enough for the sizes that take seconds per channel on the CPU")
        shifts = shifts[:3]
    nch = len(shifts)
    resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64)
    sh = np.array(shifts, np.int32)
    lay = (0, 0, 0) if B % 2 == 0 else (135, 144, 4)
    if mode == "real":
        out = np.zeros((nch, olen), np.float32)
        assert emu.emu_channels_real(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, *lay) == 0
    elif mode == "isb":
        flags = (np.arange(nch) % 3 != 2).astype(np.uint8)
        out = np.zeros((nch, olen), np.complex64)
        assert emu.emu_channels_isb(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, flags.ctypes.data, out.ctypes.data) == 0
    else:
        out = np.zeros((nch, olen), np.complex64)
        assert emu.emu_channels(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, *lay) == 0
    for i, sft in enumerate(shifts):
        kw = dict(out_type=ol.REAL) if mode == "real" else dict(isb=bool(flags[i])) if mode == "isb" else {}
        want = ol.channel(spec, in_type, P, olen, sft, resp[i], **kw)
        if np.linalg.norm(want) == 0:
            assert not out[i].any()
        else:
            assert rel(out[i], want) < 2e-6, (sft, i)


@pytest.mark.parametrize("in_type,B", [(ol.REAL, 4801), (ol.COMPLEX, 6000), (ol.COMPLEX, 6001)])
@pytest.mark.parametrize("P,olen", [(1200, 960), (300, 240), (20, 16), (600, 480)])
def test_real_output_channel_kernel(emu, in_type, B, P, olen):
    """REAL-output slaves (create_filter_output(.., REAL): src/filter.c:794-809 gather, bin (bins+1)/2 zeroed, c2r):
    wfm's composite filters.  Kernel against the restatement, REAL and COMPLEX masters, shifts in and out of range."""
    rng = np.random.default_rng(B + P)
    spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
    shifts = [0, 1, -1, 7, -7, B - P // 2, B - 3, B + 5, -(B // 2), B // 2 - P // 4, -P // 4, P] + [int(s) for s in rng.integers(-B, B, 8)]
    nch = len(shifts)
    resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64)
    sh = np.array(shifts, np.int32)
    out = np.zeros((nch, olen), np.float32)
    lay = (0, 0, 0) if B % 2 == 0 else (75, 80, 2)
    assert emu.emu_channels_real(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, out.ctypes.data, *lay) == 0
    for i, s in enumerate(shifts):
        want = ol.channel(spec, in_type, P, olen, s, resp[i], out_type=ol.REAL)
        nrm = np.linalg.norm(want)
        if nrm == 0:
            assert not out[i].any()
        else:
            assert rel(out[i], want) < 1e-6, (s, i)


@pytest.mark.parametrize("in_type,B", [(ol.REAL, 4801), (ol.COMPLEX, 6000)])
@pytest.mark.parametrize("P,olen", [(300, 240), (20, 16), (1200, 960), (600, 480)])
def test_isb_unpack_in_channel_kernel(emu, in_type, B, P, olen):
    """slave->isb (src/filter.c:895-909): LSB/USB unpacked to I/Q between gather and transform; mixed with plain channels
    in one launch (the partner bin of every register comes from another lane of the channel)."""
    rng = np.random.default_rng(B + P + 1)
    spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
    shifts = [0, 5, -5, P, B // 3, -(B // 4), B - 3, 11]
    flags = np.array([1, 1, 0, 1, 1, 1, 1, 0], np.uint8)
    nch = len(shifts)
    resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64)
    sh = np.array(shifts, np.int32)
    out = np.zeros((nch, olen), np.complex64)
    assert emu.emu_channels_isb(spec.ctypes.data, B, in_type, P, olen, nch, resp.ctypes.data, sh.ctypes.data, flags.ctypes.data, out.ctypes.data) == 0
    for i, s in enumerate(shifts):
        want = ol.channel(spec, in_type, P, olen, s, resp[i], isb=bool(flags[i]))
        nrm = np.linalg.norm(want)
        if nrm == 0:
            assert not out[i].any()
        else:
            assert rel(out[i], want) < 1e-6, (s, i, flags[i])


@pytest.mark.parametrize("B,P,olen", [(6000, 300, 240), (6001, 20, 16), (14400, 600, 480)])
def test_beam_mode_in_channel_kernel(emu, B, P, olen):
    """slave->beam (src/filter.c:756-775): antenna selection / combination weights on a COMPLEX master, mixed with plain
    channels in one launch; shifts through DC (the Re/Im special case), the band edges and the +Nyquist seam."""
    rng = np.random.default_rng(B + P + 2)
    spec = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
    h = B // 2
    shifts = [0, 3, -3, P // 2, -(P // 2), h - 10, -h + 10, h + P // 4, 1000, -1000, h - P // 2 + 1, 7]
    weights = [(1, 0), (0, 1), (0.7 + 0.2j, -0.3 + 0.6j), (1, 1j)]
    nch = len(shifts)
    on = np.array([1] * (nch - 1) + [0], np.uint8)
    ab = np.zeros((nch, 4))
    alphas, betas = [], []
    for i in range(nch):
        a, b = ol.beam_weights(*weights[i % len(weights)])
        alphas.append(a); betas.append(b); ab[i] = [a.real, a.imag, b.real, b.imag]
    resp = (rng.standard_normal((nch, P)) + 1j * rng.standard_normal((nch, P))).astype(np.complex64)
    sh = np.array(shifts, np.int32)
    out = np.zeros((nch, olen), np.complex64)
    assert emu.emu_channels_beam(spec.ctypes.data, B, P, olen, nch, resp.ctypes.data, sh.ctypes.data, ab.ctypes.data,
                                 on.ctypes.data, out.ctypes.data) == 0
    for i, s in enumerate(shifts):
        want = ol.channel_beam(spec, P, olen, s, resp[i], alphas[i], betas[i]) if on[i] else ol.channel(spec, ol.COMPLEX, P, olen, s, resp[i])
        nrm = np.linalg.norm(want)
        if nrm == 0:
            assert not out[i].any()
        else:
            assert rel(out[i], want) < 1e-6, (s, i)


@pytest.mark.parametrize("L,M", [(240, 273), (480, 545), (960, 1089), (240, 61), (300, 101), (1920, 2177),
                                 (252, 133), (660, 441), (1001, 456), (4032, 2017)])      # N = 384, 1100 (11), 1456 (7, 13), 6048 (7)
def test_mini_master_kernel(emu, L, M):
    """radiod's filter2 geometry (src/radio.c:1572-1594: N = round2(2*blocksize), M = N-L+1) and two non-power-of-two
    ones: window -> forward transform -> gather x response (+ISB) -> backward transform, one workgroup per instance,
    against the oracle's overlap-save master + same-size slave; a non-zero shift and ISB ride along."""
    N = L + M - 1
    rng = np.random.default_rng(N)
    nreq = 5
    shifts = np.array([0, 0, 7, -(N // 3), 0], np.int32)
    isb = np.array([0, 1, 0, 0, 1], np.uint8)
    resp = np.stack([ol.set_filter(N, L, N, False, lo, hi, 9.0) for lo, hi in ((-0.2, 0.2), (0.01, 0.3), (-0.4, 0.1), (-0.45, 0.45), (-0.3, -0.05))])
    streams = [ol.Stream(L, M, ol.COMPLEX) for _ in range(nreq)]
    hist = np.zeros((nreq, M - 1), np.complex64)
    for blk in range(3):
        x = (rng.standard_normal((nreq, L)) + 1j * rng.standard_normal((nreq, L))).astype(np.complex64)
        win = np.ascontiguousarray(np.concatenate([hist, x], axis=1))
        out = np.zeros((nreq, L), np.complex64)
        assert emu.emu_mini(win.ctypes.data, nreq, N, L, np.ascontiguousarray(resp).ctypes.data, shifts.ctypes.data, isb.ctypes.data, out.ctypes.data) == 0
        for i in range(nreq):
            spec = streams[i].push(x[i], f64=True)
            want = ol.channel(spec, ol.COMPLEX, N, L, int(shifts[i]), resp[i], isb=bool(isb[i]))
            assert rel(out[i], want) < 3e-6, (blk, i)
        hist = win[:, L:]


def test_every_channel_rate_of_the_reference_configs_has_a_kernel(emu):
    """`samprate = ...` of every channel group in the reference's share/*.conf (grep, 2025 tree) at radiod's block time of 20 ms
    and overlap factors 5 (default) and 2: P = rate * 0.02 * N/L.  All of them must be served -- by the register-tiled menu, by
    chan_any in LDS or by chan_any in global scratch."""
    rates = [8000, 12000, 16000, 24000, 32000, 40000, 48000, 50000, 64000, 80000, 100000, 128000, 160000, 192000, 220000, 277200,
             384000, 768000, 201600, 705600, 403200, 1536000, 161280, 1008000, 5040]      # (912 k / 921.6 k in those files are front ends)
    kinds = {}
    for fs in rates:
        for num, den in ((5, 4), (2, 1)):
            if fs % 50 or (fs // 50 * num) % den:
                continue                     # the reference itself refuses a block that is not a whole number of samples / bins
            olen = fs // 50
            P = olen * num // den
            if P > (1 << 20):
                continue                     # beyond CHZ_ANY_MAX_P the drop-in refuses loudly (none of these rates comes near)
            kinds[(fs, num, den)] = emu.emu_chan_kind(P)
    missing = [k for k, v in kinds.items() if v == 0]
    assert not missing, missing
    assert kinds[(12000, 5, 4)] == 1 and kinds[(384000, 5, 4)] == 2 and kinds[(768000, 5, 4)] == 3


def test_planner_covers_the_front_ends_people_run(emu):
    """Front-end rates of the reference's hardware drivers and example configurations (RX888 at 129.6 / 64.8 / 65.536 / 32.4 MS/s
    and its 130 / 125 / 100 MS/s options, Airspy R2 20 MS/s real, Airspy HF+ 912 k / 768 k / 384 k / 192 k, RTL-SDR 1.8 - 3.2 MS/s,
    SDRplay / HackRF / BladeRF 2 - 20 MS/s, Funcube 192 k, sig_gen at anything) at a 20 ms block with overlap 5 and 2: the planner
    must find axes for every one of them."""
    emu.emu_plan_exists.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    real = [129.6e6, 64.8e6, 65.536e6, 32.4e6, 130e6, 125e6, 100e6, 50e6, 25e6, 20e6]
    cplx = [2.4e6, 2.048e6, 1.8e6, 1.92e6, 3.2e6, 2e6, 2.5e6, 3e6, 4e6, 5e6, 6e6, 8e6, 9.6e6, 10e6, 12.5e6, 20e6,
            912e3, 921.6e3, 768e3, 384e3, 192e3, 250e3, 1.536e6]
    missing = []
    for typ, rates in ((ol.REAL, real), (ol.COMPLEX, cplx)):
        for fs in rates:
            L = int(round(fs * 0.02))
            for M in (L // 4 + 1, L + 1):
                if not emu.emu_plan_exists(L + M - 1, typ, None, 0):
                    missing.append((fs, M))
    assert not missing, missing


def test_notch_owner_names_the_thread_that_stores_the_bin(emu):
    """Round 4: spur notches of short lists are applied inside fwd_rows by the ONE thread that stores the listed bin, and the host names
    that thread (notch_owner, chz_launch.h).  For every front-end geometry the planner serves -- REAL and COMPLEX masters, two- and
    three-axis plans, pitched and natural spectrum layouts, both overlaps -- the pass's own store arithmetic must put that thread's
    output exactly at the bin's storage address: every 7th bin of the master plus its last 600 (the mirrored stores of a real master
    sit in the upper rows).  (The kernel checks the address again at run time and raises the engine's error word on a mismatch.)"""
    emu.emu_notch_owner_check.restype = C.c_long
    emu.emu_notch_owner_check.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
    real = [129.6e6, 64.8e6, 65.536e6, 32.4e6, 130e6, 125e6, 100e6, 50e6, 25e6, 20e6, 1.296e6, 576e3]
    cplx = [2.4e6, 2.048e6, 1.8e6, 1.92e6, 3.2e6, 2e6, 2.5e6, 3e6, 4e6, 5e6, 6e6, 8e6, 9.6e6, 10e6, 12.5e6, 20e6,
            912e3, 921.6e3, 768e3, 384e3, 192e3, 250e3, 1.536e6, 576e3]
    checked = 0
    for typ, rates in ((ol.REAL, real), (ol.COMPLEX, cplx)):
        for fs in rates:
            L = int(round(fs * 0.02))
            for M in (L // 4 + 1, L + 1):
                r = emu.emu_notch_owner_check(L + M - 1, typ, b"", 7, 600)
                assert r == -1, (fs, M, typ, r)
                checked += 1
    for spec in (b"16x25x36", b"81x400", b"225x144", b"144x100x225"):
        N = {b"16x25x36": 14400, b"81x400": 32400, b"225x144": 32400, b"144x100x225": 3240000}[spec]
        for typ in (ol.REAL, ol.COMPLEX):
            r = emu.emu_notch_owner_check(N, typ, spec, 1 if N < 100000 else 5, 600)
            assert r in (-1, -2), (spec, typ, r)
            checked += r == -1
    assert checked >= 70


class _DemodChan(C.Structure):           # struct DemodChan, chz_kernels.h
    _fields_ = [("channels", C.c_int), ("env", C.c_int), ("agc", C.c_int), ("encoding", C.c_int), ("snr_squelch", C.c_int),
                ("squelch_tail", C.c_int), ("tuned", C.c_int), ("on", C.c_int),
                ("samprate", C.c_double), ("headroom", C.c_double), ("threshold", C.c_double), ("recovery_rate", C.c_double),
                ("hangtime", C.c_double), ("dc_alpha", C.c_double), ("bandwidth", C.c_double), ("squelch_open", C.c_double),
                ("squelch_close", C.c_double), ("osc_phase0", C.c_double), ("osc_freq", C.c_double), ("osc_job0", C.c_uint), ("kind", C.c_int),
                ("deemph_rate", C.c_double), ("deemph_gain", C.c_double), ("threshold_extend", C.c_double),
                ("pll_enable", C.c_int), ("pll_square", C.c_int), ("pll_loop_bw", C.c_double), ("tone_freq", C.c_double),
                ("g_coeff", C.c_double), ("g_cfr", C.c_double), ("g_cfi", C.c_double), ("recov_ps", C.c_double)]


class _DemodState(C.Structure):
    _fields_ = [("gain", C.c_double), ("am_dc", C.c_double), ("n0", C.c_double), ("hangcount", C.c_int), ("squelch_state", C.c_int),
                ("squelch_open", C.c_int), ("pll_was_on", C.c_int),
                ("pm_re", C.c_double), ("pm_im", C.c_double), ("deemph_state", C.c_double), ("foffset", C.c_double), ("pdeviation", C.c_double)]


DEMOD_CASES = [dict(), dict(channels=2, encoding=ol.PCM_F32LE), dict(env=True, dc_alpha=0.002, encoding=ol.PCM_S16LE),
               dict(channels=2, env=True, dc_alpha=0.01, encoding=ol.PCM_F32BE), dict(agc=False, gain_db=30.0, shift=500.0),
               dict(snr_squelch=True, squelch_tail=2), dict(tuned=False), dict(encoding=ol.PCM_MULAW), dict(channels=2, encoding=ol.PCM_ALAW),
               dict(encoding=ol.PCM_F16LE), dict(channels=2, env=True, dc_alpha=0.01, encoding=ol.PCM_F16BE)]


# a bank whose every channel sends mono S16 (what voice channels send; big- and little-endian side by side): demod_lin_lanes then packs
# four samples of a row into one 8-byte store -- 70 channels = one full wavefront of rows and a partly filled one
S16_MONO_CASES = [dict(), dict(env=True, dc_alpha=0.002, encoding=ol.PCM_S16LE), dict(agc=False, gain_db=30.0, shift=500.0, encoding=ol.PCM_S16LE),
                  dict(snr_squelch=True, squelch_tail=2), dict(tuned=False), dict(encoding=ol.PCM_S16LE), dict(hangtime=0.3)] * 10


@pytest.mark.parametrize("path,cases", [("lanes", "mixed"), ("wave", "mixed"), ("lanes", "s16_mono")])      # demod_lin_lanes (one channel per lane) / demod_linear_tail (a wavefront per channel)
def test_linear_demodulator_kernel(emu, monkeypatch, path, cases):
    DEMOD_CASES = S16_MONO_CASES if cases == "s16_mono" else globals()["DEMOD_CASES"]
    if path == "wave":
        monkeypatch.setenv("EMU_DEMOD_WAVE", "1")
    else:
        monkeypatch.delenv("EMU_DEMOD_WAVE", raising=False)
    """demod_linear_tail (one lane per channel, the reference's own loop order) against the restated demodulator over 40
    blocks of baseband that walks every AGC branch; all seven mode combinations side by side as seven channels."""
    sizes = (C.c_int * 3)()
    emu.emu_demod_sizes(sizes)
    assert list(sizes) == [C.sizeof(_DemodChan), C.sizeof(_DemodState), C.sizeof(ol.LinStatus)]
    from test_oracle_vs_reference import _demod_case
    nblk, N, bt = (12 if cases == "s16_mono" else 40), 240, 0.02
    nch = len(DEMOD_CASES)
    r = np.random.default_rng(99)
    bbs, powers, ests, params, oracles = [], [], [], [], []
    for i, kw in enumerate(DEMOD_CASES):
        bb, power = _demod_case(np.random.default_rng(100 + i), nblk, N)
        if kw.get("snr_squelch"):
            power = power.copy(); power[20:28] = 1e-12
        bbs.append(bb); powers.append(power); ests.append(1e-8 * (1 + 0.3 * r.standard_normal(nblk)) / 12000.0)
        p = ol.lin_params(**kw); params.append(p); oracles.append(ol.LinDemod(p))
    chan = (_DemodChan * nch)(); state = (_DemodState * nch)(); status = (ol.LinStatus * nch)()
    for i, p in enumerate(params):
        c = chan[i]
        for f in ("channels", "env", "agc", "encoding", "snr_squelch", "squelch_tail", "tuned", "kind", "samprate", "headroom", "threshold",
                  "recovery_rate", "hangtime", "dc_alpha", "bandwidth", "squelch_open", "squelch_close"):
            setattr(c, f, getattr(p, f))
        c.on = 1; c.osc_phase0 = 0.0; c.osc_freq = p.shift / p.samprate; c.osc_job0 = 7
        state[i].gain = p.gain; state[i].n0 = float("nan"); state[i].squelch_state = (p.squelch_tail + 4) if not p.snr_squelch else 0
        state[i].squelch_open = 1
    pcm = np.zeros((nch, N * 8), np.uint8)
    for b in range(nblk):
        x = np.ascontiguousarray(np.stack([bbs[i][b] for i in range(nch)]))
        pw = np.array([powers[i][b] for i in range(nch)]); ne = np.array([ests[i][b] for i in range(nch)])
        assert emu.emu_demod(x.ctypes.data, pw.ctypes.data, ne.ctypes.data, chan, state, status, pcm.ctypes.data, nch, N, 7 + b, bt, None) == 0
        for i, p in enumerate(params):
            want, st = oracles[i].block(bbs[i][b], powers[i][b], ests[i][b], bt)
            got = status[i]
            assert (got.frame, got.mute, got.squelch_state) == (st.frame, st.mute, st.squelch_state), (b, i)
            assert got.gain == pytest.approx(st.gain, rel=1e-12) and got.n0 == pytest.approx(st.n0, rel=1e-14)
            assert got.output_power == pytest.approx(st.output_power, rel=2e-7 if p.env else 1e-12, abs=1e-300)
            if st.frame == ol.FRAME_DATA:
                nb = ol.pcm_bytes(p.encoding, N * p.channels)
                if p.encoding in (ol.PCM_MULAW, ol.PCM_ALAW, ol.PCM_F16LE, ol.PCM_F16BE):
                    assert np.mean(pcm[i, :nb] != want) == 0, (b, i)
                elif p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                    dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                    a, w = pcm[i, :nb].view(dt).astype(np.int32), want.view(dt).astype(np.int32)
                    assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0, (b, i)
                else:
                    dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
                    a, w = pcm[i, :nb].view(dt).astype(np.float64), want.view(dt).astype(np.float64)
                    assert np.abs(a - w).max() <= 2e-7 * max(np.abs(w).max(), 1e-30), (b, i)


FM_CASES = [dict(), dict(threshold_extend=True, encoding=ol.PCM_F32LE), dict(deemph_tc=0, encoding=ol.PCM_S16LE),
            dict(snr_squelch=True, squelch_tail=3, encoding=ol.PCM_F32BE)]


# every channel S16 (big- and little-endian side by side): demod_fm_lanes' last pass then packs four samples per 8-byte store (66 channels: a full wavefront of rows and a partly filled one)
FM_S16_CASES = [dict(), dict(deemph_tc=0, encoding=ol.PCM_S16LE), dict(threshold_extend=True), dict(snr_squelch=True, squelch_tail=3, encoding=ol.PCM_S16LE),
                dict(encoding=ol.PCM_S16LE), dict(threshold_extend=True, deemph_tc=0)] * 11


@pytest.mark.parametrize("path,cases", [("lanes", "mixed"), ("wave", "mixed"), ("lanes", "s16")])      # demod_fm_lanes (one channel per lane) / demod_linear_tail (a wavefront per channel)
def test_fm_demodulator_kernel(emu, monkeypatch, path, cases):
    FM_CASES = FM_S16_CASES if cases == "s16" else globals()["FM_CASES"]
    if path == "wave":
        monkeypatch.setenv("EMU_DEMOD_WAVE", "1")
    else:
        monkeypatch.delenv("EMU_DEMOD_WAVE", raising=False)
    """The FM branch of the demodulator kernel (demod_fm, src/fm.c; PLL / PL tone: test_fm_pll_and_tone_kernel) against the restated demodulator:
    carrier coming up out of the noise, a modulated stretch with a frequency offset, fading out through the squelch tail."""
    from test_oracle_vs_reference import _fm_case
    nblk, N, fs, bt = (18 if cases == "s16" else 36), 480, 24000.0, 0.02
    nch = len(FM_CASES)
    r = np.random.default_rng(5)
    bbs, powers, ests, params, oracles = [], [], [], [], []
    for i, kw in enumerate(FM_CASES):
        bb, power = _fm_case(np.random.default_rng(200 + i), nblk, N, fs, **({"last": 13} if cases == "s16" else {}))
        bbs.append(bb); powers.append(power); ests.append((2 * 2e-3 ** 2 / fs) * (1 + 0.1 * r.standard_normal(nblk)))
        p = ol.fm_params(**kw); params.append(p); oracles.append(ol.FmDemod(p))
    chan = (_DemodChan * nch)(); state = (_DemodState * nch)(); status = (ol.LinStatus * nch)()
    for i, p in enumerate(params):
        c = chan[i]
        for f in ("channels", "env", "agc", "encoding", "snr_squelch", "squelch_tail", "tuned", "kind", "samprate", "headroom", "threshold",
                  "recovery_rate", "hangtime", "dc_alpha", "bandwidth", "squelch_open", "squelch_close", "deemph_rate", "deemph_gain",
                  "threshold_extend"):
            setattr(c, f, getattr(p, f))
        c.on = 1
        state[i].n0 = float("nan")
    pcm = np.zeros((nch, N * 8), np.uint8)
    seen = set()
    for b in range(nblk):
        x = np.ascontiguousarray(np.stack([bbs[i][b] for i in range(nch)]))
        pw = np.array([powers[i][b] for i in range(nch)]); ne = np.array([ests[i][b] for i in range(nch)])
        assert emu.emu_demod(x.ctypes.data, pw.ctypes.data, ne.ctypes.data, chan, state, status, pcm.ctypes.data, nch, N, b, bt, None) == 0
        for i, p in enumerate(params):
            want, st = oracles[i].block(bbs[i][b], powers[i][b], ests[i][b], bt)
            got = status[i]
            assert (got.frame, got.mute, got.squelch_state) == (st.frame, st.mute, st.squelch_state), (b, i)
            assert got.snr == pytest.approx(st.snr, rel=1e-6, abs=1e-12)
            seen.add((got.frame, got.mute))
            if st.frame == ol.FRAME_DATA:
                assert got.output_power == pytest.approx(st.output_power, rel=1e-9) and got.gain == pytest.approx(st.gain, rel=1e-14)
                assert got.foffset == pytest.approx(st.foffset, rel=1e-9, abs=1e-9) and got.pdeviation == pytest.approx(st.pdeviation, rel=1e-9, abs=1e-6)
                nb = ol.pcm_bytes(p.encoding, N)
                if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                    dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                    a, w = pcm[i, :nb].view(dt).astype(np.int32), want.view(dt).astype(np.int32)
                    assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0, (b, i)
                else:
                    dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
                    a, w = pcm[i, :nb].view(dt).astype(np.float64), want.view(dt).astype(np.float64)
                    assert np.abs(a - w).max() <= 2e-7 * max(np.abs(w).max(), 1e-30), (b, i)
    assert (ol.FRAME_DATA, 0) in seen and (ol.FRAME_SILENCE, 0) in seen and (ol.FRAME_SILENCE, 1) in seen


class _DemodExt(C.Structure):            # struct DemodExt, chz_kernels.h (PllState first)
    _fields_ = [("vco_phase", C.c_uint), ("vco_step", C.c_int), ("wraps", C.c_int), ("lock", C.c_int), ("lock_count", C.c_int), ("pad0", C.c_int),
                ("bw", C.c_double), ("damping", C.c_double), ("lower", C.c_double), ("upper", C.c_double), ("u", C.c_double), ("phi", C.c_double),
                ("K1", C.c_double), ("K2", C.c_double),
                ("pll_snr", C.c_double), ("pll_cphase", C.c_double), ("foffset", C.c_double),
                ("g_s0", C.c_double), ("g_s1", C.c_double), ("old_pl_phase", C.c_double), ("tone_deviation", C.c_double),
                ("pll_rotations", C.c_int), ("pl_sample_count", C.c_int), ("tone_mute", C.c_int), ("pad", C.c_int),
                ("fm_snr", C.c_double), ("fm_noise", C.c_double), ("fm_go", C.c_int), ("fm_stage", C.c_int)]


_CHAN_FIELDS = ("channels", "env", "agc", "encoding", "snr_squelch", "squelch_tail", "tuned", "kind", "samprate", "headroom", "threshold",
                "recovery_rate", "hangtime", "dc_alpha", "bandwidth", "squelch_open", "squelch_close", "deemph_rate", "deemph_gain",
                "threshold_extend", "pll_enable", "pll_square", "pll_loop_bw")


def _check_pcm(p, got_row, want, n_samples, tol_f):
    nb = ol.pcm_bytes(p.encoding, n_samples)
    if p.encoding in (ol.PCM_F16LE, ol.PCM_F16BE, ol.PCM_MULAW, ol.PCM_ALAW):
        assert np.array_equal(got_row[:nb], want)
    elif p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
        dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
        a, w = got_row[:nb].view(dt).astype(np.int32), want.view(dt).astype(np.int32)
        assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0
    else:
        dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
        a, w = got_row[:nb].view(dt).astype(np.float64), want.view(dt).astype(np.float64)
        assert np.abs(a - w).max() <= tol_f * max(np.abs(w).max(), 1e-30)


PLL_CASES = [dict(pll=True), dict(pll=True, square=True, pll_bw=20.0, channels=2, encoding=ol.PCM_F32LE),
             dict(pll=True, env=True, dc_alpha=0.002, pll_bw=50.0, squelch_tail=0, encoding=ol.PCM_S16LE), dict()]


@pytest.mark.parametrize("path", ["lanes", "wave"])      # demod_lin_lanes (one channel per lane) / demod_linear_tail (a wavefront per channel)
def test_linear_pll_kernel(emu, monkeypatch, path):
    if path == "wave":
        monkeypatch.setenv("EMU_DEMOD_WAVE", "1")
    else:
        monkeypatch.delenv("EMU_DEMOD_WAVE", raising=False)
    """The coherent modes of the demodulator kernel (src/linear.c:83-153: PLL on one lane, lock detector, PLL squelch) against the
    restated demodulator, which is pinned to the reference's linear.c / osc.c; a channel without the PLL rides along.
    The loop is a recurrence through a truncation to a 32-bit phase word: device and restatement may differ by an LSB of that
    word now and then (2^-32 cycle), which is what the tolerances below allow."""
    assert emu.emu_demod_ext_size() == C.sizeof(_DemodExt)
    from test_oracle_vs_reference import _coherent_case
    nblk, N, bt = 90, 240, 0.02
    nch = len(PLL_CASES)
    bbs, powers, params, oracles = [], [], [], []
    for kw in PLL_CASES:
        bb, power = _coherent_case(np.random.default_rng(3), nblk, N, kw.get("square", False))
        bbs.append(bb); powers.append(power)
        p = ol.lin_params(**kw); params.append(p); oracles.append(ol.LinDemod(p))
    est = 2 * 4e-4 ** 2 / 12000.0
    chan = (_DemodChan * nch)(); state = (_DemodState * nch)(); status = (ol.LinStatus * nch)(); ext = (_DemodExt * nch)()
    emu.emu_demod_ext_init(ext, nch)
    for i, p in enumerate(params):
        c = chan[i]
        for f in _CHAN_FIELDS:
            setattr(c, f, getattr(p, f))
        c.on = 1; c.osc_freq = 0.0
        state[i].gain = p.gain; state[i].n0 = float("nan"); state[i].squelch_open = 1
        state[i].squelch_state = (p.squelch_tail + 4) if not (p.snr_squelch or p.pll_enable) else 0
    pcm = np.zeros((nch, N * 8), np.uint8)
    locked = np.zeros(nch, int)
    for b in range(nblk):
        x = np.ascontiguousarray(np.stack([bbs[i][b] for i in range(nch)]))
        pw = np.array([powers[i][b] for i in range(nch)]); ne = np.full(nch, est)
        assert emu.emu_demod(x.ctypes.data, pw.ctypes.data, ne.ctypes.data, chan, state, status, pcm.ctypes.data, nch, N, b, bt, ext) == 0
        for i, p in enumerate(params):
            want, st = oracles[i].block(bbs[i][b], powers[i][b], est, bt)
            got = status[i]
            assert (got.frame, got.mute, got.squelch_state, got.pll_lock, got.pll_rotations) == \
                (st.frame, st.mute, st.squelch_state, st.pll_lock, st.pll_rotations), (b, i)
            assert got.gain == pytest.approx(st.gain, rel=1e-7) and got.output_power == pytest.approx(st.output_power, rel=1e-6, abs=1e-300)
            if p.pll_enable:
                assert got.pll_snr == pytest.approx(st.pll_snr, rel=1e-6, abs=1e-9) and got.foffset == pytest.approx(st.foffset, rel=1e-6, abs=1e-6)
                assert abs((got.pll_cphase - st.pll_cphase + np.pi) % (2 * np.pi) - np.pi) < 1e-6
            locked[i] += got.pll_lock
            if st.frame == ol.FRAME_DATA:
                _check_pcm(p, pcm[i], want, N * p.channels, 2e-6)
    assert all(10 < locked[i] < nblk - 10 for i in range(3)) and locked[3] == 0


def test_pll_one_channel_per_lane_equals_one_lane_per_wavefront(emu, monkeypatch):
    """round 3: the carrier PLLs of a launch run in a pass of their own, one CHANNEL per lane (pll_lanes: 64 channels' blocks
    transposed through LDS tile by tile), instead of one lane of every channel's wavefront walking its block.  Same statements in
    the same order: status records, PLL state and PCM must agree BIT FOR BIT, over 150 channels (three workgroups, the last one
    ragged) of which every third has no PLL and some are FM or switched off."""
    from test_oracle_vs_reference import _coherent_case
    nblk, N, bt = 12, 240, 0.02
    nch = 150
    kws = [PLL_CASES[i % 4] for i in range(nch)]
    bb0, pw0 = _coherent_case(np.random.default_rng(3), nblk, N, False)
    bb1, pw1 = _coherent_case(np.random.default_rng(4), nblk, N, True)
    results = []
    monkeypatch.setenv("EMU_DEMOD_WAVE", "1")      # both runs through the wavefront kernel: what differs is WHERE the PLL loop runs
    for lane0 in (True, False):
        if lane0:
            monkeypatch.setenv("EMU_PLL_LANE0", "1")
        else:
            monkeypatch.delenv("EMU_PLL_LANE0", raising=False)
        chan = (_DemodChan * nch)(); state = (_DemodState * nch)(); status = (ol.LinStatus * nch)(); ext = (_DemodExt * nch)()
        emu.emu_demod_ext_init(ext, nch)
        for i, kw in enumerate(kws):
            p = ol.lin_params(**kw)
            c = chan[i]
            for f in _CHAN_FIELDS:
                setattr(c, f, getattr(p, f))
            c.on = 0 if i % 17 == 5 else 1; c.osc_freq = 0.0
            state[i].gain = p.gain; state[i].n0 = float("nan"); state[i].squelch_open = 1
            state[i].squelch_state = (p.squelch_tail + 4) if not (p.snr_squelch or p.pll_enable) else 0
        pcm = np.zeros((nch, N * 8), np.uint8)
        trace = []
        for b in range(nblk):
            x = np.ascontiguousarray(np.stack([(bb1 if kws[i].get("square") else bb0)[b] * (1 + 0.01 * (i % 7)) for i in range(nch)]).astype(np.complex64))
            pw = np.array([(pw1 if kws[i].get("square") else pw0)[b] for i in range(nch)]); ne = np.full(nch, 2 * 4e-4 ** 2 / 12000.0)
            assert emu.emu_demod(x.ctypes.data, pw.ctypes.data, ne.ctypes.data, chan, state, status, pcm.ctypes.data, nch, N, b, bt, ext) == 0
            trace.append((bytes(status), bytes(ext), pcm.tobytes()))
        results.append(trace)
    assert results[0] == results[1]


FM2_CASES = [(dict(pll=True, encoding=ol.PCM_F32LE), 0.0), (dict(pll=True, threshold_extend=True), 0.0), (dict(tone_freq=100.0), 100.0),
             (dict(tone_freq=100.0, deemph_tc=0, encoding=ol.PCM_S16LE), 0.0), (dict(tone_freq=123.0, pll=True), 100.0), (dict(), 100.0)]


@pytest.mark.parametrize("path", ["lanes", "wave"])      # demod_fm_lanes (one channel per lane) / demod_linear_tail (a wavefront per channel)
def test_fm_pll_and_tone_kernel(emu, monkeypatch, path):
    if path == "wave":
        monkeypatch.setenv("EMU_DEMOD_WAVE", "1")
    else:
        monkeypatch.delenv("EMU_DEMOD_WAVE", raising=False)
    """The PLL demodulator (src/fm.c:176-203) and the PL-tone squelch (:264-311) of the demodulator kernel against the restatement
    pinned to the reference's fm.c / osc.c / iir.c: the tone present, absent, the wrong one; a plain channel alongside."""
    from test_oracle_vs_reference import _fm_case
    nblk, N, fs, bt = 72, 480, 24000.0, 0.02
    nch = len(FM2_CASES)
    r = np.random.default_rng(5)
    bbs, powers, ests, params, oracles = [], [], [], [], []
    for i, (kw, sent) in enumerate(FM2_CASES):
        bb, power = _fm_case(np.random.default_rng(300 + i), nblk, N, fs, tone=sent, last=60)
        bbs.append(bb); powers.append(power); ests.append((2 * 2e-3 ** 2 / fs) * (1 + 0.1 * r.standard_normal(nblk)))
        p = ol.fm_params(**kw); params.append(p); oracles.append(ol.FmDemod(p))
    chan = (_DemodChan * nch)(); state = (_DemodState * nch)(); status = (ol.LinStatus * nch)(); ext = (_DemodExt * nch)()
    emu.emu_demod_ext_init(ext, nch)
    for i, p in enumerate(params):
        c = chan[i]
        for f in _CHAN_FIELDS:
            setattr(c, f, getattr(p, f))
        c.on = 1
        emu.emu_demod_tone_consts(chan, i, p.tone_freq, p.samprate)
        state[i].n0 = float("nan")
    pcm = np.zeros((nch, N * 8), np.uint8)
    data = np.zeros(nch, int)
    for b in range(nblk):
        x = np.ascontiguousarray(np.stack([bbs[i][b] for i in range(nch)]))
        pw = np.array([powers[i][b] for i in range(nch)]); ne = np.array([ests[i][b] for i in range(nch)])
        assert emu.emu_demod(x.ctypes.data, pw.ctypes.data, ne.ctypes.data, chan, state, status, pcm.ctypes.data, nch, N, b, bt, ext) == 0
        for i, p in enumerate(params):
            want, st = oracles[i].block(bbs[i][b], powers[i][b], ests[i][b], bt)
            got = status[i]
            assert (got.frame, got.mute, got.squelch_state, got.tone_mute) == (st.frame, st.mute, st.squelch_state, st.tone_mute), (b, i)
            assert got.snr == pytest.approx(st.snr, rel=1e-6, abs=1e-12)
            assert got.tone_deviation == pytest.approx(st.tone_deviation, rel=1e-6, abs=1e-9)
            if st.frame == ol.FRAME_DATA:
                data[i] += 1
                assert got.output_power == pytest.approx(st.output_power, rel=2e-6)
                assert got.foffset == pytest.approx(st.foffset, rel=1e-6, abs=1e-6) and got.pdeviation == pytest.approx(st.pdeviation, rel=1e-6, abs=1e-3)
                _check_pcm(p, pcm[i], want, N, 4e-6)
    assert data[0] > 5 and data[1] > 5 and data[2] > 5 and data[3] == 0 and data[4] == 0 and data[5] > 5


def test_fm_loops_one_channel_per_lane_equal_one_lane_per_wavefront(emu, monkeypatch):
    """FM's PLL demodulator and PL-tone detector in passes of their own, one channel per lane (fm_front_k, fm_pll_lanes, fm_tone_lanes,
    fm_finish around demod_linear_tail), against the one-kernel path where lane 0 of the channel's wavefront walks the block: the
    same statements, so status records, PCM and the loops' state must come out bit for bit the same -- 70 channels over two lane
    groups, every FM2 case, some switched off, squelch opening and closing on the way."""
    from test_oracle_vs_reference import _fm_case
    nblk, N, fs, bt = 40, 480, 24000.0, 0.02
    nch = 70
    cases = [FM2_CASES[i % len(FM2_CASES)] for i in range(nch)]
    sig = [_fm_case(np.random.default_rng(300 + k), nblk, N, fs, tone=FM2_CASES[k][1], last=30) for k in range(len(FM2_CASES))]
    r = np.random.default_rng(5)
    est = (2 * 2e-3 ** 2 / fs) * (1 + 0.1 * r.standard_normal((nch, nblk)))
    keep = ("vco_phase", "vco_step", "wraps", "lock", "lock_count", "u", "phi", "g_s0", "g_s1", "old_pl_phase", "tone_deviation", "pl_sample_count", "tone_mute")
    results = []
    monkeypatch.setenv("EMU_DEMOD_WAVE", "1")      # both runs through the wavefront kernel: what differs is WHERE the two loops run
    for lane0 in (True, False):
        if lane0:
            monkeypatch.setenv("EMU_PLL_LANE0", "1")
        else:
            monkeypatch.delenv("EMU_PLL_LANE0", raising=False)
        chan = (_DemodChan * nch)(); state = (_DemodState * nch)(); status = (ol.LinStatus * nch)(); ext = (_DemodExt * nch)()
        emu.emu_demod_ext_init(ext, nch)
        for i, (kw, _) in enumerate(cases):
            p = ol.fm_params(**kw)
            c = chan[i]
            for f in _CHAN_FIELDS:
                setattr(c, f, getattr(p, f))
            c.on = 0 if i % 19 == 7 else 1
            emu.emu_demod_tone_consts(chan, i, p.tone_freq, p.samprate)
            state[i].n0 = float("nan")
        pcm = np.zeros((nch, N * 8), np.uint8)
        trace = []
        for b in range(nblk):
            x = np.ascontiguousarray(np.stack([sig[i % len(FM2_CASES)][0][b] * np.float32(1 + 0.01 * (i % 5)) for i in range(nch)]).astype(np.complex64))
            pw = np.array([sig[i % len(FM2_CASES)][1][b] * (1 + 0.01 * (i % 5)) ** 2 for i in range(nch)]); ne = np.ascontiguousarray(est[:, b])
            assert emu.emu_demod(x.ctypes.data, pw.ctypes.data, ne.ctypes.data, chan, state, status, pcm.ctypes.data, nch, N, b, bt, ext) == 0
            trace.append((bytes(status), bytes(state), pcm.tobytes(), [[getattr(ext[i], f) for f in keep] for i in range(nch)]))
        results.append(trace)
        assert any(status[i].frame == ol.FRAME_DATA for i in range(nch))
    for b in range(nblk):
        assert results[0][b] == results[1][b], b


def random_demod_channels(seed=99, nblk=30, N=240, fs=12000.0):
    """24 channels with randomly drawn demodulator settings (linear and FM alternating) and their test signals."""
    from test_oracle_vs_reference import _demod_case, _fm_case, _coherent_case
    rng = np.random.default_rng(seed)
    encs = [ol.PCM_S16BE, ol.PCM_S16LE, ol.PCM_F32LE, ol.PCM_F32BE, ol.PCM_MULAW, ol.PCM_ALAW, ol.PCM_F16LE, ol.PCM_F16BE]
    params, oracles, bbs, powers, ests = [], [], [], [], []
    for i in range(24):
        enc = encs[int(rng.integers(0, len(encs)))]
        if i % 2 == 0:
            pll = bool(rng.integers(0, 4) == 0)
            kw = dict(channels=int(rng.integers(1, 3)), env=bool(rng.integers(0, 2)), agc=bool(rng.integers(0, 4) > 0), encoding=enc,
                      snr_squelch=bool(rng.integers(0, 3) == 0), squelch_tail=int(rng.integers(0, 4)), tuned=bool(rng.integers(0, 8) > 0),
                      headroom_db=float(rng.uniform(-25, -5)), dc_alpha=float(rng.choice([0.0, 0.002])), bandwidth=float(rng.uniform(500, 6000)),
                      gain_db=float(rng.uniform(20, 70)), pll=pll, pll_bw=float(rng.choice([20.0, 50.0])))
            p = ol.lin_params(**kw)
            bb, power = (_coherent_case(np.random.default_rng(3), 90, N, False) if pll else _demod_case(np.random.default_rng(2000 + i), nblk, N))
            bb, power = bb[:nblk], power[:nblk].copy()
            if kw["snr_squelch"]:
                power[12:18] = 1e-12
            est = 1e-8 * (1 + 0.3 * rng.standard_normal(nblk)) / fs
            orc = ol.LinDemod(p)
        else:
            tone = float(rng.choice([0.0, 0.0, 100.0]))
            kw = dict(encoding=enc, snr_squelch=bool(rng.integers(0, 3) == 0), squelch_tail=int(rng.integers(0, 4)), samprate=fs, bandwidth=8000.0,
                      threshold_extend=bool(rng.integers(0, 2)), deemph_tc=float(rng.choice([0.0, 530.5e-6])), pll=bool(rng.integers(0, 4) == 0), tone_freq=tone)
            p = ol.fm_params(**kw)
            bb, power = _fm_case(np.random.default_rng(3000 + i), nblk, N, fs, tone=tone if rng.integers(0, 2) else 0.0, last=26)
            est = (2 * 2e-3 ** 2 / fs) * (1 + 0.1 * rng.standard_normal(nblk))
            orc = ol.FmDemod(p)
        params.append(p); oracles.append(orc); bbs.append(bb); powers.append(power); ests.append(est)
    return params, oracles, bbs, powers, ests


@pytest.mark.parametrize("N", [240, 250])                # 250: the lane kernels' last tile is a partial one (10 of 16 samples)
@pytest.mark.parametrize("path", ["lanes", "wave"])      # demod_lin_lanes (one channel per lane) / demod_linear_tail (a wavefront per channel)
def test_demodulator_kernel_random_parameter_sweep(emu, monkeypatch, path, N):
    if path == "wave":
        monkeypatch.setenv("EMU_DEMOD_WAVE", "1")
    else:
        monkeypatch.delenv("EMU_DEMOD_WAVE", raising=False)
    """Twenty-four channels with randomly drawn demodulator settings -- linear and FM side by side in ONE launch, every PCM encoding,
    AGC on and off, envelope / carrier removal, squelch variants, PLLs, tone squelch -- run for 30 blocks on the emulated kernel
    against the restated demodulators (which the same kind of sweep pins to the reference's own code)."""
    from test_oracle_vs_reference import _cmp_pcm
    nblk = 30
    params, oracles, bbs, powers, ests = random_demod_channels(99, nblk, N, fs=50.0 * N)
    nch = len(params)
    chan = (_DemodChan * nch)(); state = (_DemodState * nch)(); status = (ol.LinStatus * nch)(); ext = (_DemodExt * nch)()
    emu.emu_demod_ext_init(ext, nch)
    for i, p in enumerate(params):
        c = chan[i]
        for f in _CHAN_FIELDS:
            setattr(c, f, getattr(p, f))
        c.on = 1; c.osc_freq = 0.0
        emu.emu_demod_tone_consts(chan, i, p.tone_freq, p.samprate)
        state[i].n0 = float("nan")
        if p.kind == ol.DEMOD_LINEAR:
            state[i].gain = p.gain; state[i].squelch_open = 1
            state[i].squelch_state = (p.squelch_tail + 4) if not (p.snr_squelch or p.pll_enable) else 0
    pcm = np.zeros((nch, N * 8), np.uint8)
    for b in range(nblk):
        x = np.ascontiguousarray(np.stack([bbs[i][b] for i in range(nch)]))
        pw = np.array([powers[i][b] for i in range(nch)]); ne = np.array([ests[i][b] for i in range(nch)])
        assert emu.emu_demod(x.ctypes.data, pw.ctypes.data, ne.ctypes.data, chan, state, status, pcm.ctypes.data, nch, N, b, 0.02, ext) == 0
        for i, p in enumerate(params):
            want, st = oracles[i].block(bbs[i][b], powers[i][b], ests[i][b], 0.02)
            got = status[i]
            assert (got.frame, got.mute, got.squelch_state, got.pll_lock, got.tone_mute) == (st.frame, st.mute, st.squelch_state, st.pll_lock, st.tone_mute), (b, i)
            assert got.output_power == pytest.approx(st.output_power, rel=3e-6, abs=1e-300), (b, i)
            if st.frame == ol.FRAME_DATA:
                nb = ol.pcm_bytes(p.encoding, N * p.channels)
                assert _cmp_pcm(p, pcm[i, :nb], want, 1e-4 if (p.env and p.dc_alpha) else 6e-6), (b, i)
