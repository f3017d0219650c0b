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
           [os.path.join(CSRC, f) for f in ("chz_kernels.h", "chz_launch.h", "chz_plan.h", "regfft.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", EMU_DIR, "-I", CSRC,
                        srcs[0], "-o", so], check=True)
    lib = C.CDLL(so)
    lib.emu_forward.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_double]
    lib.emu_channels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    return lib


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("N,in_type,spec,start", [
    (14400, ol.REAL, b"16x25x36", 0), (14400, ol.REAL, b"", 602), (32400, ol.REAL, b"", 0),
    (32400, ol.REAL, b"81x400", 0), (32400, ol.REAL, b"225x144", 100), (14400, ol.REAL, b"36x400", 15002),
    (14400, ol.COMPLEX, b"", 0), (14400, ol.COMPLEX, b"16x25x36", 3000), (60000, ol.COMPLEX, b"", 0),
    (162000, ol.REAL, b"", 0), (64800, ol.REAL, b"72x25x36", 65000),
    (86400, ol.REAL, b"135x640"[:0] + b"45x16x120", 0), (162000, ol.REAL, b"81x2000"[:0] + b"45x25x144", 1024), (57600, ol.REAL, b"25x16x144", 0),
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


@pytest.mark.parametrize("in_type,B", [(ol.REAL, 16201), (ol.COMPLEX, 6000), (ol.COMPLEX, 6001)])
@pytest.mark.parametrize("P,olen", [(300, 240), (600, 480), (200, 160), (400, 320), (1200, 960), (150, 120),
                                    (20, 16), (30, 24), (160, 128), (320, 256), (480, 384), (800, 640), (960, 768)])
def test_channel_kernel(emu, in_type, B, P, olen):
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
