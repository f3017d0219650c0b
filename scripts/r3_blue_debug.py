import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, __graft_entry__ as ge, oracle_lib as ol
pkg = ge.load(); ol.build()
L, M = 25920, 6481
P, olen = 85, 68
Mb = 256
rng = np.random.default_rng(1)
eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
eng.write((rng.standard_normal(L)).astype(np.float32)); eng.forward(0)
spec = eng.spectrum(0).astype(np.complex128)
shift = 1000
bank = eng.bank(P, olen, 1)
resp = (rng.standard_normal((1, P)) + 1j * rng.standard_normal((1, P))).astype(np.complex64) / P
bank.set_responses(0, resp); bank.set_shifts(0, np.array([shift], np.int32)); bank.set_active(1)
bank.execute(0); eng.sync()
got = bank.read_slot(0)[0]
# host model
fd = np.zeros(2 * P, np.float32)
import ctypes as C
Y = np.zeros(P, np.complex64)
ol.oracle().chzo_gather.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
s32 = spec.astype(np.complex64)
ol.oracle().chzo_gather(s32.ctypes.data, s32.size, ol.REAL, P, ol.COMPLEX, shift, 0, resp[0].ctypes.data, Y.ctypes.data)
n = np.arange(P)
w = np.exp(1j * np.pi * ((n * n) % (2 * P)) / P)
a = np.zeros(Mb, complex); a[:P] = np.conj(Y * w)
X1 = np.fft.ifft(a) * Mb
b = np.zeros(Mb, complex); b[:P] = np.conj(w); b[Mb - n[1:]] = np.conj(w[1:])
fb = np.fft.fft(b)
C3 = np.conj(X1) * fb
X2 = np.fft.ifft(C3) * Mb
stage = os.environ.get("STAGE", "0")
full = np.zeros(Mb, complex); full[:P] = X2[:P] * w / Mb
model = {"1": a, "2": X1, "3": C3, "4": X2, "5": fb, "6": w, "7": full[P - olen:], "8": np.full(olen, (P - olen) + 1j / Mb)}.get(stage)
if model is None:
    want = ol.channel(spec, ol.REAL, P, olen, shift, resp[0])
    print("final: norm got %.4g want %.4g" % (np.linalg.norm(got), np.linalg.norm(want)))
else:
    m = model[:olen]
    print("stage", stage, "norm got %.4g model %.4g err %.3g" % (np.linalg.norm(got), np.linalg.norm(m), np.linalg.norm(got - m) / max(np.linalg.norm(m), 1e-30)), got[:3], m[:3])
eng.close()
