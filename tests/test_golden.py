"""Golden vectors generated from the reference itself (tests/golden/make_golden.py).

CPU: the oracle restatement reproduces them.  GPU: the HIP path reproduces them.
"""
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_pkg

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


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
