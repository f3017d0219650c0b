#!/usr/bin/env python3
"""Diagnostic companion of tests/test_mini_radiod.py: run the reference link (float64 and float32 transform) and the link under test on
the standard channel table, print the A/B summary and, for every channel outside the bar, its frames side by side.
usage: python scripts/mini_radiod_ab.py [--paced] [--blocks N] [--exe PATH] [--seed S]"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mini_radiod_lib as mr

ap = argparse.ArgumentParser()
ap.add_argument("--paced", action="store_true"); ap.add_argument("--blocks", type=int, default=30); ap.add_argument("--exe", default=mr.HIP_EXE)
ap.add_argument("--seed", type=int, default=5); ap.add_argument("--repeat", type=int, default=1)
a = ap.parse_args()
FS, L, M = 1.296e6, 25920, 6481
ch = mr.standard_channels()
by = {c.ssrc: c for c in ch}
x = mr.synthesise(ch, FS, L, a.blocks, seed=a.seed)
pll = {c.ssrc for c in ch if c.preset in ("sam", "ame", "dsb")}
with tempfile.TemporaryDirectory() as tmp:
    A, _, _ = mr.run(mr.REF_EXE, tmp + "/ref", ch, x, FS, L, M, a.blocks)
    A32, _, _ = mr.run(mr.REF_EXE, tmp + "/ref32", ch, x, FS, L, M, a.blocks, env={"MINI_RADIOD_FFT_F32": "1"})
    d_self = mr.diff(A, A32)
    print("reference vs itself:", json.dumps(mr.summary(d_self)))
    for rep in range(a.repeat):
        B, meta, err = mr.run(a.exe, tmp + "/got%d" % rep, ch, x, FS, L, M, a.blocks, paced=int(a.paced))
        print("run", rep, "meta", meta)
        try:
            d = mr.diff(A, B, upto={k: v["agree"] for k, v in d_self.items()})
        except AssertionError as ex:
            print("DISCRETE MISMATCH", ex)
            ssrc = ex.args[0][0] if ex.args and isinstance(ex.args[0], tuple) else None
            if ssrc in A:
                for fa, fb in zip(A[ssrc], B[ssrc]):
                    print("  ", by[ssrc].preset, {k: (fa[k], fb[k]) for k in ("call", "next_jobnum", "block_drops", "mute", "isnull", "pll_lock", "bin_shift", "rtp_timestamp") if fa[k] != fb[k]} or "same",
                          "gain %.6g %.6g bb %.6g %.6g n0 %.6g %.6g" % (fa["gain"], fb["gain"], fa["bb_power"], fb["bb_power"], fa["n0"], fb["n0"]))
            continue
        print("link under test vs reference:", json.dumps(mr.summary(d)))
        try:
            mr.check(d, d_self, pll=pll)
            print("WITHIN THE BAR")
        except AssertionError as ex:
            print("OUTSIDE THE BAR:", ex)
        for ssrc, st in d.items():
            if st["gain_rel"] > 1e-4 or st["bb_power_rel"] > 1e-5 or (ssrc not in pll and st["float_rel"] > 1e-5):
                print("channel", ssrc, by[ssrc].preset, by[ssrc].extra, {k: (float("%.3g" % v) if isinstance(v, float) else v) for k, v in st.items()})
                for fa, fb in zip(A[ssrc], B[ssrc]):
                    print("   call %3d job %3d %s lock %d  gain %.6g %.6g  bb %.6g %.6g  n0 %.6g %.6g  out %.6g %.6g  pllsnr %.4g %.4g cphase %.5g %.5g foff %.5g %.5g" % (
                        fa["call"], fa["next_jobnum"], "N" if fa["isnull"] else "M" if fa["mute"] else "D", fa["pll_lock"], fa["gain"], fb["gain"], fa["bb_power"], fb["bb_power"],
                        fa["n0"], fb["n0"], fa["out_power"], fb["out_power"], fa["pll_snr"], fb["pll_snr"], fa["cphase"], fb["cphase"], fa["foffset"], fb["foffset"]))
