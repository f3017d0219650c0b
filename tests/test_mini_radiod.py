"""The reference's OWN callers -- radio.c (downconvert, set_channel_filter, demod_thread ...), linear.c (demod_linear), fm.c (demod_fm),
compiled unmodified from /root/reference/src -- running on the drop-in, A/B against the same objects on the reference's filter.c.

north_star: "drops in behind ka9q-radio's existing filter.h API ... so radiod, linear.c and fm.c are untouched".  tests/c/mini_radiod.c
says what is the reference's and what is stubbed; tests/c/Makefile links it twice from ONE set of caller objects:
    oracle/_ref/mini_radiod_ref        + the reference's filter.c (+ the oracle's float64 FFT provider)     -- the checker
    tests/c/_prebuilt/mini_radiod_hip  + libka9q_filter_hip.so -> libchz_hip.so (the hand-written kernels)  -- the product
Both are built by __graft_entry__.build() where /root/reference exists and travel to the GPU box.

CPU tier: the checker link runs and the channel table does what it is meant to (squelch closed on empty channels, PL tone found, PLL
locks, filter2 = 4 decimates the frame rate, commands land); the SAME callers on the drop-in's host code over the CPU stand-in engine
(tests/stub/chz_stub.cpp) equal the reference link -- the drop-in's host logic under the reference's own call sequence, no GPU.
GPU tier: the product link on the MI355X against the checker link, the bar of the round-5 review:
frame kinds / mute flags / squelch / timestamps / bin shifts / block_drops identical, float PCM <= 1e-5 rel-L2, int16 PCM <= 1 LSB on
<= 0.1 % of the samples, sig.n0 and bb_power within 1e-5."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import mini_radiod_lib as mr
import oracle_lib as ol

ROOT = mr.ROOT
REF_SRC = "/root/reference/src"
FS, L, M, NBLOCKS = 1.296e6, 25920, 6481, 30        # N = 32,400: 40 Hz bins, 20 ms blocks, overlap 5

needs_ref_exe = pytest.mark.skipif(not os.path.exists(mr.REF_EXE), reason="oracle/_ref/mini_radiod_ref not built (needs /root/reference at build time)")


def _kinds(frames):
    return "".join("N" if f["isnull"] else "M" if f["mute"] else "D" for f in frames)


def _reference_run(tmp, channels, x, nblocks=NBLOCKS, f32=False):
    return mr.run(mr.REF_EXE, os.path.join(tmp, "ref32" if f32 else "ref"), channels, x, FS, L, M, nblocks, env={"MINI_RADIOD_FFT_F32": "1"} if f32 else None)


def _pll_channels(channels):
    return {c.ssrc for c in channels if c.preset in ("sam", "ame", "dsb")}


def _ab(tmp, exe, channels, x, nblocks, geom=None, **kw):
    """reference on the float64 transform (A), reference on the float32 transform (its own spread), the link under test (B)"""
    fs, l, m = geom or (FS, L, M)
    A, _, _ = mr.run(mr.REF_EXE, os.path.join(tmp, "ref"), channels, x, fs, l, m, nblocks)
    A32, _, _ = mr.run(mr.REF_EXE, os.path.join(tmp, "ref32"), channels, x, fs, l, m, nblocks, env={"MINI_RADIOD_FFT_F32": "1"})
    B, meta, err = mr.run(exe, os.path.join(tmp, "got"), channels, x, fs, l, m, nblocks, **kw)
    d_self = mr.diff(A, A32)
    # n0_flip: estimate_noise() averages the bins under a quantile-derived threshold (src/radio.c:1840-1864) -- ONE bin crossing it moves n0 by about 1 / (bins averaged),
    # 1e-4 ... 2e-3 for a 12 kHz channel's ~1000-bin window.  Whether the REFERENCE flips against itself on a given input is luck (own spread 2e-6 on one table, 8e-3 on
    # another), so the worst frame is held to that step size at least; the bar that says the noise estimate is right is the MEDIAN (1e-5, inside check())
    s = mr.check(mr.diff(A, B, upto={k: v["agree"] for k, v in d_self.items()}), d_self, pll=_pll_channels(channels), n0_flip=2e-3)
    s["reference_vs_itself"] = {k: v for k, v in mr.summary(d_self).items() if k not in ("data", "null")}
    return s, B, meta


@needs_ref_exe
def test_reference_callers_on_the_reference_filter(tmp_path):
    """the checker link on its own: every channel thread runs its lifetime down and cleans up through close_chan(), the front end is
    shut down by the last one, and the channel table exercises what it claims to"""
    ch = mr.standard_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS)
    fr, meta, _ = _reference_run(str(tmp_path), ch, x)
    assert int(meta["channels"]) == 48 and int(meta["master_jobs"]) == NBLOCKS and int(meta["shutdowns"]) == 1 and int(meta["commands"]) == 5
    by = {c.ssrc: c for c in ch}
    for ssrc, F in fr.items():
        c = by[ssrc]
        assert all(f["block_drops"] == 0 for f in F)
        assert len(F) == (NBLOCKS // 4 if c.preset in ("cwu", "cwl") else NBLOCKS), (ssrc, c.preset, len(F))      # filter2 = 4: one frame per 4 blocks
        assert [f["next_jobnum"] for f in F] == ([4 * (i + 1) for i in range(len(F))] if c.preset in ("cwu", "cwl") else list(range(1, NBLOCKS + 1)))
    assert _kinds(fr[142]) == "N" * NBLOCKS and _kinds(fr[107]) == "N" * NBLOCKS       # FM / PM on noise: squelch never opens
    assert _kinds(fr[116]) == "N" * NBLOCKS and fr[116][-1]["pll_lock"] == 0           # SAM on noise: no lock
    assert fr[104][-1]["pll_lock"] == 1 and _kinds(fr[104]).endswith("DDDDDDDDDD")     # SAM on a carrier: locked, open
    assert fr[110][-1]["pll_lock"] == 1                                                  # AME
    k = _kinds(fr[118]); assert k.startswith("NNNNNNNNNNNN") and k.endswith("DD")       # PL tone present: opens after the 240 ms integration
    assert _kinds(fr[130]) == "N" * NBLOCKS                                             # PL tone absent: stays shut
    assert fr[105][0]["channels"] == 2 and fr[105][0]["nfloat"] == 480                   # ISB: stereo frames out of filter2
    a, b = fr[100][8], fr[100][10]
    assert a["tune_freq"] != b["tune_freq"] and b["bin_shift"] == a["bin_shift"] + 3    # the retune moved 3 bins (+ 11 Hz of remainder)
    assert fr[100][9]["remainder"] != fr[100][10]["remainder"]


@needs_ref_exe
def test_the_reference_is_not_1e_5_stable_against_its_own_transform(tmp_path):
    """why the A/B bar is relative to the reference's own spread (mini_radiod_lib.check): the SAME link, the same samples, the oracle's
    FFT provider in float64 and in float32 arithmetic (as FFTW computes) -- everything discrete stays identical, n0 and the coherent
    channels' PCM move by far more than 1e-5"""
    ch = mr.standard_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS)
    A, _, _ = _reference_run(str(tmp_path), ch, x)
    A32, _, _ = _reference_run(str(tmp_path), ch, x, f32=True)
    d = mr.diff(A, A32)
    s = mr.summary(d)
    assert s["frames_in_agreement"] >= 0.95 * s["frames"]      # the discrete outputs: (almost) everywhere the same history
    assert s["n0_rel"] > 1e-5 and s["n0_rel_median"] < 1e-5
    pll = _pll_channels(ch)
    assert max(d[k]["float_rel"] for k in pll) > 1e-5
    assert max(v["float_rel"] for k, v in d.items() if k not in pll) <= 1e-5
    assert s["bb_power_rel"] <= 1e-5 and s["lsb_max"] <= 1


def _build_stub_link(out_dir, sanitize=None):
    """the drop-in's host code (filter_hip.c, unmodified) over the CPU stand-in for libchz_hip.so, and mini-radiod's caller objects on it
    (sanitize="address": the drop-in and the stand-in instrumented, the reference's objects as they are)"""
    ol.build()
    stub = os.path.join(ROOT, "tests", "stub")
    san = ["-fsanitize=" + sanitize, "-O1"] if sanitize else ["-O2"]
    subprocess.run(["g++", "-std=c++17", "-g", "-fPIC", "-shared"] + san + [os.path.join(stub, "chz_stub.cpp"), "-o", os.path.join(out_dir, "libchz_hip.so"),
                    "-L", os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread"], check=True)
    subprocess.run(["gcc", "-g", "-std=gnu11", "-fPIC", "-shared"] + san + [os.path.join(ROOT, "ka9q-radio_amd", "csrc", "filter_hip.c"), "-o",
                    os.path.join(out_dir, "libka9q_filter_hip.so"), "-L", out_dir, "-lchz_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"], check=True)
    exe = os.path.join(out_dir, "mini_radiod_stub")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), "LIBDIR=" + out_dir, "OUT_HIP=" + exe, exe] + (["LINK_EXTRA=-fsanitize=" + sanitize] if sanitize else []), check=True)
    return exe


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_reference_callers_on_the_dropin_host_code_over_the_cpu_stand_in(tmp_path):
    exe = _build_stub_link(str(tmp_path))
    ch = mr.standard_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS)
    A, _, _ = _reference_run(str(tmp_path), ch, x)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, FS, L, M, NBLOCKS)
    assert int(meta["commands"]) == 5 and int(meta["shutdowns"]) == 1
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)       # the stand-in computes with the oracle's float64 transforms: nothing but rounding order differs
    assert s["data"] > 1000 and s["null"] > 100 and s["frames_in_agreement"] == s["frames"], s


CONFIG3 = (129.6e6, 2592000, 648001)     # BASELINE config 3: N = 3,240,000, 1024 mixed usb / cw / iq channels
CONFIG2 = (64.8e6, 1296000, 324001)      # BASELINE config 2: N = 1,620,000, 256 x 12 kHz NBFM channels


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_config3_through_the_reference_callers_on_the_dropin_host_code(tmp_path):
    """BASELINE config 3 itself -- 129.6 MS/s real, 1024 mixed usb / cw / iq channels, one REAL radiod channel thread each (demod_thread ->
    demod_linear -> downconvert; the CW third with filter2 = 4: 341 pooled inline masters) -- on the drop-in's host code over the CPU
    stand-in engine against the same objects on the reference's filter.c: 1024 threads publishing shifts, block 0, the miss path,
    filter2's leader / follower batching, lifetimes running out, close_chan() -- with nothing but rounding order between the two links."""
    exe = _build_stub_link(str(tmp_path))
    fs, l, m = CONFIG3
    ch, lines = mr.config3_channels()
    nb = 8
    x = mr.spectral_synth(lines, fs, nb * l, 0.002, 7)
    A, _, _ = mr.run(mr.REF_EXE, str(tmp_path / "ref"), ch, x, fs, l, m, nb)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, fs, l, m, nb)
    assert int(meta["channels"]) == 1024 and int(meta["master_jobs"]) == nb and int(meta["shutdowns"]) == 1
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert s["frames_in_agreement"] == s["frames"] == 342 * nb + 341 * (nb // 4) + 341 * nb, s


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_complex_front_end_and_sharded_master_on_the_dropin_host_code(tmp_path):
    """the two remaining shapes of BASELINE.json through the reference's callers on the drop-in's host code (stand-in engine): config 1 -- a
    COMPLEX 2.4 MS/s front end (write_cfilter), one IQ channel and one at a negative frequency --, and config 4's shape -- 2000 x 24 kHz
    channels (Nchannels) behind ONE master whose slaves are sharded over two stand-in devices (KA9Q_HIP_DEVICES=0,1)"""
    exe = _build_stub_link(str(tmp_path))
    ch, lines = mr.config1_channels()
    fs, l, m = 2.4e6, 48000, 12001
    x = mr.complex_synth(lines, fs, 40 * l, 0.002, 3)
    A, _, _ = mr.run(mr.REF_EXE, str(tmp_path / "ref1"), ch, x, fs, l, m, 40)
    B, meta, _ = mr.run(exe, str(tmp_path / "got1"), ch, x, fs, l, m, 40)
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert s["frames_in_agreement"] == s["frames"] == 80 and A[3001][-1]["bin_shift"] < 0
    ch, lines = mr.config4_channels()
    fs, l, m = CONFIG3
    x = mr.spectral_synth(lines, fs, 6 * l, 0.002, 11)
    A, _, _ = mr.run(mr.REF_EXE, str(tmp_path / "ref4"), ch, x, fs, l, m, 6)
    B, meta, _ = mr.run(exe, str(tmp_path / "got4"), ch, x, fs, l, m, 6, env={"CHZ_STUB_DEVICES": "2", "KA9Q_HIP_DEVICES": "0,1"})
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert s["frames_in_agreement"] == s["frames"] == 12000


def _check_churn(fr, nblocks):
    by = {c.ssrc: c for c in mr.churn_channels()}
    assert sorted(fr) == sorted(by), sorted(set(by) - set(fr))           # every channel, the late ones included, produced frames
    for ssrc, F in fr.items():
        c = by[ssrc]
        kv = dict(t.split("=", 1) for t in c.extra.split() if "=" in t)
        start, life = int(kv.get("start", 0)), int(kv.get("life", 0))
        blocks = life if life else nblocks - start
        per = 4 if c.preset in ("cwu", "cwl") else 1
        assert len(F) == blocks // per, (ssrc, c.preset, start, life, len(F))
        assert F[0]["next_jobnum"] == start + per and all(f["block_drops"] == 0 for f in F), (ssrc, F[0]["next_jobnum"], start)


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_channels_joining_and_leaving_on_the_dropin_host_code(tmp_path):
    """radiod's dynamic channels through the reference's own lifecycle code on the drop-in's host code (stand-in engine): eight channels run
    out of lifetime mid-stream (close_chan -> delete_filter_output while 50 others run), ten join mid-stream (create_filter_output on a
    running master from another thread, the bank grows, the newcomers reuse the freed Channel_list slots)"""
    exe = _build_stub_link(str(tmp_path))
    ch = mr.churn_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS, seed=21)
    A, _, _ = _reference_run(str(tmp_path), ch, x)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, FS, L, M, NBLOCKS)
    _check_churn(A, NBLOCKS)
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert int(meta["channels"]) == 58 and s["frames_in_agreement"] == s["frames"], s


def _check_spectrum(fr, wide, nblocks):
    """every analyser answered every poll; the narrowband ones with the bin counts asked for out of the block sizes spectrum.c derives"""
    assert sorted(wide) == [503] and len(wide[503]) == nblocks and all(f["nfloat"] == 128 for f in wide[503])
    for ssrc, bins in ((500, 200), (501, 256), (502, 100)):
        F = fr[ssrc]
        assert len(F) == nblocks and all(f["nfloat"] == bins and f["block_drops"] == 0 for f in F), (ssrc, len(F))
        assert [f["next_jobnum"] for f in F] == list(range(1, nblocks + 1))
        p = np.asarray(F[-1]["pcm_f"], dtype=np.float64)
        assert np.max(p) > 30 * np.median(p), ssrc             # the two lines stand out of the noise floor
    assert (fr[500][0]["olen"], fr[501][0]["olen"], fr[502][0]["olen"]) == (208, 520, 60)      # 10.4 / 26 / 3 kHz slaves: P = 261, 651, 76


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_spectrum_analysers_on_the_dropin_host_code(tmp_path):
    """src/spectrum.c's demod_spectrum() on the drop-in's host code (stand-in engine): three narrowband analysers -- COMPLEX slaves of 208, 520 and
    60 samples per block, sizes no demodulator asks for; set_filter() with a Kaiser beta of its own; the analysis transform planned through
    the drop-in's plan_complex() -- and one wideband analyser (a SPECTRUM slave as block clock), polled once per block next to 12 ordinary
    channels.  The narrowband bin data equal the reference link's; the wideband bins come from an asynchronous read of the A/D ring
    (src/spectrum.c wideband_poll) and are only counted."""
    exe = _build_stub_link(str(tmp_path))
    ch = mr.spectrum_channels()
    x = mr.spectrum_signal(ch, FS, L, NBLOCKS)
    A, _, _ = _reference_run(str(tmp_path), ch, x)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, FS, L, M, NBLOCKS)
    A, Aw = mr.split_wideband(A, ch)
    B, Bw = mr.split_wideband(B, ch)
    _check_spectrum(A, Aw, NBLOCKS)
    _check_spectrum(B, Bw, NBLOCKS)
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert int(meta["channels"]) == 16 and s["frames_in_agreement"] == s["frames"], s


WFM_SETTLE = {600: 4, 601: 4, 602: 4, 603: 4}


def _check_wfm(fr, nblocks):
    """the stereo channel decodes stereo, the mono channel and the pilot-less multiplex mono, the empty channel stays shut"""
    assert _kinds(fr[600]) == _kinds(fr[601]) == _kinds(fr[602]) == "D" * nblocks and _kinds(fr[603])[4:] == "N" * (nblocks - 4)      # (before sig.n0 has settled the empty channel's squelch may open for a frame or two, in the reference too)
    assert all(f["channels"] == 2 and f["nfloat"] == 1920 for f in fr[600][1:]) and all(f["channels"] == 1 and f["nfloat"] == 960 for f in fr[601] + fr[602][1:])
    assert all(f["olen"] == 7680 and f["block_drops"] == 0 for s in (600, 601, 602, 603) for f in fr[s])
    lr = np.asarray(fr[600][-1]["pcm_f"], dtype=np.float64).reshape(-1, 2)
    spec = np.abs(np.fft.rfft(lr * np.hanning(960)[:, None], axis=0))           # 50 Hz bins: left = 1000 Hz, right = 2600 Hz
    assert spec[20, 0] > 10 * spec[52, 0] and spec[52, 1] > 10 * spec[20, 1], "stereo separation"
    assert 60e3 < fr[600][-1]["pdeviation"] < 80e3


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_wfm_stereo_decoder_on_the_dropin_host_code(tmp_path):
    """src/wfm.c's demod_wfm() on the drop-in's host code (stand-in engine): a 384 kHz COMPLEX slave off the front end (7680 samples per block, P = 9600),
    and -- per channel -- a REAL inline master of its own (the FM composite: L = 7680, M = 7681, N = 15,360, perform_inline set AFTER creation) with a
    REAL slave (mono) and two COMPLEX slaves spun down by 19 and 38 kHz through execute_filter_output()'s shift (pilot, L - R), all decimating by 8."""
    exe = _build_stub_link(str(tmp_path))
    ch = mr.wfm_channels()
    fs, l, m = mr.WFM_GEOM
    x = mr.synthesise(ch, fs, l, NBLOCKS, seed=41)
    A, _, _ = mr.run(mr.REF_EXE, str(tmp_path / "ref"), ch, x, fs, l, m, NBLOCKS)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, fs, l, m, NBLOCKS)
    _check_wfm(A, NBLOCKS)
    s = mr.check(mr.diff(A, B, settle=WFM_SETTLE), None, float_tol=1e-6, n0_tol=1e-9)
    assert int(meta["channels"]) == 12 and s["frames_in_agreement"] == s["frames"], s
    # with the master's slaves spread over two (stand-in) devices: the front-end master is, the WFM channels' private composite masters (15,360 points, three
    # slaves each) are not -- every listed device would copy and transform their every block for nothing (filter_hip.c: SHARD_MIN_POINTS)
    B, meta, err = mr.run(exe, str(tmp_path / "got2"), ch, x, fs, l, m, NBLOCKS, env={"CHZ_STUB_DEVICES": "2", "KA9Q_HIP_DEVICES": "0,1", "KA9Q_HIP_PROFILE": "1"})
    prof = [dict(t.split("=") for t in ln.split() if "=" in t) for ln in err.splitlines() if ln.startswith("filter_hip profile:")]
    assert sorted((int(p["points"]), int(p["devices"])) for p in prof) == [(15360, 1)] * 3 + [(64800, 2)], prof     # (the empty WFM channel never opens its squelch: its composite master never runs a block and prints nothing)
    s = mr.check(mr.diff(A, B, settle=WFM_SETTLE), None, float_tol=1e-6, n0_tol=1e-9)
    assert s["frames_in_agreement"] == s["frames"], s


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_wfm_deletes_its_master_before_its_slaves_without_leaks(tmp_path):
    """src/wfm.c:290-293 ends with delete_filter_input(&composite) FIRST and the three delete_filter_output() after it; the reference's
    delete_filter_output never looks at the master (src/filter.c:943-957).  The drop-in's host code under AddressSanitizer + LeakSanitizer through the whole
    life of four WFM channels: no invalid access, and nothing of the slaves left behind (round 6: their contexts were, found by this run)."""
    probe = subprocess.run(["gcc", "-fsanitize=address", "-x", "c", "-", "-o", str(tmp_path / "probe")], input="int main(void){return 0;}", text=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("no -fsanitize=address runtime in this image")
    exe = _build_stub_link(str(tmp_path), sanitize="address")
    ch = mr.wfm_channels()
    fs, l, m = mr.WFM_GEOM
    nb = 12
    x = mr.synthesise(ch, fs, l, nb, seed=41)
    B, meta, err = mr.run(exe, str(tmp_path / "got"), ch, x, fs, l, m, nb, env={"ASAN_OPTIONS": "detect_leaks=1"})
    assert int(meta["channels"]) == 12 and "AddressSanitizer" not in err and "LeakSanitizer" not in err, err[-3000:]
    _check_wfm(B, nb)
    # ... and through restarts (a filter output deleted and created again at another size by its own thread) and an undecided small master becoming an engine
    for name, ch, (fs, l, m), xx in (("switch", mr.switch_channels(), (FS, L, M), None), ("funcube",) + (lambda cl: (cl[0], mr.FUNCUBE_GEOM, cl[1]))(mr.beam_handover_channels())):
        x = mr.complex_synth(xx, fs, NBLOCKS * l, 0.002, 5) if name == "funcube" else mr.synthesise(ch, fs, l, NBLOCKS, seed=51)
        B, meta, err = mr.run(exe, str(tmp_path / name), ch, x, fs, l, m, NBLOCKS, env={"ASAN_OPTIONS": "detect_leaks=1"}, timeout=300)
        assert "AddressSanitizer" not in err and "LeakSanitizer" not in err, (name, err[-3000:])


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_reference_callers_on_the_dropin_host_code_under_thread_sanitizer(tmp_path):
    """the drop-in's host code and the stand-in engine compiled with -fsanitize=thread under the reference's own threads (pthread calls of the
    uninstrumented caller objects are intercepted all the same): channels joining and leaving a running master, and WFM channels creating and
    deleting engines of their own from their threads -- no report"""
    probe = subprocess.run(["gcc", "-fsanitize=thread", "-x", "c", "-", "-o", str(tmp_path / "probe")], input="int main(void){return 0;}", text=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("no -fsanitize=thread runtime in this image")
    exe = _build_stub_link(str(tmp_path), sanitize="thread")
    env = {"TSAN_OPTIONS": "halt_on_error=0 report_signal_unsafe=0 exitcode=0"}
    fc, fc_lines = mr.beam_handover_channels()
    for name, ch, (fs, l, m), seed in (("churn", mr.churn_channels(), (FS, L, M), 21), ("wfm", mr.wfm_channels(), mr.WFM_GEOM, 41),
                                       ("switch", mr.switch_channels(), (FS, L, M), 51), ("funcube", fc, mr.FUNCUBE_GEOM, 5)):
        x = mr.complex_synth(fc_lines, fs, NBLOCKS * l, 0.002, seed) if name == "funcube" else mr.synthesise(ch, fs, l, NBLOCKS, seed=seed)        # (the churn table's last channel joins at block 26)
        B, meta, err = mr.run(exe, str(tmp_path / name), ch, x, fs, l, m, NBLOCKS, env=env, timeout=600)
        assert "ThreadSanitizer" not in err, (name, err[-4000:])
        assert int(meta["shutdowns"]) == 1


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_small_complex_front_end_on_the_dropin_host_code(tmp_path):
    """a Funcube dongle's front end (192 kHz COMPLEX, N = 4800) under the reference's callers on the drop-in's host code: to create_filter_input such a master
    looks like radiod's filter2; the first channel's decimating slave turns it into an engine in place, while the CW channel's filter2 = 4 next to it IS a
    pooled inline master"""
    exe = _build_stub_link(str(tmp_path))
    ch, lines = mr.funcube_channels()
    fs, l, m = mr.FUNCUBE_GEOM
    nb = 40
    x = mr.complex_synth(lines, fs, nb * l, 0.002, 5)
    A, _, _ = mr.run(mr.REF_EXE, str(tmp_path / "ref"), ch, x, fs, l, m, nb)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, fs, l, m, nb)
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert int(meta["channels"]) == 7 and s["frames_in_agreement"] == s["frames"] == 6 * nb + nb // 4 and s["data"] == 5 * nb + nb // 4, s


def _check_switches(fr, nblocks):
    """the restarted channels skipped exactly the two blocks the lock-step front end was ahead, and came back at their new size"""
    assert len(fr[700]) == nblocks - 2 and [f["olen"] for f in fr[700]][8:10] == [240, 480] and fr[700][-1]["nfloat"] == 480          # 12 -> 24 kHz after call 8
    assert len(fr[701]) == nblocks - 2 and fr[701][10]["olen"] == 240 and fr[701][11]["olen"] == 480                                    # lsb -> the fm preset
    assert fr[701][11]["next_jobnum"] == fr[701][10]["next_jobnum"] + 3
    assert [f["next_jobnum"] for f in fr[702]][:13] == list(range(1, 14)) and fr[702][-1]["nfloat"] == 960                              # -> cwu: filter2 = 4 from call 13 on, no restart
    assert all(b["next_jobnum"] - a["next_jobnum"] == 4 for a, b in zip(fr[702][13:], fr[702][14:]))
    assert fr[703][9]["channels"] == 2 and fr[703][10]["channels"] == 1 and fr[703][-1]["olen"] == 160 and len(fr[703]) == nblocks - 2   # iq -> usb (no restart), then 8 kHz
    assert len(fr[704]) == len(fr[705]) == nblocks and all(f["block_drops"] == 0 for F in fr.values() for f in F)
    assert [f["channels"] for f in fr[704]][6:9] == [1, 1, 2] and fr[704][-1]["nfloat"] == 480                                           # ISB switched on after call 7: stereo out of a filter2 that appears then


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_presets_and_sample_rates_changed_on_running_channels_on_the_dropin_host_code(tmp_path):
    """what `control` does to a running channel, through the reference's own restart path (src/radio_status.c:613-660 -> src/radio.c:940-985): a new sample rate
    or demodulator makes the channel thread delete its filter output and create one of another size on the running master; a CW preset on a running USB
    channel brings filter2 = 4 in mid-stream.  Drop-in host code over the stand-in engine against the reference link, frame by frame."""
    exe = _build_stub_link(str(tmp_path))
    ch = mr.switch_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS, seed=51)
    A, _, _ = _reference_run(str(tmp_path), ch, x)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, FS, L, M, NBLOCKS)
    _check_switches(A, NBLOCKS)
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert int(meta["commands"]) == 9 and s["frames_in_agreement"] == s["frames"] == 161, s


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_a_leaving_beam_channel_does_not_hand_its_beam_form_to_the_slave_that_takes_its_index(tmp_path):
    """round 6, found by running the reference's callers: when a slave in beam mode (src/radio.c:938-940) was deleted, the slave that took over its bank index was
    computed with the leaver's beam weights from then on -- the host's record of what is uploaded at that index was stale"""
    exe = _build_stub_link(str(tmp_path))
    ch, lines = mr.beam_handover_channels()
    fs, l, m = mr.FUNCUBE_GEOM
    nb = 32
    x = mr.complex_synth(lines, fs, nb * l, 0.002, 5)
    A, _, _ = mr.run(mr.REF_EXE, str(tmp_path / "ref"), ch, x, fs, l, m, nb)
    B, meta, _ = mr.run(exe, str(tmp_path / "got"), ch, x, fs, l, m, nb)
    assert len(A[4006]) == 10 and len(A[4007]) == nb - 6
    s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
    assert int(meta["channels"]) == 8 and s["frames_in_agreement"] == s["frames"], s


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
def test_a_device_failure_under_the_reference_callers(tmp_path):
    """the stand-in engine reports a failed device-side check at block 10 (CHZ_STUB_FAIL_JOB) while 48 channel threads of the reference's own code run: the drop-in
    replaces its engine, the blocks in flight are counted drops (zeros, src/filter.c:690-701), every thread keeps its frame count and carries on -- no hang,
    no crash, and the stateless channels are back on the reference link's samples a few blocks later"""
    exe = _build_stub_link(str(tmp_path))
    ch = mr.standard_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS)
    A, _, _ = _reference_run(str(tmp_path), ch, x)
    B, meta, err = mr.run(exe, str(tmp_path / "got"), ch, x, FS, L, M, NBLOCKS, env={"CHZ_STUB_FAIL_JOB": "10"}, timeout=300)
    assert "re-creating the engine" in err and int(meta["shutdowns"]) == 1
    assert sorted(B) == sorted(A) and all(len(B[k]) == len(A[k]) for k in A)
    drops = {F[-1]["block_drops"] for F in B.values()}
    assert drops <= {1, 2, 3} and 2 in drops, drops
    # an FM channel on noise and a squelched channel carry no state across the gap: identical frames again by block 20
    for k in (142, 107):
        assert all(a["isnull"] == b["isnull"] and a["mute"] == b["mute"] for a, b in zip(A[k][20:], B[k][20:]))


@needs_ref_exe
@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="/root/reference absent: the caller objects cannot be linked here")
@pytest.mark.parametrize("env", [{"CHZ_STUB_DEVICES": "2", "KA9Q_HIP_DEVICES": "0,1", "KA9Q_HIP_SHARD_CHANNELS": "7"},
                                 {"CHZ_STUB_DEVICES": "3", "KA9Q_HIP_DEVICES": "0,1,2", "KA9Q_HIP_SHARD_CHANNELS": "5", "KA9Q_HIP_EXCHANGE": "broadcast"}], ids=["two_devices_samples", "three_devices_broadcast"])
def test_sharded_master_under_the_reference_callers(tmp_path, env):
    """the front-end master's slaves spread over two / three stand-in devices a handful at a time (so that joiners, leavers and restarted channels land on every
    device and "where the fewest live" decides), both exchanges: channels joining and leaving, presets and sample rates changed on running channels, the 48-channel
    table -- every frame the reference link's"""
    exe = _build_stub_link(str(tmp_path))
    for name, ch, seed in (("churn", mr.churn_channels(), 21), ("switch", mr.switch_channels(), 51), ("standard", mr.standard_channels(), 5)):
        x = mr.synthesise(ch, FS, L, NBLOCKS, seed=seed)
        A, _, _ = mr.run(mr.REF_EXE, str(tmp_path / ("ref_" + name)), ch, x, FS, L, M, NBLOCKS)
        B, meta, _ = mr.run(exe, str(tmp_path / ("got_" + name)), ch, x, FS, L, M, NBLOCKS, env=env)
        s = mr.compare(A, B, float_tol=1e-6, n0_tol=1e-9)
        assert s["frames_in_agreement"] == s["frames"], (name, s)


def _hip_exe():
    if os.path.isdir(REF_SRC):          # (this container: rebuild if the sources or the libraries changed; the GPU box runs what travelled)
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), "all"], check=True)
    assert os.path.exists(mr.HIP_EXE), "tests/c/_prebuilt/mini_radiod_hip missing: __graft_entry__.build() makes it where /root/reference exists"
    assert os.path.exists(mr.REF_EXE), "oracle/_ref/mini_radiod_ref missing"
    return mr.HIP_EXE


@pytest.mark.gpu
def test_reference_callers_on_the_mi355x_match_the_reference_filter():
    """THE row-31 test: 48 channels (usb lsb cwu/cwl [filter2 = 4] am sam iq [one ISB] fm pm nfm dsb ame; S16BE and F32LE; SNR squelch,
    PLL squelch, PL tone squelch; two retunes, three set_filter()s mid-stream), 30 blocks, real demod_linear() / demod_fm() threads"""
    exe = _hip_exe()
    ch = mr.standard_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS)
    with tempfile.TemporaryDirectory() as tmp:
        s, B, meta = _ab(tmp, exe, ch, x, NBLOCKS)
    assert int(meta["commands"]) == 5 and int(meta["shutdowns"]) == 1 and int(meta["master_jobs"]) == NBLOCKS
    print("mini-radiod A/B on the device:", s)
    assert s["data"] > 1000 and s["null"] > 100


@pytest.mark.gpu
def test_reference_callers_on_the_mi355x_at_wall_clock_pace():
    """the same channel table with the front end on its own 20 ms clock (never waiting, as hardware): no channel thread of the
    reference's may be lapped, so the frames are those of the lock-step run"""
    exe = _hip_exe()
    ch = mr.standard_channels()
    nb = 100
    x = mr.synthesise(ch, FS, L, nb, seed=6)
    with tempfile.TemporaryDirectory() as tmp:
        s, B, meta = _ab(tmp, exe, ch, x, nb, paced=1)
    assert all(f["block_drops"] == 0 for F in B.values() for f in F)
    assert float(meta["seconds"]) < nb * 0.02 + 0.5
    print("mini-radiod paced A/B on the device:", s)


@pytest.mark.gpu
def test_baseline_config3_through_the_reference_callers_on_the_mi355x():
    """BASELINE config 3 -- 129.6 MS/s real (N = 3,240,000), 1024 mixed usb / cw / iq 12 kHz channels with per-channel filters -- run by the
    reference's OWN radiod code on the device: 1024 real channel threads (the CW third through filter2 = 4), in lock step and then with the
    front end on its own 20 ms clock (no drops), every frame of every channel against the same objects on the reference's filter.c."""
    exe = _hip_exe()
    ch, lines = mr.config3_channels()
    fs, l, m = CONFIG3
    with tempfile.TemporaryDirectory() as tmp:
        x = mr.spectral_synth(lines, fs, 12 * l, 0.002, 7)
        s, B, meta = _ab(tmp, exe, ch, x, 12, geom=CONFIG3)
        print("mini-radiod config 3 A/B on the device:", s)
        assert int(meta["channels"]) == 1024 and s["frames_in_agreement"] == s["frames"]
    with tempfile.TemporaryDirectory() as tmp:
        nb = 24
        x = mr.spectral_synth(lines, fs, nb * l, 0.002, 8)
        s, B, meta = _ab(tmp, exe, ch, x, nb, geom=CONFIG3, paced=1)
        print("mini-radiod config 3 paced A/B on the device:", s)
        assert all(f["block_drops"] == 0 for F in B.values() for f in F) and s["frames_in_agreement"] == s["frames"]
        assert float(meta["seconds"]) < nb * 0.02 + 1.0


@pytest.mark.gpu
def test_baseline_config2_through_the_reference_callers_on_the_mi355x():
    """BASELINE config 2 -- 64.8 MS/s real (N = 1,620,000), 256 x 12 kHz NBFM channels -- run by the reference's own demod_fm() threads on the
    device: a frequency-modulated carrier on every fourth channel (squelch opens, discriminator, threshold extension on half of them,
    de-emphasis on the pm third), noise on the others (squelch stays shut)."""
    exe = _hip_exe()
    ch, lines = mr.config2_channels()
    fs, l, m = CONFIG2
    nb = 16
    with tempfile.TemporaryDirectory() as tmp:
        x = mr.spectral_synth(lines, fs, nb * l, 0.002, 9)
        s, B, meta = _ab(tmp, exe, ch, x, nb, geom=CONFIG2)
    print("mini-radiod config 2 A/B on the device:", s)
    assert int(meta["channels"]) == 256 and s["frames_in_agreement"] == s["frames"] and s["data"] > 600 and s["null"] > 2000


@pytest.mark.gpu
def test_baseline_config1_complex_front_end_through_the_reference_callers():
    """BASELINE config 1 -- sig_gen COMPLEX 2.4 MS/s (N = 60,000), one IQ-mode channel -- and a channel on the negative side of the complex
    spectrum: write_cfilter() front end, COMPLEX master, the reference's callers on the device against the same objects on the reference's filter.c."""
    exe = _hip_exe()
    ch, lines = mr.config1_channels()
    geom = (2.4e6, 48000, 12001)
    nb = 40
    with tempfile.TemporaryDirectory() as tmp:
        x = mr.complex_synth(lines, geom[0], nb * geom[1], 0.002, 3)
        s, B, meta = _ab(tmp, exe, ch, x, nb, geom=geom)
    print("mini-radiod config 1 (complex front end) A/B on the device:", s)
    assert s["frames_in_agreement"] == s["frames"] == 2 * nb and s["data"] == 2 * nb


@pytest.mark.gpu
def test_config4_shape_sharded_behind_one_master_through_the_reference_callers():
    """BASELINE config 4's shape as far as ONE radiod reaches: 2000 x 24 kHz channels (P = 600; Nchannels = 2000, src/radio.h:356) at 129.6 MS/s,
    2000 real channel threads, the slaves of the ONE master sharded by the drop-in over KA9Q_HIP_DEVICES=0,0 (1024 per shard; two engines on the
    one device of this box: distinct devices are unmeasured), in lock step and at wall-clock pace."""
    exe = _hip_exe()
    ch, lines = mr.config4_channels()
    env = {"KA9Q_HIP_DEVICES": "0,0"}
    with tempfile.TemporaryDirectory() as tmp:
        x = mr.spectral_synth(lines, CONFIG3[0], 8 * CONFIG3[1], 0.002, 11)
        s, B, meta = _ab(tmp, exe, ch, x, 8, geom=CONFIG3, env=env)
        print("mini-radiod config 4 shape, two shards, A/B on the device:", s)
        assert int(meta["channels"]) == 2000 and s["frames_in_agreement"] == s["frames"] == 2000 * 8
    with tempfile.TemporaryDirectory() as tmp:
        nb = 20
        x = mr.spectral_synth(lines, CONFIG3[0], nb * CONFIG3[1], 0.002, 12)
        s, B, meta = _ab(tmp, exe, ch, x, nb, geom=CONFIG3, env=env, paced=1)
        print("mini-radiod config 4 shape, two shards, paced A/B on the device:", s)
        assert all(f["block_drops"] == 0 for F in B.values() for f in F) and s["frames_in_agreement"] == s["frames"]


@pytest.mark.gpu
def test_channels_joining_and_leaving_on_the_mi355x():
    """radiod's dynamic channels on the device: eight channels' lifetimes run out mid-stream (the reference's close_chan() ->
    delete_filter_output while 50 others run), ten are created mid-stream by another thread (create_filter_output on a running master: the
    drop-in registers the slave, grows and warms the bank between two blocks), every frame against the reference link"""
    exe = _hip_exe()
    ch = mr.churn_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS, seed=21)
    with tempfile.TemporaryDirectory() as tmp:
        s, B, meta = _ab(tmp, exe, ch, x, NBLOCKS)
    _check_churn(B, NBLOCKS)
    print("mini-radiod joining / leaving A/B on the device:", s)
    assert int(meta["channels"]) == 58 and s["frames_in_agreement"] == s["frames"]
    # ... and with the front end on its own 20 ms clock (which block a late-comer starts at is then a matter of wall-clock time, so nothing is compared with the
    # reference link): banks grow and are warmed, slaves register and leave while blocks go by every 20 ms -- nobody may be lapped
    nb = 100
    x = mr.synthesise(ch, FS, L, nb, seed=21)
    with tempfile.TemporaryDirectory() as tmp:
        B, meta, _ = mr.run(exe, os.path.join(tmp, "got"), ch, x, FS, L, M, nb, paced=1)
    assert sorted(B) == sorted(c.ssrc for c in ch) and all(f["block_drops"] == 0 for F in B.values() for f in F)
    assert int(meta["channels"]) == 58 and float(meta["seconds"]) < nb * 0.02 + 0.5


@pytest.mark.gpu
def test_spectrum_analysers_on_the_mi355x():
    """the reference's demod_spectrum() threads on the device: narrowband analysers whose slaves have 208, 520 and 60 samples per block (P = 261, 651
    and 76 bins: the any-length channel kernel), polled once per block, bin data against the reference link; a wideband analyser rides along
    (SPECTRUM slave: a block clock without a transform)"""
    exe = _hip_exe()
    ch = mr.spectrum_channels()
    x = mr.spectrum_signal(ch, FS, L, NBLOCKS)
    with tempfile.TemporaryDirectory() as tmp:
        A, _, _ = mr.run(mr.REF_EXE, os.path.join(tmp, "ref"), ch, x, FS, L, M, NBLOCKS)
        A32, _, _ = mr.run(mr.REF_EXE, os.path.join(tmp, "ref32"), ch, x, FS, L, M, NBLOCKS, env={"MINI_RADIOD_FFT_F32": "1"})
        B, meta, _ = mr.run(exe, os.path.join(tmp, "got"), ch, x, FS, L, M, NBLOCKS)
    (A, _), (A32, _), (B, Bw) = mr.split_wideband(A, ch), mr.split_wideband(A32, ch), mr.split_wideband(B, ch)
    _check_spectrum(B, Bw, NBLOCKS)
    d_self = mr.diff(A, A32)
    s = mr.check(mr.diff(A, B, upto={k: v["agree"] for k, v in d_self.items()}), d_self, pll=_pll_channels(ch), n0_flip=2e-3)      # (n0_flip: see _ab)
    print("mini-radiod spectrum analysers A/B on the device:", s, {k: v["float_rel"] for k, v in mr.diff(A, B).items() if k >= 500})
    assert int(meta["channels"]) == 16 and s["frames_in_agreement"] == s["frames"]


@pytest.mark.gpu
def test_wfm_stereo_decoder_on_the_mi355x():
    """the reference's demod_wfm() threads on the device: the 384 kHz channel (P = 9600) off a 2.592 MS/s front end, and every WFM channel's own REAL
    master (N = 15,360, 2:1 overlap) with its mono (REAL), pilot and L - R (COMPLEX, shifted) slaves -- four engines in one process; stereo, mono,
    pilot-less and empty channels against the reference link (the first four frames discretely only: mini_radiod_lib.diff says why)"""
    exe = _hip_exe()
    ch = mr.wfm_channels()
    fs, l, m = mr.WFM_GEOM
    x = mr.synthesise(ch, fs, l, NBLOCKS, seed=41)
    with tempfile.TemporaryDirectory() as tmp:
        A, _, _ = mr.run(mr.REF_EXE, os.path.join(tmp, "ref"), ch, x, fs, l, m, NBLOCKS)
        A32, _, _ = mr.run(mr.REF_EXE, os.path.join(tmp, "ref32"), ch, x, fs, l, m, NBLOCKS, env={"MINI_RADIOD_FFT_F32": "1"})
        B, meta, _ = mr.run(exe, os.path.join(tmp, "got"), ch, x, fs, l, m, NBLOCKS)
    _check_wfm(B, NBLOCKS)
    d_self = mr.diff(A, A32, settle=WFM_SETTLE)
    d = mr.diff(A, B, upto={k: v["agree"] for k, v in d_self.items()}, settle=WFM_SETTLE)
    n0 = {k: sorted(abs(a["n0"] - b["n0"]) / a["n0"] for a, b in zip(A[k], B[k])) for k in (600, 601, 602, 603)}
    print("n0 per frame, relative to the reference link: median / three largest", {k: (v[len(v) // 2], v[-3:]) for k, v in n0.items()})
    # estimate_noise() over a 220 kHz channel = 5500 bins: ONE bin crossing its threshold (src/radio.c:1840-1864) moves n0 by up to 1 / (bins averaged) ~ 2e-4;
    # on this input the reference happens not to flip against itself, the device does on a few frames -- the per-frame medians stay at float32's 1e-6
    assert all(v[len(v) // 2] < 1e-5 for v in n0.values())
    s = mr.check(d, d_self, pll=_pll_channels(ch), n0_flip=2e-3)
    print("mini-radiod WFM stereo A/B on the device:", s, {k: v["float_rel"] for k, v in d.items() if k >= 600}, "seconds", meta["seconds"])
    assert int(meta["channels"]) == 12 and s["frames_in_agreement"] == s["frames"]
    # the same channels with the front end on its own 20 ms clock: a WFM channel thread makes four dependent trips to the device per block
    # (its 384 kHz block; the composite transform + mono; pilot; L - R) and must not be lapped
    nb = 100
    x = mr.synthesise(ch, fs, l, nb, seed=42)
    with tempfile.TemporaryDirectory() as tmp:
        A, _, _ = mr.run(mr.REF_EXE, os.path.join(tmp, "ref"), ch, x, fs, l, m, nb)
        B, meta, err = mr.run(exe, os.path.join(tmp, "got"), ch, x, fs, l, m, nb, paced=1)
    _check_wfm(B, nb)
    d = mr.diff(A, B, settle=WFM_SETTLE)
    print("mini-radiod WFM stereo paced on the device:", {k: (v["agree"], v["frames"], v["float_rel"]) for k, v in d.items() if k >= 600}, "seconds", meta["seconds"])
    assert all(f["block_drops"] == 0 for F in B.values() for f in F) and all(v["agree"] == v["frames"] for v in d.values())
    assert max(v["float_rel"] for k, v in d.items() if k >= 600) < 1e-5 and float(meta["seconds"]) < nb * 0.02 + 0.5


@pytest.mark.gpu
def test_small_complex_front_end_through_the_reference_callers():
    """a 192 kHz COMPLEX front end (a Funcube dongle: N = 4800, small enough to be taken for a filter2 at create_filter_input) with six channels of the reference's
    own threads on the device, in lock step and at wall-clock pace: the front-end master becomes an engine in place with its first slave, the CW channel's
    filter2 = 4 stays a pooled inline master"""
    exe = _hip_exe()
    ch, lines = mr.funcube_channels()
    geom = mr.FUNCUBE_GEOM
    for nb, paced in ((40, 0), (100, 1)):
        x = mr.complex_synth(lines, geom[0], nb * geom[1], 0.002, 5 + paced)
        with tempfile.TemporaryDirectory() as tmp:
            s, B, meta = _ab(tmp, exe, ch, x, nb, geom=geom, paced=paced)
        print("mini-radiod small complex front end%s A/B on the device:" % (" paced" if paced else ""), s)
        assert int(meta["channels"]) == 7 and s["frames_in_agreement"] == s["frames"] == 6 * nb + nb // 4
        assert all(f["block_drops"] == 0 for F in B.values() for f in F)
    ch, lines = mr.beam_handover_channels()             # the beam channel leaves mid-stream and a plain channel takes its bank index
    x = mr.complex_synth(lines, geom[0], 32 * geom[1], 0.002, 5)
    with tempfile.TemporaryDirectory() as tmp:
        s, B, meta = _ab(tmp, exe, ch, x, 32, geom=geom)
    print("mini-radiod beam channel leaving A/B on the device:", s)
    assert int(meta["channels"]) == 8 and s["frames_in_agreement"] == s["frames"] and len(B[4006]) == 10


@pytest.mark.gpu
def test_presets_and_sample_rates_changed_on_running_channels_on_the_mi355x():
    """`control`'s preset / sample-rate / demodulator changes on running channels through the reference's own restart path on the device: the channel thread deletes
    its filter output and creates one of another size (P 300 -> 600, 300 -> 600 with another demodulator, 300 -> 200) on the running master -- a new bank between two
    blocks --, and a CW preset brings a pooled filter2 master in mid-stream; every frame against the reference link"""
    exe = _hip_exe()
    ch = mr.switch_channels()
    x = mr.synthesise(ch, FS, L, NBLOCKS, seed=51)
    with tempfile.TemporaryDirectory() as tmp:
        s, B, meta = _ab(tmp, exe, ch, x, NBLOCKS)
    _check_switches(B, NBLOCKS)
    print("mini-radiod preset / sample-rate changes A/B on the device:", s)
    assert int(meta["commands"]) == 9 and s["frames_in_agreement"] == s["frames"] == 161


@pytest.mark.gpu
def test_sharded_master_under_churn_and_preset_changes_on_the_mi355x():
    """KA9Q_HIP_DEVICES=0,0 with seven slaves per shard at a time: joiners, leavers and restarted channels land on both engines of the one device while the reference's
    own threads run -- channels joining / leaving, presets and sample rates changed on running channels, the 48-channel table: every frame against the reference link"""
    exe = _hip_exe()
    env = {"KA9Q_HIP_DEVICES": "0,0", "KA9Q_HIP_SHARD_CHANNELS": "7"}
    for name, ch, seed in (("churn", mr.churn_channels(), 21), ("switch", mr.switch_channels(), 51), ("standard", mr.standard_channels(), 5)):
        x = mr.synthesise(ch, FS, L, NBLOCKS, seed=seed)
        with tempfile.TemporaryDirectory() as tmp:
            s, B, meta = _ab(tmp, exe, ch, x, NBLOCKS, env=env)
        assert s["frames_in_agreement"] == s["frames"], (name, s)
