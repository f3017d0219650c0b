"""The filter.h drop-in's HOST code (ka9q-radio_amd/csrc/filter_hip.c) under ThreadSanitizer and AddressSanitizer on the CPU.

On the GPU box the drop-in is tested end to end (tests/test_dropin.py).  What those runs cannot do is watch its host-side
concurrency -- job numbering, completion signalling from a foreign thread, the miss queue with its leader and followers, bank
growth while channels run, 100+ channel pthreads against one front-end thread -- with a race detector.  Here the SAME C source
is compiled with -fsanitize=thread (and =address) against tests/stub/chz_stub.cpp, a CPU stand-in for libchz_hip.so behind the
same C ABI whose arithmetic is the oracle's and whose entry points are asynchronous like the device's, and driven by the same
radiod-style harness (tests/c/dropin_harness.c).  Outputs are checked against the oracle as in the GPU test; the sanitizer must
stay silent.  Test infrastructure only: nothing here is part of, linked into or loaded by the product."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
from test_dropin import _check, ROOT, PKG
import struct

STUB = os.path.join(ROOT, "tests", "stub")
CSRC = os.path.join(PKG, "csrc")


def _have(flag):
    src = "int main(void){return 0;}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        return subprocess.run(["gcc", flag, os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], capture_output=True).returncode == 0


_BUILT = {}


def _build(san, out_dir):
    """one build per sanitizer and test session (the three compiles take ~5 s, and 30 tests ask for them)"""
    if san in _BUILT and os.path.exists(_BUILT[san]):
        return _BUILT[san]
    out_dir = tempfile.mkdtemp(prefix="chz_stub_%s_" % san)
    _BUILT[san] = _build_once(san, out_dir)
    return _BUILT[san]


def _build_once(san, out_dir):
    ol.build()
    os.makedirs(out_dir, exist_ok=True)
    flag = "-fsanitize=" + san
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", flag, os.path.join(STUB, "chz_stub.cpp"), "-o", os.path.join(out_dir, "libchz_hip.so"),
                    "-L", os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread"], check=True)
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fPIC", "-shared", flag, "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-maybe-uninitialized",
                    "-DCHZ_EXPERIMENTS=1", os.path.join(CSRC, "filter_hip.c"), "-o", os.path.join(out_dir, "libka9q_filter_hip.so"),
                    "-L", out_dir, "-lchz_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"], check=True)
    exe = os.path.join(out_dir, "harness")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", flag, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "dropin_harness.c"), "-o", exe,
                    "-L", out_dir, "-lka9q_filter_hip", "-lchz_hip", "-Wl,-rpath," + out_dir, "-lpthread", "-lm"], check=True)
    return exe


def _run(exe, tmp, L, M, olen, plan, nblocks, x, env=None):
    open(os.path.join(tmp, "cfg.txt"), "w").write("%d %d %d %d %d %d %d\n" % (L, M, ol.REAL, olen, len(plan), nblocks, 4096))
    with open(os.path.join(tmp, "plan.bin"), "wb") as f:
        for p in plan:
            f.write(struct.pack("iiiiddddd", *p))
    x.tofile(os.path.join(tmp, "in.bin"))
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=67")
    e.update(env or {})
    r = subprocess.run([exe, tmp], capture_output=True, text=True, timeout=int(os.environ.get("STUB_TIMEOUT", "300")), env=e)
    return r


def _plan(rng, n):
    plan = []
    for i in range(n):
        shift = int(rng.integers(-12000, 12000))
        plan.append((shift, shift, 10 ** 6, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4))
    plan[0] = (2500, 2600, 3, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4)            # retune at block 3
    plan[1] = (2501, 2501, 10 ** 6, 2, 0.004, 0.25, 11.0, -0.02, 0.02)         # new filter at block 2
    plan[2] = (-7000, 7000, 5, 4, -0.4, 0.4, 11.0, 0.1, 0.3)                   # both
    return plan


@pytest.mark.parametrize("san", ["thread", "address"])
@pytest.mark.parametrize("env", [None, {"HARNESS_RETUNE_MOD": "1"}, {"KA9Q_HIP_WAKE_SHARDS": "3"}], ids=["steady", "retune_every_block", "three_wake_shards"])
def test_dropin_host_code_under_sanitizers(tmp_path, san, env):
    if not _have("-fsanitize=" + san):
        pytest.skip("no -fsanitize=%s runtime in this image" % san)
    exe = _build(san, str(tmp_path / "build"))
    L, M, olen, P = 25920, 6481, 240, 300
    nblocks, nch = 8, 96
    rng = np.random.default_rng(8)
    g = ol.SigGen(100020.0 / 1.296e6, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    x = g.generate(nblocks * L)
    plan = _plan(rng, nch)
    if env and "HARNESS_RETUNE_MOD" in env:
        plan = [(p[0], p[0] + 40 + i, 10 ** 6, 10 ** 6) + p[4:] for i, p in enumerate(plan)]      # two shifts to alternate between
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    r = _run(exe, run_dir, L, M, olen, plan, nblocks, x, env)
    report = r.stderr
    assert "WARNING: ThreadSanitizer" not in report and "ERROR: AddressSanitizer" not in report and "LeakSanitizer" not in report, report[-6000:]
    assert r.returncode == 0, (r.returncode, report[-3000:])
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, len(plan), olen)
    spec = np.fromfile(os.path.join(run_dir, "spec.bin"), np.complex64)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    _check(L, M, olen, P, plan, nblocks, out, spec, dict(zip(meta[::2], meta[1::2])), x, retune_mod=1 if env and "HARNESS_RETUNE_MOD" in env else 0)


@pytest.mark.parametrize("san", ["thread", "address"])
@pytest.mark.parametrize("env", [{"HARNESS_FILTER2": "1 -0.1 0.1 7.0 3"}, {"HARNESS_FILTER2": "4 -0.2 0.2 7.0 -1"}, {"HARNESS_REAL": "40 0.0 0.3 5.0"},
                                 {"HARNESS_ISB": "5", "HARNESS_RETUNE_MOD": "2"}, {"HARNESS_CHURN_MOD": "3", "HARNESS_RETUNE_MOD": "2"}],
                         ids=["filter2_blocking1_isb", "filter2_blocking4", "real_slave", "isb_and_retunes", "channels_leaving_and_joining"])
def test_dropin_other_paths_are_sanitizer_clean(tmp_path, san, env):
    """filter2's pooled inline masters (leader / follower batching), a REAL-output slave next to the COMPLEX ones, ISB flags flipped
    by the caller, retunes every other block: the numeric side of these paths is checked on the GPU (tests/test_dropin.py); here
    the host code must run them to the end with the race detector and the address / leak checker silent."""
    if not _have("-fsanitize=" + san):
        pytest.skip("no -fsanitize=%s runtime in this image" % san)
    exe = _build(san, str(tmp_path / "build"))
    L, M, olen = 25920, 6481, 240
    nblocks, nch = 8, 48
    rng = np.random.default_rng(9)
    g = ol.SigGen(100020.0 / 1.296e6, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    x = g.generate(nblocks * L)
    plan = _plan(rng, nch)
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    r = _run(exe, run_dir, L, M, olen, plan, nblocks, x, env)
    report = r.stderr
    assert "WARNING: ThreadSanitizer" not in report and "ERROR: AddressSanitizer" not in report and "LeakSanitizer" not in report, report[-6000:]
    assert r.returncode == 0, (r.returncode, report[-3000:])
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    assert int(meta["clock"]) == nblocks
    if "HARNESS_CHURN_MOD" not in env:
        assert meta["drops"] == "0"


def _complex_front_end(tmp_path, san, L, M, fs, nblocks=6, nch=24, env=None):
    if not _have("-fsanitize=" + san):
        pytest.skip("no -fsanitize=%s runtime in this image" % san)
    exe = _build(san, str(tmp_path / "build"))
    olen = 240
    N = L + M - 1
    P = olen * N // L                                             # 300 at overlap 5, 480 at overlap 2
    rng = np.random.default_rng(12)
    g = ol.SigGen(100020.0 / 2.4e6, 0.1, 0.01, ol.scale_ad(False, 1), False, seed=1)
    x = g.generate(nblocks * L)                                   # complex64
    reach = N // 2 - 400
    plan = [(int(rng.integers(-reach, reach)),) * 2 + (10 ** 6, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4) for _ in range(nch)]
    plan[0] = (0, 0, 10 ** 6, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4)               # the IQ channel at DC
    plan[1] = (N // 2 - 10, -(N // 2 - 10), 3, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4)   # across the +-Nyquist seam, retuned at block 3
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    open(os.path.join(run_dir, "cfg.txt"), "w").write("%d %d %d %d %d %d %d\n" % (L, M, ol.COMPLEX, olen, len(plan), nblocks, 4096))
    with open(os.path.join(run_dir, "plan.bin"), "wb") as f:
        for p in plan:
            f.write(struct.pack("iiiiddddd", *p))
    np.ascontiguousarray(x, np.complex64).tofile(os.path.join(run_dir, "in.bin"))
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=67")
    e.update(env or {})
    r = subprocess.run([exe, run_dir], capture_output=True, text=True, timeout=int(os.environ.get("STUB_TIMEOUT", "300")), env=e)
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-5000:]
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, nch, olen)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    assert meta["drops"] == "0" and int(meta["next_jobnum"]) == nblocks and int(meta["points"]) == N
    st = ol.Stream(L, M, ol.COMPLEX)
    state = np.zeros(2)
    for b in range(nblocks):
        s64 = st.push(x[b * L:(b + 1) * L], f64=True)
        dc = s64[:1].astype(np.complex64); ol.notch(state, [0], 0.01, dc); s64[0] = dc[0]
        for i, p in enumerate(plan):
            shift = p[1] if b >= p[2] else p[0]
            resp = ol.set_filter(P, olen, N, False, p[4], p[5], p[6])
            want = ol.channel(s64, ol.COMPLEX, P, olen, shift, resp)
            err = float(np.sqrt(np.mean(np.abs(out[b, i] - want) ** 2))); rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
            assert err <= 1e-5 * rms + 2e-8 * float(np.abs(s64).max()) * float(np.linalg.norm(resp)), (b, i, err, rms)


@pytest.mark.parametrize("san", ["thread", "address"])
def test_dropin_complex_master_under_sanitizers(tmp_path, san):
    """BASELINE config 1's geometry (2.4 MS/s complex front end: L = 48000, M = 12001, one IQ-mode channel among others) through the
    drop-in's host code with write_cfilter, under the sanitizers; outputs against the oracle."""
    _complex_front_end(tmp_path, san, 48000, 12001, 2.4e6)


@pytest.mark.parametrize("san", ["thread", "address"])
@pytest.mark.parametrize("geom", [(3840, 961), (5120, 1281), (3840, 3841)], ids=["funcube_192k", "airspyhf_256k", "funcube_overlap2"])
def test_dropin_small_complex_front_end_under_sanitizers(tmp_path, san, geom):
    """[r6] a COMPLEX front end small enough to look like radiod's filter2 (N <= 8192: a Funcube dongle at 192 kHz is N = 4800, an Airspy HF+ at 256 kHz
    N = 6400; at overlap 2 even M = L + 1 as filter2 has it): create_filter_input cannot tell, the master starts undecided and becomes a full engine IN
    PLACE when its first 12 kHz slave is created or its first block arrives, whichever of the front-end thread and the 24 channel threads comes first
    (rounds 2-5 refused every channel of such a front end).  Under the sanitizers, outputs against the oracle from block 0 on."""
    _complex_front_end(tmp_path, san, geom[0], geom[1], 192e3)


@pytest.mark.parametrize("san", ["thread", "address"])
@pytest.mark.parametrize("seed", [1])
def test_dropin_random_traffic_under_sanitizers(tmp_path, san, seed):
    """tests/c/dropin_fuzz.c: 24 threads doing at random what radiod's channel threads do to their slaves -- create (three output
    types, four sizes), set_filter, execute with old and new shifts, flip isb, delete and re-create, fall behind -- against a master
    that is being fed all the while.  It must end, every call must succeed, the sanitizers must stay silent."""
    if not _have("-fsanitize=" + san):
        pytest.skip("no -fsanitize=%s runtime in this image" % san)
    out_dir = os.path.dirname(_build(san, str(tmp_path / "build")))
    exe = str(tmp_path / "fuzz")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=" + san, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "dropin_fuzz.c"),
                    "-o", exe, "-L", out_dir, "-lka9q_filter_hip", "-lchz_hip", "-Wl,-rpath," + out_dir, "-lpthread", "-lm"], check=True)
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=67")
    r = subprocess.run([exe, "24", "150", str(seed)], capture_output=True, text=True, timeout=int(os.environ.get("STUB_TIMEOUT", "600")), env=e)
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-2000:])
    assert "failed 0" in r.stdout


@pytest.mark.parametrize("san", ["thread", "address"])
def test_dropin_drop_mode_and_device_noise_under_sanitizers(tmp_path, san):
    """KA9Q_HIP_INPUT_FULL=drop with a device that takes 60 ms per block under a front end that delivers one every 10 ms and never waits: the producer must not
    block; blocks that found their slot still busy are skipped -- every channel gets zeros and a counted drop for them, as a lapped
    slave does in the reference (src/filter.c:690-701) -- and every block that WAS transformed is exact, because the skipped blocks'
    samples still reached the overlap history.  The device-side estimate_noise() (include/ka9q_filter_hip_ext.h) rides along: what
    filter_hip_noise() hands a channel is the estimate of the block it just received."""
    if not _have("-fsanitize=" + san):
        pytest.skip("no -fsanitize=%s runtime in this image" % san)
    exe = _build(san, str(tmp_path / "build"))
    L, M, olen, P = 25920, 6481, 240, 300
    N = L + M - 1
    nblocks, nch = 18, 20
    fs = 1.296e6
    rng = np.random.default_rng(31)
    g = ol.SigGen(100020.0 / fs, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    x = g.generate(nblocks * L)
    plan = [(int(rng.integers(-12000, 12000)),) * 2 + (10 ** 6, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4) for _ in range(nch)]
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    env = {"KA9Q_HIP_INPUT_FULL": "drop", "CHZ_STUB_FORWARD_DELAY_MS": "60", "HARNESS_FREE_RUN": "10000", "HARNESS_NOISE": repr(fs)}
    r = _run(exe, run_dir, L, M, olen, plan, nblocks, x, env)
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, nch, olen)
    dropped = np.fromfile(os.path.join(run_dir, "dropped.bin"), np.uint8).reshape(nblocks, nch).astype(bool)
    noise = np.fromfile(os.path.join(run_dir, "noise.bin"), np.float64).reshape(nblocks, nch)
    was_skipped = np.fromfile(os.path.join(run_dir, "skipped.bin"), np.uint8).astype(bool)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    skipped = int(meta["skipped"])
    assert int(was_skipped.sum()) == skipped
    assert skipped >= 2 and int(meta["next_jobnum"]) == nblocks and int(meta["clock"]) == nblocks       # the front end ran ahead and never stood still
    assert int(meta["drops"]) == int(dropped.sum()) and dropped.sum() >= skipped * nch
    assert not dropped.all(axis=1).all() and not was_skipped.all()                                    # some blocks did get through
    st = ol.Stream(L, M, ol.REAL)
    state = np.zeros(2)
    checked = 0
    for b in range(nblocks):
        s64 = st.push(x[b * L:(b + 1) * L], f64=True)
        if was_skipped[b]:
            assert dropped[b].all() and not out[b].any()                                              # zeros and a drop for everybody
            continue                                                                                  # (the notch state did not see this block either)
        dc = s64[:1].astype(np.complex64); ol.notch(state, [0], 0.01, dc); s64[0] = dc[0]
        s32 = s64.astype(np.complex64)
        for i, p in enumerate(plan):
            if dropped[b, i]:
                assert not out[b, i].any()
                continue
            want = ol.channel(s64, ol.REAL, P, olen, p[0], ol.set_filter(P, olen, N, True, p[4], p[5], p[6]))
            err = float(np.sqrt(np.mean(np.abs(out[b, i] - want) ** 2))); rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
            assert err <= 1e-5 * rms + 2e-8 * float(np.abs(s64).max()), (b, i, err, rms)
            assert noise[b, i] == pytest.approx(ol.estimate_noise(s32, ol.REAL, P, p[0], fs), rel=1e-6), (b, i)
            checked += 1
    assert checked >= nch


def _recovery_run(tmp_path, san, env, nblocks=16, nch=20):
    exe = _build(san, str(tmp_path / "build"))
    L, M, olen = 25920, 6481, 240
    rng = np.random.default_rng(77)
    g = ol.SigGen(100020.0 / 1.296e6, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    x = g.generate(nblocks * L)
    plan = []
    for i in range(nch):
        sh = int(rng.integers(500, 12000)) * (1 if i % 2 else -1)
        plan.append((sh, sh + (37 if sh > 0 else -37), 10 ** 6, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4))     # (the second shift: HARNESS_RETUNE_MOD)
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    r = _run(exe, run_dir, L, M, olen, plan, nblocks, x, dict(env, HARNESS_RECORD_DROPS="1"))
    return r, run_dir, x, plan


@pytest.mark.parametrize("san", ["thread", "address"])
@pytest.mark.parametrize("retune", [0, 2], ids=["steady", "channels_retuning_every_other_block"])
def test_dropin_replaces_a_failed_engine_and_carries_on(tmp_path, san, retune):
    """Failure policy (round 4): the engine reports a failed device-side check from block 5 on (injected into the stand-in engine, as a
    notch ticket that ran out would).  The drop-in drops what that engine still delivers (zeros + block_drops for every slave),
    replaces the engine ONCE -- responses, shifts, notch list re-registered, the overlap history re-seated from the host ring --
    and the stream continues within 8 blocks, exact again from the first block of the new engine; one line of log, no error per block."""
    if not _have("-fsanitize=" + san):
        pytest.skip("no -fsanitize=%s runtime in this image" % san)
    nblocks, nch = 16, 20
    # (retune: half of the channels change their shift every other block, so misses are being served while the engine dies: those become
    #  drops of the lost blocks too, never an error return)
    r, run_dir, x, plan = _recovery_run(tmp_path, san, dict({"CHZ_STUB_FAIL_JOB": "5"}, **({"HARNESS_RETUNE_MOD": str(retune)} if retune else {})), nblocks, nch)
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert r.stderr.count("re-creating the engine") == 1 and r.stderr.count("execute_filter_input:") <= 1, r.stderr[-2000:]
    L, M, olen, P = 25920, 6481, 240, 300
    N = L + M - 1
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, nch, olen)
    dropped = np.fromfile(os.path.join(run_dir, "dropped.bin"), np.uint8).reshape(nblocks, nch).astype(bool)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    assert int(meta["clock"]) == nblocks and int(meta["next_jobnum"]) == nblocks
    lost = np.flatnonzero(dropped.all(axis=1))
    assert 1 <= len(lost) <= 8 and lost[0] >= 5 and np.array_equal(lost, np.arange(lost[0], lost[0] + len(lost))), lost   # one gap, recovered within 8 blocks
    # every lost block is a counted drop for everybody; a slow channel may also lose (and count) up to 3 blocks the old engine had completed but it had not fetched yet
    assert int(meta["drops"]) == int(dropped.sum()) and len(lost) * nch <= dropped.sum() <= (len(lost) + 3) * nch
    st = ol.Stream(L, M, ol.REAL)
    for b in range(nblocks):
        s64 = st.push(x[b * L:(b + 1) * L], f64=True)
        if b in lost:
            assert not out[b].any()
            continue
        for i, p in enumerate(plan):
            if dropped[b, i]:
                assert not out[b, i].any()
                continue
            shift = p[1] if (retune and ((b + i) // retune) & 1) else p[0]
            want = ol.channel(s64, ol.REAL, P, olen, shift, ol.set_filter(P, olen, N, True, p[4], p[5], p[6]))     # (channels away from the DC notch)
            err = float(np.sqrt(np.mean(np.abs(out[b, i] - want) ** 2))); rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
            assert err <= 1e-5 * rms + 2e-8 * float(np.abs(s64).max()), (b, i, err, rms)


def test_dropin_sharded_master_replaces_its_engines_when_one_device_fails(tmp_path):
    """KA9Q_HIP_DEVICES with a failure: one of the two (stand-in) devices reports a failed check from block 5 on.  The block is published
    by the LAST device's callback, so the healthy device's results of a failed block must not be handed out: every slave -- on either
    device -- gets zeros and a counted drop for the blocks the failure cost, BOTH engines are replaced once, and the stream is exact again."""
    if not _have("-fsanitize=thread"):
        pytest.skip("no -fsanitize=thread runtime in this image")
    nblocks, nch = 16, 20
    r, run_dir, x, plan = _recovery_run(tmp_path, "thread", {"CHZ_STUB_FAIL_JOB": "5", "CHZ_STUB_DEVICES": "2", "KA9Q_HIP_DEVICES": "0,1", "KA9Q_HIP_SHARD_CHANNELS": "10"}, nblocks, nch)
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert r.stderr.count("re-creating the engine") == 1, r.stderr[-2000:]
    L, M, olen, P = 25920, 6481, 240, 300
    N = L + M - 1
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, nch, olen)
    dropped = np.fromfile(os.path.join(run_dir, "dropped.bin"), np.uint8).reshape(nblocks, nch).astype(bool)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    assert int(meta["clock"]) == nblocks and int(meta["devices"]) == 2 and meta["dev_counts"] == "10:10"
    lost = np.flatnonzero(dropped.all(axis=1))
    assert 1 <= len(lost) <= 8 and lost[0] >= 5 and np.array_equal(lost, np.arange(lost[0], lost[0] + len(lost))), lost
    assert int(meta["drops"]) == int(dropped.sum()) and len(lost) * nch <= dropped.sum() <= (len(lost) + 3) * nch
    st = ol.Stream(L, M, ol.REAL)
    for b in range(nblocks):
        s64 = st.push(x[b * L:(b + 1) * L], f64=True)
        for i, p in enumerate(plan):
            if dropped[b, i]:
                assert not out[b, i].any()
                continue
            want = ol.channel(s64, ol.REAL, P, olen, p[0], ol.set_filter(P, olen, N, True, p[4], p[5], p[6]))
            err = float(np.sqrt(np.mean(np.abs(out[b, i] - want) ** 2))); rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
            assert err <= 1e-5 * rms + 2e-8 * float(np.abs(s64).max()), (b, i, err, rms)


def test_dropin_second_failure_exits_for_the_supervisor(tmp_path):
    """... and if the replacement fails as well (within 500 blocks), the process ends with EX_SOFTWARE (70) -- the reference's own answer
    to a fatal front-end / FFT error (src/radio.c:398, src/main.c:202): systemd restarts radiod."""
    if not _have("-fsanitize=address"):
        pytest.skip("no -fsanitize=address runtime in this image")
    r, _, _, _ = _recovery_run(tmp_path, "address", {"CHZ_STUB_FAIL_JOB": "4", "CHZ_STUB_FAIL_ALWAYS": "1", "ASAN_OPTIONS": "detect_leaks=0"}, 24, 8)
    assert r.returncode == 70, (r.returncode, r.stderr[-2000:])
    assert "second device failure" in r.stderr and "supervisor" in r.stderr and r.stderr.count("re-creating the engine") == 1


def test_dropin_recovers_when_the_failed_device_delivers_no_more_callbacks(tmp_path):
    """After a sticky device error the runtime delivers no more stream callbacks: the blocks in flight never complete, and a producer
    asleep on their slot would never reach the recovery (round 4's advisor finding).  The stand-in fails at block 5 AND swallows every
    callback from then on; the front end runs at its own pace and never waits for a channel.  The producer's watchdog (250 ms) must
    notice the failed check, replace the engine once, announce the blocks whose callbacks never came as dropped (zeros + a counted drop
    for every slave -- nobody is left asleep), and the stream must be exact again afterwards."""
    if not _have("-fsanitize=thread"):
        pytest.skip("no -fsanitize=thread runtime in this image")
    nblocks, nch = 20, 12
    r, run_dir, x, plan = _recovery_run(tmp_path, "thread", {"CHZ_STUB_FAIL_JOB": "5", "CHZ_STUB_LOSE_CALLBACKS": "1", "HARNESS_PACED_US": "20000",
                                                            "CHZ_STUB_FORWARD_DELAY_MS": "60"}, nblocks, nch)       # (a device 3 x slower than the stream: the producer is asleep on a busy slot when the failure is raised)
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert r.stderr.count("re-creating the engine") == 1, r.stderr[-2000:]
    L, M, olen, P = 25920, 6481, 240, 300
    N = L + M - 1
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, nch, olen)
    dropped = np.fromfile(os.path.join(run_dir, "dropped.bin"), np.uint8).reshape(nblocks, nch).astype(bool)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    assert int(meta["clock"]) == nblocks and int(meta["next_jobnum"]) == nblocks       # everybody got to the end: nobody was left asleep on a lost block
    # (a paced front end that never waits: WHICH call of a slave saw a drop is not the block number -- the counts and the delivered data are checked)
    assert int(meta["drops"]) == int(dropped.sum()) and nch <= dropped.sum() <= 16 * nch, dropped.sum(axis=1)
    st = ol.Stream(L, M, ol.REAL)
    wants = []
    for b in range(nblocks):
        s64 = st.push(x[b * L:(b + 1) * L], f64=True)
        wants.append([ol.channel(s64, ol.REAL, P, olen, p[0], ol.set_filter(P, olen, N, True, p[4], p[5], p[6])) for p in plan])
    exact = 0
    for i in range(nch):
        # every call's output is either zeros (a counted drop) or EXACTLY some block of the stream, in order
        b = 0
        for call in range(nblocks):
            if dropped[call, i]:
                assert not out[call, i].any()
                continue
            while b < nblocks:
                want = wants[b][i]
                err = float(np.sqrt(np.mean(np.abs(out[call, i] - want) ** 2))); rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
                b += 1
                if err <= 1e-5 * rms + 1e-9:
                    exact += 1
                    break
            else:
                raise AssertionError("channel %d call %d: output matches no later block of the stream" % (i, call))
    assert exact >= nch * (nblocks - 16)


def test_dropin_exits_when_the_device_is_wedged_without_an_error(tmp_path):
    """... and a device that neither completes anything nor reports anything: no recovery is possible from inside the process --
    after KA9Q_HIP_WEDGED_MS (default 10 s; 1.5 s here) the producer ends the process with EX_SOFTWARE through _exit, with its locks
    held and a dozen channel threads asleep (the reference's fatal path, src/main.c:202)."""
    exe = _build_plain(str(tmp_path / "build"))
    L, M, olen = 25920, 6481, 240
    x = np.zeros(12 * L, np.float32)
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    r = _run(exe, run_dir, L, M, olen, _plan(np.random.default_rng(1), 12), 12, x, {"CHZ_STUB_WEDGE_JOB": "4", "KA9Q_HIP_WEDGED_MS": "1500", "HARNESS_PACED_US": "20000"})
    assert r.returncode == 70, (r.returncode, r.stderr[-1500:])
    assert "has not completed a block" in r.stderr and "supervisor" in r.stderr


@pytest.mark.parametrize("ndev,san,exchange", [(3, "thread", "samples"), (2, "address", "samples"), (3, "thread", "broadcast"), (2, "address", "broadcast")])
def test_dropin_sharded_over_fake_devices(tmp_path, ndev, san, exchange):
    """KA9Q_HIP_DEVICES: ONE master behind filter.h, its slaves spread over 2 and 3 (stand-in) devices -- BASELINE config 4's shape
    scaled down (24 kHz channels, P = 600, a disjoint contiguous block of channels per device), driven by the radiod-style C harness.
    Every device transforms the block's samples itself; a block is complete when the last device's callback has run; retunes, a new
    filter and channels leaving / joining cross device boundaries.  Outputs of EVERY channel against the oracle, the host-visible
    fdomain[] from device 0, zero drops, the slaves where SURVEY 8e puts them (creation order, KA9Q_HIP_SHARD_CHANNELS at a time);
    ThreadSanitizer / AddressSanitizer silent."""
    if not _have("-fsanitize=" + san):
        pytest.skip("no -fsanitize=%s runtime in this image" % san)
    exe = _build(san, str(tmp_path / "build"))
    L, M, olen, P = 25920, 6481, 480, 600
    nblocks, nch = 8, 24 * ndev
    rng = np.random.default_rng(80 + ndev)
    g = ol.SigGen(100020.0 / 1.296e6, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    x = g.generate(nblocks * L)
    plan = _plan(rng, nch)
    # exchange: every device transforms the samples itself (default), or device 0 transforms and its spectrum slot is handed to the
    # others (KA9Q_HIP_EXCHANGE=broadcast: in-process grouped ncclBroadcast on the device; a queue hand-over in the stand-in)
    env = {"CHZ_STUB_DEVICES": str(ndev), "KA9Q_HIP_DEVICES": ",".join(str(i) for i in range(ndev)), "KA9Q_HIP_SHARD_CHANNELS": "24", "KA9Q_HIP_EXCHANGE": exchange}
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    r = _run(exe, run_dir, L, M, olen, plan, nblocks, x, env)
    report = r.stderr
    assert "WARNING: ThreadSanitizer" not in report and "ERROR: AddressSanitizer" not in report and "LeakSanitizer" not in report, report[-6000:]
    assert r.returncode == 0, (r.returncode, report[-3000:])
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, len(plan), olen)
    spec = np.fromfile(os.path.join(run_dir, "spec.bin"), np.complex64)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    assert int(meta["devices"]) == ndev and [int(v) for v in meta["dev_counts"].split(":")] == [24] * ndev
    _check(L, M, olen, P, plan, nblocks, out, spec, meta, x)


def test_dropin_sharded_refuses_a_device_that_is_not_there(tmp_path):
    exe = _build("address", str(tmp_path / "build"))
    L, M, olen = 25920, 6481, 240
    x = np.zeros(2 * L, np.float32)
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    r = _run(exe, run_dir, L, M, olen, _plan(np.random.default_rng(1), 4), 2, x, {"CHZ_STUB_DEVICES": "2", "KA9Q_HIP_DEVICES": "0,5"})
    assert r.returncode == 3 and "device 5" in r.stderr                # create_filter_input fails loudly, nothing runs on fewer devices than asked
    r = _run(exe, run_dir, L, M, olen, _plan(np.random.default_rng(1), 4), 2, x, {"CHZ_STUB_DEVICES": "2", "KA9Q_HIP_DEVICES": "0,0", "KA9Q_HIP_EXCHANGE": "broadcast"})
    # an RCCL clique cannot hold one device twice: round 6's ladder falls back to the sample exchange instead of refusing to start
    assert r.returncode == 0 and "listed twice" in r.stderr and "falling back to KA9Q_HIP_EXCHANGE=samples" in r.stderr, (r.returncode, r.stderr[-800:])


@pytest.mark.parametrize("fault", [{"CHZ_STUB_FAIL_COMM": "1"}, {"CHZ_STUB_FAIL_BCAST_CALL": "7"}], ids=["no_clique_at_start", "a_broadcast_fails_mid_stream"])
def test_dropin_exchange_ladder_broadcast_to_samples(tmp_path, fault):
    """SURVEY 8e behind filter.h, fail-safe (round 6): KA9Q_HIP_EXCHANGE=broadcast that cannot form its clique comes up with the sample
    exchange; a broadcast that FAILS mid-stream (an RCCL error out of the engine) takes the recovery path -- engines re-created, the blocks
    in flight counted as drops -- and the replacement engines exchange samples.  Three stand-in devices; every channel's output of the
    blocks after the recovery against the oracle."""
    exe = _build_plain(str(tmp_path / "build"))
    L, M, olen, P = 25920, 6481, 480, 600
    ndev, nblocks = 3, 16
    nch = 24 * ndev
    rng = np.random.default_rng(91)
    g = ol.SigGen(100020.0 / 1.296e6, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    x = g.generate(nblocks * L)
    plan = [(int(rng.integers(-12000, 12000)),) * 2 + (10 ** 6, 10 ** 6, -0.4, 0.4, 11.0, -0.4, 0.4) for _ in range(nch)]
    env = {"CHZ_STUB_DEVICES": str(ndev), "KA9Q_HIP_DEVICES": "0,1,2", "KA9Q_HIP_SHARD_CHANNELS": "24", "KA9Q_HIP_EXCHANGE": "broadcast", "HARNESS_RECORD_DROPS": "1"}
    env.update(fault)
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    r = _run(exe, run_dir, L, M, olen, plan, nblocks, x, env)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, nch, olen)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    if "CHZ_STUB_FAIL_COMM" in fault:
        assert "falling back to KA9Q_HIP_EXCHANGE=samples" in r.stderr and meta["drops"] == "0"
        first_good = 0
    else:
        assert "the new engines exchange samples instead" in r.stderr and int(meta["drops"]) > 0
        first_good = 12                                     # the overlap history is re-seated from the host ring: exact again well before this
    st = ol.Stream(L, M, ol.REAL)
    state = np.zeros(2)
    for b in range(nblocks):
        s64 = st.push(x[b * L:(b + 1) * L], f64=True)
        dc = s64[:1].astype(np.complex64); ol.notch(state, [0], 0.01, dc); s64[0] = dc[0]
        if b < first_good:
            continue
        for i, pl in enumerate(plan):
            if first_good and abs(pl[0]) <= P // 2 + 1:
                continue                    # a channel over the DC bin: the spur notch's state (alpha 0.01, a 100-block memory) restarted with the new engines
            resp = ol.set_filter(P, olen, L + M - 1, True, -0.4, 0.4, 11.0)
            want = ol.channel(s64, ol.REAL, P, olen, pl[0], resp)
            err = float(np.sqrt(np.mean(np.abs(out[b, i] - want) ** 2))); rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
            assert err <= 1e-4 * rms + 1e-9, (b, i, err, rms)


def _build_plain(out_dir):
    if "plain" in _BUILT and os.path.exists(_BUILT["plain"]):
        return _BUILT["plain"]
    out_dir = tempfile.mkdtemp(prefix="chz_stub_plain_")
    _BUILT["plain"] = _build_plain_once(out_dir)
    return _BUILT["plain"]


def _build_plain_once(out_dir):
    ol.build()
    os.makedirs(out_dir, exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", os.path.join(STUB, "chz_stub.cpp"), "-o", os.path.join(out_dir, "libchz_hip.so"),
                    "-L", os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread"], check=True)
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-maybe-uninitialized",
                    "-DCHZ_EXPERIMENTS=1", os.path.join(CSRC, "filter_hip.c"), "-o", os.path.join(out_dir, "libka9q_filter_hip.so"),
                    "-L", out_dir, "-lchz_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"], check=True)
    exe = os.path.join(out_dir, "harness")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "dropin_harness.c"), "-o", exe,
                    "-L", out_dir, "-lka9q_filter_hip", "-lchz_hip", "-Wl,-rpath," + out_dir, "-lpthread", "-lm"], check=True)
    return exe


@pytest.mark.parametrize("ndev,nch", [(3, 3072), (8, 8192)])
def test_dropin_config4_channel_count_sharded(tmp_path, ndev, nch):
    """BASELINE config 4's channel count behind filter.h: 8192 x 24 kHz channels (P = 600; the master scaled to 1.296 MS/s so that the CPU
    stand-in finishes), ONE master, 8192 channel pthreads, slaves spread over 8 stand-in devices (and 3840 over 3).  With 8 devices every device
    owns exactly one contiguous 1024-block of channels in creation order; with fewer the blocks fill up and the rest is
    balanced.  256 sampled channels (the first and last of every device's share among them) x 3 blocks against the oracle, zero drops."""
    exe = _build_plain(str(tmp_path / "build"))
    L, M, olen, P = 25920, 6481, 480, 600
    N = L + M - 1
    nblocks = 3
    rng = np.random.default_rng(4)
    g = ol.SigGen(100020.0 / 1.296e6, 0.1, 0.01, ol.scale_ad(True, 1), True, seed=1)
    x = g.generate(nblocks * L)
    plan = [(int(s), int(s), 10 ** 6, 10 ** 6, -10000 / 24000, 10000 / 24000, 11.0, -10000 / 24000, 10000 / 24000) for s in rng.integers(-15000, 15000, nch)]
    env = {"CHZ_STUB_DEVICES": str(ndev), "KA9Q_HIP_DEVICES": ",".join(str(i) for i in range(ndev)), "STUB_TIMEOUT": "600"}
    run_dir = str(tmp_path / "run"); os.makedirs(run_dir)
    os.environ["STUB_TIMEOUT"] = "600"
    try:
        r = _run(exe, run_dir, L, M, olen, plan, nblocks, x, env)
    finally:
        os.environ.pop("STUB_TIMEOUT", None)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    out = np.fromfile(os.path.join(run_dir, "out.bin"), np.complex64).reshape(nblocks, nch, olen)
    meta = open(os.path.join(run_dir, "meta.txt")).read().split()
    meta = dict(zip(meta[::2], meta[1::2]))
    counts = [int(v) for v in meta["dev_counts"].split(":")]
    assert int(meta["devices"]) == ndev and sum(counts) == nch and meta["drops"] == "0" and int(meta["clock"]) == nblocks
    if ndev == 8:
        assert counts == [1024] * 8
    else:
        assert min(counts) >= 1024 and max(counts) - min(counts) <= 1
    # (channel threads register in whatever order the scheduler starts them: WHICH slave sits where is not checked, every output is)
    edges = [0, nch - 1] + [k * 1024 + d for k in range(1, nch // 1024) for d in (-1, 0)]
    sample = sorted(set(edges) | set(int(v) for v in rng.choice(nch, 256 - len(edges), replace=False)))
    st = ol.Stream(L, M, ol.REAL)
    state = np.zeros(2)
    resp = ol.set_filter(P, olen, N, True, -10000 / 24000, 10000 / 24000, 11.0)
    for b in range(nblocks):
        s64 = st.push(x[b * L:(b + 1) * L], f64=True)
        dc = s64[:1].astype(np.complex64); ol.notch(state, [0], 0.01, dc); s64[0] = dc[0]
        peak = float(np.abs(s64).max())
        for i in sample:
            want = ol.channel(s64, ol.REAL, P, olen, plan[i][0], resp)
            err = float(np.sqrt(np.mean(np.abs(out[b, i] - want) ** 2)))
            rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
            assert err <= 1e-5 * rms + 2e-8 * peak * float(np.linalg.norm(resp)), (b, i, err, rms)
