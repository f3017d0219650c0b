"""The engine's HOST code on the CPU.

`ka9q-radio_amd/csrc/chz_engine.hip` -- lanes, issuing threads, per-slot descriptors, response-row recycling, the demodulator
stream, inline-master pools, every C-ABI entry point -- is compiled UNMODIFIED with g++ against tests/hipemu (the kernels run on the
fiber emulator, the HIP runtime calls on a synchronous stand-in, tests/hipemu/hip/hip_host_stub.h) into a test-only library behind
the same C ABI.  The parity tests that normally need an MI355X (tests/test_gpu_parity.py, test_gpu_pipeline.py, test_golden.py) are
then run against it in a child process: what they check on the device -- outputs against the oracle, through the C ABI -- they check
here for the engine's orchestration, without a GPU.  Not a fallback: the library carries a marker symbol and engine.py refuses to load
it unless CHZ_ALLOW_EMULATED_ENGINE=1 (set here and nowhere else)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu")
CSRC = os.path.join(ROOT, "ka9q-radio_amd", "csrc")
LIB = os.path.join(EMU, "libchz_hip_emu.so")

# what cannot run there: sizes that take minutes on the emulator, hipGraph capture, RCCL, tests that talk to libamdhip64 themselves or
# wait on the device-side ticket, and the two long demodulator scenarios (run by scripts/engine_emulated.sh)
SKIP = ("full_size or config3 or config2 or 2592000 or 1296000 or soak or rccl or comm_rendezvous or graph or runs_out or config4 or "
        "noise_and_conversion or beyond_the_lds or 400000 or 2600000 or 2500000 or coherent_modes or linear_demodulator_on_the_device or "
        "random_operations_on_the_device or null_stream or "
        # 75 s on the emulator for what tests/test_kernels_emulated.py::test_fm_loops_one_channel_per_lane_equal_one_lane_per_wavefront
        # pins at kernel level in 10 s (the engine-level comparison stays a GPU test)
        "fm_lane_passes_equal")


@pytest.fixture(scope="module")
def emulated_engine():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(EMU, "hip", f) for f in os.listdir(os.path.join(EMU, "hip"))]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-DHIPEMU", "-DHIPEMU_HOST", "-I", EMU, "-I", CSRC, "-x", "c++",
                        os.path.join(CSRC, "chz_engine.hip"), "-o", LIB, "-lpthread", "-ldl"], check=True)
    return LIB


def test_product_binding_refuses_the_emulated_library(emulated_engine):
    code = "import sys; sys.path.insert(0, %r); from conftest import load_pkg; load_pkg().engine.lib()" % os.path.join(ROOT, "tests")
    env = dict(os.environ, CHZ_LIB=emulated_engine)
    env.pop("CHZ_ALLOW_EMULATED_ENGINE", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing to use it as the product" in r.stderr


def test_engine_orchestration_on_the_emulator(emulated_engine):
    env = dict(os.environ, CHZ_LIB=emulated_engine, CHZ_ALLOW_EMULATED_ENGINE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_pipeline.py"),
                        os.path.join(ROOT, "tests", "test_golden.py"), "-m", "gpu", "-q", "-x", "--timeout", "300", "-p", "no:cacheprovider", "-k", "not (%s)" % SKIP,
                        "-n", str(max(1, min(6, (os.cpu_count() or 2) - 1)))],      # the tests are independent processes' worth of work: pytest-xdist
                       capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, (tail, r.stdout[-3000:], r.stderr[-1500:])
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 60, tail


def test_dropin_on_the_emulated_engine(emulated_engine, tmp_path):
    """The filter.h drop-in linked against the CPU build of the engine, driven by the radiod-style C harness (front-end thread, one
    pthread per channel, retunes, new filters, REAL slave, ISB, filter2's pooled inline masters, WFM-sized channels): the whole host
    path -- filter_hip.c on top of chz_engine.hip on top of the kernels -- against the oracle, without a GPU."""
    import shutil
    libdir = str(tmp_path / "lib")
    os.makedirs(libdir)
    shutil.copy(emulated_engine, os.path.join(libdir, "libchz_hip.so"))
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-maybe-uninitialized",
                    os.path.join(CSRC, "filter_hip.c"), "-o", os.path.join(libdir, "libka9q_filter_hip.so"), "-L", libdir, "-lchz_hip",
                    "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"], check=True)
    env = dict(os.environ, KA9Q_TEST_LIBDIR=libdir)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_dropin.py"), "-m", "gpu", "-q", "-x", "--timeout", "300", "-p", "no:cacheprovider",
                        "-k", "not (config3 or c_example or sharded or reference_header or runs_out or wall_clock or full_rate or clique or survives_exit or wider_than or packetd)"], capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, (tail, r.stdout[-3000:], r.stderr[-1500:])
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 6, tail


@pytest.mark.parametrize("seed", [1, 2])
def test_engine_random_operations(emulated_engine, seed):
    """tests/engine_fuzz_child.py: a model-based random walk over the engine's C ABI -- retunes, response swaps, ISB flags, active
    counts, banks destroyed and re-created, blocks run pipelined or one by one -- with the last block of every run checked against
    the oracle for a sample of channels of both banks."""
    env = dict(os.environ, CHZ_LIB=emulated_engine, CHZ_ALLOW_EMULATED_ENGINE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "engine_fuzz_child.py"), str(seed), "100"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "FUZZ ok" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])


def _build_driver(libpath, out, san=None):
    cmd = ["gcc", "-O1", "-g", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "engine_driver.c"), "-o", out,
           "-L", os.path.dirname(libpath), "-l:" + os.path.basename(libpath), "-Wl,-rpath," + os.path.dirname(libpath), "-lpthread"]
    if san:
        cmd.insert(1, "-fsanitize=" + san)
    subprocess.run(cmd, check=True)


def test_engine_threading_driver(emulated_engine, tmp_path):
    """tests/c/engine_driver.c from plain C: blocks pipelined from several issuing threads, the notch and demodulator hand-overs, a tuned
    bank with demodulators, retunes and response swaps between runs, inline masters executed from two threads -- must run to the end."""
    exe = str(tmp_path / "driver")
    _build_driver(emulated_engine, exe)
    for thr in ("1", "2", "4"):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, CHZ_ENQ_THREADS=thr))
        assert r.returncode == 0 and "driver ok blocks 32" in r.stdout, (thr, r.stdout[-300:], r.stderr[-1500:])


@pytest.mark.skipif(os.environ.get("CHZ_TEST_TSAN_ENGINE") != "1", reason="~6 min: set CHZ_TEST_TSAN_ENGINE=1 (TSAN=1 scripts/engine_emulated.sh does)")
def test_engine_threading_driver_under_thread_sanitizer(tmp_path):
    """The same driver with the ENGINE's host code (and the emulated kernels, whose fibers are announced to the race detector) built with
    -fsanitize=thread: 2 and 4 issuing threads.  Clean in round 2 after one finding (Bank::last_slot written by two issuing threads)."""
    lib = str(tmp_path / "libchz_hip_emu_tsan.so")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=thread", "-DHIPEMU", "-DHIPEMU_HOST", "-I", EMU, "-I", CSRC, "-x", "c++",
                    os.path.join(CSRC, "chz_engine.hip"), "-o", lib, "-lpthread", "-ldl"], check=True)
    exe = str(tmp_path / "driver_tsan")
    _build_driver(lib, exe, "thread")
    for thr in ("2", "4"):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=1500,
                           env=dict(os.environ, CHZ_ENQ_THREADS=thr, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66"))
        assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-5000:]
        assert r.returncode == 0 and "driver ok blocks 32" in r.stdout, (thr, r.stdout[-300:], r.stderr[-1500:])
