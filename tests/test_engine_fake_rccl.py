"""The engine's multi-GPU code with MORE THAN ONE rank, without a GPU.

bench.py --gpus N and a C host of BASELINE config 4 go through chz_comm_create + chz_run_blocks_sharded (ka9q-radio_amd/csrc/
chz_comm.inc).  On a one-GPU box that code only ever sees a world of one; the multi-rank data movement is otherwise covered by the
Python restatement in tests/test_distributed_gloo.py.  Here the real thing runs: the engine's host code compiled for the CPU
(tests/test_engine_emulated.py), RCCL replaced by an in-process stand-in whose ranks are threads (tests/stub/fake_rccl.cpp, bound
through CHZ_RCCL_LIB exactly as librccl would be), two and three ranks, the root with its two issuing threads, whole-slot broadcast
and row-range exchange, every rank's channels against the oracle."""
import os
import subprocess
import sys

import pytest

from test_engine_emulated import emulated_engine, ROOT, EMU  # noqa: F401

FAKE = os.path.join(ROOT, "tests", "stub", "libfake_rccl.so")


@pytest.fixture(scope="module")
def fake_rccl():
    src = os.path.join(ROOT, "tests", "stub", "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(src) > os.path.getmtime(FAKE):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-I", EMU, src, "-o", FAKE, "-lpthread"], check=True)
    return FAKE


@pytest.mark.parametrize("world,threads", [(2, "2"), (3, "1")])
def test_sharded_block_loop_with_several_ranks(emulated_engine, fake_rccl, world, threads):  # noqa: F811
    env = dict(os.environ, CHZ_LIB=emulated_engine, CHZ_ALLOW_EMULATED_ENGINE="1", CHZ_RCCL_LIB=fake_rccl, CHZ_ENQ_THREADS=threads)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl_ranks.py"), str(world), "11"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "fake-rccl ranks ok: world=%d" % world in r.stdout


def test_rendezvous_file_carries_the_launch_id(emulated_engine, fake_rccl):  # noqa: F811
    """round 3 advisor: a restarted rank that comes up before rank 0 must not read the id a crashed launch left at the same path"""
    env = dict(os.environ, CHZ_LIB=emulated_engine, CHZ_ALLOW_EMULATED_ENGINE="1", CHZ_RCCL_LIB=fake_rccl)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl_ranks.py"), "rendezvous", "0"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "rendezvous ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
