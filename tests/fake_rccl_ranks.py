"""Child process of tests/test_engine_fake_rccl.py: `world` ranks as THREADS of this process, each with its own engine (the CPU build,
tests/hipemu/libchz_hip_emu.so) and its own communicator over tests/stub/fake_rccl.cpp.  Rank 0 owns the A/D ring; every rank runs
its shard of the channels after the hand-over (whole-slot broadcast, then row ranges) through chz_run_blocks_sharded -- the same
entry point bench.py's --gpus N leg and a C host use -- and each rank's channels are checked against the oracle.  TEST
INFRASTRUCTURE: needs CHZ_LIB / CHZ_ALLOW_EMULATED_ENGINE / CHZ_RCCL_LIB from the parent."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_pkg  # noqa: E402
import oracle_lib as ol  # noqa: E402
from test_gpu_parity import check_channel  # noqa: E402


def main(world, nblk):
    pkg = load_pkg()
    L, M, P, olen, total = 25920, 6481, 300, 240, 22
    N = L + M - 1
    rng = np.random.default_rng(77)
    ring = rng.standard_normal(8 * L).astype(np.float32)
    resp = pkg.filterapi.design_response(P, olen, N, True, -0.3, 0.3, 11.0)
    shifts_all = np.array([900 + 517 * i for i in range(total)], np.int32)
    shifts_all[3] = -shifts_all[3]
    uid = pkg.engine.comm_unique_id()
    errors, results = [], [None] * world
    gate = threading.Barrier(world)

    def rank_main(rank):
        try:
            eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
            if rank == 0:
                eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
            first, last = pkg.sharding.shard_channels(total, rank, world)
            n = last - first
            bank = eng.bank(P, olen, max(n, 1))
            bank.set_responses(0, np.stack([resp] * max(n, 1)))
            bank.set_shifts(0, shifts_all[first:last] if n else np.zeros(1, np.int32)); bank.set_active(n)
            bank.enable_noise(50.0 * L)
            eng.set_notches([125, 0], 0.01)
            comm = pkg.engine.Comm(rank, world, uid, device=0)
            got = comm.allreduce_max([float(rank), -float(rank)])
            assert got.tolist() == [float(world - 1), 0.0], got
            comm.barrier()
            na, pitch, off = eng.spec_layout
            rows = [pkg.sharding.needed_rows(shifts_all[slice(*pkg.sharding.shard_channels(total, r, world))], P, eng.bins, na, noise=True) for r in range(world)]
            out = {}
            # row ranges first: what a peer's kernels read outside the rows it was sent is NOT the spectrum yet
            t = eng.run_blocks_sharded(comm, 0, nblk, rows=([r[0] for r in rows], [r[1] for r in rows]))
            assert t.blocks == nblk
            out["rows"] = [bank.read_slot(s).copy() for s in range(4)]
            out["noise"] = [bank.read_noise(s).copy() for s in range(4)] if n else [np.zeros(0)] * 4
            gate.wait()
            if rank == 0:
                eng.set_notches([125, 0], 0.01)                             # restart the recurrence for the second run
            eng.run_blocks_sharded(comm, 0, nblk)                           # whole-slot broadcast
            out["broadcast"] = [bank.read_slot(s).copy() for s in range(4)]
            spec = eng.spectrum((nblk - 1) % 4).copy()
            gate.wait()
            # SURVEY 8e's alternative: the block's new samples travel, every rank transforms them itself (mode 2).  The peers' rings
            # start empty, so their first window differs from the root's; from the second block on every rank sees the same samples
            # (the notch recurrence keeps a memory of block 0, but bins 0 and 125 lie outside every channel and noise window here)
            eng.run_blocks_sharded(comm, 0, nblk, samples=True)
            out["samples"] = [bank.read_slot(s).copy() for s in range(4)]
            out["samples_noise"] = [bank.read_noise(s).copy() for s in range(4)] if n else [np.zeros(0)] * 4
            comm.barrier()
            comm.close(); eng.close()
            results[rank] = (first, last, out, spec)
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                gate.abort()
            except Exception:
                pass

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errors, errors
    assert all(r is not None for r in results), "a rank did not finish"

    st = ol.Stream(L, M, ol.REAL)
    st.push(ring[7 * L:8 * L])
    nstate = np.zeros(4)
    spectra = []
    for j in range(nblk):
        s = st.push(ring[(j % 8) * L:(j % 8 + 1) * L])
        ol.notch(nstate, [125, 0], 0.01, s)
        spectra.append(s)
    # every rank holds the ROOT's spectrum bit for bit after a broadcast
    for r in range(1, world):
        np.testing.assert_array_equal(results[r][3], results[0][3])
    checked = 0
    for first, last, out, _ in results:
        for mode in ("broadcast", "rows", "samples"):
            for j in range(max(0, nblk - 4), nblk):
                got = out[mode][j % 4]
                for c in range(last - first):
                    check_channel(got[c], ol.channel(spectra[j], ol.REAL, P, olen, int(shifts_all[first + c]), resp))
                    checked += 1
        # estimate_noise() on the peers reads its 1000-bin windows out of the rows that were shipped
        for j in range(max(0, nblk - 4), nblk):
            for c in range(last - first):
                want = ol.estimate_noise(spectra[j], ol.REAL, P, int(shifts_all[first + c]), 50.0 * L)
                assert abs(out["noise"][j % 4][c] - want) <= 1e-6 * want, (first + c, j, out["noise"][j % 4][c], want)
                assert abs(out["samples_noise"][j % 4][c] - want) <= 1e-6 * want, (first + c, j, out["samples_noise"][j % 4][c], want)
        # and both hand-overs give the same samples
        for s in range(4):
            np.testing.assert_array_equal(out["rows"][s][:last - first], out["broadcast"][s][:last - first])
    print("fake-rccl ranks ok: world=%d blocks=%d channel-blocks checked=%d" % (world, nblk, checked))


def rendezvous():
    """chz_comm_create_file with a launch id (env CHZ_LAUNCH_ID): rank 1 comes up FIRST and finds the fresh left-over of another launch
    (same path, another id) -- it must keep waiting for rank 0's file instead of joining a communicator that will never form."""
    import tempfile
    import time
    pkg = load_pkg()
    os.environ["CHZ_LAUNCH_ID"] = "launch-42"
    path = os.path.join(tempfile.mkdtemp(), "chz_id")
    open(path, "wb").write(b"\x55" * 128 + b"launch-41".ljust(64, b"\0"))          # a crashed launch's file, seconds old
    done, errors = {}, []

    def rank_main(rank):
        try:
            c = pkg.engine.Comm(rank, 2, device=0, path=path, timeout_s=30.0)
            c.barrier()
            done[rank] = time.monotonic()
            c.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    t1 = threading.Thread(target=rank_main, args=(1,)); t1.start()
    time.sleep(1.0)
    assert 1 not in done and not errors, ("rank 1 took the stale file", done, errors)
    t0 = threading.Thread(target=rank_main, args=(0,)); t0.start()
    t0.join(60); t1.join(60)
    assert not errors and 0 in done and 1 in done, (errors, done)
    print("rendezvous ok")


if __name__ == "__main__":
    if sys.argv[1] == "rendezvous":
        rendezvous()
    else:
        main(int(sys.argv[1]), int(sys.argv[2]))
