"""GPU tests of the engine's pipelining machinery (round 2): the event-ordered notch recurrence under hostile
stream-to-queue mappings, retunes and filter swaps that never drain the pipeline, and RCCL behind the C ABI.
Everything is compared with the CPU oracle (oracle/), which is pinned to the reference's own filter.c.
"""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_pkg
from test_gpu_parity import check_channel, rel, SPEC_REL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    if p.engine.lib().chz_device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests cannot run (there is no CPU fallback)")
    ol.build()
    return p


def notch_per_entry(state, bins, alphas, spec):
    """apply_notch_filters (src/filter.c:464-474) with one gain per entry: the oracle walks one entry at a time."""
    for i, (b, a) in enumerate(zip(bins, alphas)):
        ol.notch(state[2 * i:2 * i + 2], [b], a, spec)


def run_cyclic(pkg, L, M, ring, bins, alphas, nblk, seed_check=True):
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    eng.set_notches(bins, alphas)
    eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
    eng.run_blocks(0, nblk)
    got = eng.spectrum((nblk - 1) % 4)
    eng.close()
    return got


def oracle_cyclic(L, M, ring, bins, alphas, nblk):
    st = ol.Stream(L, M, ol.REAL)
    st.push(ring[7 * L:8 * L])                     # primes the history with the samples before block 0
    state = np.zeros(2 * len(bins))
    want = None
    for j in range(nblk):
        want = st.push(ring[(j % 8) * L:(j % 8 + 1) * L])
        notch_per_entry(state, bins, alphas, want)
    return want


def test_notch_per_entry_alpha_duplicates_and_zero_gain(pkg):
    # struct notch_state carries its own alpha (src/filter.h:42-46); a bin named twice is filtered twice, in list
    # order; alpha = 0 (the calloc'd DC sentinel of a full spur list, src/radio.c:601-620) is a no-op
    L, M = 25920, 6481
    nblk = 23
    rng = np.random.default_rng(7)
    ring = (rng.standard_normal(8 * L) + 0.4).astype(np.float32)
    bins = [125, 4000, 125, 16000, 9000, 0]
    alphas = [0.05, 0.01, 0.2, 0.003, 0.0, 0.02]
    got = run_cyclic(pkg, L, M, ring, bins, alphas, nblk)
    want = oracle_cyclic(L, M, ring, bins, alphas, nblk)
    for b in set(bins):
        assert abs(got[b] - want[b]) <= 5e-6 * abs(want[b]) + 5e-3, b
    assert rel(got, want) <= SPEC_REL
    # the zero-gain entry really left its bin alone
    plain = oracle_cyclic(L, M, ring, [0], [0.0], nblk)
    assert abs(got[9000] - plain[9000]) <= 5e-6 * abs(plain[9000]) + 5e-3


def test_two_engines_with_notches_run_concurrently(pkg):
    # two masters in one process (wfm composite filters, several front ends): 8 streams plus the runtime's own
    # compete for the hardware queues; the recurrence of each engine must still be the sequential one
    L, M = 25920, 6481
    nblk = 61
    rng = np.random.default_rng(8)
    rings = [(rng.standard_normal(8 * L) + 0.3 * (i + 1)).astype(np.float32) for i in range(2)]
    bins = [[0], [300, 0]]
    alphas = [[0.01], [0.1, 0.02]]
    got = [None, None]

    def work(i):
        got[i] = run_cyclic(pkg, L, M, rings[i], bins[i], alphas[i], nblk)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        want = oracle_cyclic(L, M, rings[i], bins[i], alphas[i], nblk)
        for b in bins[i]:
            assert abs(got[i][b] - want[b]) <= 5e-6 * abs(want[b]) + 5e-3, (i, b)
        assert rel(got[i], want) <= SPEC_REL


_QUEUE_SCRIPT = r"""
import sys, os
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import numpy as np
import oracle_lib as ol
from conftest import load_pkg
from test_gpu_pipeline import run_cyclic, oracle_cyclic
pkg = load_pkg(); ol.build()
L, M, nblk = 25920, 6481, 45
rng = np.random.default_rng(9)
ring = (rng.standard_normal(8 * L) + 0.4).astype(np.float32)
bins, alphas = [125, 16000, 0], [0.05, 0.02, 0.01]
got = run_cyclic(pkg, L, M, ring, bins, alphas, nblk)
want = oracle_cyclic(L, M, ring, bins, alphas, nblk)
for b in bins:
    assert abs(got[b] - want[b]) <= 5e-6 * abs(want[b]) + 5e-3, b
print("ok")
"""


@pytest.mark.parametrize("queues,threads", [(1, 2), (2, 2), (1, 1), (4, 4)])
def test_notch_order_survives_any_queue_mapping(queues, threads):
    # GPU_MAX_HW_QUEUES squeezes the engine's 4 streams into fewer hardware queues, CHZ_ENQ_THREADS changes who issues
    # what: with ordering by events there is nothing on the device that could wait for a kernel queued behind it
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(queues), CHZ_ENQ_THREADS=str(threads))
    script = _QUEUE_SCRIPT % {"tests": os.path.join(ROOT, "tests"), "root": ROOT}
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]


def _engine_with_bank(pkg, L, M, P, olen, nch, rng):
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    ring = rng.standard_normal(8 * L).astype(np.float32)
    eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
    bank = eng.bank(P, olen, nch)
    return eng, bank, ring


def _spectra(L, M, ring, nblk):
    st = ol.Stream(L, M, ol.REAL)
    st.push(ring[7 * L:8 * L])
    return [st.push(ring[(j % 8) * L:(j % 8 + 1) * L], f64=True) for j in range(nblk)]


def test_retune_takes_effect_in_stream_order_without_draining(pkg):
    # blocks are enqueued back to back with NO synchronisation; between them channels are retuned (shift), one gets a
    # new filter, one flips to ISB.  Every block must come out with the settings that were current when IT was
    # enqueued -- blocks in flight keep the old ones (per-slot descriptors, response rows swapped, not overwritten).
    L, M, P, olen, nch = 25920, 6481, 300, 240, 24
    rng = np.random.default_rng(10)
    eng, bank, ring = _engine_with_bank(pkg, L, M, P, olen, nch, rng)
    N = L + M - 1
    respA = np.stack([pkg.filterapi.design_response(P, olen, N, True, -0.3, 0.3, 11.0)] * nch)
    respB = pkg.filterapi.design_response(P, olen, N, True, 0.02, 0.2, 6.0)
    bank.set_responses(0, respA)
    shifts = np.array([500 + 137 * i for i in range(nch)], np.int32)
    bank.set_shifts(0, shifts)
    bank.set_active(nch)
    history = []                                   # per block: (shifts, which channels use respB, isb flags)
    cur_shift, cur_b, cur_isb = shifts.copy(), set(), np.zeros(nch, np.uint8)
    nblk = 4                                       # one block per slot, all in flight together
    for j in range(nblk):
        if j == 1:
            cur_shift = cur_shift.copy(); cur_shift[3] = -4000; cur_shift[17] = 9000
            bank.set_shifts(3, cur_shift[3:4]); bank.set_shifts(17, cur_shift[17:18])
        if j == 2:
            cur_b = {5}
            bank.set_responses(5, respB[None, :])
            cur_isb = cur_isb.copy(); cur_isb[8] = 1
            bank.set_isb(8, cur_isb[8:9])
        if j == 3:
            cur_shift = cur_shift.copy(); cur_shift[3] = 2500
            bank.set_shifts(3, cur_shift[3:4])
        history.append((cur_shift.copy(), set(cur_b), cur_isb.copy()))
        eng.step(j)                                # asynchronous
    eng.sync()
    spectra = _spectra(L, M, ring, nblk)
    for j in range(nblk):
        out = bank.read_slot(j % 4)
        sh, useb, isb = history[j]
        for c in range(nch):
            r = respB if c in useb else respA[c]
            want = ol.channel(spectra[j], ol.REAL, P, olen, int(sh[c]), r, isb=bool(isb[c]))
            check_channel(out[c], want)
    eng.close()


def test_sparse_retunes_in_a_large_bank(pkg):
    # retunes at the two ends and in the middle of a 20,000-channel bank, a dozen scattered ones on top, between blocks that are
    # never drained: the per-slot descriptor copies are refreshed range by range (a short list of dirty ranges per slot, merged
    # only beyond 16 islands; round 2 kept ONE interval and re-uploaded everything in between) -- every touched channel, its
    # neighbours and a sample of the rest must come out with the settings current when the block was enqueued
    L, M, P, olen, nch = 25920, 6481, 20, 16, 20000
    rng = np.random.default_rng(14)
    eng, bank, ring = _engine_with_bank(pkg, L, M, P, olen, nch, rng)
    N = L + M - 1
    resp = pkg.filterapi.design_response(P, olen, N, True, -0.3, 0.3, 5.0)
    for c0 in range(0, nch, 4000):
        bank.set_responses(c0, np.stack([resp] * 4000))
    shifts = (200 + (np.arange(nch) * 7) % 12000).astype(np.int32)
    bank.set_shifts(0, shifts)
    bank.set_active(nch)
    cur = shifts.copy()
    history = []
    touched = set()
    nblk = 6
    for j in range(nblk):
        if j >= 1:
            edits = [0, nch - 1, nch // 2] if j % 2 else []
            edits += [int(c) for c in rng.integers(0, nch, 12 if j == 3 else 2)]      # block 3: more islands than the list holds
            cur = cur.copy()
            for c in edits:
                cur[c] = int(rng.integers(-12000, 12000))
                bank.set_shifts(c, cur[c:c + 1])
                touched.add(c)
        history.append(cur.copy())
        eng.step(j)                                # asynchronous: nothing is drained in between
    eng.sync()
    spectra = _spectra(L, M, ring, nblk)
    check = sorted(set(list(touched) + [min(nch - 1, c + 1) for c in touched] + [max(0, c - 1) for c in touched] + list(range(0, nch, 997))))
    for j in range(nblk - 4, nblk):                # the last block of every slot
        out = bank.read_slot(j % 4)
        for c in check:
            want = ol.channel(spectra[j], ol.REAL, P, olen, int(history[j][c]), resp)
            check_channel(out[c], want)
    eng.close()


def test_response_swaps_recycle_spare_rows(pkg):
    # more filter changes than the bank has spare rows, while blocks keep flowing: rows are recycled behind fences
    L, M, P, olen, nch = 25920, 6481, 300, 240, 8
    rng = np.random.default_rng(11)
    eng, bank, ring = _engine_with_bank(pkg, L, M, P, olen, nch, rng)
    N = L + M - 1
    resp = [pkg.filterapi.design_response(P, olen, N, True, -0.3, 0.3, 11.0) for _ in range(nch)]
    bank.set_responses(0, np.stack(resp))
    shifts = np.array([700 + 211 * i for i in range(nch)], np.int32)
    bank.set_shifts(0, shifts); bank.set_active(nch)
    nblk = 60                                      # 60 swaps > 16 spare rows
    for j in range(nblk):
        c = j % nch
        hi = 0.05 + 0.4 * ((j * 7) % 10) / 10
        resp[c] = pkg.filterapi.design_response(P, olen, N, True, -hi, hi, 11.0)
        bank.set_responses(c, resp[c][None, :])
        eng.step(j)
    eng.sync()
    spectra = _spectra(L, M, ring, nblk)
    out = bank.read_slot((nblk - 1) % 4)
    for c in range(nch):
        check_channel(out[c], ol.channel(spectra[nblk - 1], ol.REAL, P, olen, int(shifts[c]), resp[c]))
    eng.close()


def test_rccl_behind_the_c_abi_single_rank(pkg):
    # chz_comm_* / chz_spectrum_broadcast / _exchange_rows / chz_run_blocks_sharded with one rank: the collective
    # calls really go through RCCL (ncclCommInitRank, ncclBroadcast, grouped send/recv) on the slot streams and the
    # sharded block loop gives exactly what the plain one gives.  Multi-rank data movement: tests/test_distributed_gloo.py.
    L, M, P, olen, nch = 25920, 6481, 300, 240, 16
    rng = np.random.default_rng(12)
    eng, bank, ring = _engine_with_bank(pkg, L, M, P, olen, nch, rng)
    N = L + M - 1
    bank.set_responses(0, np.stack([pkg.filterapi.design_response(P, olen, N, True, -0.3, 0.3, 11.0)] * nch))
    shifts = np.array([400 + 300 * i for i in range(nch)], np.int32)
    bank.set_shifts(0, shifts); bank.set_active(nch)
    eng.set_notches([0], 0.01)
    comm = pkg.engine.Comm(0, 1, pkg.engine.comm_unique_id(), device=0)
    assert comm.allreduce_max([3.5, -1.0]).tolist() == [3.5, -1.0]
    comm.barrier()
    eng.run_blocks(0, 9)
    plain = [bank.read_slot(s).copy() for s in range(4)]
    spec_plain = eng.spectrum(0)
    eng.set_notches([0], 0.01)                     # reset the recurrence
    t = eng.run_blocks_sharded(comm, 0, 9)
    assert t.blocks == 9 and t.total_ms > 0
    for s in range(4):
        np.testing.assert_array_equal(bank.read_slot(s), plain[s])
    np.testing.assert_array_equal(eng.spectrum(0), spec_plain)
    na, pitch, off = eng.spec_layout
    rows = pkg.sharding.needed_rows(shifts, P, eng.bins, na)
    eng.set_notches([0], 0.01)
    eng.run_blocks_sharded(comm, 0, 9, rows=([rows[0]], [rows[1]]))
    for s in range(4):
        np.testing.assert_array_equal(bank.read_slot(s), plain[s])
    comm.broadcast_spectrum(eng, 2); comm.exchange_rows(eng, 1, [rows[0]], [rows[1]])
    eng.sync()
    comm.close()
    eng.close()


def test_comm_rendezvous_file_single_rank(pkg, tmp_path):
    path = tmp_path / "chz_id"
    path.write_bytes(b"\x55" * 128)                       # a previous launch's left-over must not be taken for this launch's id
    comm = pkg.engine.Comm(0, 1, device=0, path=str(path))
    assert not path.exists()                               # rank 0 removes its file once every rank has joined
    comm.barrier()
    comm.close()
    comm = pkg.engine.Comm(0, 1, device=0, path=str(path))   # and the path can be used again
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("L,M", [(240, 273), (480, 545), (960, 1089), (1920, 2177), (300, 101)])
def test_mini_master_pool_matches_oracle(pkg, L, M):
    # radiod's filter2 = 1 / 4 geometries (src/radio.c:1572-1594; share/presets.conf:204,223,297): 300 independent inline
    # masters of one pool served by ONE launch per block; per instance the overlap-save answer of the oracle
    N = L + M - 1
    rng = np.random.default_rng(N + 1)
    ninst = 300
    pool = pkg.engine.MiniPool(L, M, ninst)
    insts = [pool.add() for _ in range(ninst)]
    assert sorted(insts) == list(range(ninst))
    with pytest.raises(pkg.engine.ChzError):
        pool.add()                                        # full: loud, not silent
    kinds = [(-0.2, 0.2), (0.01, 0.3), (-0.4, 0.1), (-0.45, 0.45), (-0.3, -0.05)]
    resp = [ol.set_filter(N, L, N, False, lo, hi, 9.0) for lo, hi in kinds]
    for i in insts:
        pool.set_response(i, resp[i % 5])
    isb = np.array([(i % 7) == 3 for i in range(ninst)], np.uint8)
    check = [0, 3, 10, 11, 150, 299]
    streams = {i: ol.Stream(L, M, ol.COMPLEX) for i in check}
    hist = np.zeros((ninst, M - 1), np.complex64)
    try:
        for blk in range(3):
            x = (rng.standard_normal((ninst, L)) + 1j * rng.standard_normal((ninst, L))).astype(np.complex64)
            win = np.concatenate([hist, x], axis=1)
            out = pool.execute(insts, win, isb=isb)
            for i in check:
                spec = streams[i].push(x[i], f64=True)
                want = ol.channel(spec, ol.COMPLEX, N, L, 0, resp[i % 5], isb=bool(isb[i]))
                check_channel(out[i], want)
            hist = win[:, L:]
        pool.release(insts[5]); assert pool.add() == insts[5]
    finally:
        pool.close()


def test_mini_master_rejects_what_it_cannot_do(pkg):
    for L, M in ((240, 272), (5000, 5001)):               # N = 511 (prime factor 7 x 73), N = 10000 > 8192
        with pytest.raises(pkg.engine.ChzError):
            pkg.engine.MiniPool(L, M, 4)


# ------------------------------------------------------------------------------------------------
# the linear demodulator behind the channel outputs
# ------------------------------------------------------------------------------------------------
def _demod_engine(pkg, ring, L, M, P, olen, cases, fs_out):
    from test_kernels_emulated import DEMOD_CASES                      # the seven mode combinations
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    eng.write(ring[:8 * L - (M - 1)]); eng.write(ring[8 * L - (M - 1):])
    nch = len(cases)
    bank = eng.bank(P, olen, nch)
    N = L + M - 1
    bank.set_responses(0, np.stack([pkg.filterapi.design_response(P, olen, N, True, -0.12, 0.12, 11.0)] * nch))
    shifts = np.array([2500 + 3 * i for i in range(nch)], np.int32)      # all around the test carrier
    bank.set_tuning(0, 0, shifts, np.array([-(3.7 + i) / fs_out for i in range(nch)]))
    bank.set_active(nch)
    bank.enable_noise(50.0 * L)
    params = [ol.lin_params(**kw) for kw in cases]
    bank.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in params], 0.02)
    return eng, bank, params


# demod_lin_lanes (what banks of >= 65536 channels get) / demod_linear_tail / demod_lin_lanes behind a channel kernel that stages its rows
# through LDS and leaves the AGC's slice-energy peak there (round 4: what banks of >= 65536 channels REALLY get -- the demodulator then reads
# the baseband once; channel 4 of DEMOD_CASES has a post-detection shift and still walks it twice)
@pytest.mark.parametrize("lin_path", ["lanes", "wave", "lanes_staged"])
def test_linear_demodulator_on_the_device(pkg, monkeypatch, lin_path):
    monkeypatch.setenv("CHZ_DEMOD_WAVE", "1" if lin_path == "wave" else "0")
    if lin_path == "lanes_staged":
        monkeypatch.setenv("CHZ_CHAN_STAGE", "1")
    """chan_ifft (+ fine tuning, bb_power) -> noise_est -> demod_linear_tail, block after block through the C ABI.  The
    oracle's demodulator (pinned to the reference's linear.c) is fed exactly what the device stage was fed -- the channel
    outputs, bb_power and noise estimate read back from the same slot -- and must produce the same frames; then the same
    40 blocks run pipelined over 4 streams and two issuing threads must end in the same AGC state."""
    from test_kernels_emulated import DEMOD_CASES
    L, M, P, olen, fs_out = 25920, 6481, 300, 240, 12000.0
    nblk = 40
    rng = np.random.default_rng(77)
    t = np.arange(8 * L)
    env = np.ones(8 * L); env[:3 * L] = 0.02; env[5 * L:5 * L + L // 3] = 8.0; env[6 * L:] = 0.1       # level steps walk the AGC branches
    ring = ((0.05 * np.cos(2 * np.pi * (2501.3 / (L + M - 1)) * t) * env) + 1e-4 * rng.standard_normal(8 * L)).astype(np.float32)
    eng, bank, params = _demod_engine(pkg, ring, L, M, P, olen, DEMOD_CASES, fs_out)
    oracles = [ol.LinDemod(p) for p in params]
    seen = set()
    try:
        for b in range(nblk):
            eng.step(b)
            out = bank.read_slot(b % 4); power = bank.read_power(b % 4); noise = bank.read_noise(b % 4)
            pcm, status = bank.read_pcm(b % 4)
            for i, p in enumerate(params):
                want, st = oracles[i].block(out[i], power[i], noise[i], 0.02)
                got = status[i]
                assert (got.frame, got.mute, got.squelch_state) == (st.frame, st.mute, st.squelch_state), (b, i)
                assert got.gain == pytest.approx(st.gain, rel=1e-9) and got.n0 == pytest.approx(st.n0, rel=1e-12)
                assert got.output_power == pytest.approx(st.output_power, rel=1e-6, abs=1e-300)
                seen.add((i, got.frame, got.mute))
                if st.frame == ol.FRAME_DATA:
                    nb = ol.pcm_bytes(p.encoding, olen * p.channels)
                    if p.encoding in (ol.PCM_MULAW, ol.PCM_ALAW, ol.PCM_F16LE, ol.PCM_F16BE):
                        assert np.mean(pcm[i, :nb] != want) == 0, (b, i)
                    elif p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                        dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                        a, w = pcm[i, :nb].view(dt).astype(np.int32), want.view(dt).astype(np.int32)
                        assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0, (b, i)
                    else:
                        dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
                        a, w = pcm[i, :nb].view(dt).astype(np.float64), want.view(dt).astype(np.float64)
                        assert np.abs(a - w).max() <= 1e-6 * max(np.abs(w).max(), 1e-30), (b, i)
        final = [(s.gain, s.n0, s.squelch_state) for s in bank.read_pcm((nblk - 1) % 4)[1]]
    finally:
        eng.close()
    assert (6, ol.FRAME_DATA, 1) in seen                                  # the untuned channel is muted but still demodulated
    eng2, bank2, _ = _demod_engine(pkg, ring, L, M, P, olen, DEMOD_CASES, fs_out)
    try:
        t = eng2.run_blocks(0, nblk)
        st2 = bank2.read_pcm((nblk - 1) % 4)[1]
        for i, (g, n0, sq) in enumerate(final):
            assert st2[i].gain == pytest.approx(g, rel=1e-9) and st2[i].n0 == pytest.approx(n0, rel=1e-12) and st2[i].squelch_state == sq
        it = eng2.run_blocks(nblk, 8, instrument=True)
        assert it.demod_n == 8 and it.demod_ms > 0
    finally:
        eng2.close()


def test_partial_rerun_leaves_the_demodulators_alone(pkg):
    """chz_bank_execute_range(bank, job, 0, n < active) -- the drop-in's miss path re-running a few channels of a block that has
    already been demodulated -- must not step anybody's AGC / squelch / noise smoothing a second time nor touch the block's PCM:
    only a launch over the whole bank is 'the block' (round 2 advisor finding)."""
    from test_kernels_emulated import DEMOD_CASES
    L, M, P, olen, fs_out = 25920, 6481, 300, 240, 12000.0
    rng = np.random.default_rng(78)
    t = np.arange(8 * L)
    ring = ((0.05 * np.cos(2 * np.pi * (2501.3 / (L + M - 1)) * t)) + 1e-4 * rng.standard_normal(8 * L)).astype(np.float32)
    finals = []
    for rerun in (False, True):
        eng, bank, params = _demod_engine(pkg, ring, L, M, P, olen, DEMOD_CASES[:4], fs_out)
        try:
            for b in range(6):
                eng.step(b)
                if rerun and b in (2, 5):
                    eng.sync()
                    pcm0, st0 = bank.read_pcm(b % 4)
                    bank.execute_range(b, 0, 1); bank.execute_range(b, 0, 3)
                    eng.sync()
                    pcm1, st1 = bank.read_pcm(b % 4)
                    assert np.array_equal(pcm0, pcm1)
                    assert [(s.gain, s.n0, s.squelch_state, s.frame) for s in st0] == [(s.gain, s.n0, s.squelch_state, s.frame) for s in st1]
            eng.sync()
            pcm, st = bank.read_pcm(5 % 4)
            finals.append((pcm.copy(), [(s.gain, s.n0, s.squelch_state, s.frame, s.output_power) for s in st]))
        finally:
            eng.close()
    assert np.array_equal(finals[0][0], finals[1][0]) and finals[0][1] == finals[1][1]


def test_mini_master_of_8192_points(pkg):
    # two 8192-point buffers are 128 KB of LDS: the kernel has to ask for more than the default 64 KB (round 2 advisor finding)
    L, M = 6144, 2049
    N = L + M - 1
    rng = np.random.default_rng(N)
    pool = pkg.engine.MiniPool(L, M, 3)
    insts = [pool.add() for _ in range(3)]
    resp = ol.set_filter(N, L, N, False, -0.2, 0.3, 9.0)
    for i in insts:
        pool.set_response(i, resp)
    stream = ol.Stream(L, M, ol.COMPLEX)
    hist = np.zeros((3, M - 1), np.complex64)
    try:
        for blk in range(2):
            x = (rng.standard_normal((3, L)) + 1j * rng.standard_normal((3, L))).astype(np.complex64)
            x[1:] = x[0]
            win = np.concatenate([hist, x], axis=1)
            out = pool.execute(insts, win)
            want = ol.channel(stream.push(x[0], f64=True), ol.COMPLEX, N, L, 0, resp)
            for i in range(3):
                check_channel(out[i], want)
            hist = win[:, L:]
    finally:
        pool.close()


@pytest.mark.parametrize("fm_path", ["lanes", "wave"])      # demod_fm_lanes (what banks of >= 65536 channels get) / demod_linear_tail
def test_fm_demodulator_on_the_device(pkg, monkeypatch, fm_path):
    monkeypatch.setenv("CHZ_DEMOD_WAVE", "1" if fm_path == "wave" else "0")
    """24 kHz NBFM channels behind the channelizer: an FM carrier that comes up out of the noise, is modulated with an offset,
    and fades out through the squelch tail.  The oracle's demod_fm restatement (pinned to the reference's fm.c) gets exactly
    what the device stage got (channel outputs, bb_power, noise estimate read back) and must produce the same frames."""
    from test_kernels_emulated import FM_CASES
    L, M, P, olen, fs_in, fs_out = 25920, 6481, 600, 480, 1.296e6, 24000.0
    nblk = 36
    rng = np.random.default_rng(91)
    t = np.arange(nblk * L)
    fc, dev, fmod = 100000.0 + 350.0, 3000.0, 1000.0
    level = np.full(nblk * L, 0.05); level[:6 * L] = 0.0; level[26 * L:] = 0.0
    level[22 * L:26 * L] = 0.05 * np.linspace(1, 0.02, 4 * L)
    x = (level * np.cos(2 * np.pi * fc * t / fs_in - (dev / fmod) * np.cos(2 * np.pi * fmod * t / fs_in)) +
         2e-3 * rng.standard_normal(nblk * L)).astype(np.float32)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    nch = len(FM_CASES)
    bank = eng.bank(P, olen, nch)
    N = L + M - 1
    bank.set_responses(0, np.stack([pkg.filterapi.design_response(P, olen, N, True, -8000 / fs_out, 8000 / fs_out, 11.0)] * nch))
    shifts = np.full(nch, 2500, np.int32)
    bank.set_tuning(0, 0, shifts, np.zeros(nch))
    bank.set_active(nch)
    bank.enable_noise(fs_in)
    params = [ol.fm_params(**kw) for kw in FM_CASES]
    bank.set_pcm_stride(4 * olen)
    bank.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in params], 0.02)
    oracles = [ol.FmDemod(p) for p in params]
    seen = set()
    try:
        for b in range(nblk):
            eng.write(x[b * L:(b + 1) * L])
            eng.step(b)
            out = bank.read_slot(b % 4); power = bank.read_power(b % 4); noise = bank.read_noise(b % 4)
            pcm, status = bank.read_pcm(b % 4)
            for i, p in enumerate(params):
                want, st = oracles[i].block(out[i], power[i], noise[i], 0.02)
                got = status[i]
                assert (got.frame, got.mute, got.squelch_state) == (st.frame, st.mute, st.squelch_state), (b, i)
                assert got.snr == pytest.approx(st.snr, rel=1e-5, abs=1e-9)
                seen.add((got.frame, got.mute))
                if st.frame == ol.FRAME_DATA:
                    assert got.output_power == pytest.approx(st.output_power, rel=1e-6)
                    assert got.foffset == pytest.approx(st.foffset, rel=1e-6, abs=1e-6) and got.pdeviation == pytest.approx(st.pdeviation, rel=1e-5, abs=1e-3)
                    nb = ol.pcm_bytes(p.encoding, olen)
                    if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                        dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                        a, w = pcm[i, :nb].view(dt).astype(np.int32), want.view(dt).astype(np.int32)
                        assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0, (b, i)
                    else:
                        dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
                        a, w = pcm[i, :nb].view(dt).astype(np.float64), want.view(dt).astype(np.float64)
                        assert np.abs(a - w).max() <= 2e-6 * max(np.abs(w).max(), 1e-30), (b, i)
    finally:
        eng.close()
    assert (ol.FRAME_DATA, 0) in seen and (ol.FRAME_SILENCE, 0) in seen and (ol.FRAME_SILENCE, 1) in seen


@pytest.mark.gpu
@pytest.mark.parametrize("lin_path", ["lanes", "wave"])     # demod_lin_lanes (what banks of >= 65536 channels get) / demod_linear_tail
def test_coherent_modes_and_tone_squelch_on_the_device(pkg, monkeypatch, lin_path):
    monkeypatch.setenv("CHZ_DEMOD_WAVE", "1" if lin_path == "wave" else "0")
    """The sequential stages of the demodulator kernel through the C ABI: the carrier-tracking PLL of the linear demodulator
    (plain and squaring loop) with its lock detector and squelch, FM's PLL demodulator and the PL-tone squelch (tone present /
    absent).  Two banks on one engine -- 12 kHz linear channels and 24 kHz FM channels -- share the demodulator stream.  The
    restated demodulators (pinned to the reference's linear.c / fm.c / osc.c / iir.c) get exactly what the device stages got."""
    from test_kernels_emulated import PLL_CASES, FM2_CASES, _check_pcm
    L, M, fs_in = 25920, 6481, 1.296e6
    N = L + M - 1
    nblk = 90
    rng = np.random.default_rng(123)
    t = np.arange(nblk * L)
    on_a = (t < 50 * L).astype(np.float64)             # there from the start: a wide loop left alone with noise can run off to a false lock
    am = 0.05 * on_a * (1 + 0.5 * np.sin(2 * np.pi * 400 * t / fs_in)) * np.cos(2 * np.pi * (100000.0 + 30.0) * t / fs_in + 0.7)
    bpsk = 0.05 * on_a * np.sign(np.sin(2 * np.pi * 31.25 * t / fs_in + 0.3)) * np.cos(2 * np.pi * (140000.0 + 5.0) * t / fs_in)
    lvl = np.full(nblk * L, 0.05); lvl[:6 * L] = 0.0; lvl[60 * L:] = 0.0; lvl[56 * L:60 * L] = 0.05 * np.linspace(1, 0.02, 4 * L)
    mod = (3000.0 / 1000.0) * np.cos(2 * np.pi * 1000.0 * t / fs_in)
    fm_tone = lvl * np.cos(2 * np.pi * 200350.0 * t / fs_in - mod - (600.0 / 100.0) * np.cos(2 * np.pi * 100.0 * t / fs_in))
    fm_plain = lvl * np.cos(2 * np.pi * 300350.0 * t / fs_in - mod)
    x = (am + bpsk + fm_tone + fm_plain + 4e-4 * rng.standard_normal(nblk * L)).astype(np.float32)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    # bank A: linear, 12 kHz; cases 0, 2, 3 listen to the AM carrier, case 1 (squaring loop) to the BPSK one
    lin = [ol.lin_params(**kw) for kw in PLL_CASES]
    A = eng.bank(300, 240, len(lin))
    A.set_responses(0, np.stack([pkg.filterapi.design_response(300, 240, N, True, -2950 / 12000.0, 2950 / 12000.0, 11.0)] * len(lin)))
    A.set_tuning(0, 0, np.array([2500, 3500, 2500, 2500], np.int32), np.zeros(len(lin)))
    A.set_active(len(lin)); A.enable_noise(fs_in); A.set_pcm_stride(8 * 240)
    A.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in lin], 0.02)
    # bank B: FM, 24 kHz; the cases that were sent the tone listen at 200.35 kHz, the others at 300.35 kHz
    fm = [ol.fm_params(**kw) for kw, _ in FM2_CASES]
    B = eng.bank(600, 480, len(fm))
    B.set_responses(0, np.stack([pkg.filterapi.design_response(600, 480, N, True, -8000 / 24000.0, 8000 / 24000.0, 11.0)] * len(fm)))
    B.set_tuning(0, 0, np.array([5000 if sent else 7500 for _, sent in FM2_CASES], np.int32), np.zeros(len(fm)))
    B.set_active(len(fm)); B.enable_noise(fs_in); B.set_pcm_stride(4 * 480)
    B.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in fm], 0.02)
    o_lin = [ol.LinDemod(p) for p in lin]; o_fm = [ol.FmDemod(p) for p in fm]
    locked = np.zeros(len(lin), int); data = np.zeros(len(fm), int)
    dev = []
    try:
        for b in range(nblk):
            eng.write(x[b * L:(b + 1) * L])
            eng.step(b)
            s = b % 4
            out = A.read_slot(s); power = A.read_power(s); noise = A.read_noise(s); pcm, status = A.read_pcm(s)
            for i, p in enumerate(lin):
                want, st = o_lin[i].block(out[i], power[i], noise[i], 0.02)
                got = status[i]
                # Once the carrier is gone a PLL chews on noise, where the phase detector's +-pi wrap turns an ulp of difference
                # between two arithmetic libraries into a different cycle slip: beyond that point only what does not hang on the
                # loop's trajectory is compared (frame type, squelch sequencer, lock detector, AGC gain).
                strict = b < 50 or not p.pll_enable
                locked[i] += got.pll_lock
                assert (got.frame, got.mute, got.squelch_state, got.pll_lock) == (st.frame, st.mute, st.squelch_state, st.pll_lock), (b, i)
                assert got.gain == pytest.approx(st.gain, rel=1e-6)
                if not strict:
                    continue
                assert got.pll_rotations == st.pll_rotations, (b, i)
                assert got.output_power == pytest.approx(st.output_power, rel=1e-5, abs=1e-300)
                if p.pll_enable:
                    dev.append((abs(got.pll_snr - st.pll_snr) / max(abs(st.pll_snr), 1e-3), abs(got.foffset - st.foffset),
                                abs((got.pll_cphase - st.pll_cphase + np.pi) % (2 * np.pi) - np.pi), b, i, st.pll_snr, st.pll_lock))
                if st.frame == ol.FRAME_DATA:
                    _check_pcm(p, pcm[i], want, 240 * p.channels, 4e-6)
            out = B.read_slot(s); power = B.read_power(s); noise = B.read_noise(s); pcm, status = B.read_pcm(s)
            pcm2, fl = B.read_pcm_flags(s)                       # the one-byte-per-channel status of the same block
            assert np.array_equal(pcm2, pcm)
            assert [int(f) for f in fl] == [(st_.frame & 1) | (st_.mute & 1) << 1 | (st_.pll_lock & 1) << 2 | (st_.tone_mute & 1) << 3 for st_ in status]
            for i, p in enumerate(fm):
                want, st = o_fm[i].block(out[i], power[i], noise[i], 0.02)
                got = status[i]
                assert (got.frame, got.mute, got.squelch_state, got.tone_mute) == (st.frame, st.mute, st.squelch_state, st.tone_mute), (b, i)
                assert got.snr == pytest.approx(st.snr, rel=1e-5, abs=1e-9)
                assert got.tone_deviation == pytest.approx(st.tone_deviation, rel=1e-5, abs=1e-6)
                if st.frame == ol.FRAME_DATA:
                    data[i] += 1
                    assert got.output_power == pytest.approx(st.output_power, rel=4e-6)
                    assert got.foffset == pytest.approx(st.foffset, rel=1e-5, abs=1e-5)
                    _check_pcm(p, pcm[i], want, 480, 8e-6)
    finally:
        eng.close()
    # with a carrier to hold on to, device and restatement walk the same trajectory (measured: 2e-14 relative on the SNR, 3e-13 Hz on
    # the frequency offset, identical VCO phase words)
    assert max(d[0] for d in dev) < 1e-9 and max(d[1] for d in dev) < 1e-9 and max(d[2] for d in dev) < 1e-6
    assert all(10 < locked[i] < nblk - 10 for i in range(3)) and locked[3] == 0        # the loops locked on their carriers
    assert data[0] > 5 and data[1] > 5 and data[2] > 5 and data[3] == 0 and data[4] == 0 and data[5] > 5


@pytest.mark.gpu
def test_fm_lane_passes_equal_the_one_kernel_path_on_the_device(pkg, monkeypatch):
    """FM's PLL demodulator and PL-tone detector at one channel per lane (passes of their own around the demodulator kernel) against
    the one-kernel path, where lane 0 of each channel's wavefront walks the block (CHZ_PLL_LANE0=1): both are the same statements
    compiled twice, and every status record and PCM byte must agree -- 70 channels, two lane groups, squelch opening and closing."""
    from test_kernels_emulated import FM2_CASES
    L, M, fs_in = 25920, 6481, 1.296e6
    N = L + M - 1
    nblk, nch = 44, 70
    rng = np.random.default_rng(321)
    t = np.arange(nblk * L)
    lvl = np.full(nblk * L, 0.05); lvl[:5 * L] = 0.0; lvl[34 * L:] = 0.0; lvl[30 * L:34 * L] = 0.05 * np.linspace(1, 0.02, 4 * L)
    mod = (3000.0 / 1000.0) * np.cos(2 * np.pi * 1000.0 * t / fs_in)
    fm_tone = lvl * np.cos(2 * np.pi * 200350.0 * t / fs_in - mod - (600.0 / 100.0) * np.cos(2 * np.pi * 100.0 * t / fs_in))
    fm_plain = lvl * np.cos(2 * np.pi * 300350.0 * t / fs_in - mod)
    x = (fm_tone + fm_plain + 4e-4 * rng.standard_normal(nblk * L)).astype(np.float32)
    cases = [FM2_CASES[i % len(FM2_CASES)] for i in range(nch)]
    fm = [ol.fm_params(**kw) for kw, _ in cases]
    runs = []
    for lane0 in ("1", None):
        if lane0:
            monkeypatch.setenv("CHZ_PLL_LANE0", lane0)
        else:
            monkeypatch.delenv("CHZ_PLL_LANE0", raising=False)
        eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
        try:
            B = eng.bank(600, 480, nch)
            B.set_responses(0, np.stack([pkg.filterapi.design_response(600, 480, N, True, -8000 / 24000.0, 8000 / 24000.0, 11.0)] * nch))
            B.set_tuning(0, 0, np.array([5000 if sent else 7500 for _, sent in cases], np.int32), np.zeros(nch))
            B.set_active(nch); B.enable_noise(fs_in); B.set_pcm_stride(4 * 480)
            B.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in fm], 0.02)
            trace, frames = [], 0
            for b in range(nblk):
                eng.write(x[b * L:(b + 1) * L])
                eng.step(b)
                pcm, status = B.read_pcm(b % 4)
                frames += sum(1 for st in status if st.frame == ol.FRAME_DATA)
                trace.append((pcm.tobytes(), bytes(status)))
            runs.append(trace)
            assert frames > 200
        finally:
            eng.close()
    for b in range(nblk):
        assert runs[0][b] == runs[1][b], b


@pytest.mark.gpu
def test_filter2_between_channelizer_and_demodulator(pkg):
    """radiod with `filter2 = 1` (src/radio.c:1572-1594): channel block -> the channel's private second filter (chz_mini_*) -> the
    fine-tuning tail (radiod's own, here the restated one) -> demodulator.  On the device: a bank whose demodulators do NOT run
    behind the channel kernel (chz_bank_demod_auto 0), the pooled inline masters, chz_bank_write_block + chz_bank_demod.  Against
    the same chain on the oracle, block after block; a second, ordinary bank keeps demodulating automatically alongside."""
    L, M, P, olen, fs_out = 25920, 6481, 300, 240, 12000.0
    N = L + M - 1
    nblk, nch = 14, 4
    rng = np.random.default_rng(31)
    t = np.arange(nblk * L)
    x = (0.05 * np.cos(2 * np.pi * (2500.3 / N) * t) * (1 + 0.4 * np.sin(2 * np.pi * 3e-4 * t)) + 1e-4 * rng.standard_normal(nblk * L)).astype(np.float32)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    resp = np.stack([ol.set_filter(P, olen, N, True, -0.4, 0.4, 9.0)] * nch).astype(np.complex64)
    shifts = np.full(nch, 2500, np.int32)                      # 2500 % overlap factor == 0, no fine offset: the tuning tail is the identity
    params = [ol.lin_params(), ol.lin_params(env=True, dc_alpha=0.002, encoding=ol.PCM_S16LE), ol.lin_params(channels=2, encoding=ol.PCM_F32LE),
              ol.lin_params(agc=False, gain_db=30.0)]
    dp = [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in params]
    banks = []
    for _ in range(2):
        b = eng.bank(P, olen, nch)
        b.set_responses(0, resp); b.set_tuning(0, 0, shifts, np.zeros(nch)); b.set_active(nch); b.enable_noise(1.296e6); b.set_pcm_stride(8 * olen)
        b.set_demod(0, 0, dp, 0.02)
        banks.append(b)
    f2bank, plain = banks
    f2bank.demod_auto(False)
    # filter2: L2 = 240, N2 = round2(2 * 240) = 512 -> M2 = 273; a 2.4 kHz low-pass per channel
    L2, M2 = olen, 512 - olen + 1
    pool = pkg.engine.MiniPool(L2, M2, nch)
    insts = [pool.add() for _ in range(nch)]
    r2 = ol.set_filter(512, L2, 512, False, -0.1, 0.1, 7.0)
    for i in insts:
        pool.set_response(i, r2)
    hist = np.zeros((nch, M2 - 1), np.complex64)
    st2 = [ol.Stream(L2, M2, ol.COMPLEX) for _ in range(nch)]
    o_f2 = [ol.LinDemod(p) for p in params]; o_plain = [ol.LinDemod(p) for p in params]
    try:
        for b in range(nblk):
            eng.write(x[b * L:(b + 1) * L]); eng.step(b)
            s = b % 4
            raw = f2bank.read_slot(s); noise = f2bank.read_noise(s)
            win = np.concatenate([hist, raw], axis=1)
            filt = pool.execute(insts, win)                                   # device: all four second filters in one launch
            hist = win[:, L2:]
            pw = np.array([np.mean(np.abs(f.astype(np.complex128)) ** 2) for f in filt])      # the power sum of the tuning tail
            f2bank.inject(s, filt, pw, noise)
            f2bank.demod_only(b)
            pcm, status = f2bank.read_pcm(s)
            pcm_p, status_p = plain.read_pcm(s)
            raw_p = plain.read_slot(s); pw_p = plain.read_power(s); noise_p = plain.read_noise(s)
            for i, p in enumerate(params):
                spec2 = st2[i].push(raw[i], f64=True)
                want_f = ol.channel(spec2, ol.COMPLEX, 512, L2, 0, r2)
                assert np.linalg.norm(filt[i] - want_f) <= 3e-6 * max(np.linalg.norm(want_f), 1e-12)
                for orc, got, gp, inp, ipw, ins in ((o_f2[i], status[i], pcm[i], filt[i], pw[i], noise[i]),
                                                    (o_plain[i], status_p[i], pcm_p[i], raw_p[i], pw_p[i], noise_p[i])):
                    want, st = orc.block(inp, ipw, ins, 0.02)
                    assert (got.frame, got.mute, got.squelch_state) == (st.frame, st.mute, st.squelch_state), (b, i)
                    assert got.gain == pytest.approx(st.gain, rel=1e-9) and got.output_power == pytest.approx(st.output_power, rel=1e-6, abs=1e-300)
                    if st.frame == ol.FRAME_DATA:
                        nb = ol.pcm_bytes(p.encoding, olen * p.channels)
                        if p.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                            dt = ">i2" if p.encoding == ol.PCM_S16BE else "<i2"
                            a, w = gp[:nb].view(dt).astype(np.int32), want.view(dt).astype(np.int32)
                            assert np.abs(a - w).max() <= 1 and np.mean(a != w) == 0, (b, i)
                        else:
                            a, w = gp[:nb].view("<f4").astype(np.float64), want.view("<f4").astype(np.float64)
                            assert np.abs(a - w).max() <= 1e-6 * max(np.abs(w).max(), 1e-30), (b, i)
    finally:
        pool.close(); eng.close()


@pytest.mark.gpu
def test_soak_free_running_with_retunes_and_response_swaps():
    """scripts/soak.py, short: 40,000 blocks of config 3 free-running over 4 streams with channels retuning and responses being
    swapped all the way; the last block must equal, bit for bit, what a fresh engine with the final settings produces."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak.py"), "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["final_block_bit_identical_to_fresh_engine"] is True and j["blocks"] == 40008


_TINY_BUDGET = r"""
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
L, M = 25920, 6481
eng = pkg.engine.Engine(L, M, pkg.engine.REAL, ring_blocks=8)
x = np.random.default_rng(0).standard_normal(8 * L).astype(np.float32)
eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
b = eng.bank(300, 240, 64); b.set_responses(0, np.ones((64, 300), np.complex64) / 300); b.set_shifts(0, np.arange(64, dtype=np.int32) * 50 + 100); b.set_active(64)
eng.set_notches([0, 7], 0.01)
t0 = time.time()
try:
    eng.run_blocks(0, 4000)
    print("RESULT ok %%.2f" %% (time.time() - t0))
except pkg.engine.ChzError as ex:
    print("RESULT failed %%.2f %%s" %% (time.time() - t0, str(ex)[:80]))
"""


@pytest.mark.gpu
def test_a_notch_wait_that_runs_out_fails_fast_and_loudly():
    """With a wait budget of a microsecond some block's notch kernel will not see its predecessor in time: the run must then
    END quickly (the tombstone lets everything queued behind give up at once) and SAY so -- never hang, never publish a wrong
    recurrence.  (Should every wait happen to be shorter than the budget, the run simply succeeds.)"""
    env = dict(os.environ, CHZ_NOTCH_WAIT_MS="0.001")
    r = subprocess.run([sys.executable, "-c", _TINY_BUDGET % (ROOT, ROOT)], capture_output=True, text=True, timeout=120, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, (r.stdout[-300:], r.stderr[-800:])
    kind, secs = line[0].split()[1], float(line[0].split()[2])
    assert secs < 20.0
    if kind == "failed":
        assert "ordering failed" in line[0]


@pytest.mark.gpu
def test_channels_sharing_response_rows(pkg):
    """chz_bank_create_shared: 3 response rows for 48 channels.  Bit-identical to an ordinary bank given the same responses
    channel by channel; a channel re-pointed to another row (no drain) follows from its next block on; per-channel
    chz_bank_set_responses is refused."""
    L, M, P, olen = 25920, 6481, 300, 240
    N = L + M - 1
    nch = 48
    rng = np.random.default_rng(4)
    rows3 = np.stack([pkg.filterapi.design_response(P, olen, N, True, lo, hi, 11.0) for lo, hi in ((0.004, 0.25), (-0.02, 0.02), (-0.4, 0.4))]).astype(np.complex64)
    which = (np.arange(nch) % 3).astype(np.int32)
    shifts = rng.integers(-12000, 12000, nch).astype(np.int32)
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    try:
        shared = eng.bank(P, olen, nch, shared_rows=3); plain = eng.bank(P, olen, nch)
        shared.set_row_responses(0, rows3); shared.set_rows(0, which); shared.set_shifts(0, shifts); shared.set_active(nch)
        plain.set_responses(0, rows3[which]); plain.set_shifts(0, shifts); plain.set_active(nch)
        with pytest.raises(pkg.engine.ChzError):
            shared.set_responses(0, rows3[:1])
        with pytest.raises(pkg.engine.ChzError):
            shared.set_rows(0, np.array([3], np.int32))
        x = (rng.standard_normal(8 * L) * 0.05).astype(np.float32)
        eng.write(x[:8 * L - (M - 1)]); eng.write(x[8 * L - (M - 1):])
        eng.run_blocks(0, 6)
        assert np.array_equal(shared.read_slot(5 % 4).view(np.uint32), plain.read_slot(5 % 4).view(np.uint32))
        which[5] = 2; which[17] = 0
        shared.set_rows(5, which[5:6]); shared.set_rows(17, which[17:18])          # like a retune: in stream order, nothing drains
        plain.set_responses(5, rows3[which[5:6]]); plain.set_responses(17, rows3[which[17:18]])
        eng.run_blocks(6, 7)
        assert np.array_equal(shared.read_slot(12 % 4).view(np.uint32), plain.read_slot(12 % 4).view(np.uint32))
        assert np.abs(shared.read_slot(12 % 4)).max() > 0
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lin_path", ["lanes", "wave"])     # demod_lin_lanes (what banks of >= 65536 channels get) / demod_linear_tail
def test_demodulators_random_parameter_sweep_on_the_device(pkg, monkeypatch, lin_path):
    monkeypatch.setenv("CHZ_DEMOD_WAVE", "1" if lin_path == "wave" else "0")
    """The 24 randomly configured channels of tests/test_kernels_emulated.py::random_demod_channels (linear and FM mixed in one bank,
    all encodings, PLLs, tone squelch) on the device: blocks handed to the demodulator stage through chz_bank_write_block +
    chz_bank_demod, frames against the restated demodulators."""
    from test_kernels_emulated import random_demod_channels
    from test_oracle_vs_reference import _cmp_pcm
    nblk, N = 30, 240
    params, oracles, bbs, powers, ests = random_demod_channels(424242, nblk, N)
    nch = len(params)
    eng = pkg.engine.Engine(25920, 6481, ol.REAL, ring_blocks=8)
    try:
        bank = eng.bank(300, N, nch)
        bank.set_responses(0, np.ones((nch, 300), np.complex64) / 300)
        bank.set_tuning(0, 0, np.full(nch, 2500, np.int32), np.zeros(nch))
        bank.set_active(nch); bank.enable_noise(1.296e6); bank.set_pcm_stride(8 * N)
        bank.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(p, f) for f, _ in ol.LinParams._fields_]) for p in params], 0.02)
        bank.demod_auto(False)
        for b in range(nblk):
            slot = b % 4
            bank.inject(slot, np.stack([bbs[i][b] for i in range(nch)]), np.array([powers[i][b] for i in range(nch)]), np.array([ests[i][b] for i in range(nch)]))
            bank.demod_only(b)
            pcm, status = bank.read_pcm(slot)
            for i, p in enumerate(params):
                want, st = oracles[i].block(bbs[i][b], powers[i][b], ests[i][b], 0.02)
                got = status[i]
                strict = not (p.pll_enable and b >= 20)            # (a PLL without a carrier is chaotic; see test_coherent_modes...)
                assert (got.frame, got.mute, got.squelch_state) == (st.frame, st.mute, st.squelch_state), (b, i)
                if strict:
                    assert got.output_power == pytest.approx(st.output_power, rel=1e-5, abs=1e-300), (b, i)
                    if st.frame == ol.FRAME_DATA:
                        nb = ol.pcm_bytes(p.encoding, N * p.channels)
                        assert _cmp_pcm(p, pcm[i, :nb], want, 1e-4 if (p.env and p.dc_alpha) else 8e-6), (b, i)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_engine_random_operations_on_the_device(seed):
    """The model-based random walk of tests/engine_fuzz_child.py on the device (there the operations really are asynchronous: retunes
    and response swaps land between blocks in flight)."""
    env = dict(os.environ)
    env.pop("CHZ_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "engine_fuzz_child.py"), str(seed), "300"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "FUZZ ok" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])


@pytest.mark.gpu
@pytest.mark.parametrize("queues", ["1", "2", "4"])
def test_stream_set_up_survives_null_stream_traffic(queues):
    """scripts/null_stream_soak.py: what bench.py runs in one process (free-running lanes with the ticketed notch, staged descriptor
    refreshes, response swaps, the 8f chain of all three modes with its demodulator and PCM copy streams), twice, while another thread
    hammers the legacy null stream through torch.  CHZ_OWN_QUEUES=1 (shipped: the two tail streams CU-masked = BLOCKING streams),
    =2 (every lane masked: round 5's run that never came back -- the notch is now ordered by HIP events there, always) and =4 (priority
    streams, non-blocking).  A deadlock shows up as the time-out."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "null_stream_soak.py")], capture_output=True, text=True, timeout=420,
                       env=dict(os.environ, CHZ_OWN_QUEUES=queues))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(out["rounds"]) == 2 and out["null_stream_ops"] > 100, out
