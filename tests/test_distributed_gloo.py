"""world_size-2 gloo test of the multi-GPU orchestration (CPU only).

The sharding / pipelining logic bench.py uses at N > 1 (ka9q-radio_amd/sharding.py) is
exercised with the oracle standing in for the device kernels: rank 0 "owns the front
end", the block spectrum is broadcast over gloo, each rank runs its channel shard, and
the concatenated result must equal the single-process answer.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from conftest import load_pkg

L, M, OLEN, P, NCH, NBLK = 11520, 2881, 240, 300, 11, 3


def _inputs():
    rng = np.random.default_rng(2024)
    x = rng.standard_normal(NBLK * L).astype(np.float32)
    shifts = [int(s) for s in rng.integers(-7000, 7000, NCH)]
    resp = [ol.set_filter(P, OLEN, L + M - 1, True, -0.3, 0.3, 11.0) for _ in range(NCH)]
    return x, shifts, resp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    x, shifts, resp = _inputs()
    first, last = pkg.sharding.shard_channels(NCH, rank, world)
    bins = (L + M - 1) // 2 + 1
    slots = [torch.zeros(2 * bins, dtype=torch.float32) for _ in range(4)]
    stream = ol.Stream(L, M, ol.REAL) if rank == 0 else None
    outs = {}

    def forward(job):
        spec = stream.push(x[job * L:(job + 1) * L])
        slots[job % 4].copy_(torch.from_numpy(spec.view(np.float32)))

    def broadcast(job):
        return dist.broadcast(slots[job % 4], src=0, async_op=True)

    def channels(job):
        spec = slots[job % 4].numpy().view(np.complex64)
        outs[job] = [ol.channel(spec, ol.REAL, P, OLEN, shifts[c], resp[c]) for c in range(first, last)]

    pkg.sharding.pipelined_blocks(range(NBLK), rank == 0, forward, broadcast, channels)
    q.put((rank, first, last, {j: np.stack(v) if v else np.zeros((0, OLEN), np.complex64) for j, v in outs.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _subband_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    x, _, resp = _inputs()
    bins = (L + M - 1) // 2 + 1
    na = 45                                             # rows of 45 bins, natural pitch
    # channels sharded by frequency: rank r owns a contiguous band (one negative-shift channel each to
    # exercise the mirrored read)
    shifts = [[300 + 40 * c for c in range(5)] + [-250], [4000 + 55 * c for c in range(4)] + [-5100, bins - 200]][rank]
    mine = pkg.sharding.needed_rows(shifts, P, bins, na)
    all_rows = [None] * world
    dist.all_gather_object(all_rows, mine)
    nrows = (bins + na - 1) // na
    assert pkg.sharding.plan_exchange(all_rows, nrows) == "subband"
    slot = torch.zeros(2 * nrows * na, dtype=torch.float32)
    if rank == 0:
        spec = ol.Stream(L, M, ol.REAL).push(x[:L])
        slot[:2 * bins].copy_(torch.from_numpy(spec.view(np.float32)))
        ops = [dist.P2POp(dist.isend, slot[2 * na * lo:2 * na * hi], r) for r, (lo, hi) in enumerate(all_rows) if r != 0]
    else:
        lo, hi = mine
        ops = [dist.P2POp(dist.irecv, slot[2 * na * lo:2 * na * hi], 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    got = slot.numpy().view(np.complex64)[:bins]
    outs = np.stack([ol.channel(got, ol.REAL, P, OLEN, s, resp[0]) for s in shifts])
    q.put((rank, shifts, outs, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_subband_exchange_delivers_every_bin_a_rank_reads(oracle_built):
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_subband_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, _, resp = _inputs()
    spec = ol.Stream(L, M, ol.REAL).push(x[:L])
    for rank, shifts, outs, mine in results:
        want = np.stack([ol.channel(spec, ol.REAL, P, OLEN, s, resp[0]) for s in shifts])
        np.testing.assert_array_equal(outs, want)
        if rank == 1:
            assert mine[1] - mine[0] < ((L + M - 1) // 2 + 1) // 45      # really a sub-band


def test_needed_rows_and_exchange_plan():
    pkg = load_pkg()
    nr = pkg.sharding.needed_rows
    assert nr([1000], 300, 16201, 45) == (max((1000 - 151) // 45 - 1, 0), (1000 + 151 + 44) // 45 + 1)
    assert nr([-1000], 300, 16201, 45) == nr([1000], 300, 16201, 45)           # inverted spectrum reads the same bins
    assert nr([10, 16190], 300, 16201, 45) == (0, (16201 + 44) // 45)           # clipped to the spectrum
    assert nr([], 300, 16201, 45) == (0, 0)
    # with the noise estimator on, the rank reads max(P, 1000) bins around |shift|, clamped to the spectrum (src/radio.c:1794-1816)
    assert nr([5000], 300, 16201, 45, noise=True) == ((5000 - 500) // 45 - 1, (5000 + 500 + 44) // 45 + 1)
    assert nr([100], 300, 16201, 45, noise=True) == (0, (1000 + 44) // 45 + 1)
    assert nr([16100], 300, 16201, 45, noise=True)[0] == (16201 - 1000) // 45 - 1
    assert pkg.sharding.plan_exchange([(0, 100)], 100) == "none"
    assert pkg.sharding.plan_exchange([(0, 100), (0, 30), (40, 80)], 100) == "subband"
    assert pkg.sharding.plan_exchange([(0, 100), (0, 30), (10, 90)], 100) == "broadcast"


def test_shard_ranges_tile_exactly():
    pkg = load_pkg()
    for total in (0, 1, 7, 1024, 8192, 8191):
        for world in (1, 2, 3, 8):
            edges = [pkg.sharding.shard_channels(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pkg.sharding.shard_channels(4, 2, 2)


def test_two_rank_broadcast_pipeline_matches_single_process(oracle_built):
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda r: r[0])
    assert results[0][1] == 0 and results[-1][2] == NCH and results[0][2] == results[1][1]
    x, shifts, resp = _inputs()
    st = ol.Stream(L, M, ol.REAL)
    for job in range(NBLK):
        spec = st.push(x[job * L:(job + 1) * L])
        want = np.stack([ol.channel(spec, ol.REAL, P, OLEN, shifts[c], resp[c]) for c in range(NCH)])
        got = np.concatenate([r[3][job] for r in results])
        np.testing.assert_array_equal(got, want)


def _samples_worker(rank, world, port, q):
    """SURVEY 8e's alternative hand-over (round 4, chz_run_blocks_sharded mode 2): the block's L new samples travel, every rank
    keeps its own overlap history and runs the forward transform itself."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    x, shifts, resp = _inputs()
    first, last = pkg.sharding.shard_channels(NCH, rank, world)
    stream = ol.Stream(L, M, ol.REAL)                      # EVERY rank has a master: same zero history before the first block
    blocks = [torch.zeros(L, dtype=torch.float32) for _ in range(4)]
    outs = {}
    spectra = {}

    def forward(job):                                       # root only: the A/D delivers block `job` into the root's ring
        blocks[job % 4].copy_(torch.from_numpy(x[job * L:(job + 1) * L]))

    def broadcast(job):
        return dist.broadcast(blocks[job % 4], src=0, async_op=True)

    def channels(job):
        spec = stream.push(blocks[job % 4].numpy())
        spectra[job] = spec
        outs[job] = [ol.channel(spec, ol.REAL, P, OLEN, shifts[c], resp[c]) for c in range(first, last)]

    pkg.sharding.pipelined_blocks(range(NBLK), rank == 0, forward, broadcast, channels)
    q.put((rank, first, last, {j: np.stack(v) if v else np.zeros((0, OLEN), np.complex64) for j, v in outs.items()}, spectra[NBLK - 1]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_exchange_matches_single_process(oracle_built):
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_samples_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda r: r[0])
    x, shifts, resp = _inputs()
    st = ol.Stream(L, M, ol.REAL)
    for job in range(NBLK):
        spec = st.push(x[job * L:(job + 1) * L])
        want = np.stack([ol.channel(spec, ol.REAL, P, OLEN, shifts[c], resp[c]) for c in range(NCH)])
        got = np.concatenate([r[3][job] for r in results])
        np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(results[0][4], results[1][4])     # every rank computed the same spectrum from the same samples
