"""Channel sharding across the GPUs of one node (one process per GPU).

The channel axis is the natural data-parallel axis (thread-per-channel in the
reference, src/radio.c:996): slaves never interact, they only read the shared block
spectrum.  One rank owns the front end and runs the forward transform; the spectrum
slot is broadcast (RCCL over xGMI on the GPU, gloo in the CPU tests) and every rank
runs a disjoint, contiguous channel subset.  The loop below is a two-stage software
pipeline: block j's broadcast is in flight while block j+1's forward transform runs,
and block j's channels start as soon as its broadcast has landed.
"""


def shard_channels(total, rank, world):
    """Contiguous, balanced [first, last) channel range of `rank` (ranges tile [0,total))."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def pipelined_blocks(jobs, is_root, forward, broadcast, channels):
    """Run `jobs` (iterable of job numbers).

    forward(job)            root only: transform the block into slot job % 4
    broadcast(job) -> work  every rank: start the (async) broadcast of slot job % 4; the
                            returned object has .wait() which orders later work after it
    channels(job)           every rank: run this rank's channels on slot job % 4
    """
    pending = None
    for job in jobs:
        if is_root:
            forward(job)
        work = broadcast(job)
        if pending is not None:
            pending[0].wait()
            channels(pending[1])
        pending = (work, job)
    if pending is not None:
        pending[0].wait()
        channels(pending[1])
