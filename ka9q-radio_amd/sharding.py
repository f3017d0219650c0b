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


MIN_NOISE_BINS = 1000        # Min_noise_bins, src/radio.c:73


def needed_rows(shifts, P, master_bins, na, margin_rows=1, in_type=2, noise=False):
    """Rows [lo, hi) of the spectrum (rows of `na` bins, see SpecLayout) that the channels with these
    shifts read.  REAL master (in_type 2): bins |shift|-P/2 .. |shift|+P/2, a negative shift reading the same
    bins mirrored (src/filter.c:856-892).  COMPLEX master (in_type 1): a negative shift reads bins
    master_bins+shift at the TOP of the spectrum (src/filter.c:728-793), so the needed set can be two
    intervals; one covering interval cannot describe that, and the exchange must not silently ship the
    wrong rows -- use the whole-slot broadcast for complex masters.
    noise=True: the rank also runs estimate_noise() on these channels, whose window of max(P, 1000) bins around
    |shift| (clamped to the spectrum, src/radio.c:1794-1816) is usually wider than the channel itself."""
    if in_type != 2:
        raise ValueError("needed_rows describes REAL masters only; broadcast the whole slot for a COMPLEX master")
    lo_bin, hi_bin = master_bins, 0
    for s in shifts:
        a, b = abs(int(s)) - P // 2 - 1, abs(int(s)) + (P + 1) // 2 + 1
        lo_bin, hi_bin = min(lo_bin, a), max(hi_bin, b)
        if noise:
            nb = max(P, MIN_NOISE_BINS)
            w = min(max(abs(int(s)) - nb // 2, 0), master_bins - nb)
            lo_bin, hi_bin = min(lo_bin, w), max(hi_bin, w + nb)
    lo_bin, hi_bin = max(lo_bin, 0), min(hi_bin, master_bins)
    if hi_bin <= lo_bin:
        return 0, 0
    nrows = (master_bins + na - 1) // na
    return max(lo_bin // na - margin_rows, 0), min((hi_bin + na - 1) // na + margin_rows, nrows)


def plan_exchange(all_rows, nrows, threshold=0.6):
    """'subband' if every peer needs less than `threshold` of the spectrum rows, else 'broadcast'.
    all_rows: list over ranks of (lo, hi)."""
    if len(all_rows) <= 1:
        return "none"
    worst = max((hi - lo) for lo, hi in all_rows[1:]) if len(all_rows) > 1 else 0
    return "subband" if worst < threshold * nrows else "broadcast"
