"""Host-side mirror of the reference's filter.h interface, driving the HIP engine.

Same names, argument meaning and error behaviour as src/filter.h:99-118 so the
parity tests read like code written against the reference:

    master = create_filter_input(L, M, REAL)               # src/filter.c:186
    slave  = create_filter_output(master, olen, COMPLEX)   # src/filter.c:298
    set_filter(slave, low, high, kaiser_beta)              # src/filter.c:968
    write_rfilter(master, samples)                         # src/filter.c:1114
    execute_filter_output(slave, shift); slave.output      # src/filter.c:663

Everything numeric on the hot path (forward transform, gather x response, backward
transform) runs in the gfx950 kernels through the C ABI (ka9q-radio_amd/engine.py).
set_filter is host work in the reference as well (rare, tiny) and is float64 numpy
here.  All slaves of a master that share (P, olen) form one device "bank" and are
executed by ONE launch per block: the first execute_filter_output of a block runs the
whole bank with every slave's last-known shift (speculation);
a slave that arrives with a different shift triggers a re-run of the bank.
"""
import numpy as np

from . import engine as _eng

NONE, COMPLEX, REAL, SPECTRUM = 0, 1, 2, 3   # enum filtertype, src/filter.h:29-34
ND = 4                                       # src/filter.h:48


# ----------------------------------------------------------------------------
# filter design (host side; src/filter.c:968-1045, src/window.c:217-254, src/misc.c:416-427)
# ----------------------------------------------------------------------------
def i0(z):
    """Modified Bessel function I0: power series, <= 40 terms, 1e-12 relative stop."""
    t = 0.25 * z * z
    term, s = t, 1.0 + t
    for k in range(2, 40):
        term *= t / (k * k)
        s += term
        if term < 1e-12 * s:
            break
    return s


def make_kaiserf(M, beta):
    w = np.empty(M, np.float32)
    inv = 1.0 / i0(beta)
    pc = 2.0 / (M - 1)
    for n in range(M // 2):
        p = pc * n - 1.0
        v = np.float32(i0(beta * np.sqrt(1.0 - p * p)) * inv)
        w[n] = v
        w[M - 1 - n] = v
    if M & 1:
        w[(M - 1) // 2] = 1.0
    return w


def design_response(P, olen, master_points, master_real, low, high, kaiser_beta, out_type=COMPLEX):
    """Frequency response exactly as set_filter leaves it in slave->response."""
    if out_type == REAL:
        low, high = abs(low), abs(high)
    if low > high:
        low, high = high, low
    low = min(max(low, -0.5), 0.5)
    high = min(max(high, -0.5), 0.5)
    M = P - olen + 1
    if M < 2:
        return None
    bw2 = 0.0001 if high == low else abs(high - low) / 2
    center = (high + low) / 2
    win = make_kaiserf(M, kaiser_beta)
    g = float(np.sum(win.astype(np.float64)))
    win = win * np.float32(M / g)                                       # normalize_windowf
    n = np.arange(M, dtype=np.float64) - (M - 1) / 2
    r = win.astype(np.float64) * 2 * bw2 * np.sinc(2 * bw2 * n)         # np.sinc(x) = sin(pi x)/(pi x)
    wsum = float(np.sum(r))
    ph = 2 * center * n                                                 # half-turns
    taps = ((np.cos(np.pi * ph) + 1j * np.sin(np.pi * ph)) * r).astype(np.complex64)
    gain = (np.sqrt(2.0) if master_real else 1.0) / (wsum * master_points)
    taps = (taps.astype(np.complex128) * gain).astype(np.complex64)
    full = np.zeros(P, np.complex128)
    full[:M] = taps
    return np.fft.fft(full).astype(np.complex64)


# ----------------------------------------------------------------------------
# master / slave objects with the reference's caller-visible fields
# ----------------------------------------------------------------------------
class FilterIn:
    """struct filter_in (src/filter.h:49-74), caller-visible subset."""

    def __init__(self):
        self.init = False


class FilterOut:
    """struct filter_out (src/filter.h:76-97), caller-visible subset."""

    def __init__(self):
        self.init = False
        self.master = None


class _BankState:
    def __init__(self, bank):
        self.bank = bank
        self.slaves = []
        self.shifts = np.zeros(bank.capacity, np.int32)
        self.result_job = None
        self.results = None
        self.dirty_shifts = True
        self.isb = np.zeros(bank.capacity, np.uint8)          # slave->isb flags as uploaded
        self.beam = [None] * bank.capacity                    # (alpha, beta) as uploaded, None = beam off


def create_filter_input(L, M, in_type, device=0, plan="", master=None):
    """src/filter.c:186-269.  Returns a FilterIn, or None where the reference returns -1."""
    if in_type not in (REAL, COMPLEX):
        return None                      # SPECTRUM / NONE are not valid input types (:228-234)
    N = L + M - 1
    bins = N if in_type == COMPLEX else N // 2 + 1
    if L <= 0 or M <= 0 or bins < 2:
        return None
    m = master if master is not None else FilterIn()
    if m.init and m.ilen == L and m.impulse_length == M and m.in_type == in_type:
        return m                         # nothing changed (:191-192)
    if m.init:
        m._engine.close()
    m._engine = _eng.Engine(L, M, in_type, device=device, plan=plan)
    m.in_type, m.points, m.ilen, m.bins, m.impulse_length = in_type, N, L, bins, M
    m.wcnt = 0
    m.next_jobnum = 0
    m.sample_index = 0
    m.samples_by_job = [0] * ND
    m.completed_jobs = [None] * ND
    m.notches = None
    m._banks = {}
    m.init = True
    return m


def delete_filter_input(master):
    if master is None:
        return -1
    if getattr(master, "init", False):
        master._engine.close()
    master.__dict__.clear()
    master.init = False
    return 0


def set_notches(master, bins, alpha=0.01):
    """Equivalent of radio.c filling master->notches (src/radio.c:601-620); DC last."""
    master.notches = list(bins)
    master._engine.set_notches(bins, alpha)


def execute_filter_input(master):
    """src/filter.c:558-651: transform the next window into slot jobnum % ND."""
    if master is None:
        return -1
    job = master.next_jobnum
    master.next_jobnum = (master.next_jobnum + 1) & 0xFFFFFFFF
    master.samples_by_job[job % ND] = master.sample_index
    master.sample_index += master.ilen
    master._engine.forward(job)
    master.completed_jobs[job % ND] = job
    return 0


def _write(master, samples, dtype):
    if master is None:
        return -1
    samples = np.ascontiguousarray(samples, dtype)
    size = samples.shape[0]
    per = 4 if dtype == np.float32 else 8
    if (master.wcnt + size) * per >= master._engine.ring_blocks * master.ilen * per:
        return -1                        # would overrun the ring (:1096-1097,:1117-1118)
    executed = 0
    # hand the samples over block by block so a forward transform never sees a
    # half-written window
    pos = 0
    while pos < size:
        take = min(size - pos, master.ilen - master.wcnt)
        master._engine.write(samples[pos:pos + take])
        pos += take
        master.wcnt += take
        if master.wcnt >= master.ilen:
            master.wcnt -= master.ilen
            execute_filter_input(master)
            executed = 1
    return executed


def write_rfilter(master, samples):
    """src/filter.c:1114-1134 (returns 1 if a block was transformed, 0 if not, -1 on overrun)."""
    return _write(master, samples, np.float32)


def write_cfilter(master, samples):
    """src/filter.c:1093-1113."""
    return _write(master, samples, np.complex64)


def create_filter_output(master, olen, out_type, slave=None):
    """src/filter.c:298-415.  None where the reference returns -1."""
    if master is None or (out_type != SPECTRUM and olen <= 0):
        return None
    s = slave if slave is not None else FilterOut()
    if s.init and s.master is master and s.olen == olen and s.out_type == out_type:
        s.next_jobnum = master.next_jobnum
        return s
    if out_type == SPECTRUM:
        olen = 0
    N, L = master.points, master.ilen
    if (olen * N) % L != 0:
        return None                      # "Invalid filter output length" (:313-316)
    s.master, s.out_type, s.olen = master, out_type, olen
    s.points = olen * N // L
    s.block_drops = 0
    s.sample_index = 0
    s.response = None
    s.output = None
    s.isb = False
    s.beam = False
    set_filter_weights(s, 1.0, 0.0)                         # defaults select the A input only (:341)
    if out_type in (COMPLEX, REAL):
        real = out_type == REAL
        if real and s.points % 2:
            return None                  # the c2r kernels need an even block size
        s.bins = s.points // 2 + 1 if real else s.points          # (:346,374)
        key = (s.points, olen, real)
        st = master._banks.get(key)
        if st is None or len(st.slaves) >= st.bank.capacity:
            cap = 64 if st is None else st.bank.capacity * 2
            new = _BankState(master._engine.bank(s.points, olen, cap, real=real))
            if st is not None:            # grow: move the existing slaves over
                for old in st.slaves:
                    old._bank, old._index = new, len(new.slaves)
                    new.slaves.append(old)
                    if old.response is not None:
                        new.bank.set_responses(old._index, old.response)
                new.shifts[:len(st.slaves)] = st.shifts[:len(st.slaves)]
                new.isb[:len(st.slaves)] = st.isb[:len(st.slaves)]
                if new.isb.any():
                    new.bank.set_isb(0, new.isb[:len(st.slaves)])
                st.bank.destroy()         # the old, smaller device bank
            master._banks[key] = st = new
        s._bank, s._index = st, len(st.slaves)
        st.slaves.append(s)
        st.bank.set_active(len(st.slaves))
        st.result_job = None
    elif out_type == SPECTRUM:
        s.bins = 0                        # block clock only, no buffers (:368-371)
        s._bank = None
    else:
        return None
    s.next_jobnum = master.next_jobnum
    s.init = True
    return s


def delete_filter_output(slave):
    if slave is None:
        return -1
    st = getattr(slave, "_bank", None)
    if st is not None and slave in st.slaves:
        # compact the bank: the last slave takes the freed index
        last = st.slaves[-1]
        idx = slave._index
        if last is not slave:
            st.slaves[idx] = last
            last._index = idx
            if last.response is not None:
                st.bank.set_responses(idx, last.response)
            st.shifts[idx] = st.shifts[len(st.slaves) - 1]
            st.dirty_shifts = True
            st.isb[idx] = st.isb[len(st.slaves) - 1]
            if st.isb.any():
                st.bank.set_isb(0, st.isb[:len(st.slaves)])
        st.slaves.pop()
        st.bank.set_active(len(st.slaves))
        st.result_job = None
    slave.__dict__.clear()
    slave.init = False
    slave.master = None
    return 0


def set_filter_weights(slave, i_weight, q_weight):
    """src/filter.c:922-929."""
    if slave is None:
        return -1
    slave.alpha = 0.5 * complex(i_weight) - 1j * complex(q_weight)
    slave.beta = 0.5 * complex(i_weight) + 1j * complex(q_weight)
    return 0


def set_filter(slave, low, high, kaiser_beta):
    """src/filter.c:968-1045."""
    if slave is None or not getattr(slave, "init", False) or slave.master is None:
        return -1
    if np.isnan(low) or np.isnan(high) or np.isnan(kaiser_beta):
        return -1
    resp = design_response(slave.points, slave.olen, slave.master.points, slave.master.in_type == REAL,
                           low, high, kaiser_beta, slave.out_type)
    if resp is None:
        return -1
    return set_response(slave, resp)


def set_response(slave, resp):
    """Install an arbitrary response (what callers writing slave->response directly do)."""
    slave.response = np.ascontiguousarray(resp, np.complex64)
    if slave._bank is not None:
        slave._bank.bank.set_responses(slave._index, slave.response)
        slave._bank.result_job = None
    return 0


def execute_filter_output(slave, shift):
    """src/filter.c:663-921.  0 on success (slave.output holds olen samples), -1 on bad args."""
    if slave is None:
        return -1
    master = slave.master
    if master is None:
        return -1
    # single-threaded host mirror == the reference's same-thread shortcut (:681-683)
    newest = (master.next_jobnum - 1) & 0xFFFFFFFF
    behind = (newest - slave.next_jobnum) & 0xFFFFFFFF
    if behind >= 0x80000000:
        raise RuntimeError("execute_filter_output would block: no spectrum for this job yet")
    if behind >= ND:                      # lapped: zeros + drop count (:690-701)
        slave.block_drops += 1
        slave.next_jobnum = (slave.next_jobnum + 1) & 0xFFFFFFFF
        if slave.out_type == COMPLEX:
            slave.output = np.zeros(slave.olen, np.complex64)
        elif slave.out_type == REAL:
            slave.output = np.zeros(slave.olen, np.float32)
        return 0
    job = slave.next_jobnum
    slave.sample_index = master.samples_by_job[job % ND]
    slave.next_jobnum = (slave.next_jobnum + 1) & 0xFFFFFFFF
    if slave.out_type == SPECTRUM or slave.response is None:
        return 0                          # (:715-718)
    st = slave._bank
    shift = int(shift)
    if st.shifts[slave._index] != shift:
        st.shifts[slave._index] = shift
        st.dirty_shifts = True
        st.result_job = None
    want_beam = (complex(slave.alpha), complex(slave.beta)) if (slave.out_type == COMPLEX and getattr(slave, "beam", False)
                                                               and master.in_type == COMPLEX) else None
    if want_beam != st.beam[slave._index]:                    # callers set slave->beam and the weights directly (src/radio.c:938-940)
        st.beam[slave._index] = want_beam
        a, b = want_beam if want_beam else (0j, 0j)
        st.bank.set_beam(slave._index, [a], [b], [1 if want_beam else 0])
        st.result_job = None
    if slave.out_type == COMPLEX and bool(getattr(slave, "isb", False)) != bool(st.isb[slave._index]):   # callers set slave->isb directly (src/radio.c:1586)
        st.isb[slave._index] = 1 if slave.isb else 0
        st.bank.set_isb(0, st.isb[:len(st.slaves)])
        st.result_job = None
    if st.result_job != job:
        if st.dirty_shifts:
            st.bank.set_shifts(0, st.shifts[:len(st.slaves)])
            st.dirty_shifts = False
        st.bank.execute(job % ND)
        st.results = st.bank.read(0, len(st.slaves))
        st.result_job = job
    slave.output = st.results[slave._index]
    return 0
