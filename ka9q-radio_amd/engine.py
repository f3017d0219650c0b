"""ctypes binding of include/chz_engine.h (libchz_hip.so).

There is deliberately no CPU fallback here: if the HIP library is missing or no
MI355X is visible, loading / engine creation raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHZ_LIB") or os.path.join(HERE, "libchz_hip.so")   # CHZ_LIB: A/B builds only

COMPLEX, REAL = 1, 2
ND = 4

_vp, _i, _u, _d, _l = C.c_void_p, C.c_int, C.c_uint, C.c_double, C.c_long


class ChzInfo(C.Structure):
    _fields_ = [("L", _i), ("M", _i), ("N", _i), ("in_type", _i), ("bins", _i), ("ring_blocks", _i),
                ("Na", _i), ("Nb", _i), ("Nc", _i), ("n_banks", _i), ("lanes", _i), ("spec_elems", _l),
                ("spec_na", _i), ("spec_pitch", _i), ("spec_off", _i), ("plan", C.c_char * 320)]


class ChzTiming(C.Structure):
    _fields_ = [("total_ms", _d), ("blocks", _i),
                ("first_ms", _d), ("cols_ms", _d), ("rows_ms", _d), ("notch_ms", _d), ("chan_ms", _d),
                ("first_n", _i), ("cols_n", _i), ("rows_n", _i), ("notch_n", _i), ("chan_n", _i), ("enqueue_ms", _d),
                ("fix_ms", _d), ("fix_n", _i), ("demod_ms", _d), ("demod_n", _i)]


class DemodParams(C.Structure):
    """chz_demod_params: the chan_t members src/linear.c reads (linear amplitudes / power ratios)."""
    _fields_ = [("channels", _i), ("env", _i), ("agc", _i), ("encoding", _i), ("snr_squelch", _i), ("squelch_tail", _i),
                ("tuned", _i), ("kind", _i),
                ("samprate", _d), ("headroom", _d), ("threshold", _d), ("recovery_rate", _d), ("hangtime", _d), ("dc_alpha", _d),
                ("bandwidth", _d), ("shift", _d), ("squelch_open", _d), ("squelch_close", _d), ("gain", _d),
                ("deemph_rate", _d), ("deemph_gain", _d), ("threshold_extend", _d),
                ("pll_enable", _i), ("pll_square", _i), ("pll_loop_bw", _d), ("tone_freq", _d)]


class DemodStatus(C.Structure):
    _fields_ = [("frame", _i), ("mute", _i), ("squelch_state", _i), ("pll_lock", _i),
                ("output_power", _d), ("gain", _d), ("n0", _d), ("snr", _d), ("foffset", _d), ("pdeviation", _d),
                ("pll_snr", _d), ("pll_cphase", _d), ("tone_deviation", _d), ("pll_rotations", _i), ("tone_mute", _i)]


PCM_S16BE, PCM_S16LE, PCM_F32LE, PCM_F32BE, PCM_MULAW, PCM_ALAW, PCM_F16LE, PCM_F16BE = 0, 1, 2, 3, 4, 5, 6, 7


# every symbol include/chz_engine.h declares (checked by tests/test_host_logic.py and __graft_entry__.build())
SYMBOLS = [
    "chz_last_error", "chz_process_exiting", "chz_device_count", "chz_engine_create", "chz_engine_destroy", "chz_engine_info",
    "chz_engine_set_stream", "chz_sync", "chz_input_write", "chz_input_write_device", "chz_input_ring",
    "chz_forward", "chz_slot_stream", "chz_set_notches", "chz_spectrum_read", "chz_spectrum_device", "chz_spectrum_attach",
    "chz_bank_create", "chz_bank_create_shared", "chz_bank_set_rows", "chz_bank_set_row_responses", "chz_bank_set_responses", "chz_bank_set_shifts", "chz_bank_set_active",
    "chz_bank_execute", "chz_bank_execute_range", "chz_bank_destroy", "chz_bank_read", "chz_bank_read_async",
    "chz_spectrum_read_async", "chz_host_callback", "chz_host_alloc", "chz_host_free", "chz_host_register", "chz_host_unregister",
    "chz_bank_output_device", "chz_bank_write_block", "chz_bank_demod", "chz_bank_demod_auto", "chz_bank_read_pcm_flags_async", "chz_bank_pcm_wait", "chz_step", "chz_run_blocks", "chz_gather_descriptor",
    "chz_bank_set_tuning", "chz_bank_read_power", "chz_bank_read_power_async",
    "chz_input_write_i16", "chz_input_write_i16_device", "chz_input_stats",
    "chz_bank_enable_noise", "chz_bank_read_noise", "chz_bank_read_noise_async", "chz_bank_create_real", "chz_bank_set_isb", "chz_bank_set_beam",
    "chz_set_notches_alpha", "chz_slot_sync", "chz_engine_check", "chz_input_seek", "chz_input_mark", "chz_input_mark_wait", "chz_engine_notch_order",
    "chz_bank_set_demod", "chz_bank_pcm_stride", "chz_bank_set_pcm_stride", "chz_bank_read_pcm", "chz_bank_read_pcm_async",
    "chz_comm_unique_id", "chz_comm_create", "chz_comm_create_file", "chz_comm_destroy", "chz_comm_rank", "chz_comm_world",
    "chz_mini_create", "chz_mini_destroy", "chz_mini_capacity", "chz_mini_add", "chz_mini_release", "chz_mini_set_response", "chz_mini_execute",
    "chz_comm_barrier", "chz_comm_allreduce_max", "chz_spectrum_broadcast", "chz_spectrum_exchange_rows", "chz_run_blocks_sharded",
    "chz_comm_create_local", "chz_spectrum_broadcast_local", "chz_set_option",
]

# the CHZ_* variables tests and scripts have always used to force a code path or arm a test hook: the shipped library does not read
# them (its options are set through chz_set_option, include/chz_engine.h) -- this mirror translates them whenever an engine, a pool
# or a communicator is created, so that a variable set by a test (monkeypatch.setenv) means what it always meant
ENV_OPTIONS = {"CHZ_CHAN_STAGE": "chan_stage", "CHZ_NOISE_ENERGY": "noise_energy", "CHZ_DEMOD_WAVE": "demod_wave", "CHZ_ENQ_THREADS": "enq_threads",
               "CHZ_GRAPH_BLOCKS": "graph_blocks", "CHZ_NOTCH_FOLD": "notch_fold", "CHZ_NOISE_HINT": "noise_hint", "CHZ_PLL_LANE0": "pll_lane0",
               "CHZ_NOTCH_WAIT_MS": "notch_wait_ms", "CHZ_FAULT_TICKET_SKEW": "fault_ticket_skew", "CHZ_ALLOW_FAULT_INJECTION": "allow_fault_injection",
               "CHZ_LAUNCH_ID": "launch_id"}


def apply_env_options():
    L = lib()
    for env, opt in ENV_OPTIONS.items():
        v = os.environ.get(env)
        _check(L.chz_set_option(opt.encode(), v.encode() if v is not None else None))

_lib = None


def lib():
    """Load libchz_hip.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libchz_hip.so is missing: build it with `make -C ka9q-radio_amd/csrc` "
                               "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        # tests/hipemu can build the engine's HOST code for the CPU (kernels on a fiber emulator) so that its orchestration is
        # testable without a GPU; such a library carries this symbol, and nothing but a test that says so may load it
        if hasattr(L, "chz_emulated_build") and os.environ.get("CHZ_ALLOW_EMULATED_ENGINE") != "1":
            raise RuntimeError("%s is a CPU-emulated TEST build of the engine: refusing to use it as the product "
                               "(there is no CPU fallback)" % LIB_PATH)
        L.chz_last_error.restype = C.c_char_p
        L.chz_set_option.argtypes = [C.c_char_p, C.c_char_p]
        L.chz_device_count.restype = _i
        L.chz_engine_create.argtypes = [C.POINTER(_vp), _i, _i, _i, _i, C.c_char_p, _i]
        L.chz_engine_destroy.argtypes = [_vp]; L.chz_engine_destroy.restype = None
        L.chz_engine_info.argtypes = [_vp, C.POINTER(ChzInfo)]
        L.chz_engine_set_stream.argtypes = [_vp, _vp]
        L.chz_sync.argtypes = [_vp]
        L.chz_input_write.argtypes = [_vp, _vp, _l]
        L.chz_input_write_device.argtypes = [_vp, _vp, _l]
        L.chz_input_ring.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_l)]
        L.chz_input_write_i16.argtypes = [_vp, _vp, _l, C.c_float, _i]
        L.chz_input_write_i16_device.argtypes = [_vp, _vp, _l, C.c_float, _i]
        L.chz_input_stats.argtypes = [_vp, _i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)]
        L.chz_forward.argtypes = [_vp, _u]
        L.chz_set_notches.argtypes = [_vp, _vp, _i, _d]
        L.chz_set_notches_alpha.argtypes = [_vp, _vp, _vp, _i]
        L.chz_slot_sync.argtypes = [_vp, _i]
        L.chz_engine_check.argtypes = [_vp]
        L.chz_bank_set_demod.argtypes = [_vp, _i, _u, _i, _i, _vp, _d]
        L.chz_bank_pcm_stride.argtypes = [_vp, _i]
        L.chz_bank_set_pcm_stride.argtypes = [_vp, _i, _i]
        L.chz_bank_read_pcm.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp]
        L.chz_bank_read_pcm_async.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp]
        L.chz_mini_create.argtypes = [C.POINTER(_vp), _i, _i, _i, _i]
        L.chz_mini_destroy.argtypes = [_vp]; L.chz_mini_destroy.restype = None
        L.chz_mini_capacity.argtypes = [_vp]
        L.chz_mini_add.argtypes = [_vp]
        L.chz_mini_release.argtypes = [_vp, _i]
        L.chz_mini_set_response.argtypes = [_vp, _i, _vp]
        L.chz_mini_execute.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _vp]
        L.chz_comm_unique_id.argtypes = [_vp]
        L.chz_comm_create.argtypes = [C.POINTER(_vp), _i, _i, _vp, _i]
        L.chz_comm_create_file.argtypes = [C.POINTER(_vp), _i, _i, C.c_char_p, _i, _d]
        L.chz_comm_destroy.argtypes = [_vp]; L.chz_comm_destroy.restype = None
        L.chz_comm_rank.argtypes = [_vp]
        L.chz_comm_world.argtypes = [_vp]
        L.chz_comm_barrier.argtypes = [_vp]
        L.chz_comm_allreduce_max.argtypes = [_vp, _vp, _i]
        L.chz_spectrum_broadcast.argtypes = [_vp, _vp, _i, _i]
        L.chz_spectrum_exchange_rows.argtypes = [_vp, _vp, _i, _i, _vp, _vp]
        L.chz_run_blocks_sharded.argtypes = [_vp, _vp, _i, _i, _vp, _vp, _u, _i, C.POINTER(ChzTiming)]
        L.chz_spectrum_read.argtypes = [_vp, _i, _vp]
        L.chz_spectrum_device.argtypes = [_vp, _i, C.POINTER(_vp)]
        L.chz_slot_stream.argtypes = [_vp, _i, C.POINTER(_vp)]
        L.chz_spectrum_attach.argtypes = [_vp, _i, _vp]
        L.chz_bank_create.argtypes = [_vp, _i, _i, _i]
        L.chz_bank_create_real.argtypes = [_vp, _i, _i, _i]
        L.chz_bank_set_isb.argtypes = [_vp, _i, _i, _i, _vp]
        L.chz_bank_set_beam.argtypes = [_vp, _i, _i, _i, _vp, _vp]
        L.chz_bank_set_responses.argtypes = [_vp, _i, _i, _i, _vp]
        L.chz_bank_set_shifts.argtypes = [_vp, _i, _i, _i, _vp]
        L.chz_bank_set_active.argtypes = [_vp, _i, _i]
        L.chz_bank_execute.argtypes = [_vp, _i, _u]
        L.chz_bank_read.argtypes = [_vp, _i, _i, _i, _vp]
        L.chz_bank_execute_range.argtypes = [_vp, _i, _u, _i, _i]
        L.chz_bank_set_tuning.argtypes = [_vp, _i, _u, _i, _i, _vp, _vp, _vp]
        L.chz_bank_read_power.argtypes = [_vp, _i, _i, _i, _i, _vp]
        L.chz_bank_read_power_async.argtypes = [_vp, _i, _i, _i, _i, _vp]
        L.chz_bank_enable_noise.argtypes = [_vp, _i, _d]
        L.chz_bank_read_noise.argtypes = [_vp, _i, _i, _i, _i, _vp]
        L.chz_bank_read_noise_async.argtypes = [_vp, _i, _i, _i, _i, _vp]
        L.chz_bank_destroy.argtypes = [_vp, _i]
        L.chz_bank_read_async.argtypes = [_vp, _i, _i, _i, _i, _vp]
        L.chz_bank_output_device.argtypes = [_vp, _i, _i, C.POINTER(_vp)]
        L.chz_step.argtypes = [_vp, _u]
        L.chz_run_blocks.argtypes = [_vp, _u, _i, _i, _i, C.POINTER(ChzTiming)]
        L.chz_gather_descriptor.argtypes = [_i, _i, _i, _i, C.POINTER(_i * 6)]
        _lib = L
    return _lib


class ChzError(RuntimeError):
    pass


def _check(r):
    if r < 0:
        raise ChzError(lib().chz_last_error().decode())
    return r


def gather_descriptor(in_type, master_bins, P, shift):
    out = (_i * 6)()
    lib().chz_gather_descriptor(in_type, master_bins, P, int(shift), C.byref(out))
    return tuple(out)


class Engine:
    """One master (input half) on one GPU."""

    def __init__(self, L, M, in_type, device=0, plan="", ring_blocks=0):
        self._h = _vp()
        apply_env_options()
        _check(lib().chz_engine_create(C.byref(self._h), L, M, in_type, device,
                                       plan.encode() if plan else None, ring_blocks))
        info = ChzInfo()
        _check(lib().chz_engine_info(self._h, C.byref(info)))
        self.L, self.M, self.N, self.in_type, self.bins = info.L, info.M, info.N, info.in_type, info.bins
        self.ring_blocks = info.ring_blocks
        self.plan = info.plan.decode()
        self.axes = (info.Na, info.Nb, info.Nc)
        self.lanes = info.lanes
        self.spec_elems = info.spec_elems
        self.spec_layout = (info.spec_na, info.spec_pitch, info.spec_off)
        self.banks = []

    def close(self):
        if self._h:
            lib().chz_engine_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- input ---------------------------------------------------------------
    def write(self, samples):
        dt = np.float32 if self.in_type == REAL else np.complex64
        samples = np.ascontiguousarray(samples, dt)
        _check(lib().chz_input_write(self._h, samples.ctypes.data, samples.shape[0]))
        _check(lib().chz_sync(self._h))   # numpy buffer may be freed by the caller

    def write_i16(self, samples, scale, randomize=False):
        """Raw A/D samples (rx888.c's convert() + write_rfilter): int16, scaled on the device."""
        s = np.ascontiguousarray(samples, np.int16).reshape(-1)
        _check(lib().chz_input_write_i16(self._h, s.ctypes.data, s.size, float(scale), 1 if randomize else 0))
        self.sync()       # the caller's buffer may be pageable / reused

    def input_stats(self, slot):
        """(sum x^2, clipped samples) over the new samples of the block last transformed into `slot`."""
        en, cl = C.c_ulonglong(0), C.c_uint(0)
        _check(lib().chz_input_stats(self._h, slot, C.byref(en), C.byref(cl)))
        return en.value, cl.value

    def write_device(self, dev_ptr, n):
        _check(lib().chz_input_write_device(self._h, dev_ptr, n))

    def ring(self):
        p, n = _vp(), _l()
        _check(lib().chz_input_ring(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_stream(self, hip_stream):
        _check(lib().chz_engine_set_stream(self._h, hip_stream))
        info = ChzInfo()
        _check(lib().chz_engine_info(self._h, C.byref(info)))
        self.lanes = info.lanes

    # -- forward ---------------------------------------------------------------
    def forward(self, job):
        _check(lib().chz_forward(self._h, job & 0xFFFFFFFF))

    def set_notches(self, bins, alpha=0.01):
        """alpha: one gain for the whole list, or one per entry (struct notch_state, src/filter.h:42-46)."""
        bins = np.ascontiguousarray(bins, np.int32)
        if np.ndim(alpha) == 0:
            _check(lib().chz_set_notches(self._h, bins.ctypes.data if len(bins) else None, len(bins), float(alpha)))
        else:
            al = np.ascontiguousarray(alpha, np.float64).reshape(-1)
            assert al.shape[0] == bins.shape[0]
            _check(lib().chz_set_notches_alpha(self._h, bins.ctypes.data if len(bins) else None,
                                               al.ctypes.data if len(bins) else None, len(bins)))

    def check(self):
        """Raises once a device-side consistency check (the notch ticket) has failed."""
        _check(lib().chz_engine_check(self._h))

    def slot_sync(self, slot):
        _check(lib().chz_slot_sync(self._h, slot))

    def run_blocks_sharded(self, comm, job0, nblocks, root=0, rows=None, samples=False):
        """BASELINE config 4: the root transforms, the spectrum travels over RCCL (whole slot, or the row ranges
        rows = (lo[world], hi[world])), every rank runs its own banks.  samples=True: the block's new samples travel
        instead and every rank transforms them itself (SURVEY 8e's alternative)."""
        t = ChzTiming()
        if samples:
            _check(lib().chz_run_blocks_sharded(self._h, comm._h, root, 2, None, None, job0 & 0xFFFFFFFF, nblocks, C.byref(t)))
        elif rows is None:
            _check(lib().chz_run_blocks_sharded(self._h, comm._h, root, 0, None, None, job0 & 0xFFFFFFFF, nblocks, C.byref(t)))
        else:
            lo = np.ascontiguousarray(rows[0], np.int32); hi = np.ascontiguousarray(rows[1], np.int32)
            _check(lib().chz_run_blocks_sharded(self._h, comm._h, root, 1, lo.ctypes.data, hi.ctypes.data,
                                                job0 & 0xFFFFFFFF, nblocks, C.byref(t)))
        return t

    def spectrum(self, slot):
        out = np.zeros(self.bins, np.complex64)
        _check(lib().chz_spectrum_read(self._h, slot, out.ctypes.data))
        return out

    def spectrum_ptr(self, slot):
        p = _vp()
        _check(lib().chz_spectrum_device(self._h, slot, C.byref(p)))
        return p.value

    def slot_stream(self, slot):
        p = _vp()
        _check(lib().chz_slot_stream(self._h, slot, C.byref(p)))
        return p.value or 0

    def attach_spectrum(self, slot, dev_ptr):
        _check(lib().chz_spectrum_attach(self._h, slot, dev_ptr))

    # -- banks -----------------------------------------------------------------
    def bank(self, P, olen, capacity, real=False, shared_rows=0):
        b = Bank(self, P, olen, capacity, real, shared_rows)
        self.banks.append(b)
        return b

    def step(self, job):
        _check(lib().chz_step(self._h, job & 0xFFFFFFFF))

    def sync(self):
        _check(lib().chz_sync(self._h))

    def run_blocks(self, job0, nblocks, graph=False, instrument=False):
        t = ChzTiming()
        _check(lib().chz_run_blocks(self._h, job0 & 0xFFFFFFFF, nblocks, 1 if graph else 0,
                                    1 if instrument else 0, C.byref(t)))
        return t


class Bank:
    def __init__(self, eng, P, olen, capacity, real=False, shared_rows=0):
        self.eng, self.P, self.olen, self.capacity, self.real = eng, P, olen, capacity, bool(real)
        L = lib()
        if shared_rows:                                                            # channels name one of `shared_rows` response rows
            L.chz_bank_create_shared.argtypes = [_vp, _i, _i, _i, _i]
            L.chz_bank_set_rows.argtypes = [_vp, _i, _i, _i, _vp]
            L.chz_bank_set_row_responses.argtypes = [_vp, _i, _i, _i, _vp]
            self.id = _check(L.chz_bank_create_shared(eng._h, P, olen, capacity, shared_rows))
        else:
            make = L.chz_bank_create_real if real else L.chz_bank_create          # REAL- or COMPLEX-output slaves
            self.id = _check(make(eng._h, P, olen, capacity))
        self._dtype = np.float32 if real else np.complex64
        self.active = 0

    def destroy(self):
        """Free the bank's device arrays (the id stays reserved)."""
        if self.id is not None and self.eng._h:
            _check(lib().chz_bank_destroy(self.eng._h, self.id))
        self.id = None

    def set_rows(self, ch0, rows):
        r = np.ascontiguousarray(rows, np.int32)
        _check(lib().chz_bank_set_rows(self.eng._h, self.id, ch0, r.shape[0], r.ctypes.data))

    def set_row_responses(self, row0, resp):
        r = np.ascontiguousarray(resp, np.complex64)
        assert r.ndim == 2 and r.shape[1] == self.P
        _check(lib().chz_bank_set_row_responses(self.eng._h, self.id, row0, r.shape[0], r.ctypes.data))

    def set_responses(self, ch0, resp):
        resp = np.ascontiguousarray(resp, np.complex64).reshape(-1, self.P)
        _check(lib().chz_bank_set_responses(self.eng._h, self.id, ch0, resp.shape[0], resp.ctypes.data))

    def set_shifts(self, ch0, shifts):
        shifts = np.ascontiguousarray(shifts, np.int32).reshape(-1)
        _check(lib().chz_bank_set_shifts(self.eng._h, self.id, ch0, shifts.shape[0], shifts.ctypes.data))

    def set_isb(self, ch0, flags):
        """slave->isb per channel (src/filter.c:895-909)."""
        f = np.ascontiguousarray(flags, np.uint8).reshape(-1)
        _check(lib().chz_bank_set_isb(self.eng._h, self.id, ch0, f.shape[0], f.ctypes.data))

    def set_beam(self, ch0, alpha, beta, on):
        """slave->beam with slave->alpha / ->beta per channel (src/filter.c:756-775)."""
        alpha = np.asarray(alpha, np.complex128).reshape(-1); beta = np.asarray(beta, np.complex128).reshape(-1)
        ab = np.ascontiguousarray(np.stack([alpha.real, alpha.imag, beta.real, beta.imag], axis=1), np.float64)
        f = np.ascontiguousarray(on, np.uint8).reshape(-1)
        assert ab.shape[0] == f.shape[0]
        _check(lib().chz_bank_set_beam(self.eng._h, self.id, ch0, f.shape[0], ab.ctypes.data, f.ctypes.data))

    def set_active(self, n):
        _check(lib().chz_bank_set_active(self.eng._h, self.id, n))
        self.active = n

    def execute(self, job):
        """Run the bank on the spectrum of block `job` (slot job % 4; a bare slot number is fine for an untuned bank)."""
        _check(lib().chz_bank_execute(self.eng._h, self.id, job & 0xFFFFFFFF))

    def execute_range(self, job, ch0, n):
        """The same for channels [ch0, ch0+n) only (the slow path of one retuned channel)."""
        _check(lib().chz_bank_execute_range(self.eng._h, self.id, job & 0xFFFFFFFF, ch0, n))

    def set_tuning(self, job, ch0, shifts, freq, rate=None):
        """downconvert()'s tuning update (src/radio.c:1479-1497) taking effect at block `job`:
        shifts[i] bins, freq[i] = -remainder/samprate cycles/sample, rate[i] = doppler_rate/samprate^2."""
        shifts = np.ascontiguousarray(shifts, np.int32).reshape(-1)
        freq = np.ascontiguousarray(freq, np.float64).reshape(-1)
        assert freq.shape == shifts.shape
        rp = None
        if rate is not None:
            rate = np.ascontiguousarray(rate, np.float64).reshape(-1)
            assert rate.shape == shifts.shape
            rp = rate.ctypes.data
        _check(lib().chz_bank_set_tuning(self.eng._h, self.id, job & 0xFFFFFFFF, ch0, shifts.shape[0],
                                         shifts.ctypes.data, freq.ctypes.data, rp))

    def set_demod(self, job, ch0, params, blocktime=0.02):
        """demod_linear()'s per-block work (src/linear.c) for channels ch0.. from block `job`; params: list of DemodParams."""
        arr = (DemodParams * len(params))(*params)
        _check(lib().chz_bank_set_demod(self.eng._h, self.id, job & 0xFFFFFFFF, ch0, len(params), arr, float(blocktime)))

    def set_pcm_stride(self, nbytes):
        _check(lib().chz_bank_set_pcm_stride(self.eng._h, self.id, int(nbytes)))

    def read_pcm(self, slot, ch0=0, n=None):
        """(pcm uint8[n][stride], status DemodStatus[n]) of the block last demodulated on `slot`."""
        if n is None:
            n = self.active - ch0
        stride = _check(lib().chz_bank_pcm_stride(self.eng._h, self.id))
        pcm = np.zeros((n, stride), np.uint8)
        st = (DemodStatus * n)()
        _check(lib().chz_bank_read_pcm(self.eng._h, self.id, slot, ch0, n, pcm.ctypes.data, st))
        return pcm, st

    def inject(self, slot, samples, bb_power=None, n0=None, ch0=0):
        """chz_bank_write_block: complex64[n][olen] (+ float64[n] bb_power, noise estimates) into the slot's output image."""
        x = np.ascontiguousarray(samples, np.complex64)
        pw = None if bb_power is None else np.ascontiguousarray(bb_power, np.float64)
        ne = None if n0 is None else np.ascontiguousarray(n0, np.float64)
        L = lib()
        L.chz_bank_write_block.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp, _vp]
        _check(L.chz_bank_write_block(self.eng._h, self.id, slot, ch0, x.shape[0], x.ctypes.data,
                                      None if pw is None else pw.ctypes.data, None if ne is None else ne.ctypes.data))

    def demod_auto(self, on):
        L = lib()
        L.chz_bank_demod_auto.argtypes = [_vp, _i, _i]
        _check(L.chz_bank_demod_auto(self.eng._h, self.id, 1 if on else 0))

    def demod_only(self, job, slot=None):
        """chz_bank_demod: the demodulator stage alone over what `slot` (default job % 4) holds."""
        L = lib()
        L.chz_bank_demod.argtypes = [_vp, _i, C.c_uint, _i]
        _check(L.chz_bank_demod(self.eng._h, self.id, job, job % 4 if slot is None else slot))

    def read_pcm_flags(self, slot, ch0=0, n=None):
        """(pcm uint8[n][stride], flags uint8[n]): chz_bank_read_pcm_flags_async + chz_sync of the demodulator stream."""
        if n is None:
            n = self.active - ch0
        stride = _check(lib().chz_bank_pcm_stride(self.eng._h, self.id))
        pcm = np.zeros((n, stride), np.uint8); fl = np.zeros(n, np.uint8)
        L = lib()
        L.chz_bank_read_pcm_flags_async.argtypes = [_vp, _i, _i, _i, _i, _vp, _vp]
        _check(L.chz_bank_read_pcm_flags_async(self.eng._h, self.id, slot, ch0, n, pcm.ctypes.data, fl.ctypes.data))
        self.eng.sync()
        return pcm, fl

    def enable_noise(self, samprate):
        """estimate_noise() (src/radio.c:1783-1866) on the device after every block; samprate = front-end rate in Hz."""
        _check(lib().chz_bank_enable_noise(self.eng._h, self.id, float(samprate)))

    def read_noise(self, slot, ch0=0, n=None):
        if n is None:
            n = self.active - ch0
        out = np.zeros(n, np.float64)
        _check(lib().chz_bank_read_noise(self.eng._h, self.id, slot, ch0, n, out.ctypes.data))
        return out

    def read_power(self, slot, ch0=0, n=None):
        """chan->sig.bb_power of the block last executed on `slot` (src/radio.c:1516-1520)."""
        if n is None:
            n = self.active - ch0
        out = np.zeros(n, np.float64)
        _check(lib().chz_bank_read_power(self.eng._h, self.id, slot, ch0, n, out.ctypes.data))
        return out

    def read(self, ch0=0, n=None):
        if n is None:
            n = self.active - ch0
        out = np.zeros((n, self.olen), self._dtype)
        _check(lib().chz_bank_read(self.eng._h, self.id, ch0, n, out.ctypes.data))
        return out

    def read_slot(self, slot, ch0=0, n=None):
        """Outputs of the block last executed on spectrum slot `slot` (every slot keeps its own image)."""
        if n is None:
            n = self.active - ch0
        out = np.zeros((n, self.olen), self._dtype)
        _check(lib().chz_bank_read_async(self.eng._h, self.id, slot, ch0, n, out.ctypes.data))
        self.eng.sync()
        return out

    def output_ptr(self, slot=0):
        p = _vp()
        _check(lib().chz_bank_output_device(self.eng._h, self.id, slot, C.byref(p)))
        return p.value


COMM_ID_BYTES = 128


def comm_unique_id():
    """Rank 0: the 128-byte id every rank passes to Comm()."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(lib().chz_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """One RCCL communicator per process (one process per GPU), behind the engine's C ABI."""

    def __init__(self, rank, world, uid=None, device=0, path=None, timeout_s=120.0):
        self._h = _vp()
        if path is not None:
            apply_env_options()
            _check(lib().chz_comm_create_file(C.byref(self._h), rank, world, path.encode(), device, float(timeout_s)))
        else:
            buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(uid)
            _check(lib().chz_comm_create(C.byref(self._h), rank, world, buf, device))
        self.rank, self.world = rank, world

    def barrier(self):
        _check(lib().chz_comm_barrier(self._h))

    def allreduce_max(self, values):
        v = np.ascontiguousarray(values, np.float64).reshape(-1).copy()
        _check(lib().chz_comm_allreduce_max(self._h, v.ctypes.data, v.shape[0]))
        return v

    def broadcast_spectrum(self, eng, slot, root=0):
        _check(lib().chz_spectrum_broadcast(eng._h, self._h, slot, root))

    def exchange_rows(self, eng, slot, lo, hi, root=0):
        lo = np.ascontiguousarray(lo, np.int32); hi = np.ascontiguousarray(hi, np.int32)
        _check(lib().chz_spectrum_exchange_rows(eng._h, self._h, slot, root, lo.ctypes.data, hi.ctypes.data))

    def close(self):
        if self._h:
            lib().chz_comm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MiniPool:
    """Pool of small inline masters of one geometry (radiod's filter2, src/radio.c:1572-1594)."""

    def __init__(self, L, M, capacity, device=0):
        self._h = _vp()
        _check(lib().chz_mini_create(C.byref(self._h), L, M, capacity, device))
        self.L, self.M, self.N, self.capacity = L, M, L + M - 1, capacity

    def add(self):
        return _check(lib().chz_mini_add(self._h))

    def release(self, inst):
        _check(lib().chz_mini_release(self._h, inst))

    def set_response(self, inst, resp):
        r = np.ascontiguousarray(resp, np.complex64).reshape(-1)
        assert r.shape[0] == self.N
        _check(lib().chz_mini_set_response(self._h, inst, r.ctypes.data))

    def execute(self, insts, windows, shifts=None, isb=None):
        """windows: [n][N] complex64 (M-1 old + L new samples each) -> [n][L] complex64."""
        insts = np.ascontiguousarray(insts, np.int32)
        n = insts.shape[0]
        win = np.ascontiguousarray(windows, np.complex64).reshape(n, self.N)
        out = np.zeros((n, self.L), np.complex64)
        wp = (_vp * n)(*[win[i].ctypes.data for i in range(n)])
        op = (_vp * n)(*[out[i].ctypes.data for i in range(n)])
        sh = np.ascontiguousarray(shifts, np.int32) if shifts is not None else None
        fl = np.ascontiguousarray(isb, np.uint8) if isb is not None else None
        _check(lib().chz_mini_execute(self._h, n, insts.ctypes.data, wp, sh.ctypes.data if sh is not None else None,
                                      fl.ctypes.data if fl is not None else None, op))
        return out

    def close(self):
        if self._h:
            lib().chz_mini_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
