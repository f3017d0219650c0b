"""ctypes bindings for the CPU oracle (TEST INFRASTRUCTURE).

Two libraries live under oracle/:
  liboracle.so          project restatement (oracle/chz_oracle.c, oracle/dft.c)
  _ref/libka9q_ref.so   the reference's own filter.c & friends, compiled unmodified
                        from /root/reference/src by oracle/Makefile (prebuilt file
                        travels to the GPU box; /root/reference itself does not)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

COMPLEX, REAL, SPECTRUM = 1, 2, 3

_vp, _i, _d, _u64 = C.c_void_p, C.c_int, C.c_double, C.c_uint64


def build(force=False):
    """(Re)build liboracle.so always-if-stale, and _ref when /root/reference exists."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"] + (["-B"] if force else []),
                   check=True, stdout=subprocess.DEVNULL)


def _fptr(a):
    return a.ctypes.data_as(C.c_void_p)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.chzo_i0.restype = _d; L.chzo_i0.argtypes = [_d]
        L.chzo_make_kaiser.argtypes = [_vp, _i, _d]
        L.chzo_set_filter.argtypes = [_i, _i, _i, _i, _i, _d, _d, _d, _vp]
        L.chzo_forward.argtypes = [_vp, _i, _i, _vp]
        L.chzo_forward_f64.argtypes = [_vp, _i, _i, _vp]
        L.chzo_notch.argtypes = [_vp, _vp, _i, _d, _vp]
        L.chzo_gather.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]
        L.chzo_channel.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]
        L.chzo_channel_f64.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]
        L.chzo_siggen_create.restype = _vp
        L.chzo_siggen_create.argtypes = [_d, _d, _d, _d, _i, _u64]
        L.chzo_siggen_delete.argtypes = [_vp]
        L.chzo_siggen_generate.argtypes = [_vp, _vp, C.c_long]
        L.chzo_scale_ad.restype = _d; L.chzo_scale_ad.argtypes = [_d, _d, _d, _i, _i]
        L.chzo_compute_tuning.argtypes = [_i, _d, _d, _vp, _vp]
        L.chzo_stream_create.restype = _vp; L.chzo_stream_create.argtypes = [_i, _i, _i]
        L.chzo_stream_delete.argtypes = [_vp]
        L.chzo_stream_bins.argtypes = [_vp]
        L.chzo_stream_push.argtypes = [_vp, _vp, _vp]
        L.chzo_stream_push_f64.argtypes = [_vp, _vp, _vp]
        L.chzo_gather_beam.argtypes = [_vp, _i, _i, _i, _vp, _d, _d, _d, _d, _vp]
        L.chzo_channel_beam.argtypes = [_vp, _i, _i, _i, _i, _vp, _d, _d, _d, _d, _vp]
        L.chzo_estimate_noise.restype = _d
        L.chzo_estimate_noise.argtypes = [_vp, _i, _i, _i, _i, _d]
        L.chzo_convert_i16.argtypes = [_vp, _i, C.c_float, _i, _vp, _vp]
        L.chzo_downconv_create.restype = _vp
        L.chzo_downconv_delete.argtypes = [_vp]
        L.chzo_downconv_block.restype = _d
        L.chzo_downconv_block.argtypes = [_vp, _i, _d, _d, _d, _i, _i, _vp, _i]
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref.so"))


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref.so"))
        L.refchz_master_create.restype = _vp; L.refchz_master_create.argtypes = [_i, _i, _i, _i]
        L.refchz_master_delete.argtypes = [_vp]
        L.refchz_master_bins.argtypes = [_vp]
        L.refchz_master_points.argtypes = [_vp]
        L.refchz_master_next_jobnum.restype = C.c_uint; L.refchz_master_next_jobnum.argtypes = [_vp]
        L.refchz_master_set_notches.argtypes = [_vp, _vp, _i, _d]
        L.refchz_master_write.argtypes = [_vp, _vp, _i]
        L.refchz_master_spectrum.argtypes = [_vp, C.c_uint, _vp]
        L.refchz_chan_create.restype = _vp; L.refchz_chan_create.argtypes = [_vp, _i, _i]
        L.refchz_chan_delete.argtypes = [_vp]
        L.refchz_chan_points.argtypes = [_vp]
        L.refchz_chan_bins.argtypes = [_vp]
        L.refchz_chan_drops.restype = C.c_uint; L.refchz_chan_drops.argtypes = [_vp]
        L.refchz_chan_set_isb.argtypes = [_vp, _i]
        L.refchz_chan_set_filter.argtypes = [_vp, _d, _d, _d]
        L.refchz_chan_response.argtypes = [_vp, _vp]
        L.refchz_chan_set_response.argtypes = [_vp, _vp]
        L.refchz_chan_execute.argtypes = [_vp, _i, _vp]
        L.refchz_chan_fdomain.argtypes = [_vp, _vp]
        L.refchz_chan_set_beam.argtypes = [_vp, _d, _d, _d, _d]
        L.refchz_chan_weights.argtypes = [_vp, _vp]
        L.refchz_make_kaiserf.argtypes = [_vp, _i, _d]
        L.refchz_i0.restype = _d; L.refchz_i0.argtypes = [_d]
        L.refchz_bench.restype = _d
        L.refchz_bench.argtypes = [_vp, _vp, _vp, _i, _vp, _i, _i, _i]
        if hasattr(L, "refchz_bench_blocks"):
            L.refchz_bench_blocks.restype = _d
            L.refchz_bench_blocks.argtypes = [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i]
        if hasattr(L, "refchz_reset_drops"):
            L.refchz_reset_drops.restype = None
            L.refchz_reset_drops.argtypes = [_vp, _i]
        L.refchz_fft_times.argtypes = [_vp, _vp, _vp]
        L.refsig_create.restype = _vp; L.refsig_create.argtypes = [_d, _d, _d, _d, _i, C.c_ulonglong]
        L.refsig_delete.argtypes = [_vp]
        L.refsig_generate.argtypes = [_vp, _vp, C.c_long]
        L.oracle_fft_set_precision.argtypes = [_i]
        L.refdc_create.restype = _vp
        L.refdc_delete.argtypes = [_vp]
        L.refdc_block.restype = _d
        L.refdc_block.argtypes = [_vp, _i, _d, _d, _d, _i, _i, _vp, _i]
        _ref = L
    return _ref


# ----------------------------------------------------------------------------
# numpy-level conveniences over the restatement
# ----------------------------------------------------------------------------

def set_filter(P, olen, master_points, master_real, low, high, beta, out_type=COMPLEX):
    resp = np.zeros(P, np.complex64)
    r = oracle().chzo_set_filter(P, olen, master_points, int(master_real), out_type,
                                 low, high, beta, _fptr(resp))
    if r != 0:
        raise ValueError("chzo_set_filter failed")
    return resp


def forward(window, in_type, f64=False):
    window = np.ascontiguousarray(window, np.float32 if in_type == REAL else np.complex64)
    N = window.shape[0]
    bins = N // 2 + 1 if in_type == REAL else N
    if f64:
        out = np.zeros(bins, np.complex128)
        oracle().chzo_forward_f64(_fptr(window), N, in_type, _fptr(out))
    else:
        out = np.zeros(bins, np.complex64)
        oracle().chzo_forward(_fptr(window), N, in_type, _fptr(out))
    return out


def channel(spectrum, in_type, P, olen, shift, response, out_type=COMPLEX, isb=False):
    response = np.ascontiguousarray(response, np.complex64)
    if spectrum.dtype == np.complex128:
        out = np.zeros(olen, np.complex128)
        r = oracle().chzo_channel_f64(_fptr(spectrum), spectrum.shape[0], in_type, P, olen, out_type,
                                      int(shift), int(isb), _fptr(response), _fptr(out))
    else:
        spectrum = np.ascontiguousarray(spectrum, np.complex64)
        out = np.zeros(olen, np.complex64 if out_type == COMPLEX else np.float32)
        r = oracle().chzo_channel(_fptr(spectrum), spectrum.shape[0], in_type, P, olen, out_type,
                                  int(shift), int(isb), _fptr(response), _fptr(out))
    if r != 0:
        raise ValueError("chzo_channel failed")
    return out


def gather(spectrum, in_type, s_bins, shift, response, out_type=COMPLEX, isb=False):
    spectrum = np.ascontiguousarray(spectrum, np.complex64)
    response = np.ascontiguousarray(response, np.complex64)
    fd = np.zeros(s_bins + 1, np.complex64)
    r = oracle().chzo_gather(_fptr(spectrum), spectrum.shape[0], in_type, s_bins, out_type,
                             int(shift), int(isb), _fptr(response), _fptr(fd))
    if r != 0:
        raise ValueError("chzo_gather failed")
    return fd[:s_bins]


class SigGen:
    """Deterministic sig_gen stream (restatement)."""

    def __init__(self, cycles_per_sample, amplitude, noise, scale, isreal=True, seed=1):
        self.isreal = isreal
        self.h = oracle().chzo_siggen_create(cycles_per_sample, amplitude, noise, scale, int(isreal), seed)

    def generate(self, n):
        out = np.zeros(n, np.float32 if self.isreal else np.complex64)
        oracle().chzo_siggen_generate(self.h, _fptr(out), n)
        return out

    def __del__(self):
        try:
            oracle().chzo_siggen_delete(self.h)
        except Exception:
            pass


class Stream:
    """Overlap-save master restatement: push L samples, get the block spectrum."""

    def __init__(self, L, M, in_type):
        self.L, self.M, self.in_type = L, M, in_type
        self.h = oracle().chzo_stream_create(L, M, in_type)
        if not self.h:
            raise ValueError("bad stream parameters")
        self.bins = oracle().chzo_stream_bins(self.h)
        self.N = L + M - 1

    def push(self, samples, f64=False):
        samples = np.ascontiguousarray(samples, np.float32 if self.in_type == REAL else np.complex64)
        assert samples.shape[0] == self.L
        if f64:
            out = np.zeros(self.bins, np.complex128)
            oracle().chzo_stream_push_f64(self.h, _fptr(samples), _fptr(out))
        else:
            out = np.zeros(self.bins, np.complex64)
            oracle().chzo_stream_push(self.h, _fptr(samples), _fptr(out))
        return out

    def __del__(self):
        try:
            oracle().chzo_stream_delete(self.h)
        except Exception:
            pass


def notch(state, bins, alpha, spectrum):
    bins = np.ascontiguousarray(bins, np.int32)
    oracle().chzo_notch(_fptr(state), _fptr(bins), len(bins), alpha, _fptr(spectrum))


def scale_ad(isreal=True, bitspersample=1, rf_gain=float("nan"), rf_atten=float("nan"), cal=float("nan")):
    return oracle().chzo_scale_ad(rf_gain, rf_atten, cal, int(isreal), bitspersample)


def compute_tuning(N, samprate, freq):
    s = C.c_int(0); rem = C.c_double(0)
    r = oracle().chzo_compute_tuning(N, samprate, freq, C.byref(s), C.byref(rem))
    return r, s.value, rem.value


# ----------------------------------------------------------------------------
# the reference itself (oracle/_ref)
# ----------------------------------------------------------------------------

class RefMaster:
    def __init__(self, L, M, in_type, worker_threads=0, internal_threads=1):
        self.lib = ref()
        self.L, self.M, self.in_type = L, M, in_type
        # radiod's fft-internal-threads (src/radio.c:296): filter.c hands it to fftwf_plan_with_nthreads() when it plans (src/filter.c:131-133)
        C.c_int.in_dll(self.lib, "N_internal_threads").value = int(internal_threads)
        self.h = self.lib.refchz_master_create(L, M, in_type, worker_threads)
        if not self.h:
            raise ValueError("create_filter_input failed")
        self.bins = self.lib.refchz_master_bins(self.h)
        self.N = self.lib.refchz_master_points(self.h)
        self.chans = []

    def write(self, samples):
        samples = np.ascontiguousarray(samples, np.float32 if self.in_type == REAL else np.complex64)
        return self.lib.refchz_master_write(self.h, _fptr(samples), samples.shape[0])

    def jobnum(self):
        return self.lib.refchz_master_next_jobnum(self.h)

    def spectrum(self, jobnum=None):
        if jobnum is None:
            jobnum = self.jobnum() - 1
        out = np.zeros(self.bins, np.complex64)
        self.lib.refchz_master_spectrum(self.h, jobnum & 0xFFFFFFFF, _fptr(out))
        return out

    def set_notches(self, bins, alpha=0.01):
        bins = np.ascontiguousarray(bins, np.int32)
        self.lib.refchz_master_set_notches(self.h, _fptr(bins), len(bins), alpha)

    def channel(self, olen, out_type=COMPLEX):
        c = RefChan(self, olen, out_type)
        self.chans.append(c)
        return c

    def close(self):
        for c in self.chans:
            c.close()
        self.chans = []
        if self.h:
            self.lib.refchz_master_delete(self.h)
            self.h = None


class RefChan:
    def __init__(self, master, olen, out_type):
        self.lib = master.lib
        self.olen, self.out_type = olen, out_type
        self.h = self.lib.refchz_chan_create(master.h, olen, out_type)
        if not self.h:
            raise ValueError("create_filter_output failed")
        self.points = self.lib.refchz_chan_points(self.h)
        self.bins = self.lib.refchz_chan_bins(self.h)

    def set_filter(self, low, high, beta):
        return self.lib.refchz_chan_set_filter(self.h, low, high, beta)

    def set_isb(self, isb):
        self.lib.refchz_chan_set_isb(self.h, int(isb))

    def set_beam(self, i_weight, q_weight):
        """out.beam = true + set_filter_weights (src/radio.c:938-940); returns (alpha, beta) as the reference stored them."""
        i_weight, q_weight = complex(i_weight), complex(q_weight)
        self.lib.refchz_chan_set_beam(self.h, i_weight.real, i_weight.imag, q_weight.real, q_weight.imag)
        ab = np.zeros(4)
        self.lib.refchz_chan_weights(self.h, _fptr(ab))
        return complex(ab[0], ab[1]), complex(ab[2], ab[3])

    def response(self):
        out = np.zeros(self.points, np.complex64)
        if self.lib.refchz_chan_response(self.h, _fptr(out)) < 0:
            return None
        return out

    def set_response(self, resp):
        resp = np.ascontiguousarray(resp, np.complex64)
        assert resp.shape[0] == self.points
        self.lib.refchz_chan_set_response(self.h, _fptr(resp))

    def execute(self, shift):
        out = np.zeros(self.olen, np.complex64 if self.out_type == COMPLEX else np.float32)
        r = self.lib.refchz_chan_execute(self.h, int(shift), _fptr(out))
        if r != 0:
            raise RuntimeError("execute_filter_output returned %d" % r)
        return out

    def fdomain(self):
        out = np.zeros(self.bins, np.complex64)
        self.lib.refchz_chan_fdomain(self.h, _fptr(out))
        return out

    def drops(self):
        return self.lib.refchz_chan_drops(self.h)

    def close(self):
        if self.h:
            self.lib.refchz_chan_delete(self.h)
            self.h = None


class RefSigGen:
    def __init__(self, cycles_per_sample, amplitude, noise, scale, isreal=True, seed=1):
        self.lib = ref()
        self.isreal = isreal
        self.h = self.lib.refsig_create(cycles_per_sample, amplitude, noise, scale, int(isreal), seed)

    def generate(self, n):
        out = np.zeros(n, np.float32 if self.isreal else np.complex64)
        self.lib.refsig_generate(self.h, _fptr(out), n)
        return out

    def __del__(self):
        try:
            self.lib.refsig_delete(self.h)
        except Exception:
            pass


def beam_weights(i_weight, q_weight):
    """set_filter_weights (src/filter.c:922-929): alpha = 0.5 i_w - j q_w, beta = 0.5 i_w + j q_w."""
    i_weight, q_weight = complex(i_weight), complex(q_weight)
    return 0.5 * i_weight - 1j * q_weight, 0.5 * i_weight + 1j * q_weight


def channel_beam(spectrum, P, olen, shift, response, alpha, beta):
    """COMPLEX master -> COMPLEX slave in beam mode (src/filter.c:756-775)."""
    sp = np.ascontiguousarray(spectrum, np.complex64)
    response = np.ascontiguousarray(response, np.complex64)
    out = np.zeros(olen, np.complex64)
    r = oracle().chzo_channel_beam(_fptr(sp), sp.size, P, olen, int(shift), _fptr(response),
                                   alpha.real, alpha.imag, beta.real, beta.imag, _fptr(out))
    if r != 0:
        raise ValueError("chzo_channel_beam failed")
    return out


def estimate_noise(spectrum, in_type, s_bins, shift, samprate):
    """estimate_noise() of src/radio.c:1783-1866 on a complex64 master spectrum."""
    sp = np.ascontiguousarray(spectrum, np.complex64)
    return oracle().chzo_estimate_noise(_fptr(sp), sp.size, in_type, int(s_bins), int(shift), float(samprate))


def convert_i16(samples, scale, randomize=False):
    """rx888.c convert(): returns (float32 samples, energy, clips)."""
    s = np.ascontiguousarray(samples, np.int16)
    out = np.zeros(s.size, np.float32)
    en = C.c_uint64(0)
    clips = oracle().chzo_convert_i16(_fptr(s), s.size, float(scale), 1 if randomize else 0, _fptr(out), C.byref(en))
    return out, en.value, clips


_ref_radio = None
_ref_rx888 = None


def have_ref_radio():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_radio.so"))


def have_ref_rx888():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_rx888.so"))


def ref_radio():
    """The reference's own radio.c (oracle/ref_radio_wrap.c): estimate_noise / quantile / quickselect."""
    global _ref_radio
    if _ref_radio is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_radio.so"))
        L.refradio_quickselect.restype = _d; L.refradio_quickselect.argtypes = [_vp, _i, _i]
        L.refradio_quantile.restype = _d; L.refradio_quantile.argtypes = [_vp, _i, _d]
        L.refradio_estimate_noise.restype = _d; L.refradio_estimate_noise.argtypes = [_vp, _i, _i, _i, _i, _d]
        _ref_radio = L
    return _ref_radio


def ref_estimate_noise(spectrum, in_type, s_bins, shift, samprate):
    sp = np.ascontiguousarray(spectrum, np.complex64)
    return ref_radio().refradio_estimate_noise(_fptr(sp), sp.size, in_type, int(s_bins), int(shift), float(samprate))


def ref_rx888():
    """The reference's own rx888.c (oracle/ref_rx888_wrap.c): convert / convert_avx2."""
    global _ref_rx888
    if _ref_rx888 is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_rx888.so"))
        for f in (L.refrx_convert, L.refrx_convert_avx2):
            f.argtypes = [_vp, _vp, _i, C.c_float, _vp, _i]
        _ref_rx888 = L
    return _ref_rx888


def ref_convert_i16(samples, scale, randomize=False, avx2=False):
    """(float32 samples, energy, clips) from the reference's convert() / convert_avx2(); None if avx2 is asked for and absent."""
    s = np.ascontiguousarray(samples, np.int16)
    raw = np.zeros(s.size + 16, np.float32)
    off = (-raw.ctypes.data // 4) % 8                # 32-byte aligned output, as convert_avx2 asserts
    out = raw[off:off + s.size]
    en = C.c_uint64(0)
    fn = ref_rx888().refrx_convert_avx2 if avx2 else ref_rx888().refrx_convert
    clips = fn(_fptr(out), _fptr(s), s.size, float(scale), C.byref(en), 1 if randomize else 0)
    if clips < 0:
        return None
    return out.copy(), en.value, clips


# ---- linear demodulator + PCM packing -------------------------------------------------------------
PCM_S16BE, PCM_S16LE, PCM_F32LE, PCM_F32BE, PCM_MULAW, PCM_ALAW, PCM_F16LE, PCM_F16BE = 0, 1, 2, 3, 4, 5, 6, 7
FRAME_DATA, FRAME_SILENCE = 0, 1


class LinParams(C.Structure):
    """chzo_lindemod_params (oracle/chz_oracle.h); the names are the chan_t members src/linear.c reads."""
    _fields_ = [("channels", _i), ("env", _i), ("agc", _i), ("encoding", _i), ("snr_squelch", _i), ("squelch_tail", _i),
                ("tuned", _i), ("kind", _i),
                ("samprate", _d), ("headroom", _d), ("threshold", _d), ("recovery_rate", _d), ("hangtime", _d), ("dc_alpha", _d),
                ("bandwidth", _d), ("shift", _d), ("squelch_open", _d), ("squelch_close", _d), ("gain", _d),
                ("deemph_rate", _d), ("deemph_gain", _d), ("threshold_extend", _d),
                ("pll_enable", _i), ("pll_square", _i), ("pll_loop_bw", _d), ("tone_freq", _d)]


class LinStatus(C.Structure):
    _fields_ = [("frame", _i), ("mute", _i), ("squelch_state", _i), ("pll_lock", _i),
                ("output_power", _d), ("gain", _d), ("n0", _d), ("snr", _d), ("foffset", _d), ("pdeviation", _d),
                ("pll_snr", _d), ("pll_cphase", _d), ("tone_deviation", _d), ("pll_rotations", _i), ("tone_mute", _i)]


DEMOD_LINEAR, DEMOD_FM = 0, 1


def lin_params(channels=1, env=False, agc=True, encoding=PCM_S16BE, snr_squelch=False, squelch_tail=1, tuned=True, samprate=12000.0,
               headroom_db=-15.0, threshold_db=-15.0, recovery_db_per_s=20.0, hangtime=1.1, dc_alpha=0.0, bandwidth=2950.0, shift=0.0,
               squelch_open_db=8.0, squelch_close_db=7.0, gain_db=50.0, pll=False, square=False, pll_bw=100.0):
    """Defaults follow src/modes.c:40-60,224-246 (dB2voltage / dB2power as there)."""
    v = lambda db: 10 ** (db / 20.0)
    return LinParams(channels, int(env), int(agc), encoding, int(snr_squelch), squelch_tail, int(tuned), DEMOD_LINEAR, float(samprate), v(headroom_db),
                     v(threshold_db), v(recovery_db_per_s), float(hangtime), float(dc_alpha), float(bandwidth), float(shift),
                     10 ** (squelch_open_db / 10.0), 10 ** (squelch_close_db / 10.0), v(gain_db), 0.0, 0.0, 0.0,
                     int(pll), int(square), float(pll_bw), 0.0)


def fm_params(encoding=PCM_S16BE, snr_squelch=False, squelch_tail=1, samprate=24000.0, headroom_db=-15.0, bandwidth=16000.0,
              squelch_open=6.3, squelch_close=4.0, threshold_extend=False, deemph_tc=530.5e-6, deemph_gain_db=12.0, pll=False, tone_freq=0.0):
    """NBFM as src/fm.c:38-44 and src/modes.c set it up: squelch thresholds are power ratios, de-emphasis
    rate = -expm1(-1 / (tc * samprate)) (0 = flat FM), gain from dB."""
    rate = -np.expm1(-1.0 / (deemph_tc * samprate)) if deemph_tc else 0.0
    return LinParams(1, 0, 0, encoding, int(snr_squelch), squelch_tail, 1, DEMOD_FM, float(samprate), 10 ** (headroom_db / 20.0),
                     0.0, 0.0, 0.0, 0.0, float(bandwidth), 0.0, float(squelch_open), float(squelch_close), 1.0,
                     float(rate), 10 ** (deemph_gain_db / 20.0) if deemph_tc else 0.0, 1.0 if threshold_extend else 0.0,
                     int(pll), 0, 0.0, float(tone_freq))


def f32_to_f16_bits(x):
    """chzo_f32_to_f16 over an array (the restated float -> binary16 conversion)."""
    x = np.ascontiguousarray(x, np.float32)
    fn = oracle().chzo_f32_to_f16
    fn.argtypes = [C.c_float]; fn.restype = C.c_ushort
    return np.array([fn(float(v)) for v in x], np.uint16) if x.size < 4096 else _f16_many(x)


def _f16_many(x):
    # through the oracle's PCM packer, which loops in C
    lib = oracle()
    if not hasattr(lib, "_f16_pack_ready"):
        lib.chzo_pcm_pack.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]; lib.chzo_pcm_pack.restype = None
        lib._f16_pack_ready = True
    out = np.zeros(x.size, np.uint16)
    lib.chzo_pcm_pack(PCM_F16LE, x.ctypes.data, x.size, out.ctypes.data)
    return out


def ref_f16():
    """The reference's own export_f16_* / import_f16_* (oracle/_ref/libka9q_ref_f16.so), or None where it could not be built."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libka9q_ref_f16.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_export_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; lib.ref_export_f16.restype = None
    lib.ref_import_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; lib.ref_import_f16.restype = None
    return lib


def pcm_bytes(encoding, nsamples):
    return (1 if encoding in (PCM_MULAW, PCM_ALAW) else 2 if encoding in (PCM_S16BE, PCM_S16LE, PCM_F16LE, PCM_F16BE) else 4) * nsamples


class LinDemod:
    """Restated per-block work of demod_linear() (oracle/chz_oracle.c:chzo_lindemod_block)."""

    def __init__(self, params):
        L = oracle()
        L.chzo_lindemod_create.restype = _vp; L.chzo_lindemod_create.argtypes = [C.POINTER(LinParams)]
        L.chzo_lindemod_delete.argtypes = [_vp]
        L.chzo_lindemod_set_params.argtypes = [_vp, C.POINTER(LinParams)]
        L.chzo_lindemod_block.argtypes = [_vp, _vp, _i, _d, _d, _d, _vp, C.POINTER(LinStatus)]
        self.p = params
        self._h = L.chzo_lindemod_create(C.byref(params))

    def set_params(self, params):
        self.p = params
        oracle().chzo_lindemod_set_params(self._h, C.byref(params))

    def block(self, samples, bb_power, n0_est, blocktime=0.02):
        """samples: complex64[N] from downconvert(); returns (pcm bytes or None, LinStatus)."""
        buf = np.ascontiguousarray(samples, np.complex64).copy()
        n = buf.shape[0]
        pcm = np.zeros(pcm_bytes(self.p.encoding, n * self.p.channels), np.uint8)
        st = LinStatus()
        oracle().chzo_lindemod_block(self._h, _fptr(buf), n, float(bb_power), float(n0_est), float(blocktime), _fptr(pcm), C.byref(st))
        return (pcm if st.frame == FRAME_DATA else None), st

    def __del__(self):
        try:
            oracle().chzo_lindemod_delete(self._h)
        except Exception:
            pass


class FmDemod:
    """Restated per-block work of demod_fm() (oracle/chz_oracle.c:chzo_fmdemod_block)."""

    def __init__(self, params):
        L = oracle()
        L.chzo_fmdemod_create.restype = _vp; L.chzo_fmdemod_create.argtypes = [C.POINTER(LinParams)]
        L.chzo_fmdemod_delete.argtypes = [_vp]
        L.chzo_fmdemod_block.argtypes = [_vp, _vp, _i, _d, _d, _d, _vp, C.POINTER(LinStatus)]
        self.p = params
        self._h = L.chzo_fmdemod_create(C.byref(params))

    def block(self, samples, bb_power, n0_est, blocktime=0.02):
        buf = np.ascontiguousarray(samples, np.complex64)
        n = buf.shape[0]
        pcm = np.zeros(pcm_bytes(self.p.encoding, n), np.uint8)
        st = LinStatus()
        oracle().chzo_fmdemod_block(self._h, _fptr(buf), n, float(bb_power), float(n0_est), float(blocktime), _fptr(pcm), C.byref(st))
        return (pcm if st.frame == FRAME_DATA else None), st

    def __del__(self):
        try:
            oracle().chzo_fmdemod_delete(self._h)
        except Exception:
            pass


def have_ref_fm():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_fm.so"))


_ref_fm = None


def ref_fm_run(params, baseband, bb_power, n0_smoothed, blocktime=0.02):
    """The reference's OWN demod_fm() (oracle/ref_fm_wrap.c).  Returns dict of per-block arrays."""
    global _ref_fm
    if _ref_fm is None:
        _ref_fm = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_fm.so"))
        _ref_fm.reffm_run.argtypes = [C.POINTER(LinParams), _d, _i, _i, _vp, _vp, _vp, _vp, _i] + [_vp] * 8
    bb = np.ascontiguousarray(baseband, np.complex64)
    nb, n = bb.shape
    stride = pcm_bytes(params.encoding, n)
    out = dict(pcm=np.zeros((nb, stride), np.uint8), frame=np.zeros(nb, np.int32), mute=np.zeros(nb, np.int32), power=np.zeros(nb),
               gain=np.zeros(nb), snr=np.zeros(nb), foffset=np.zeros(nb), pdev=np.zeros(nb), tonedev=np.zeros(nb))
    bp = np.ascontiguousarray(bb_power, np.float64); n0 = np.ascontiguousarray(n0_smoothed, np.float64)
    r = _ref_fm.reffm_run(C.byref(params), float(blocktime), nb, n, _fptr(bb), _fptr(bp), _fptr(n0), _fptr(out["pcm"]), stride,
                          _fptr(out["frame"]), _fptr(out["mute"]), _fptr(out["power"]), _fptr(out["gain"]), _fptr(out["snr"]),
                          _fptr(out["foffset"]), _fptr(out["pdev"]), _fptr(out["tonedev"]))
    assert r == 0
    return out


def have_ref_linear():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_linear.so"))


_ref_linear = None


def ref_linear_run(params, baseband, bb_power, n0_smoothed, blocktime=0.02, pll_out=None):
    """The reference's OWN demod_linear() (oracle/ref_linear_wrap.c) over nblocks blocks of baseband[nblocks][N];
    n0_smoothed = chan->sig.n0 as downconvert() leaves it.  Returns (pcm[nblocks][bytes], frame, mute, out_power, gain);
    pll_out (float64[nblocks][5], optional) receives chan->pll.snr, .lock, .cphase, .rotations and chan->sig.foffset per block."""
    global _ref_linear
    if _ref_linear is None:
        _ref_linear = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libka9q_ref_linear.so"))
        _ref_linear.reflin_run.argtypes = [C.POINTER(LinParams), _d, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]
    bb = np.ascontiguousarray(baseband, np.complex64)
    nb, n = bb.shape
    stride = pcm_bytes(params.encoding, n * params.channels)
    pcm = np.zeros((nb, stride), np.uint8)
    frame = np.zeros(nb, np.int32); mute = np.zeros(nb, np.int32)
    power = np.zeros(nb); gain = np.zeros(nb)
    bp = np.ascontiguousarray(bb_power, np.float64); n0 = np.ascontiguousarray(n0_smoothed, np.float64)
    r = _ref_linear.reflin_run(C.byref(params), float(blocktime), nb, n, _fptr(bb), _fptr(bp), _fptr(n0), _fptr(pcm), stride,
                               _fptr(frame), _fptr(mute), _fptr(power), _fptr(gain), _fptr(pll_out) if pll_out is not None else None)
    assert r == 0
    return pcm, frame, mute, power, gain


class Downconv:
    """Per-channel tail of downconvert() (src/radio.c:1476-1520).  which = "oracle" (restatement) or
    "ref" (the reference's own osc.c / cispi with the glue statements restated in ref_driver.c)."""

    def __init__(self, L, M, out_samprate, which="oracle"):
        self.L, self.M, self.fs = L, M, float(out_samprate)
        self.which = which
        self._lib = oracle() if which == "oracle" else ref()
        self._h = self._lib.chzo_downconv_create() if which == "oracle" else self._lib.refdc_create()

    def block(self, samples, shift, remainder, doppler_rate=0.0):
        """samples: complex64[olen] as execute_filter_output leaves them; returns (rotated, bb_power)."""
        buf = np.ascontiguousarray(samples, dtype=np.complex64).copy()
        fn = self._lib.chzo_downconv_block if self.which == "oracle" else self._lib.refdc_block
        pw = fn(self._h, int(shift), float(remainder), self.fs, float(doppler_rate), self.L, self.M,
                _fptr(buf), buf.size)
        return buf, pw

    def close(self):
        if self._h:
            (self._lib.chzo_downconv_delete if self.which == "oracle" else self._lib.refdc_delete)(self._h)
            self._h = None

    __del__ = close
