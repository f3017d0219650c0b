"""Oracle checks of SAMPLED channels of very large banks (test infrastructure; checker only).

The banks that carry the large numbers -- 70 k ... 20 M channels, where the engine picks its other
code paths by size (demod_lin_lanes / demod_fm_lanes / pll_lanes from 65,536 channels, the |X|^2 image
of noise_est and the staged stores of chan_ifft from 16,384) -- cannot be compared channel by channel
in reasonable time, so a sample is: the first and last channel, the channels either side of every kind
of workgroup boundary (12 channels per chan_ifft workgroup at P=300, 4 per noise_est workgroup, 64 per
lane-per-channel demodulator workgroup), the last partially filled groups, and random ones in between.

Used by tests/test_gpu_scale.py and by bench.py's c_rt / next_rows / c_rt_pcie legs AFTER their timed
regions (never inside one).  Everything is compared with the oracle (tests/oracle_lib.py):
  channel outputs   ol.channel() on the device's own block spectrum (+ the restated downconvert() tail
                    for tuned banks), tolerance = tests/test_gpu_parity.py's: err_rms <= 1e-5 rms + floor
  noise estimates   ol.estimate_noise() on the same spectrum, 1e-12 relative (src/radio.c:1783-1866)
  demodulators      ol.LinDemod / ol.FmDemod (pinned to the reference's linear.c / fm.c) fed exactly what the
                    device stage was fed; frames / squelch states equal, integer PCM bit-exact
"""
import numpy as np

import oracle_lib as ol

REL_L2 = 1e-5
MAXABS_RMS = 1e-4
FLOOR = 2e-8


def sample_channels(nch, k, seed=0, groups=(12, 4, 64, 3)):
    """k (or a few more) distinct channel indices below nch: ends, workgroup edges of every group size, random."""
    want = {0, nch - 1}
    for g in groups:
        for base in (g, (nch // 2 // g) * g, ((nch - 1) // g) * g):
            for d in (-1, 0, 1):
                want.add(base + d)
    want = {int(c) for c in want if 0 <= c < nch}
    rng = np.random.default_rng(seed)
    while len(want) < min(k, nch):
        want.add(int(rng.integers(0, nch)))
    return sorted(want)


def channel_error(got, want, floor):
    """(relative rms error, ok) under the parity tests' rule: err_rms <= REL_L2*rms + floor, max|err| <= MAXABS_RMS*rms + 6 floor."""
    want = np.asarray(want)
    rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
    if rms == 0.0:
        return 0.0, bool(np.abs(got).max() == 0)
    d = np.abs(got - want)
    err = float(np.sqrt(np.mean(d ** 2)))
    ok = err <= REL_L2 * rms + floor and float(d.max()) <= MAXABS_RMS * rms + 6 * floor
    return err / rms, bool(ok)


def check_plain(eng, bank, slot, chans, shift_of, resp_of, in_type=ol.REAL, out_host=None):
    """Channel outputs of an UNTUNED bank on `slot` against ol.channel() on the device's own spectrum of that slot.
    out_host: complex64[nch][olen] already in host memory (the c_rt_pcie leg's pinned buffer) instead of device reads.
    -> dict(verified_channels, max_rel_err, failed=[channel, ...])"""
    spec = eng.spectrum(slot)
    spec64 = spec.astype(np.complex128)
    smax = float(np.abs(spec).max())
    worst, failed = 0.0, []
    for ch in chans:
        got = out_host[ch] if out_host is not None else bank.read_slot(slot, ch, 1)[0]
        resp = resp_of(ch)
        want = ol.channel(spec64, in_type, bank.P, bank.olen, int(shift_of(ch)), resp)
        rel, ok = channel_error(got, want, FLOOR * smax * float(np.linalg.norm(resp)))
        worst = max(worst, rel)
        if not ok or not np.all(np.isfinite(got)):
            failed.append(int(ch))
    return {"verified_channels": len(chans), "max_rel_err": worst, "failed": failed, "highest_channel_checked": int(max(chans))}


def _pcm_mismatches(p, got_row, want, n_samples):
    """number of PCM samples that differ (integer and half-float encodings: any bit; float32: beyond 2e-6 of the block's peak)"""
    nb = ol.pcm_bytes(p.encoding, n_samples)
    if p.encoding in (ol.PCM_F32LE, ol.PCM_F32BE):
        dt = ">f4" if p.encoding == ol.PCM_F32BE else "<f4"
        a, w = got_row[:nb].view(dt).astype(np.float64), want.view(dt).astype(np.float64)
        return int(np.sum(np.abs(a - w) > 2e-6 * max(np.abs(w).max(), 1e-30)))
    width = 1 if p.encoding in (ol.PCM_MULAW, ol.PCM_ALAW) else 2
    a = np.frombuffer(got_row[:nb].tobytes(), np.uint8 if width == 1 else np.uint16)
    w = np.frombuffer(np.asarray(want, np.uint8).tobytes(), np.uint8 if width == 1 else np.uint16)
    return int(np.sum(a != w))


class ChainChecker:
    """SURVEY 8f's chain behind sampled channels of one tuned bank: chan_ifft (+ downconvert() tail) -> noise_est -> demodulator.

    Blocks must be handed over in order from the block at which the sampled channels' demodulators were (re)started;
    `pre_blocks` = blocks the tuning has been running for before that (the rotation is a closed form in the block number on
    the device and a stepped oscillator in the reference: the restated oscillators are stepped through them on zeros)."""

    def __init__(self, L, M, fs_in, fs_out, P, olen, chans, shifts, rems, resp_of, params, in_type=ol.REAL, pre_blocks=0, strict_pll=True):
        self.L, self.M, self.fs_in, self.P, self.olen, self.in_type = L, M, fs_in, P, olen, in_type
        self.chans = list(chans)
        self.shifts = {c: int(s) for c, s in zip(self.chans, shifts)}
        self.rems = {c: float(r) for c, r in zip(self.chans, rems)}
        self.resp_of = resp_of
        self.params = {c: p for c, p in zip(self.chans, params)}
        self.dc = {c: ol.Downconv(L, M, fs_out, "oracle") for c in self.chans}
        zeros = np.zeros(olen, np.complex64)
        for c in self.chans:
            for _ in range(pre_blocks):
                self.dc[c].block(zeros, self.shifts[c], self.rems[c], 0.0)
        self.dm = {c: (ol.FmDemod(p) if p.kind == ol.DEMOD_FM else ol.LinDemod(p)) for c, p in self.params.items()}
        self.strict_pll = strict_pll
        self.blocks = 0
        self.max_rel_err = 0.0
        self.noise_max_rel = 0.0
        self.pcm_mismatches = 0
        self.pcm_samples = 0
        self.status_mismatches = 0
        self.data_frames = 0
        self.failed = []

    def block(self, spec, out, power, noise, pcm, status):
        """spec: the device's spectrum of the block (complex64[bins]); out/power/noise/pcm/status: dicts channel -> what the
        device produced for that channel (complex64[olen], float, float, uint8[stride], DemodStatus-like)."""
        spec64 = spec.astype(np.complex128)
        smax = float(np.abs(spec).max())
        for c in self.chans:
            resp = self.resp_of(c)
            raw = ol.channel(spec64, self.in_type, self.P, self.olen, self.shifts[c], resp)
            want, wpw = self.dc[c].block(raw.astype(np.complex64), self.shifts[c], self.rems[c], 0.0)
            # the rotation was applied to a float32 copy of the float64 channel: compare under the same tolerance
            rel, ok = channel_error(out[c], want, FLOOR * smax * float(np.linalg.norm(resp)))
            self.max_rel_err = max(self.max_rel_err, rel)
            bad = not ok or not np.all(np.isfinite(out[c]))
            if wpw > 0 and abs(power[c] - wpw) > 1e-4 * wpw + 1e-30:
                bad = True
            wn = ol.estimate_noise(spec, self.in_type, self.P, self.shifts[c], self.fs_in)
            nrel = abs(noise[c] - wn) / max(abs(wn), 1e-300)
            self.noise_max_rel = max(self.noise_max_rel, nrel)
            if nrel > 1e-12:
                bad = True
            p = self.params[c]
            wpcm, st = self.dm[c].block(out[c], power[c], noise[c], 0.02)
            got = status[c]
            same = (got.frame, got.mute, got.squelch_state) == (st.frame, st.mute, st.squelch_state)
            if p.kind != ol.DEMOD_FM:
                same = same and got.pll_lock == st.pll_lock and abs(got.gain - st.gain) <= 1e-6 * abs(st.gain)
                if not p.pll_enable:
                    same = same and abs(got.gain - st.gain) <= 1e-9 * abs(st.gain) and abs(got.n0 - st.n0) <= 1e-12 * abs(st.n0)
                elif self.strict_pll:
                    same = same and got.pll_rotations == st.pll_rotations
            else:
                same = same and abs(got.snr - st.snr) <= 1e-5 * abs(st.snr) + 1e-9
            if not same:
                self.status_mismatches += 1
                bad = True
            if st.frame == ol.FRAME_DATA and same:
                self.data_frames += 1
                n = self.olen * (1 if p.kind == ol.DEMOD_FM else p.channels)
                mm = _pcm_mismatches(p, pcm[c], wpcm, n)
                self.pcm_mismatches += mm
                self.pcm_samples += n
                if mm:
                    bad = True
            if bad and c not in self.failed:
                self.failed.append(int(c))
        self.blocks += 1

    def result(self):
        return {"verified_channels": len(self.chans), "verified_blocks": self.blocks, "highest_channel_checked": int(max(self.chans)),
                "max_rel_err": self.max_rel_err, "noise_max_rel_err": self.noise_max_rel,
                "pcm_mismatches": self.pcm_mismatches, "pcm_samples_compared": self.pcm_samples, "data_frames": self.data_frames,
                "status_mismatches": self.status_mismatches, "failed": self.failed}


def read_sampled(bank, slot, chans, pcm_host=None, flags_host=None):
    """what the device left for the sampled channels on `slot`: dicts channel -> value"""
    out, power, noise, pcm, status = {}, {}, {}, {}, {}
    for c in chans:
        out[c] = bank.read_slot(slot, c, 1)[0]
        power[c] = float(bank.read_power(slot, c, 1)[0])
        noise[c] = float(bank.read_noise(slot, c, 1)[0])
        row, st = bank.read_pcm(slot, c, 1)
        status[c] = st[0]
        pcm[c] = row[0] if pcm_host is None else pcm_host[c]
    return out, power, noise, pcm, status
