"""Child process of tests/test_engine_emulated.py::test_engine_random_operations: a model-based random walk over the engine's C ABI
(on whatever library CHZ_LIB names -- the CPU build in the test, an MI355X when run by hand on the GPU box).

Two banks on one REAL master.  At random: retune a range of channels, swap responses, flip ISB flags, change the active count, run a
few blocks pipelined over the lanes or one by one, re-create a bank.  A Python-side model knows what every channel should be doing;
after every run the LAST block's outputs are compared with the oracle (float64 spectrum of the same samples, the model's shift /
response / isb per channel)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as ol
from conftest import load_pkg
from test_gpu_parity import check_channel, noise_floor


def main(seed, iters):
    pkg = load_pkg()
    rng = np.random.default_rng(seed)
    L, M = 25920, 6481
    N = L + M - 1
    eng = pkg.engine.Engine(L, M, ol.REAL, ring_blocks=8)
    st = ol.Stream(L, M, ol.REAL)
    notch_state = np.zeros(2)
    eng.set_notches([0], 0.01)
    geo = [(300, 240, 40), (600, 480, 24)]
    lib = {P: np.stack([ol.set_filter(P, olen, N, True, lo, hi, 9.0) for lo, hi in ((-0.35, 0.35), (0.01, 0.3), (-0.3, -0.02), (-0.1, 0.1), (-0.45, 0.2))]).astype(np.complex64)
           for P, olen, _ in geo}

    def fresh(i):
        P, olen, cap = geo[i]
        b = eng.bank(P, olen, cap)
        m = dict(bank=b, P=P, olen=olen, cap=cap, shift=rng.integers(-12000, 12000, cap).astype(np.int32), which=rng.integers(0, 5, cap), isb=np.zeros(cap, np.uint8), active=cap)
        b.set_responses(0, lib[P][m["which"]]); b.set_shifts(0, m["shift"]); b.set_active(cap)
        return m
    models = [fresh(0), fresh(1)]
    # a third bank with fine tuning (the tail of downconvert(): rotator, block phase, shift-change kick) and a fourth whose channels
    # SHARE three response rows; both follow every block
    fs_in, fs_out = 1.296e6, 12000.0
    nt = 8
    tuned = eng.bank(300, 240, nt)
    t_resp = lib[300][np.zeros(nt, int)]
    tuned.set_responses(0, t_resp); tuned.set_active(nt)
    t_f = 50e3 + rng.uniform(0, 500e3, nt); t_shift = np.zeros(nt, np.int32); t_rem = np.zeros(nt)
    for ch in range(nt):
        _, t_shift[ch], t_rem[ch] = ol.compute_tuning(N, fs_in, t_f[ch])
    tuned.set_tuning(0, 0, t_shift, -t_rem / fs_out, np.zeros(nt))
    dcs = [ol.Downconv(L, M, fs_out, "oracle") for _ in range(nt)]
    # ... and demodulators behind it (AGC / envelope / I-Q, different encodings): their state is a recurrence over blocks too
    tuned.enable_noise(fs_in); tuned.set_pcm_stride(8 * 240)
    dkw = [dict(), dict(env=True, dc_alpha=0.002, encoding=ol.PCM_S16LE), dict(channels=2, encoding=ol.PCM_F32LE), dict(agc=False, gain_db=30.0, encoding=ol.PCM_MULAW)]
    dpar = [ol.lin_params(**dkw[ch % 4]) for ch in range(nt)]
    tuned.set_demod(0, 0, [pkg.engine.DemodParams(*[getattr(q, f) for f, _ in ol.LinParams._fields_]) for q in dpar], 0.02)
    dem = [ol.LinDemod(q) for q in dpar]
    shared = eng.bank(300, 240, 30, shared_rows=3)
    s_rows = rng.integers(0, 3, 30).astype(np.int32); s_shift = rng.integers(-12000, 12000, 30).astype(np.int32)
    shared.set_row_responses(0, lib[300][:3]); shared.set_rows(0, s_rows); shared.set_shifts(0, s_shift); shared.set_active(30)
    job = 0
    spec = None
    checks = 0
    for it in range(iters):
        op = int(rng.integers(0, 100))
        m = models[int(rng.integers(0, 2))]
        b, cap = m["bank"], m["cap"]
        if op < 30:                                         # retune a range
            c0 = int(rng.integers(0, cap)); n = int(rng.integers(1, min(12, cap - c0) + 1))
            m["shift"][c0:c0 + n] = rng.integers(-17000, 17000, n)
            b.set_shifts(c0, m["shift"][c0:c0 + n])
        elif op < 48:                                       # new filters for a few channels (spare rows, fences)
            c0 = int(rng.integers(0, cap)); n = int(rng.integers(1, min(6, cap - c0) + 1))
            m["which"][c0:c0 + n] = rng.integers(0, 5, n)
            b.set_responses(c0, lib[m["P"]][m["which"][c0:c0 + n]])
        elif op < 58:                                       # ISB flags
            c0 = int(rng.integers(0, cap)); n = int(rng.integers(1, min(8, cap - c0) + 1))
            m["isb"][c0:c0 + n] = rng.integers(0, 2, n)
            b.set_isb(c0, m["isb"][c0:c0 + n])
        elif op < 64:
            m["active"] = int(rng.integers(1, cap + 1)); b.set_active(m["active"])
        elif op < 68:                                       # the bank goes away and comes back (new device arrays, same id space)
            idx = models.index(m)
            b.destroy()
            models[idx] = fresh(idx)
        elif op < 74:                                       # a tuned channel moves (takes effect at the next block)
            ch = int(rng.integers(0, nt))
            t_f[ch] += rng.uniform(-3e3, 3e3)
            _, t_shift[ch], t_rem[ch] = ol.compute_tuning(N, fs_in, t_f[ch])
            tuned.set_tuning(job, ch, t_shift[ch:ch + 1], -t_rem[ch:ch + 1] / fs_out, np.zeros(1))
        elif op < 78:                                       # a sharing channel names another row / moves
            c0 = int(rng.integers(0, 30)); n = int(rng.integers(1, min(5, 30 - c0) + 1))
            s_rows[c0:c0 + n] = rng.integers(0, 3, n); s_shift[c0:c0 + n] = rng.integers(-12000, 12000, n)
            shared.set_rows(c0, s_rows[c0:c0 + n]); shared.set_shifts(c0, s_shift[c0:c0 + n])
        else:                                               # run: pipelined over the lanes, or block by block
            k = int(rng.integers(1, 5))                    # at most ND blocks: every slot of the run can still be read afterwards
            xs = (rng.standard_normal(k * L) * 0.05).astype(np.float32)
            stepped = bool(rng.integers(0, 2))
            if not stepped:
                # the ring holds 8 blocks: write ahead, then run them in one go
                for j in range(k):
                    eng.write(xs[j * L:(j + 1) * L])
                eng.run_blocks(job, k)
            for j in range(k):
                if stepped:
                    eng.write(xs[j * L:(j + 1) * L]); eng.step(job + j)
                spec = st.push(xs[j * L:(j + 1) * L], f64=True)
                dc = spec[:1].astype(np.complex64); ol.notch(notch_state, [0], 0.01, dc); spec[0] = dc[0]
                # the tuned bank's oscillators are a recurrence over blocks in the reference: the model follows every block; its
                # outputs can be looked at block by block when stepping, and for the last block of a pipelined run
                look = stepped or j == k - 1
                got = tuned.read_slot((job + j) % 4) if look else None
                for ch in range(nt):
                    ideal = ol.channel(spec, ol.REAL, 300, 240, int(t_shift[ch]), t_resp[ch])
                    ideal, _ = dcs[ch].block(ideal, t_shift[ch], t_rem[ch], 0.0)
                    if look:
                        check_channel(got[ch], ideal, noise_floor(spec, t_resp[ch])); checks += 1
            for j in range(k):                              # the demodulators, block after block, on what the device stage was fed
                slot = (job + j) % 4
                d_in = tuned.read_slot(slot); d_pw = tuned.read_power(slot); d_n0 = tuned.read_noise(slot)
                pcm, status = tuned.read_pcm(slot)
                for ch in range(nt):
                    want, stt = dem[ch].block(d_in[ch], d_pw[ch], d_n0[ch], 0.02)
                    got = status[ch]
                    assert (got.frame, got.mute, got.squelch_state) == (stt.frame, stt.mute, stt.squelch_state), (it, j, ch)
                    assert abs(got.gain - stt.gain) <= 1e-9 * abs(stt.gain) and abs(got.output_power - stt.output_power) <= 1e-5 * abs(stt.output_power) + 1e-300, (it, j, ch)
                    if stt.frame == ol.FRAME_DATA:
                        q = dpar[ch]; nb = ol.pcm_bytes(q.encoding, 240 * q.channels)
                        a_, w_ = pcm[ch, :nb], want
                        if q.encoding == ol.PCM_MULAW:
                            assert np.mean(a_ != w_) < 0.02, (it, j, ch)
                        elif q.encoding in (ol.PCM_S16BE, ol.PCM_S16LE):
                            dt = ">i2" if q.encoding == ol.PCM_S16BE else "<i2"
                            assert np.abs(a_.view(dt).astype(np.int32) - w_.view(dt).astype(np.int32)).max() <= 1, (it, j, ch)
                        else:
                            assert np.abs(a_.view("<f4").astype(np.float64) - w_.view("<f4").astype(np.float64)).max() <= 1e-4 * max(float(np.abs(w_.view("<f4")).max()), 1e-30), (it, j, ch)
                    checks += 1
            job += k
            sh_out = shared.read_slot((job - 1) % 4)
            for ch in rng.choice(30, size=5, replace=False):
                want = ol.channel(spec, ol.REAL, 300, 240, int(s_shift[ch]), lib[300][s_rows[ch]])
                check_channel(sh_out[ch], want, noise_floor(spec, lib[300][s_rows[ch]])); checks += 1
            last = (job - 1) % 4
            for mm in models:
                out = mm["bank"].read_slot(last)
                for ch in rng.choice(mm["active"], size=min(6, mm["active"]), replace=False):
                    resp = lib[mm["P"]][mm["which"][ch]]
                    want = ol.channel(spec, ol.REAL, mm["P"], mm["olen"], int(mm["shift"][ch]), resp, isb=bool(mm["isb"][ch]))
                    check_channel(out[ch], want, noise_floor(spec, resp))
                    checks += 1
    eng.check()
    eng.close()
    print("FUZZ ok iterations %d blocks %d channel checks %d" % (iters, job, checks))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
