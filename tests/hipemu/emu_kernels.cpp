// tests/hipemu/emu_kernels.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the unmodified kernels of ka9q-radio_amd/csrc/chz_kernels.h on the CPU
// through the fiber emulator, behind a flat C API for ctypes.  Used by
// tests/test_kernels_emulated.py (-m "not gpu") to check butterfly wiring and
// index math against the oracle before the kernels ever see a GPU.
#include <hip/hip_runtime.h>
#include <vector>
#include <cstring>
#include "chz_launch.h"
#include "chz_finetune.h"

using namespace chz;

static const float2* F2(const std::vector<f2>& v) { return reinterpret_cast<const float2*>(v.data()); }

extern "C" {

// ring: input ring on the host (floats for REAL, float pairs for COMPLEX); ring_len in floats
static int emu_forward_impl(const float* ring, const short* ring16, float scale16, int derand, int new_from,
                            unsigned long long* energy, unsigned* clips,
                            long ring_len, long start, int N, int in_type, const char* spec,
                            float* spectrum, char* desc, int desc_len,
                            const int* notch_bins, double* notch_state, int n_notch, double notch_alpha);
int emu_forward(const float* ring, long ring_len, long start, int N, int in_type, const char* spec,
                float* spectrum, char* desc, int desc_len,
                const int* notch_bins, double* notch_state, int n_notch, double notch_alpha) {
  return emu_forward_impl(ring, nullptr, 1.f, 0, 0, nullptr, nullptr, ring_len, start, N, in_type, spec, spectrum, desc, desc_len,
                          notch_bins, notch_state, n_notch, notch_alpha);
}
// int16 ring (REAL masters): energy/clips = sums over window indices >= new_from
int emu_forward_i16(const short* ring16, float scale, int derand, int new_from, long ring_len, long start, int N, const char* spec,
                    float* spectrum, unsigned long long* energy, unsigned* clips) {
  return emu_forward_impl(nullptr, ring16, scale, derand, new_from, energy, clips, ring_len, start, N, CHZ_IN_REAL, spec, spectrum,
                          nullptr, 0, nullptr, nullptr, 0, 0.0);
}
static int emu_forward_impl(const float* ring, const short* ring16, float scale16, int derand, int new_from,
                            unsigned long long* energy, unsigned* clips,
                            long ring_len, long start, int N, int in_type, const char* spec,
                            float* spectrum, char* desc, int desc_len,
                            const int* notch_bins, double* notch_state, int n_notch, double notch_alpha) {
  FwdPlan p;
  if (!build_fwd_plan(N, in_type, spec, p)) return -1;
  if (desc) { strncpy(desc, p.desc.c_str(), (size_t)desc_len - 1); desc[desc_len - 1] = 0; }
  std::vector<float2> buf((size_t)p.Ra * p.inner);
  if (in_type == CHZ_IN_REAL) {
    FirstRealParams a{};
    a.ring = ring; a.ring_len = ring_len; a.start = start; a.buf = buf.data(); a.inner = p.inner;
    a.T = p.T1; a.Ra = p.Ra; a.padk = p.padk1;
    a.tw_sub = F2(p.tw_sub_a); a.tw_tile = F2(p.tw1_tile); a.tw_col = F2(p.tw1_col);
#if CHZ_TW_SHUFFLE
    if (p.T1 != 16) return -3;
#endif
    std::vector<unsigned long long> en((size_t)p.grid1 * (p.block1 / 64), 0ull);
    std::vector<unsigned> cl(en.size(), 0u);
    if (ring16) {
      a.ring16 = ring16; a.scale16 = scale16; a.derand = derand; a.new_from = new_from;
      a.energy_part = en.data(); a.clip_part = cl.data();
    }
    if (launch_first_real(p.ra, p.grid1, p.block1, p.lds1, nullptr, a)) return -2;
    if (ring16) {
      unsigned long long se = 0; unsigned sc = 0;
      for (size_t i = 0; i < en.size(); i++) { se += en[i]; sc += cl[i]; }
      if (energy) *energy = se;
      if (clips) *clips = sc;
    }
  } else {
    ColsParams a{};
    a.in = reinterpret_cast<const float2*>(ring); a.in_len = ring_len / 2; a.in_start = start / 2;
    a.out = buf.data(); a.rows = 1; a.inner = p.inner; a.T = p.T1; a.padk = p.padk1;
    a.tw_sub = F2(p.tw_sub_a); a.tw_tile = F2(p.tw1_tile); a.tw_col = F2(p.tw1_col);
    if (launch_cols(p.ra, p.grid1, p.block1, p.lds1, nullptr, a)) return -2;
  }
  if (p.Nb > 1) {
    ColsParams b{};
    b.in = buf.data(); b.in_len = 0; b.in_start = 0; b.out = buf.data();
    b.rows = p.Ra; b.inner = p.Nc; b.T = p.T2; b.padk = p.padk2;
    b.tw_sub = F2(p.tw_sub_b); b.tw_tile = F2(p.tw2_tile); b.tw_col = F2(p.tw2_col);
    if (!getenv("EMU_NO_TWFULL")) b.tw_full = F2(p.tw2_full);
    if (launch_cols(p.rb, p.grid2, p.block2, p.lds2, nullptr, b)) return -3;
  }
  RowsParams c{};
  std::vector<float2> spec_dev((size_t)p.spec_elems, make_float2(0.f, 0.f));
  c.lay = SpecLayout{p.Na, p.spec_pitch, p.spec_off}; c.ka_shift = p.ka_shift;
  c.buf = buf.data(); c.spec = spec_dev.data();
  c.Ra = p.Ra; c.Na = p.Na; c.Nb = p.Nb; c.Ta = p.Ta; c.ld = p.ld3; c.padg = p.padg3; c.N = p.N;
  c.mirror = in_type == CHZ_IN_REAL; c.tw_sub = F2(p.tw_sub_c);
  if (launch_rows(p.rc, p.grid3, p.block3, p.lds3, nullptr, c)) return -4;
  if (n_notch > 0) {           // K2 as the engine runs it: its own tiny kernel right after the last pass
    NotchTables nt = notch_tables(notch_bins, n_notch, c.lay);
    std::vector<double> alpha((size_t)n_notch, notch_alpha);
    NotchFixParams q{};
    q.spec = spec_dev.data(); q.addr = nt.addr.data(); q.next = nt.next.data(); q.head = nt.head.data();
    q.alpha = alpha.data(); q.state = notch_state; q.n = n_notch;
    if (getenv("EMU_NOTCH_INLINE")) nt.fill_inline(q, alpha.data());
    if (launch_notch_fix(nullptr, q)) return -5;
  }
  float2* out = reinterpret_cast<float2*>(spectrum);
  for (long k = 0; k < p.bins; k++) out[k] = spec_dev[(size_t)spec_addr(c.lay, k)];   // back to natural order
  return 0;
}

// which kernel serves a channel size: 1 register-tiled menu, 2 chan_any in LDS, 3 chan_any in global scratch, 0 none
int emu_chan_kind(int P) { ChanGeom g; if (!build_chan_geom(P, g)) return 0; return !g.any ? 1 : (g.big ? 3 : 2); }
// does the planner find axes for this master?  (desc receives the plan string)
int emu_plan_exists(int N, int in_type, char* desc, int len) {
  FwdPlan p;
  if (!build_fwd_plan(N, in_type, "", p)) return 0;
  if (desc && len > 0) { strncpy(desc, p.desc.c_str(), (size_t)len - 1); desc[len - 1] = 0; }
  return 1;
}
// the division-free storage index of chan_ifft against the plain definition, for EVERY bin of a master: returns the first
// disagreeing bin or -1; -2 if chan_layout refuses the layout
long emu_spec_index_check(int na, int pitch, int off, long bins) {
  ChanParams c{};
  SpecLayout lay{na, pitch, off};
  if (!chan_layout(c, lay, bins)) return -2;
  for (long k = 0; k < bins; k++)
    if ((long)spec_index(lay.off, c.magic, c.dpitch, (int)k) != spec_addr(lay, k)) return k;
  return -1;
}
// K2 inside fwd_rows (round 4): notch_owner() names the workgroup / thread / output index that stores a listed bin; here the pass's own
// store arithmetic (fwd_rows in chz_kernels.h, restated line by line) says where THAT thread's K2-th output goes and whether the store
// happens at all.  Walks bins 0, step, 2*step, ... and the last `tail` bins of the planned master; returns the first bin whose owner
// would store somewhere else (or not at all), -1 if every probed bin is stored by exactly the named thread, -2 without a plan.
long emu_notch_owner_check(int N, int in_type, const char* spec, int step, int tail) {
  FwdPlan p;
  if (!build_fwd_plan(N, in_type, spec ? spec : "", p)) return -2;
  const bool real = in_type == CHZ_IN_REAL;
  const SpecLayout lay{p.Na, p.spec_pitch, p.spec_off};
  const int R1 = p.rc.r1;
  const long xrows = (long)p.Nb * p.Nc, half = (long)p.N >> 1;
  auto check = [&](long bin) -> bool {
    int wg = 0, tid = 0, k2 = 0;
    if (!notch_owner(p, real, (int)bin, &wg, &tid, &k2)) return false;
    // the kernel's own indices for (wg, tid, k2)
    const int kb = wg % p.Nb, at = wg / p.Nb;
    const int a0 = at * p.Ta - p.ka_shift;
    if (tid >= R1 * p.Ta) return false;
    const int k1 = tid / p.Ta, r = tid - k1 * p.Ta;
    const int ka = a0 + r;
    if (ka < 0 || ka >= p.Ra) return false;
    const bool selfconj = (ka == 0) || (2 * ka == p.Na);
    const long x0 = kb + (long)p.Nb * k1, xs = (long)p.Nb * R1;
    const long kk0 = ka + (long)p.Na * x0, kks = (long)p.Na * xs;
    const long d0 = x0 * lay.pitch + lay.off + ka, ds = xs * lay.pitch;
    const long m0 = (xrows - 1 - x0) * lay.pitch + lay.off + (p.Na - ka);
    const bool direct = !real || kk0 + k2 * kks <= half;
    const long addr = direct ? d0 + k2 * ds : m0 - k2 * ds;
    const bool stored = direct || !selfconj;
    return stored && addr == spec_addr(lay, bin);
  };
  const long bins = real ? p.N / 2 + 1 : p.N;
  if (step < 1) step = 1;
  for (long b = 0; b < bins; b += step) if (!check(b)) return b;
  for (long b = bins - tail < 0 ? 0 : bins - tail; b < bins; b++) if (!check(b)) return b;
  return -1;
}
int emu_chan_desc(int in_type, int m_bins, int P, int shift, int* out6) {
  ChanDescH d = make_chan_desc(in_type, m_bins, P, shift);
  out6[0] = d.t0; out6[1] = d.cnt; out6[2] = d.src0; out6[3] = d.dir; out6[4] = d.conj; out6[5] = d.wrap;
  return 0;
}

int emu_channels(const float* spec, int m_bins, int in_type, int P, int olen, int nch,
                 const float* resp, const int* shifts, float* out, int lay_na, int lay_pitch, int lay_off) {
  ChanGeom g;
  if (!build_chan_geom(P, g)) return -1;
  SpecLayout lay{lay_na > 0 ? lay_na : m_bins, lay_na > 0 ? lay_pitch : m_bins, lay_na > 0 ? lay_off : 0};
  std::vector<float2> spec_dev((size_t)((long)(m_bins / lay.na + 2) * lay.pitch + 16), make_float2(0.f, 0.f));
  for (long k = 0; k < m_bins; k++) spec_dev[(size_t)spec_addr(lay, k)] = reinterpret_cast<const float2*>(spec)[k];
  std::vector<ChanDesc> desc((size_t)nch);
  for (int i = 0; i < nch; i++) {
    ChanDescH h = make_chan_desc(in_type, m_bins, P, shifts[i]);
    desc[i] = ChanDesc{h.t0, h.cnt, h.src0, h.dir, h.conj, h.wrap, i, shifts[i]};
  }
  ChanParams c{};
  c.spec = spec_dev.data(); if (!chan_layout(c, lay, m_bins)) return -9;
  c.resp = reinterpret_cast<const float2*>(resp);
  c.desc = desc.data(); c.out = reinterpret_cast<float2*>(out); c.ch0 = 0; c.nch = nch; c.olen = olen;
  c.tw_sub = F2(g.tw_sub);
  const int per_block = g.wpb * g.cpw;
  const int grid = (nch + per_block - 1) / per_block;
  c.stage = getenv("CHZ_CHAN_STAGE") ? atoi(getenv("CHZ_CHAN_STAGE")) : 0;
  if (g.any) { c.m_bins = m_bins; c.m_real = in_type == CHZ_IN_REAL; std::vector<float2> scr(g.big ? (size_t)nch * 2 * g.lb : 0); return launch_chan_any(g, nch, nullptr, c, F2(g.tw_any), false, g.big ? scr.data() : nullptr); }
  return launch_chan(g.r, grid, g.wpb * 64, g.lds, nullptr, c);
}

// COMPLEX slaves with per-channel ISB flags (src/filter.c:895-909)
int emu_channels_isb(const float* spec, int m_bins, int in_type, int P, int olen, int nch,
                     const float* resp, const int* shifts, const unsigned char* isb, float* out) {
  ChanGeom g;
  if (!build_chan_geom(P, g)) return -1;
  SpecLayout lay{m_bins, m_bins, 0};
  std::vector<ChanDesc> desc((size_t)nch);
  for (int i = 0; i < nch; i++) {
    ChanDescH h = make_chan_desc(in_type, m_bins, P, shifts[i]);
    desc[i] = ChanDesc{h.t0, h.cnt, h.src0, h.dir, h.conj, h.wrap, i, shifts[i]};
  }
  ChanParams c{};
  c.spec = reinterpret_cast<const float2*>(spec); if (!chan_layout(c, lay, m_bins)) return -9;
  c.resp = reinterpret_cast<const float2*>(resp);
  c.desc = desc.data(); c.out = reinterpret_cast<float2*>(out); c.ch0 = 0; c.nch = nch; c.olen = olen;
  c.tw_sub = F2(g.tw_sub); c.isb = isb;
  if (g.any) { c.m_bins = m_bins; c.m_real = in_type == CHZ_IN_REAL; std::vector<float2> scr(g.big ? (size_t)nch * 2 * g.lb : 0); return launch_chan_any(g, nch, nullptr, c, F2(g.tw_any), false, g.big ? scr.data() : nullptr); }
  const int per_block = g.wpb * g.cpw;
  const int grid = (nch + per_block - 1) / per_block;
  return launch_chan(g.r, grid, g.wpb * 64, g.lds, nullptr, c);
}

// COMPLEX master, COMPLEX slaves in beam mode (src/filter.c:756-775): ab = 4 doubles per channel
int emu_channels_beam(const float* spec, int m_bins, int P, int olen, int nch, const float* resp, const int* shifts,
                      const double* ab, const unsigned char* on, float* out) {
  ChanGeom g;
  if (!build_chan_geom(P, g)) return -1;
  SpecLayout lay{m_bins, m_bins, 0};
  std::vector<ChanDesc> desc((size_t)nch);
  std::vector<BeamDesc> bd((size_t)nch);
  for (int i = 0; i < nch; i++) {
    ChanDescH h = make_chan_desc(CHZ_IN_COMPLEX, m_bins, P, shifts[i]);
    desc[i] = ChanDesc{h.t0, h.cnt, h.src0, h.dir, h.conj, h.wrap, i, shifts[i]};
    bd[i] = BeamDesc{ab[4 * i], ab[4 * i + 1], ab[4 * i + 2], ab[4 * i + 3], on[i] ? 1 : 0, 0};
  }
  ChanParams c{};
  c.spec = reinterpret_cast<const float2*>(spec); if (!chan_layout(c, lay, m_bins)) return -9;
  c.resp = reinterpret_cast<const float2*>(resp);
  c.desc = desc.data(); c.out = reinterpret_cast<float2*>(out); c.ch0 = 0; c.nch = nch; c.olen = olen;
  c.tw_sub = F2(g.tw_sub); c.beam = bd.data();
  const int per_block = g.wpb * g.cpw;
  const int grid = (nch + per_block - 1) / per_block;
  return launch_chan(g.r, grid, g.wpb * 64, g.lds, nullptr, c);
}

// REAL-output slaves: out = [nch][olen] floats
int emu_channels_real(const float* spec, int m_bins, int in_type, int P, int olen, int nch,
                      const float* resp, const int* shifts, float* out, int lay_na, int lay_pitch, int lay_off) {
  ChanGeom g;
  if (!build_chan_geom(P, g)) return -1;
  SpecLayout lay{lay_na > 0 ? lay_na : m_bins, lay_na > 0 ? lay_pitch : m_bins, lay_na > 0 ? lay_off : 0};
  std::vector<float2> spec_dev((size_t)((long)(m_bins / lay.na + 2) * lay.pitch + 16), make_float2(0.f, 0.f));
  for (long k = 0; k < m_bins; k++) spec_dev[(size_t)spec_addr(lay, k)] = reinterpret_cast<const float2*>(spec)[k];
  ChanParams c{};
  c.spec = spec_dev.data(); if (!chan_layout(c, lay, m_bins)) return -9;
  c.resp = reinterpret_cast<const float2*>(resp);
  std::vector<ChanDesc> desc((size_t)nch);
  for (int i = 0; i < nch; i++) desc[(size_t)i] = ChanDesc{0, 0, 0, 1, 0, 0, i, shifts[i]};   // the REAL-output gather reads row and shift only
  c.desc = desc.data(); c.m_bins = m_bins; c.m_real = in_type == CHZ_IN_REAL;
  c.out = reinterpret_cast<float2*>(out); c.ch0 = 0; c.nch = nch; c.olen = olen;
  c.tw_sub = F2(g.tw_sub);
  if (g.any) { std::vector<float2> scr(g.big ? (size_t)nch * 2 * g.lb : 0); return launch_chan_any(g, nch, nullptr, c, F2(g.tw_any), true, g.big ? scr.data() : nullptr); }
  const int per_block = g.wpb * g.cpw;
  const int grid = (nch + per_block - 1) / per_block;
  return launch_chan_real(g.r, grid, g.wpb * 64, g.lds, nullptr, c);
}

// --- estimate_noise() kernel ------------------------------------------------------------------------
int emu_noise(const float* spec, int m_bins, int in_type, int s_bins, int nch, const int* shifts, double samprate, double* n0,
              int lay_na, int lay_pitch, int lay_off) {
  SpecLayout lay{lay_na > 0 ? lay_na : m_bins, lay_na > 0 ? lay_pitch : m_bins, lay_na > 0 ? lay_off : 0};
  std::vector<float2> spec_dev((size_t)((long)(m_bins / lay.na + 2) * lay.pitch + 16), make_float2(0.f, 0.f));
  for (long k = 0; k < m_bins; k++) spec_dev[(size_t)spec_addr(lay, k)] = reinterpret_cast<const float2*>(spec)[k];
  NoiseParams q = noise_params(m_bins, in_type == CHZ_IN_REAL, s_bins, samprate);
  if (q.nbins > m_bins) return -1;
  std::vector<ChanDesc> desc((size_t)nch);
  for (int i = 0; i < nch; i++) desc[(size_t)i] = ChanDesc{0, 0, 0, 1, 0, 0, i, shifts[i]};
  q.spec = spec_dev.data(); q.lay = lay; q.desc = desc.data(); q.n0 = n0; q.ch0 = 0; q.nch = nch;
  { ChanParams cp{}; if (!chan_layout(cp, lay, m_bins)) return -2; q.magic = cp.magic; q.dpitch = cp.dpitch; }
  return launch_noise(nch, nullptr, q);   // -1: window larger than the compiled sorts
}

// --- fine tuning: host bookkeeping (chz_finetune.h) + the channel kernel's epilogue -----------------
void* emu_fine_create(int nch) { return new std::vector<FineHost>((size_t)nch); }
void emu_fine_delete(void* h) { delete static_cast<std::vector<FineHost>*>(h); }
void emu_fine_retune(void* h, int ch, unsigned job, int olen, int V, int shift, double freq, double rate) {
  fine_retune((*static_cast<std::vector<FineHost>*>(h))[(size_t)ch], job, olen, V, shift, freq, rate);
}
// like emu_channels for block `job`, every channel rotated by its oscillator; power[nch] = mean |y|^2
int emu_channels_tuned(const float* spec, int m_bins, int in_type, int P, int olen, int nch,
                       const float* resp, const int* shifts, float* out, void* fine, int V, unsigned job, double* power) {
  ChanGeom g;
  if (!build_chan_geom(P, g)) return -1;
  SpecLayout lay{m_bins, m_bins, 0};
  std::vector<ChanDesc> desc((size_t)nch);
  std::vector<FineDesc> fd((size_t)nch);
  const std::vector<FineHost>& fh = *static_cast<std::vector<FineHost>*>(fine);
  for (int i = 0; i < nch; i++) {
    ChanDescH h = make_chan_desc(in_type, m_bins, P, shifts[i]);
    desc[i] = ChanDesc{h.t0, h.cnt, h.src0, h.dir, h.conj, h.wrap, i, shifts[i]};
    fd[(size_t)i] = fine_desc(fh[(size_t)i], V, g.r.r1);
  }
  ChanParams c{};
  c.spec = reinterpret_cast<const float2*>(spec); if (!chan_layout(c, lay, m_bins)) return -9;
  c.resp = reinterpret_cast<const float2*>(resp);
  c.desc = desc.data(); c.out = reinterpret_cast<float2*>(out); c.ch0 = 0; c.nch = nch; c.olen = olen;
  c.tw_sub = F2(g.tw_sub);
  c.fine = fd.data(); c.power = power; c.job = job; fine_launch(c, V, job);
  c.stage = getenv("CHZ_CHAN_STAGE") ? atoi(getenv("CHZ_CHAN_STAGE")) : 0;
  const int per_block = g.wpb * g.cpw;
  const int grid = (nch + per_block - 1) / per_block;
  return launch_chan(g.r, grid, g.wpb * 64, g.lds, nullptr, c);
}

// --- small inline masters (filter2): one workgroup per request, window -> olen output samples --------------
int emu_mini(const float* windows, int nreq, int N, int olen, const float* resp, const int* shifts, const unsigned char* isb, float* out) {
  MiniParams p{};
  if (!mini_factor(N, p.radix, &p.nstages)) return -1;
  std::vector<f2> tw((size_t)N);
  for (int k = 0; k < N; k++) tw[(size_t)k] = root_of_unity(k, N, -1);
  std::vector<MiniReq> req((size_t)nreq);
  for (int i = 0; i < nreq; i++) {
    ChanDescH h = make_chan_desc(CHZ_IN_COMPLEX, N, N, shifts[i]);
    req[(size_t)i].d = ChanDesc{h.t0, h.cnt, h.src0, h.dir, h.conj, h.wrap, i, shifts[i]};
    req[(size_t)i].isb = isb[i] != 0; req[(size_t)i].pad = 0;
  }
  p.in = reinterpret_cast<const float2*>(windows); p.out = reinterpret_cast<float2*>(out); p.req = req.data();
  p.resp = reinterpret_cast<const float2*>(resp); p.tw = F2(tw); p.N = N; p.olen = olen;
  const int threads = N / 4 >= 256 ? 256 : (N / 4 >= 64 ? (N / 4 + 63) / 64 * 64 : 64);
  hipLaunchKernelGGL(mini_ovs, dim3(nreq), dim3(threads), sizeof(float2) * 2 * (size_t)N, nullptr, p);
  return 0;
}

// --- linear demodulator tail: chan/state/status are the kernel's own structs (DemodChan, DemodState, DemodStatus) ----------
int emu_demod(const float* in, const double* power, const double* n0, const void* chan, void* state, void* status, unsigned char* pcm,
              int nch, int olen, unsigned job, double blocktime, void* ext) {
  DemodParams d{};
  d.in = reinterpret_cast<const float2*>(in); d.power = power; d.n0 = n0;
  { DemodChan* cw = static_cast<DemodChan*>(const_cast<void*>(chan));      // the host-computed constant, as chz_bank_set_demod leaves it
    for (int i = 0; i < nch; i++) cw[i].recov_ps = std::pow(cw[i].recovery_rate, 1.0 / cw[i].samprate); }
  d.chan = static_cast<const DemodChan*>(chan); d.state = static_cast<DemodState*>(state); d.status = static_cast<DemodStatus*>(status);
  d.ext = static_cast<DemodExt*>(ext); d.flags = nullptr;
  d.pcm = pcm; d.ch0 = 0; d.nch = nch; d.olen = olen; d.pcm_stride = olen * 8; d.job = job; d.blocktime = blocktime; d.power_alpha = 0.10;
  // the coherent modes' PLLs run one channel per lane in a pass of their own when the scratch block exists (as in the engine);
  // EMU_PLL_LANE0=1 keeps round 2's path (lane 0 of the channel's wavefront) for the A/B test
  std::vector<float2> mix;
  if (ext != nullptr && !getenv("EMU_PLL_LANE0")) {
    mix.assign((size_t)nch * olen, make_float2(0.f, 0.f)); d.mix = mix.data();
    for (int i = 0; i < nch; i++) {
      const DemodChan& c = d.chan[i];
      if (!c.on) continue;
      d.lin_pll |= c.kind == 0 && c.pll_enable; d.fm_pll |= c.kind == 1 && c.pll_enable; d.fm_tone |= c.kind == 1 && c.tone_freq != 0;
    }
  }
  // the linear demodulators run one channel per lane (as in the engine); EMU_DEMOD_WAVE=1 keeps them on the wavefront kernel
  { bool any_lin = false, fm_pll = false, fm_nopll = false, pll_lin = false;
    for (int i = 0; i < nch; i++) { const DemodChan& c = d.chan[i]; if (!c.on) continue; any_lin |= c.kind == 0; fm_pll |= c.kind == 1 && c.pll_enable; fm_nopll |= c.kind == 1 && !c.pll_enable; pll_lin |= c.kind == 0 && c.pll_enable; }
    const bool lanes_on = !getenv("EMU_DEMOD_WAVE");
    std::vector<float2>* mp = &mix;
    if (fm_nopll && lanes_on && mp->empty() && !getenv("EMU_PLL_LANE0")) { mp->assign((size_t)nch * olen, make_float2(0.f, 0.f)); d.mix = mp->data(); }
    d.lin_lanes = (any_lin && lanes_on) ? 1 : 0;
    d.fm_lanes = (fm_nopll && lanes_on && d.mix != nullptr) ? 1 : 0;
    d.wave_any = (fm_pll || (fm_nopll && !d.fm_lanes) || (any_lin && !d.lin_lanes) || (pll_lin && d.mix == nullptr)) ? 1 : 0; }
  return launch_demod(nullptr, d);
}
int emu_demod_sizes(int* out3) { out3[0] = (int)sizeof(DemodChan); out3[1] = (int)sizeof(DemodState); out3[2] = (int)sizeof(DemodStatus); return 0; }
// the host-side records of the coherent modes / tone squelch, as chz_bank_set_demod makes them
int emu_demod_ext_size() { return (int)sizeof(DemodExt); }
int emu_demod_ext_init(void* ext, int n) { DemodExt* x = static_cast<DemodExt*>(ext); for (int i = 0; i < n; i++) x[i] = demod_ext_init(); return 0; }
int emu_demod_tone_consts(void* chan, int i, double tone_freq, double samprate) { demod_tone_consts(tone_freq, samprate, static_cast<DemodChan*>(chan)[i]); return 0; }

}  // extern "C"
