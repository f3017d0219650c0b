// chz_launch.h -- runtime (R1,R2) -> compiled kernel instantiation dispatch.
// Shared by chz_engine.hip (hipcc, real launches on a HIP stream) and the CPU
// test harness (tests/hipemu), so both exercise the same template instances.
#pragma once
#include <cstring>
#include <cmath>
#include <cmath>
#include "chz_kernels.h"
#include "chz_plan.h"

namespace chz {

// Plain launch, or -- when an event pair is supplied -- hipExtLaunchKernelGGL, whose events carry
// the dispatch packet's own begin/end timestamps (the same clock rocprofv3 --kernel-trace reads), so
// per-kernel times measured in-process agree with the profiler.
// CHZ_EXIT_SCOPE: the engine library (chz_engine.hip) stops issuing work to the runtime once the process has begun to exit -- see chz_exit there.
// Other includers (the CPU emulation of the tests) launch unconditionally.
#ifndef CHZ_EXIT_SCOPE
#define CHZ_EXIT_SCOPE(name) struct { bool ok; } name = {true}
#endif
#define CHZ_LAUNCH(kern, grid, block, lds, s, ev0, ev1, p)                                              \
  do {                                                                                                  \
    CHZ_EXIT_SCOPE(_xs);                                                                                \
    if (!_xs.ok) break;                                                                                 \
    if ((ev0) || (ev1)) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (unsigned)(lds), s, (ev0), (ev1), 0, p); \
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, p);                                \
  } while (0)

inline int launch_first_real(Radix2 r, dim3 grid, int block, size_t lds, hipStream_t s, const FirstRealParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { CHZ_LAUNCH((fwd_first_real<a, b>), grid, block, lds, s, e0, e1, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_cols(Radix2 r, dim3 grid, int block, size_t lds, hipStream_t s, const ColsParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { CHZ_LAUNCH((fwd_cols<a, b>), grid, block, lds, s, e0, e1, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
inline int launch_rows(Radix2 r, dim3 grid, int block, size_t lds, hipStream_t s, const RowsParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { CHZ_LAUNCH((fwd_rows<a, b>), grid, block, lds, s, e0, e1, p); return 0; }
  CHZ_FWD_MENU(X)
#undef X
  return -1;
}
// the spectrum layout as the channel kernels want it: 1/na for the float split (REAL-output and beam paths), and the
// multiply-high reciprocal of na for the division-free index of chan_ifft.  Returns false if the reciprocal would not be exact
// over this master's bins (cannot happen for compiled axis lengths: na <= 400 and bins < 2^23).
inline bool chan_layout(ChanParams& c, const SpecLayout& lay, long bins) {
  c.lay = lay; c.inv_na = 1.0f / (float)lay.na;
  c.dpitch = lay.pitch - lay.na;
  const unsigned long long m = ((1ull << 32) + (unsigned long long)lay.na - 1) / (unsigned long long)lay.na;      // ceil(2^32 / na)
  c.magic = (unsigned)(m > 0xffffffffull ? 0xffffffffull : m);
  if (c.dpitch == 0) return true;                          // natural order: the quotient is multiplied by zero
  if (c.dpitch < 0 || c.dpitch >= (1 << 24) || bins >= (1 << 24)) return false;
  const unsigned long long e = m * (unsigned long long)lay.na - (1ull << 32);                                        // < na
  return e * (unsigned long long)(bins > 0 ? bins - 1 : 0) < (1ull << 32);
}
inline int launch_chan(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const ChanParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { \
    if (p.isb || p.beam) { CHZ_LAUNCH((chan_ifft<a, b, 2>), grid, block, lds, s, e0, e1, p); } \
    else if (p.fine || p.power) { CHZ_LAUNCH((chan_ifft<a, b, 1>), grid, block, lds, s, e0, e1, p); } \
    else { CHZ_LAUNCH((chan_ifft<a, b, 0>), grid, block, lds, s, e0, e1, p); } \
    return 0; }
  CHZ_CHAN_MENU(X)
#undef X
  return -1;
}
inline int launch_chan_real(Radix2 r, int grid, int block, size_t lds, hipStream_t s, const ChanParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
#define X(a, b) if (r.r1 == a && r.r2 == b) { if constexpr ((a * b) % 2 == 0) { CHZ_LAUNCH((chan_c2r<a, b>), grid, block, lds, s, e0, e1, p); return 0; } }
  CHZ_CHAN_MENU(X)
#undef X
  return -1;
}
// a kernel that wants more than the default 64 KB of dynamic LDS has to say so once (up to the CU's 160 KB)
inline int big_lds_prepare(const void* kern) {
#if defined(__HIPCC__) && !defined(HIPEMU)
  return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 0 : -1;
#else
  (void)kern; return 0;
#endif
}
// any P (prime factors up to 13) without a register-tiled kernel: one workgroup per channel (chan_any); needs the large-LDS attribute once
inline int chan_any_prepare() {
#if defined(__HIPCC__) && !defined(HIPEMU)
  static int done = -1;
  if (done < 0) done = hipFuncSetAttribute(reinterpret_cast<const void*>(chan_any<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 1 : 0;
  return done == 1 ? 0 : -1;
#else
  return 0;
#endif
}
inline int launch_chan_any(const ChanGeom& g, int nch, hipStream_t s, const ChanParams& p, const float2* tw, bool real_out, float2* scratch = nullptr,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  if (nch <= 0) return 0;
  if (g.big && !scratch) return -1;
  AnyParams q{};
  q.c = p; q.real_out = real_out ? 1 : 0; q.scratch = scratch;
  q.P = g.P;
  q.m.N = g.lb; q.m.olen = p.olen; q.m.nstages = g.nstages; q.m.tw = tw;
  for (int i = 0; i < g.nstages; i++) q.m.radix[i] = g.radix[i];
  if (g.big) { CHZ_LAUNCH(chan_any<true>, nch, g.any_threads, 0, s, e0, e1, q); }
  else { CHZ_LAUNCH(chan_any<false>, nch, g.any_threads, g.lds, s, e0, e1, q); }
  return 0;
}
inline int launch_notch_fix(hipStream_t s, const NotchFixParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  if (p.n <= 0) return 0;
  if (p.n > 1024) return -1;
  CHZ_LAUNCH(notch_fix, 1, (p.n + 63) / 64 * 64, 0, s, e0, e1, p);
  return 0;
}
// Host side of K2: the reference walks its list in order (src/filter.c:464-474); entries naming a bin a second time are
// chained behind the first so one lane applies them in list order.
struct NotchTables {
  std::vector<int> addr, next, head;
  // small lists ride in the kernel arguments
  void fill_inline(NotchFixParams& q, const double* alpha) const {
    const int n = (int)addr.size();
    q.inl = n <= CHZ_NOTCH_INLINE;
    if (!q.inl) return;
    for (int i = 0; i < n; i++) { q.i_addr[i] = addr[(size_t)i]; q.i_next[i] = (signed char)next[(size_t)i]; q.i_head[i] = (signed char)head[(size_t)i]; q.i_alpha[i] = alpha[i]; }
  }
};
inline NotchTables notch_tables(const int* bins, int n, const SpecLayout& lay) {
  NotchTables t;
  t.addr.resize((size_t)n); t.next.assign((size_t)n, -1); t.head.assign((size_t)n, 1);
  for (int i = 0; i < n; i++) {
    t.addr[(size_t)i] = (int)spec_addr(lay, bins[i]);
    for (int j = i - 1; j >= 0; j--)
      if (bins[j] == bins[i]) { t.next[(size_t)j] = i; t.head[(size_t)i] = 0; break; }
  }
  return t;
}
// K2 folded into fwd_rows: which thread of which workgroup stores listed bin `bin`, and as which second-layer output.  Restates the
// pass's own index arithmetic (fwd_rows in chz_kernels.h): workgroup = at * Nb + kb holds ka in [at*Ta - ka_shift, ...) at one kb,
// thread = k1 * Ta + r stores bins ka + Na * (kb + Nb * (k1 + R1 * K2)); for a real master the rows ka > Na/2 are never computed:
// those bins leave as the conjugate mirror image of bin N - k, written by the thread that holds row Na - ka, column xrows - 1 - x.
// Returns false for a bin nobody stores (cannot happen for bin < bins); the kernel checks the storage index again before it acts.
inline bool notch_owner(const FwdPlan& p, bool real, int bin, int* wg, int* tid, int* k2) {
  const int R1 = p.rc.r1, R2 = p.rc.r2;
  const long xrows = (long)p.Nb * p.Nc;
  long ka = bin % p.Na, x = bin / p.Na;
  if (real && ka >= p.Ra) { ka = p.Na - ka; x = xrows - 1 - x; }        // the mirror image's owner
  if (ka < 0 || ka >= p.Ra || x < 0 || x >= xrows) return false;
  const long kb = x % p.Nb, kc = x / p.Nb;
  const long k1 = kc % R1, K2 = kc / R1;
  if (K2 >= R2) return false;
  const long at = (ka + p.ka_shift) / p.Ta, r = (ka + p.ka_shift) % p.Ta;
  *wg = (int)(at * p.Nb + kb); *tid = (int)(k1 * p.Ta + r); *k2 = (int)K2;
  return *wg < p.grid3 && *tid < p.block3;
}
// the pass's notch block for the engine's list (n <= CHZ_NOTCH_INLINE entries): owner workgroups into the kernel arguments, the
// per-entry owner records into `own` (uploaded by the caller); false: some bin has no owner (the notch_fix kernel keeps the list)
inline bool rows_notch_fill(RowsNotch& nf, std::vector<NotchOwn>& own, const FwdPlan& p, bool real, const int* bins, int n) {
  nf = RowsNotch{}; own.clear();
  if (n < 1 || n > CHZ_NOTCH_INLINE) return false;
  for (int i = 0; i < n; i++) {
    int wg = 0, tid = 0, k2 = 0;
    if (!notch_owner(p, real, bins[i], &wg, &tid, &k2) || k2 > 31) { nf = RowsNotch{}; own.clear(); return false; }
    // one thread remembers ONE listed bin: two different bins stored by the same thread (Na*Nb*R1 bins apart) keep the kernel
    for (int j = 0; j < i; j++) if (bins[j] != bins[i] && own[(size_t)j].wg == wg && own[(size_t)j].tid == tid) { nf = RowsNotch{}; own.clear(); return false; }
    nf.wg[i] = wg;
    own.push_back(NotchOwn{wg, tid, k2, 0});
    bool seen = false;
    for (int j = 0; j < i; j++) seen = seen || nf.wg[j] == wg;
    if (!seen) nf.nwg++;
  }
  nf.n = n;
  return true;
}
inline int launch_demod(hipStream_t s, const DemodParams& p_in, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  if (p_in.nch <= 0) return 0;
  DemodParams p = p_in;
  p.fm_alpha = -std::expm1(-p.blocktime / 1.0);
  const size_t lds = 16 * (size_t)p.olen;                                          // one wavefront per channel; LDS: N doubles + N complex
  if (lds > 160 * 1024) return -1;                                                 // blocks of more than 10240 samples do not fit a CU
  if (lds > 64 * 1024) {
    static int big = -1;
    if (big < 0) big = big_lds_prepare(reinterpret_cast<const void*>(demod_linear_tail)) == 0 ? 1 : 0;
    if (big != 1) return -1;
  }
  // The sample-by-sample recurrences run one channel per lane in passes of their own, around the wavefront-per-channel kernel (only
  // when the caller provided the scratch block: the engine does as soon as a channel of the bank asks for one of them).  Timed
  // together when instrumented: the first dispatch starts the clock, the last stops it.
  const bool lanes = p.mix != nullptr;
  const bool lin = lanes && p.lin_pll, fpll = lanes && p.fm_pll, ftone = lanes && p.fm_tone;
  const bool linl = p.lin_lanes != 0, fml = lanes && p.fm_lanes != 0, wave = p.wave_any != 0 || !(linl || fml);
  if ((fpll || ftone) && 8 * (size_t)p.olen > 64 * 1024) {
    static int big2 = -1;
    if (big2 < 0) big2 = (big_lds_prepare(reinterpret_cast<const void*>(fm_front_k)) == 0 && big_lds_prepare(reinterpret_cast<const void*>(fm_finish)) == 0) ? 1 : 0;
    if (big2 != 1) return -1;
  }
  const size_t tl = sizeof(float2) * 64 * (PLL_TILE + 1);
  const int groups = (p.nch + 63) / 64;
  // dispatches of this call, in order; the first one carries e0, the last one e1
  const int total = (lin ? 1 : 0) + (linl ? 1 : 0) + (fml ? 1 : 0) + (wave ? (fpll ? 2 : 0) + 1 + (ftone ? 2 : 0) : 0);
  int k = 0;
  auto E0 = [&]() { return k == 0 ? e0 : (hipEvent_t) nullptr; };
  auto E1 = [&]() { return k == total - 1 ? e1 : (hipEvent_t) nullptr; };
  if (lin) { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(pll_lanes, groups, 64, tl, s, a, b, p); k++; }
  if (linl) { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(demod_lin_lanes, groups, 64, sizeof(float2) * 64 * (LIN_TILE + 1) + sizeof(LinRow) * 64, s, a, b, p); k++; }
  if (fml) { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(demod_fm_lanes, groups, 64, sizeof(float2) * 64 * (LIN_TILE + 1) + sizeof(LinRow) * 64, s, a, b, p); k++; }
  if (wave) {
    if (fpll) {
      { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(fm_front_k, p.nch, 64, 8 * (size_t)p.olen, s, a, b, p); k++; }
      { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(fm_pll_lanes, groups, 64, sizeof(float2) * 64 * (FMP_TILE + 1), s, a, b, p); k++; }
    }
    { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(demod_linear_tail, p.nch, 64, lds, s, a, b, p); k++; }
    if (ftone) {
      { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(fm_tone_lanes, groups, 64, sizeof(float) * 64 * (FM_TILE + 1), s, a, b, p); k++; }
      { hipEvent_t a = E0(), b = E1(); CHZ_LAUNCH(fm_finish, p.nch, 64, 8 * (size_t)p.olen, s, a, b, p); k++; }
    }
  }
  return 0;
}
// host side of the coherent modes and the PL-tone squelch: what init_pll() (src/osc.c:130-136, called once when the demodulator
// starts, src/linear.c:41) and the tone set-up of demod_fm() (src/fm.c:50-61) leave behind
inline DemodExt demod_ext_init() {
  DemodExt x; std::memset(&x, 0, sizeof x);
  pll_init(x.pll);
  x.tone_mute = 1;                                     // muted until the tone has been seen (src/fm.c:61)
  return x;
}
inline void demod_tone_consts(double tone_freq, double samprate, DemodChan& c) {          // init_goertzel(), src/iir.c:32-39
  c.tone_freq = tone_freq;
  const double f = tone_freq / (double)(int)samprate;
  c.g_coeff = 2 * std::cos(2 * M_PI * f); c.g_cfr = std::cos(2 * M_PI * f); c.g_cfi = -std::sin(2 * M_PI * f);
}
// |X|^2 of every bin of a slot in bin order (+ CHZ_ENERGY_TAIL repeated bins): the input of the EN noise kernels
inline size_t spec_energy_floats(int bins) { return (size_t)bins + CHZ_ENERGY_TAIL + 64; }
inline void launch_spec_energy(const float2* spec, float* energy, int bins, const SpecLayout& lay, unsigned magic, int dpitch, hipStream_t s,
                               hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  EnergyParams q{spec, energy, bins, lay.off, magic, dpitch};
  long grid = ((long)bins + 255) / 256; if (grid > 2048) grid = 2048;
  CHZ_LAUNCH(spec_energy, (int)grid, 256, 0, s, e0, e1, q);
}
inline void launch_desc_push(const PushParams& p, hipStream_t s) {
  size_t words = 0;
  for (int g = 0; g < p.nseg; g++) words += (p.seg[g].bytes + 3) / 4;
  long grid = ((long)words + 1023) / 1024; if (grid < 1) grid = 1; if (grid > 128) grid = 128;
  CHZ_LAUNCH(desc_push, (int)grid, 256, 0, s, nullptr, nullptr, p);
}
inline int launch_noise(int nch, hipStream_t s, const NoiseParams& p, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  const int grid = (nch + 3) / 4;          // four wavefronts = four channels per workgroup
  if (p.energy) {
    if (p.nsort == 1024) { CHZ_LAUNCH((noise_est<16, true>), grid, 256, 0, s, e0, e1, p); return 0; }
    if (p.nsort == 2048) { CHZ_LAUNCH((noise_est<32, true>), grid, 256, 0, s, e0, e1, p); return 0; }
    return -1;
  }
  if (p.nsort == 1024) { CHZ_LAUNCH((noise_est<16>), grid, 256, 0, s, e0, e1, p); return 0; }
  if (p.nsort == 2048) { CHZ_LAUNCH((noise_est<32>), grid, 256, 0, s, e0, e1, p); return 0; }
  return -1;
}
// host side of K5: window size, sort size and the constant factor of estimate_noise() (src/radio.c:73-76,1840-1865)
inline NoiseParams noise_params(int m_bins, bool real, int s_bins, double samprate) {
  NoiseParams p{};
  p.m_bins = m_bins; p.real = real ? 1 : 0;
  p.nbins = s_bins < 1000 ? 1000 : s_bins;
  p.nsort = p.nbins <= 1024 ? 1024 : (p.nbins <= 2048 ? 2048 : 0);   // 0: no kernel (launch_noise fails)
  const double NQ = 0.10, N_cutoff = 1.5;
  const double z = N_cutoff * (-std::log(1.0 - NQ));
  const double correction = 1.0 / (1.0 - z * std::exp(-z) / (1.0 - std::exp(-z)));
  p.scale = correction / ((double)m_bins * samprate);
  return p;
}
}  // namespace chz
