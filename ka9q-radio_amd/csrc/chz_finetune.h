// chz_finetune.h -- host-side bookkeeping of the per-channel fine-tuning oscillator.
//
// radiod's downconvert() (src/radio.c:1476-1520) keeps, per channel, a complex rotator that is stepped once
// per output sample (struct osc, src/osc.c:28-70) plus three statements of block-phase bookkeeping.  On
// the GPU the rotation is applied in the channel kernel's epilogue from a CLOSED FORM in the block number
// (FineDesc, chz_kernels.h), so blocks in flight on different streams never share mutable state.  This
// header does the part that happens only when a channel is (re)tuned: evaluate the closed form at the
// block where the change takes effect and re-base it, following the reference's statements one by one.
#pragma once
#include <cmath>
#include "chz_kernels.h"

namespace chz {

struct FineHost {
  // the device descriptor's base
  double phase0 = 0.0;      // cycles, phase before block job0's phase_adjust
  double feff = 0.0;        // phase of osc.phasor_step at the base (cycles/sample)
  double rate = 0.0;        // phase of osc.phasor_step_step (cycles/sample^2)
  unsigned job0 = 0;
  int adj_num = 0;          // (shift mod V), per-block phase_adjust = adj_num/V cycles
  // what the reference compares against
  double set_freq = 0.0, set_rate = 0.0;   // osc.freq, osc.rate (src/osc.h:13-14)
  int bin_shift = -1000999;                // chan->filter.bin_shift start value (src/modes.c:266)
  bool osc_init = false;                   // is_phasor_init(osc.phasor) (src/osc.c:20-24)
  bool on = false;
};

static inline double frac1(double x) { return x - std::floor(x); }

// phase (cycles, before that block's phase_adjust) and step frequency at the start of block `job`
static inline void fine_state_at(const FineHost& h, unsigned job, int olen, int V, double* phase, double* feff) {
  const unsigned kbu = job - h.job0;
  const double kb = (double)kbu, g = kb * (double)olen;
  // kb phase_adjust multiplications so far: (kb * adj_num mod V)/V, exact in integers
  const unsigned r = (unsigned)(((unsigned long long)(kbu % (unsigned)V) * (unsigned)h.adj_num) % (unsigned)V);
  double hi = g * h.feff, lo = std::fma(g, h.feff, -hi);
  hi -= std::rint(hi);
  // sample index g is reached after g steps whose phases are feff + i*rate, i = 1..g (src/osc.c:64-68)
  double q = 0.5 * h.rate * g * (g + 1.0);
  q -= std::rint(q);
  *phase = frac1(h.phase0 + (double)r / (double)V + hi + lo + q);
  *feff = h.feff + g * h.rate;
}

// A tuning update taking effect at block `job` (job - job0 < 2^31): the statements of src/radio.c:1479-1496.
//   freq = -remainder / output samprate, rate = doppler_rate / samprate^2 (the arguments of set_osc, :1481)
//   V    = 1 + L/(M-1), the master's overlap factor (:1492)
static inline void fine_retune(FineHost& h, unsigned job, int olen, int V, int shift, double freq, double rate) {
  double phase = 0.0, feff = 0.0;
  if (h.on) fine_state_at(h, job, olen, V, &phase, &feff);
  double cur_rate = h.rate;
  // set_osc (src/osc.c:28-47): a fresh oscillator starts at phasor 1, freq 0, rate 0
  if (!h.osc_init) { phase = 0.0; feff = 0.0; cur_rate = 0.0; h.set_freq = 0.0; h.set_rate = 0.0; h.osc_init = true; }
  if (freq != h.set_freq) { h.set_freq = freq; feff = freq; }
  if (rate != h.set_rate) { h.set_rate = rate; cur_rate = rate; }
  int adj = h.adj_num;
  if (shift != h.bin_shift) {                                           // src/radio.c:1491-1496
    adj = ((shift % V) + V) % V;                                        // cispi(2*(shift % V)/V): whole turns drop out
    phase += 0.5 * ((double)(shift - h.bin_shift) / (-2.0 * (double)(V - 1)));   // cispi(x) turns by x/2 cycles
    h.bin_shift = shift;
  }
  h.phase0 = frac1(phase); h.feff = feff; h.rate = cur_rate; h.job0 = job; h.adj_num = adj; h.on = true;
}

// stride: the register-tiled channel kernel's lanes hold every stride-th output sample (its R1); 0 for kernels that do not use the step
static inline FineDesc fine_desc(const FineHost& h, int V, int stride = 0) {
  FineDesc d;
  d.phase0 = h.phase0; d.freq = h.feff; d.rate = h.rate; d.job0 = h.job0; d.adj_num = h.adj_num; d.V = V; d.on = h.on ? 1 : 0;
  d.job0m = (int)(h.job0 % (unsigned)V); d.stride = stride;
  const double a = 2.0 * M_PI * std::remainder((double)stride * h.feff, 1.0);
  d.step_c = std::cos(a); d.step_s = std::sin(a);
  return d;
}
// the per-launch integers of the block phase correction (ChanParams::fine_*)
static inline void fine_launch(ChanParams& c, int V, unsigned job) {
  c.fine_V = V; c.fine_invV = 1.0 / (double)V; c.fine_jobm = (int)(job % (unsigned)V); c.fine_wrapm = (int)(4294967296ull % (unsigned long long)V);
}

}  // namespace chz
