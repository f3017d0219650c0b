/* oracle/ref_driver.c -- TEST INFRASTRUCTURE (CPU oracle), not a product path.
 *
 * A flat C handle API over the *reference's own* filter.h (compiled from
 * /root/reference/src where it lies, see oracle/Makefile) so Python tests can
 * drive it through ctypes without mirroring struct layouts.  This file is
 * project code; it only *calls* the reference:
 *   create_filter_input   src/filter.c:186      write_rfilter/cfilter src/filter.c:1093,1114
 *   create_filter_output  src/filter.c:298      execute_filter_output src/filter.c:663
 *   set_filter            src/filter.c:968      delete_filter_*       src/filter.c:930,943
 *
 * With N_worker_threads == 0 the master runs "perform_inline"
 * (src/filter.c:205,562-600): the forward transform executes on the calling
 * thread and the same-thread shortcut (src/filter.c:681-683) makes
 * execute_filter_output pick the newest block without waiting, which gives a
 * deterministic single-threaded oracle.  refchz_bench_* below uses the worker
 * thread mode radiod uses.
 */
#define _GNU_SOURCE 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <complex.h>
#include "filter.h"     /* the reference's, via -iquote /root/reference/src */
#include "window.h"

int Verbose = 0;        /* normally defined by main.c; misc.c references it */
const char *App_path = "oracle";

/* libbsd headers are absent from the image; misc.c wants these two. */
size_t strlcpy(char *dst, const char *src, size_t siz) {
  size_t n = strlen(src);
  if (siz) { size_t c = n >= siz ? siz - 1 : n; memcpy(dst, src, c); dst[c] = 0; }
  return n;
}
size_t strlcat(char *dst, const char *src, size_t siz) {
  size_t d = strnlen(dst, siz);
  if (d == siz) return siz + strlen(src);
  return d + strlcpy(dst + d, src, siz - d);
}

struct refchz_master { struct filter_in in; };
struct refchz_chan { struct filter_out out; };

/* in_type / out_type use the reference's enum filtertype numbering:
   1 = COMPLEX, 2 = REAL, 3 = SPECTRUM (src/filter.h:29-34) */
static int Workers_running;   /* run_fft threads alive in this process */
void *refchz_master_create(int L, int M, int in_type, int worker_threads) {
  /* fft_init() (src/filter.c:1047) starts N_worker_threads workers exactly once
     per process; later masters asking for more get them started here through
     the exported run_fft (src/filter.h:107). */
  static int first = 1;
  N_worker_threads = worker_threads;
  if (first) { first = 0; Workers_running = worker_threads; }
  while (Workers_running < worker_threads) {
    pthread_t t; pthread_create(&t, NULL, run_fft, NULL); Workers_running++;
  }
  struct refchz_master *m = calloc(1, sizeof *m);
  if (!m) return NULL;
  if (create_filter_input(&m->in, L, M, (enum filtertype)in_type) != 0) { free(m); return NULL; }
  return m;
}
void refchz_master_delete(void *h) {
  struct refchz_master *m = h;
  if (!m) return;
  delete_filter_input(&m->in);
  free(m);
}
int refchz_master_bins(void *h) { return ((struct refchz_master *)h)->in.bins; }
int refchz_master_points(void *h) { return ((struct refchz_master *)h)->in.points; }
unsigned refchz_master_next_jobnum(void *h) { return ((struct refchz_master *)h)->in.next_jobnum; }

/* bins: array of n bin indices, LAST one must be 0 (DC), as radio.c builds it
   (src/radio.c:601-620); alpha as there. */
int refchz_master_set_notches(void *h, const int *bins, int n, double alpha) {
  struct refchz_master *m = h;
  struct notch_state *ns = calloc((size_t)n, sizeof *ns);
  if (!ns) return -1;
  for (int i = 0; i < n; i++) { ns[i].bin = bins[i]; ns[i].alpha = alpha; ns[i].state = 0; }
  free(m->in.notches);
  m->in.notches = ns;
  return 0;
}

/* feed n samples (floats for REAL, interleaved re/im pairs for COMPLEX).
   returns what write_rfilter/write_cfilter returns (1 = a block was transformed) */
int refchz_master_write(void *h, const float *samples, int n) {
  struct refchz_master *m = h;
  if (m->in.in_type == REAL) return write_rfilter(&m->in, samples, n);
  return write_cfilter(&m->in, (const float complex *)samples, n);
}

/* copy spectrum of job `jobnum` (must be one of the last ND) into out (2*bins floats) */
int refchz_master_spectrum(void *h, unsigned jobnum, float *out) {
  struct refchz_master *m = h;
  memcpy(out, m->in.fdomain[jobnum % ND], sizeof(float complex) * (size_t)m->in.bins);
  return m->in.bins;
}

void *refchz_chan_create(void *h, int olen, int out_type) {
  struct refchz_master *m = h;
  struct refchz_chan *c = calloc(1, sizeof *c);
  if (!c) return NULL;
  if (create_filter_output(&c->out, &m->in, olen, (enum filtertype)out_type) != 0) { free(c); return NULL; }
  return c;
}
void refchz_chan_delete(void *h) {
  struct refchz_chan *c = h;
  if (!c) return;
  delete_filter_output(&c->out);
  free(c);
}
int refchz_chan_points(void *h) { return ((struct refchz_chan *)h)->out.points; }
int refchz_chan_bins(void *h) { return ((struct refchz_chan *)h)->out.bins; }
unsigned refchz_chan_drops(void *h) { return ((struct refchz_chan *)h)->out.block_drops; }
void refchz_chan_set_isb(void *h, int isb) { ((struct refchz_chan *)h)->out.isb = isb != 0; }

int refchz_chan_set_filter(void *h, double low, double high, double beta) {
  return set_filter(&((struct refchz_chan *)h)->out, low, high, beta);
}
/* copy the frequency response (points complex for COMPLEX out, bins for REAL) */
int refchz_chan_response(void *h, float *out) {
  struct refchz_chan *c = h;
  if (!c->out.response) return -1;
  memcpy(out, c->out.response, sizeof(float complex) * (size_t)c->out.points);
  return c->out.points;
}
/* install an arbitrary response (points complex), as callers that bypass
   set_filter do; ownership passes to the reference (freed in delete) */
int refchz_chan_set_response(void *h, const float *resp) {
  struct refchz_chan *c = h;
  float complex *r = NULL;
  if (posix_memalign((void **)&r, 64, sizeof(float complex) * (size_t)c->out.points)) return -1;
  memcpy(r, resp, sizeof(float complex) * (size_t)c->out.points);
  pthread_mutex_lock(&c->out.response_mutex);
  float complex *old = c->out.response;
  c->out.response = r;
  pthread_mutex_unlock(&c->out.response_mutex);
  free(old);
  return 0;
}

/* run the channel on the newest block; copies olen outputs (complex pairs for
   COMPLEX out, floats for REAL out) */
int refchz_chan_execute(void *h, int shift, float *out) {
  struct refchz_chan *c = h;
  int r = execute_filter_output(&c->out, shift);
  if (r != 0) return r;
  if (c->out.out_type == COMPLEX) memcpy(out, c->out.output.c, sizeof(float complex) * (size_t)c->out.olen);
  else if (c->out.out_type == REAL) memcpy(out, c->out.output.r, sizeof(float) * (size_t)c->out.olen);
  return 0;
}
/* beam mode (src/radio.c:938-940): the caller sets out.beam and the weights after create.  The reference never
   initialises fdomain and its beam branch has no trailing zero fill (src/filter.c:756-775), so the scratch vector is
   cleared once here to make the untouched bins comparable. */
int refchz_chan_set_beam(void *h, double iw_re, double iw_im, double qw_re, double qw_im) {
  struct refchz_chan *c = h;
  c->out.beam = true;
  memset(c->out.fdomain, 0, sizeof(float complex) * (size_t)c->out.bins);
  return set_filter_weights(&c->out, iw_re + I * iw_im, qw_re + I * qw_im);
}
void refchz_chan_weights(void *h, double *ab) {
  struct refchz_chan *c = h;
  ab[0] = creal(c->out.alpha); ab[1] = cimag(c->out.alpha); ab[2] = creal(c->out.beta); ab[3] = cimag(c->out.beta);
}
/* the gathered & weighted frequency-domain vector that fed the last IFFT */
int refchz_chan_fdomain(void *h, float *out) {
  struct refchz_chan *c = h;
  memcpy(out, c->out.fdomain, sizeof(float complex) * (size_t)c->out.bins);
  return c->out.bins;
}

/* window.c / misc.c pieces exposed for pinning the restatement */
int refchz_make_kaiserf(float *w, int M, double beta) { return make_kaiserf(w, M, beta); }
double refchz_i0(double z) { return i0(z); }

/* ------------------------------------------------------------------------
 * CPU baseline leg (bench.py "cpu_baseline", kind "reference"): the reference's
 * filter.c driven the way radiod drives it -- forward transforms on
 * N_worker_threads worker thread(s) (src/filter.c:485-555), channels on a pool
 * of threads each looping execute_filter_output over a fixed channel subset
 * (thread-per-channel, src/radio.c:996, without spawning thousands of threads).
 * Timing with CLOCK_MONOTONIC as src/filter.c:500-520 does.
 * ------------------------------------------------------------------------ */
struct bench_pool_arg {
  struct refchz_chan **ch; const int *shift; int first, last; int blocks;
};
static void *bench_pool_thread(void *a) {
  struct bench_pool_arg *p = a;
  for (int b = 0; b < p->blocks; b++)
    for (int i = p->first; i < p->last; i++)
      execute_filter_output(&p->ch[i]->out, p->shift[i]);   /* blocks until the block's spectrum exists */
  return NULL;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* Feed `blocks` blocks of L samples taken cyclically from `ring` (ring_blocks*L
   samples) and run every channel on every block.  Returns elapsed seconds. */
double refchz_bench(void *mh, void **chans, const int *shifts, int nchan,
                    const float *ring, int ring_blocks, int blocks, int pool_threads) {
  struct refchz_master *m = mh;
  int L = m->in.ilen;
  int per = m->in.in_type == REAL ? 1 : 2;
  if (pool_threads < 1) pool_threads = 1;
  if (pool_threads > nchan) pool_threads = nchan > 0 ? nchan : 1;
  pthread_t *th = calloc((size_t)pool_threads, sizeof *th);
  struct bench_pool_arg *args = calloc((size_t)pool_threads, sizeof *args);
  /* channels created earlier start at the master's current job number */
  for (int i = 0; i < nchan; i++)
    ((struct refchz_chan *)chans[i])->out.next_jobnum = m->in.next_jobnum;
  double t0 = now_s();
  for (int t = 0; t < pool_threads; t++) {
    args[t].ch = (struct refchz_chan **)chans; args[t].shift = shifts; args[t].blocks = blocks;
    args[t].first = (int)((long)nchan * t / pool_threads);
    args[t].last = (int)((long)nchan * (t + 1) / pool_threads);
    pthread_create(&th[t], NULL, bench_pool_thread, &args[t]);
  }
  for (int b = 0; b < blocks; b++) {
    const float *src = ring + (size_t)(b % ring_blocks) * (size_t)L * per;
    /* keep at most ND-1 blocks in flight so no channel is ever lapped (src/filter.c:690-701) */
    for (;;) {
      unsigned slowest = m->in.next_jobnum;
      for (int i = 0; i < nchan; i++) {
        unsigned nj = *(volatile unsigned *)&((struct refchz_chan *)chans[i])->out.next_jobnum;
        if ((int)(nj - slowest) < 0) slowest = nj;   /* wrap-safe minimum */
      }
      if (nchan == 0 || (int)(m->in.next_jobnum - slowest) < ND - 1) break;
      struct timespec ts = {0, 20000}; nanosleep(&ts, NULL);
    }
    if (m->in.in_type == REAL) write_rfilter(&m->in, src, L);
    else write_cfilter(&m->in, (const float complex *)src, L);
  }
  for (int t = 0; t < pool_threads; t++) pthread_join(th[t], NULL);
  double t1 = now_s();
  free(th); free(args);
  return t1 - t0;
}

/* The same run with a clock on every block: stats[0] = worst and stats[1] = mean time between the completions of two consecutive
   blocks (a block is complete when the LAST pool thread has run its channels on it), stats[2] = worst time from handing a block's
   samples to write_?filter to its completion, all in ms, over blocks skip.. only (the first ones fill the pipeline).  This is the
   CPU side of bench.py's C_rt definition: a channel count is sustained if EVERY block completes within 20 ms of the one before. */
struct bench_pool_arg2 { struct bench_pool_arg a; double *done; };
static void *bench_pool_thread2(void *v) {
  struct bench_pool_arg2 *p = v;
  for (int b = 0; b < p->a.blocks; b++) {
    for (int i = p->a.first; i < p->a.last; i++)
      execute_filter_output(&p->a.ch[i]->out, p->a.shift[i]);
    p->done[b] = now_s();
  }
  return NULL;
}
/* pace_us > 0: the front end hands a block over every pace_us microseconds of WALL CLOCK (absolute deadlines) and never waits for a
   channel, as a real A/D front end does (src/sig_gen.c:357-362, src/rx888.c): a channel that falls ND blocks behind is lapped and
   counts a drop (src/filter.c:686-701); stats[3] = total block_drops of all channels over the run.  This is "sustained in real time"
   exactly as radiod experiences it.  pace_us == 0: free-running with back-pressure (never more than ND-1 blocks ahead). */
double refchz_bench_blocks(void *mh, void **chans, const int *shifts, int nchan, const float *ring, int ring_blocks, int blocks,
                           int pool_threads, int skip, double *stats, int pace_us) {
  struct refchz_master *m = mh;
  int L = m->in.ilen;
  int per = m->in.in_type == REAL ? 1 : 2;
  if (pool_threads < 1) pool_threads = 1;
  if (pool_threads > nchan) pool_threads = nchan > 0 ? nchan : 1;
  pthread_t *th = calloc((size_t)pool_threads, sizeof *th);
  struct bench_pool_arg2 *args = calloc((size_t)pool_threads, sizeof *args);
  double *done = calloc((size_t)pool_threads * (size_t)blocks, sizeof *done), *in = calloc((size_t)blocks, sizeof *in);
  for (int i = 0; i < nchan; i++)
    ((struct refchz_chan *)chans[i])->out.next_jobnum = m->in.next_jobnum;
  double t0 = now_s();
  for (int t = 0; t < pool_threads; t++) {
    args[t].a.ch = (struct refchz_chan **)chans; args[t].a.shift = shifts; args[t].a.blocks = blocks;
    args[t].a.first = (int)((long)nchan * t / pool_threads);
    args[t].a.last = (int)((long)nchan * (t + 1) / pool_threads);
    args[t].done = done + (size_t)t * (size_t)blocks;
    pthread_create(&th[t], NULL, bench_pool_thread2, &args[t]);
  }
  for (int b = 0; b < blocks; b++) {
    const float *src = ring + (size_t)(b % ring_blocks) * (size_t)L * per;
    if (pace_us > 0) {
      double const due = t0 + (double)(b + 1) * (double)pace_us * 1e-6;        /* the block's last sample arrives now */
      struct timespec d = {.tv_sec = (time_t)due, .tv_nsec = (long)((due - (double)(time_t)due) * 1e9)};
      while (clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &d, NULL) != 0) { }
    } else for (;;) {
      unsigned slowest = m->in.next_jobnum;
      for (int i = 0; i < nchan; i++) {
        unsigned nj = *(volatile unsigned *)&((struct refchz_chan *)chans[i])->out.next_jobnum;
        if ((int)(nj - slowest) < 0) slowest = nj;
      }
      if (nchan == 0 || (int)(m->in.next_jobnum - slowest) < ND - 1) break;
      struct timespec ts = {0, 20000}; nanosleep(&ts, NULL);
    }
    in[b] = now_s();
    if (m->in.in_type == REAL) write_rfilter(&m->in, src, L);
    else write_cfilter(&m->in, (const float complex *)src, L);
  }
  for (int t = 0; t < pool_threads; t++) pthread_join(th[t], NULL);
  double t1 = now_s();
  double worst = 0, sum = 0, lat = 0, prev = 0; int n = 0;
  for (int b = 0; b < blocks; b++) {
    double d = 0;
    for (int t = 0; t < pool_threads; t++) if (done[(size_t)t * (size_t)blocks + b] > d) d = done[(size_t)t * (size_t)blocks + b];
    if (b >= skip && b > 0) {
      double const iv = d - prev;
      if (iv > worst) worst = iv;
      sum += iv; n++;
      if (d - in[b] > lat) lat = d - in[b];
    }
    prev = d;
  }
  unsigned long drops = 0;
  for (int i = 0; i < nchan; i++) drops += ((struct refchz_chan *)chans[i])->out.block_drops;
  if (stats) { stats[0] = worst * 1e3; stats[1] = n ? sum / n * 1e3 : 0; stats[2] = lat * 1e3; stats[3] = (double)drops; }
  free(th); free(args); free(done); free(in);
  return t1 - t0;
}

/* between two paced probes: block_drops is a running total per slave (src/filter.c:700), the next probe counts from zero */
void refchz_reset_drops(void **chans, int nchan) {
  for (int i = 0; i < nchan; i++) ((struct refchz_chan *)chans[i])->out.block_drops = 0;
}

extern int64_t Min_fft_time, Max_fft_time, Avg_fft_time;
void refchz_fft_times(long long *mn, long long *mx, long long *avg) { *mn = Min_fft_time; *mx = Max_fft_time; *avg = Avg_fft_time; }

/* ------------------------------------------------------------------------
 * Deterministic sig_gen stream.  The plugin itself (src/sig_gen.c) cannot be
 * built here (libsamplerate, radio.h), so the per-sample law of its CW branch
 * (src/sig_gen.c:291-296 real, :321-326 complex) is restated around the
 * reference's OWN oscillator (src/osc.c:28-70) and noise source
 * (src/gauss.c:95-111, xoshiro256** seeded with 1), both compiled unmodified.
 * ------------------------------------------------------------------------ */
#include "osc.h"
extern _Thread_local xoshiro256ss_state Rand_state;   /* src/gauss.c:21 */

struct refsig {
  struct osc carrier;
  xoshiro256ss_state rng;
  double amplitude, noise, scale;
  int isreal;
};

void *refsig_create(double cycles_per_sample, double amplitude, double noise, double scale,
                    int isreal, unsigned long long seed) {
  struct refsig *s = calloc(1, sizeof *s);
  if (!s) return NULL;
  set_osc(&s->carrier, cycles_per_sample, 0.0);        /* src/sig_gen.c:220-224 */
  xoshiro256ss_seed(&s->rng, seed);                     /* src/gauss.c:99 uses seed 1 */
  s->amplitude = amplitude; s->noise = noise; s->scale = scale; s->isreal = isreal;
  return s;
}
void refsig_delete(void *h) { free(h); }

/* n samples: n floats (real) or n (re,im) pairs (complex) */
void refsig_generate(void *h, float *out, long n) {
  struct refsig *s = h;
  Rand_state = s->rng;
  if (s->isreal) {
    for (long i = 0; i < n; i++) {
      double samp = s->amplitude * creal(step_osc(&s->carrier)) + s->noise * real_gauss();
      out[i] = samp * s->scale;
    }
  } else {
    float complex *o = (float complex *)out;
    for (long i = 0; i < n; i++) {
      double complex samp = s->amplitude * step_osc(&s->carrier) + s->noise * complex_gauss();
      o[i] = samp * s->scale;
    }
  }
  s->rng = Rand_state;
}

/* ---------------------------------------------------------------------------
 * downconvert() tail.  radio.c cannot be built here (iniparser, opus, libusb ...), so the dozen
 * statements of src/radio.c:1476-1520 that follow execute_filter_output() are restated around the
 * reference's OWN oscillator and cispi (src/osc.c:28-70, src/sincospi.c), both compiled unmodified.
 * ------------------------------------------------------------------------ */
struct refdc {
  struct osc fine;
  int bin_shift;
  double remainder;
  double complex phase_adjust;
};
void *refdc_create(void) {
  struct refdc *d = calloc(1, sizeof *d);
  if (!d) return NULL;
  d->remainder = NAN;        /* src/modes.c:265 */
  d->bin_shift = -1000999;   /* src/modes.c:266 */
  return d;
}
void refdc_delete(void *h) { free(h); }
double refdc_block(void *h, int shift, double remainder, double out_samprate, double doppler_rate,
                   int L, int M, float *buf, int olen) {
  struct refdc *d = h;
  float complex *x = (float complex *)buf;
  if (shift != d->bin_shift || isnan(d->remainder) || remainder != d->remainder) {
    set_osc(&d->fine, -remainder / out_samprate, doppler_rate / (out_samprate * out_samprate));
    d->remainder = remainder;
  }
  if (shift != d->bin_shift) {
    const int V = 1 + (L / (M - 1));
    d->phase_adjust = cispi(2.0 * (shift % V) / (double)V);
    d->fine.phasor *= cispi((shift - d->bin_shift) / (-2.0 * (V - 1)));
    d->bin_shift = shift;
  }
  d->fine.phasor *= d->phase_adjust;
  for (int n = 0; n < olen; n++) x[n] *= step_osc(&d->fine);
  double energy = 0;
  for (int n = 0; n < olen; n++) energy += cnrmf(x[n]);
  return olen ? energy / olen : 0;
}
