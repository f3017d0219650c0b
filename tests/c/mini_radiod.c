/* tests/c/mini_radiod.c -- TEST INFRASTRUCTURE.  The reference's OWN callers on a filter.h implementation of the linker's choice.
 *
 * north_star: "drops in behind ka9q-radio's existing filter.h API ... so radiod, linear.c and fm.c are untouched".  This program is the
 * evidence: it is linked from the reference's radio.c, linear.c, fm.c, spectrum.c, modes.c, osc.c, misc.c, iir.c, rtp.c, sched.c, sincospi.c,
 * window.c, compiled UNMODIFIED from where they lie (tests/c/Makefile; never copied), plus this one translation unit, TWICE:
 *     oracle/_ref/mini_radiod_ref        + the reference's own filter.c on the oracle's FFT provider  (the checker)
 *     tests/c/_prebuilt/mini_radiod_hip  + libka9q_filter_hip.so                                      (the product under test)
 * Everything between the A/D samples and the PCM frames is the reference's code: lookup_or_create_chan() / start_demod() /
 * demod_thread() (src/radio.c:862-998), one pthread per channel in demod_linear() (src/linear.c:21-375) or demod_fm()
 * (src/fm.c:19-345), each calling downconvert() (src/radio.c:1410-1524): compute_tuning, execute_filter_output (:1460),
 * estimate_noise on the master's host fdomain[] (:1469, :1783-1866), the fine-tuning rotator and phase_adjust (:1476-1501), filter2
 * through set_channel_filter() (:1503-1513, :1559-1612), set_freq() (:1099), response() (:1525), set_defaults() (src/modes.c:209).
 *
 * What this file supplies is what radiod's OTHER files would (none of them on the path SURVEY section 8 scopes):
 *   - the config loader (src/radio.c:166 loadconfig, src/modes.c:315 loadpreset need iniparser, absent here): channels come from a text
 *     file of key=value pairs with loadpreset()'s key names and unit conversions, the test holds the presets (share/presets.conf);
 *   - the front end driver (setup / start / shutdown entry points, src/radio.c:502-590): samples from a file, written IN PLACE at
 *     Frontend.in.input_write_pointer and handed over with write_rfilter(&Frontend.in,NULL,L) as src/rx888.c:800 / src/sig_gen.c:296 do;
 *   - send_output() (src/audio.c:41: RTP + Opus): captures the frame, packs PCM with the reference's own src/import.h, keeps the
 *     timestamp / silent bookkeeping of src/audio.c:46-55,62-74,185-186;
 *   - decode_radio_commands() / send_radio_status() (src/radio_status.c needs libbsd headers + TLV tables): a two-command decoder
 *     that does what src/radio_status.c:241 (RADIO_FREQUENCY) and :640-659 (new filter edges) do, with the reference's set_freq() /
 *     set_channel_filter(); the commands are queued on chan->commands[] under chan->status.lock exactly where radiod's status
 *     thread queues them, but by the channel's own thread at a chosen frame so that both links see them at the same block;
 *   - opus_encoder_destroy: referenced by demod_thread()'s clean-up, never reached (abort).
 * Round 6 also links the reference's spectrum.c: demod_spectrum() in narrowband mode (a COMPLEX slave of whatever block size its rbw / bin count
 * ask for, set_filter(), downconvert(), its own analysis transform through plan_complex()) and in wideband mode (a SPECTRUM slave as block clock,
 * the raw A/D ring read through input_write_pointer); a poll command every block (as `control` polls), the bin data captured where the status
 * packet would carry it.  And wfm.c: demod_wfm() takes a 384 kHz COMPLEX slave off the front end and owns a REAL inline master of its own (the FM
 * composite, 2:1 overlap) with three slaves -- mono (REAL), pilot and L-R (COMPLEX, spun down by 19 / 38 kHz through their shifts).
 * Determinism: the front end thread writes block b only when every channel has taken block b-2 (no drops by construction, in either
 * link), or -- "paced 1" -- on its own wall clock at Blocktime intervals without ever waiting, as hardware does.
 *
 *   mini_radiod <dir>      reads <dir>/cfg.txt, <dir>/in.f32; writes <dir>/frames.bin, <dir>/meta.txt
 */
#define _GNU_SOURCE 1
#include <assert.h>
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "misc.h"
#include "filter.h"
#include "radio.h"
#include "osc.h"
#include "import.h"

int Verbose = 0;                       /* src/main.c */
/* the oracle's FFT provider (the checker link's transform; in the product link only spectrum.c's analysis FFTs reach it, through the drop-in's
   plan_complex() forwarding to the process's fftwf_* as it would to libfftw3f) can do its arithmetic in float32 (oracle/fftw_shim.c) -- MINI_RADIOD_FFT_F32=1 runs the
   reference on a float32 transform, as it would on FFTW, to show how far two CORRECT transforms move the reference's own outputs */
extern void oracle_fft_set_precision(int) __attribute__((weak));
extern int Overlap;                    /* src/radio.c:128, not in radio.h */

/* ---- never reached (see the header) ---- */
void opus_encoder_destroy(OpusEncoder *e) { (void)e; fprintf(stderr, "mini_radiod: no Opus in this image\n"); abort(); }

/* ---- per-channel capture ---- */
struct event { int frame; int kind; double a, b; };      /* kind 'F': retune to a Hz; 'W': filter edges a..b Hz; 'M': the preset change sw[(int)a]; 'P': spectrum poll */
#define MAXEV 4
struct capture {
  uint32_t ssrc;
  int calls;                        /* send_output calls so far */
  unsigned char *buf; size_t len, cap;
  struct event ev[MAXEV]; int nev;
  char *sw[MAXEV]; int nsw;         /* "key~value,key~value,...": what a PRESET / OUTPUT_SAMPRATE / DEMOD_TYPE command changes (kind 'M') */
  atomic_int status_calls; int commands;
  int poll;                         /* a spectrum channel: a poll command every block */
  int start, life;                  /* joins when `start` blocks have been written (0: before the front end starts); lives for `life` blocks (0: to the end of the run) */
  int slot;                         /* its Channel_list[] entry */
  double freq; chan_t *tmpl;        /* what it is created from (late joiners: by the front-end thread) */
};
static struct capture Cap[2 * Nchannels];  /* in creation order */
static int FE_is_paced(void), FE_slack(void), FE_is_done(void);
static int SlotCap[Nchannels];         /* Channel_list[] slot -> Cap[] index of the channel that lives there now (a slot is reused after close_chan()) */
static int Nchan;                      /* channels this run has started so far */
#define CAP_OF(chan) (&Cap[SlotCap[(chan) - Channel_list]])

struct frame_hdr {                     /* one per send_output() call; tests/test_mini_radiod.py reads this layout */
  uint32_t ssrc, call, next_jobnum, block_drops;
  int32_t frames, channels, mute, isnull, encoding, bin_shift, pll_lock, squelch_silent;
  uint32_t rtp_timestamp, pcm_bytes, nfloat, olen;
  double n0, bb_power, out_power, gain, pll_snr, fm_snr, foffset, pdeviation, cphase, remainder, tone_deviation, tune_freq;
};

static void cap_append(struct capture *c, const void *p, size_t n) {
  if (c->len + n > c->cap) {
    c->cap = (c->len + n) * 2 + 4096;
    c->buf = realloc(c->buf, c->cap);
    if (!c->buf) { perror("realloc"); abort(); }
  }
  memcpy(c->buf + c->len, p, n); c->len += n;
}

/* src/audio.c:41 send_output(): the demodulators call it once per block (src/linear.c:340,344,366; src/fm.c:172,307,335) */
int send_output(chan_t *restrict const chan, float const *restrict buffer, int frames, bool const mute) {
  struct capture *c = CAP_OF(chan);
  if (chan == NULL || frames <= 0 || chan->output.channels == 0 || chan->output.samprate == 0) return 0;      /* src/audio.c:43-44 */
  struct frame_hdr h;
  memset(&h, 0, sizeof h);
  h.ssrc = chan->output.rtp.ssrc; h.call = (uint32_t)c->calls; h.next_jobnum = chan->filter.out.next_jobnum;
  h.block_drops = chan->filter.out.block_drops;
  h.frames = frames; h.channels = chan->output.channels; h.mute = mute; h.isnull = buffer == NULL; h.encoding = chan->output.encoding;
  h.bin_shift = chan->filter.bin_shift; h.pll_lock = chan->pll.lock; h.squelch_silent = chan->output.silent;
  h.rtp_timestamp = chan->output.rtp.timestamp; h.olen = (uint32_t)chan->filter.out.olen;
  h.n0 = chan->sig.n0; h.bb_power = chan->sig.bb_power; h.out_power = chan->output.power; h.gain = chan->output.gain;
  h.pll_snr = chan->pll.snr; h.fm_snr = chan->fm.snr; h.foffset = chan->sig.foffset; h.pdeviation = chan->fm.pdeviation;
  h.cphase = chan->pll.cphase; h.remainder = chan->filter.remainder; h.tone_deviation = chan->fm.tone_deviation; h.tune_freq = chan->tune.freq;
  uint8_t pcm[8 * 4096];
  int const samples = frames * chan->output.channels;
  if (!(mute || buffer == NULL)) {
    if ((size_t)samples * 4 > sizeof pcm) { fprintf(stderr, "mini_radiod: frame of %d samples\n", samples); abort(); }
    uint8_t *ndp = NULL;
    switch (chan->output.encoding) {                           /* src/audio.c:117-135 */
    case MULAW: ndp = export_mulaw(pcm, buffer, samples); break;
    case ALAW: ndp = export_alaw(pcm, buffer, samples); break;
    case S16BE: ndp = export_s16_be(pcm, buffer, samples); break;
    case S16LE: ndp = export_s16_le(pcm, buffer, samples); break;
    case F32BE: ndp = export_f32_be(pcm, buffer, samples); break;
    case F32LE: ndp = export_f32_le(pcm, buffer, samples); break;
    default: fprintf(stderr, "mini_radiod: encoding %d is outside this test\n", chan->output.encoding); abort();
    }
    h.pcm_bytes = (uint32_t)(ndp - pcm); h.nfloat = (uint32_t)samples;
  }
  cap_append(c, &h, sizeof h);
  if (h.nfloat) { cap_append(c, buffer, sizeof(float) * h.nfloat); cap_append(c, pcm, h.pcm_bytes); }
  /* the scheduled commands of this frame: queued where radiod's status thread queues them (src/radio_status.c:98-117) */
  for (int i = 0; i < c->nev; i++) {
    if (c->ev[i].frame != c->calls) continue;
    pthread_mutex_lock(&chan->status.lock);
    for (int q = 0; q < CQLEN; q++) {
      if (chan->commands[q].buffer != NULL) continue;
      struct event *e = malloc(sizeof *e);
      *e = c->ev[i];
      chan->commands[q].buffer = (uint8_t *)e; chan->commands[q].length = sizeof *e;
      break;
    }
    pthread_mutex_unlock(&chan->status.lock);
  }
  c->calls++;
  /* the bookkeeping the demodulators and response() can see (src/audio.c:46-55, 62-74, 185-186, 200) */
  chan->output.rtp.timestamp += (uint32_t)frames;
  if (mute || buffer == NULL) { chan->output.silent = true; return 0; }
  chan->output.silent = false;
  chan->output.samples += (uint64_t)frames;
  return frames;
}

static int apply_kv(chan_t *chan, struct capture *c, char const *k, char const *v);
/* src/radio_status.c:133 decode_radio_commands(): the commands this test sends */
bool decode_radio_commands(chan_t *chan, uint8_t const *buffer, int length) {
  if (length != (int)sizeof(struct event)) return false;
  struct event e; memcpy(&e, buffer, sizeof e);
  struct capture *c = CAP_OF(chan);
  c->commands++;
  chan->status.packets_in++;
  /* NOT mirrored: "chan->lifetime = chan->lifestart" (src/radio_status.c:141) -- the lifetime counter is this test's block budget */
  if (e.kind == 'P') {
    /* a spectrum poll (what `control` sends every update): the reply carries the bin data demod_spectrum() computes after this block's
       downconvert() -- and the next poll is already waiting: queued on the OTHER slot (the caller frees this one when we return), so
       that exactly one poll is answered per block, in both links alike */
    for (int q = 0; q < CQLEN; q++) {
      if (chan->commands[q].buffer == (uint8_t *)buffer || chan->commands[q].buffer != NULL) continue;
      struct event *n = malloc(sizeof *n);
      *n = e;
      chan->commands[q].buffer = (uint8_t *)n; chan->commands[q].length = sizeof *n;
      break;
    }
  } else if (e.kind == 'M') {
    /* a PRESET (src/radio_status.c:168-181: loadpreset()) / OUTPUT_SAMPRATE (:215-230) / DEMOD_TYPE (:310-319) command, and what the decoder does about it
       afterwards (:613-660): a new sample rate or demodulator asks for a RESTART -- the demodulator returns, demod_thread() deletes the channel's
       filter output and starts the (new) demodulator, which creates one of the new size (src/radio.c:940-985) --, new edges for new filters */
    int const old_rate = chan->output.samprate, old_demod = chan->demod_type, old_blocking = chan->filter2.blocking, old_channels = chan->output.channels;
    bool const old_isb = chan->filter2.out.isb;
    double const old_lo = chan->filter.min_IF, old_hi = chan->filter.max_IF, old_beta = chan->filter.kaiser_beta;
    char *list = strdup(c->sw[(int)e.a]);
    char *save = NULL;                     /* (strtok_r: channel threads decode their commands side by side) */
    for (char *tok = strtok_r(list, ",", &save); tok != NULL; tok = strtok_r(NULL, ",", &save)) {
      char *sep = strchr(tok, '~');
      if (sep == NULL) continue;
      *sep = 0;
      if (apply_kv(chan, c, tok, sep + 1) != 0) { fprintf(stderr, "mini_radiod: switch %s~%s\n", tok, sep + 1); abort(); }
    }
    free(list);
    if (chan->filter2.out.isb && !old_isb) {               /* ISB being turned on (:636-646): stereo output, filter2 forced on */
      if (old_channels != 2) chan->output.channels = 2;
      if (chan->filter2.blocking == 0) chan->filter2.blocking = 1;
    }
    if (chan->output.samprate != old_rate || (int)chan->demod_type != old_demod) {
      /* RESTART.  The new filter output will start at the master's job counter of that moment (src/filter.c:413), i.e. it skips what the front end has written
         meanwhile.  For two links to skip the SAME blocks, the lock-step front end is let run as far ahead as it may (it stops `slack` blocks ahead of the slowest
         channel, which this one now is) before the demodulator returns, and it does not move while a running channel has no filter output (all_channels_took).
         The skipped blocks come off this test's block budget (the "lifetime"), so that the channel still ends with the input. */
      unsigned const mine = chan->filter.out.next_jobnum;
      while (!FE_is_paced() && (int)(*(volatile unsigned *)&Frontend.in.next_jobnum - (mine + FE_slack())) < 0 && !FE_is_done()) usleep(50);
      int const skipped = (int)(*(volatile unsigned *)&Frontend.in.next_jobnum - mine);
      if (skipped > 0 && chan->lifetime > skipped) chan->lifetime -= skipped;
      return true;
    }
    if (chan->filter.min_IF != old_lo || chan->filter.max_IF != old_hi || chan->filter.kaiser_beta != old_beta || chan->filter2.blocking != old_blocking) {
      set_channel_filter(chan);
      set_freq(chan, chan->tune.freq);
      chan->filter.remainder = NAN;
    }
    int const pt = pt_from_info(chan->output.samprate, chan->output.channels, chan->output.encoding);      /* :663-677 */
    if (pt != -1) chan->output.rtp.type = pt;
  } else if (e.kind == 'F') {
    set_freq(chan, e.a);                                   /* src/radio_status.c:241 */
  } else if (e.kind == 'W') {
    chan->filter.min_IF = e.a; chan->filter.max_IF = e.b;  /* src/radio_status.c:262-281 */
    set_channel_filter(chan);                              /* src/radio_status.c:653-659 */
    set_freq(chan, chan->tune.freq);
    chan->filter.remainder = NAN;
  }
  return false;
}
int send_radio_status(struct sockaddr const *sock, struct frontend const *frontend, chan_t *chan) {
  (void)sock; (void)frontend;
  struct capture *c = CAP_OF(chan);
  if ((chan->demod_type == SPECT_DEMOD || chan->demod_type == SPECT2_DEMOD) && chan->spectrum.bin_data != NULL && chan->spectrum.bin_count > 0) {
    /* src/radio_status.c:880-905 packs chan->spectrum.bin_data into the status packet: captured as a frame of bin_count floats */
    struct frame_hdr h;
    memset(&h, 0, sizeof h);
    h.ssrc = chan->output.rtp.ssrc; h.call = (uint32_t)c->calls++; h.next_jobnum = chan->filter.out.next_jobnum; h.block_drops = chan->filter.out.block_drops;
    h.frames = chan->spectrum.bin_count; h.channels = 1; h.bin_shift = chan->filter.bin_shift; h.olen = (uint32_t)chan->filter.out.olen;
    h.nfloat = (uint32_t)chan->spectrum.bin_count; h.encoding = 0;
    h.n0 = chan->sig.n0; h.bb_power = chan->sig.bb_power; h.gain = (double)chan->spectrum.fft_n; h.out_power = (double)chan->output.samprate;
    h.remainder = chan->filter.remainder; h.tune_freq = chan->tune.freq;
    cap_append(c, &h, sizeof h);
    cap_append(c, chan->spectrum.bin_data, sizeof(float) * h.nfloat);
  }
  atomic_fetch_add(&c->status_calls, 1);
  return 0;
}

/* ---- the front end "driver" ---- */
static struct {
  int L, M, nblocks, paced, slack, isreal;
  double samprate;
  float *samples;                      /* [nblocks][L] real, or [nblocks][L] complex (interleaved) for a complex front end */
  pthread_t thread;
  atomic_int go, done;
  double worst_wait_ms;
} FE;

static int FE_is_paced(void) { return FE.paced; }
static int FE_slack(void) { return FE.slack; }
static int FE_is_done(void) { return atomic_load(&FE.done); }
static int Ncfg;                       /* channels of the configuration file */
static int create_channel(int ci);
static bool all_channels_took(uint32_t job) {        /* every running channel has next_jobnum >= job */
  for (int i = 0; i < Nchannels; i++) {
    chan_t *ch = &Channel_list[i];
    if (ch->state != CHANNEL_RUNNING) continue;
    if (ch->filter.out.master == NULL) return false;              /* between delete_filter_output and create_filter_output of a restart (decode_radio_commands 'M') */
    if (ch->filter.out.master != &Frontend.in) continue;
    if ((int32_t)(*(volatile unsigned int *)&ch->filter.out.next_jobnum - job) < 0) return false;
  }
  return true;
}

static void *fe_thread(void *arg) {
  struct frontend *fe = arg;
  pthread_setname("mini-fe");
  while (!atomic_load(&FE.go)) usleep(200);
  struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int b = 0; b < FE.nblocks; b++) {
    if (FE.paced) {                                                  /* hardware pace: absolute deadlines, never waits for anybody */
      struct timespec t = t0;
      int64_t ns = (int64_t)llrint((b + 1) * Blocktime * 1e9) + t0.tv_nsec;
      t.tv_sec += ns / 1000000000; t.tv_nsec = ns % 1000000000;
      clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &t, NULL);
    } else if (b >= FE.slack) {
      while (!all_channels_took((uint32_t)(b - FE.slack + 1))) usleep(100);
    }
    /* dynamic channels: created while the stream runs, as radiod creates one when a client asks for an unknown ssrc (src/radio_status.c:60-96);
       the test wants both links to attach the newcomer at the SAME block, so the front end waits until it is in its loop */
    for (int i = 0; i < Ncfg; i++)
      if (Cap[i].start == b && b > 0) {
        if (create_channel(i) != 0) abort();
        for (int spins = 0; atomic_load(&Cap[i].status_calls) == 0; spins++) { if (spins > 300000) { fprintf(stderr, "mini_radiod: late channel %u never came up\n", Cap[i].ssrc); abort(); } usleep(100); }
      }
    int r;
    if (FE.isreal) {
      float *wptr = fe->in.input_write_pointer.r;                    /* src/rx888.c:730-800: convert in place, then hand over */
      memcpy(wptr, FE.samples + (size_t)b * FE.L, sizeof(float) * (size_t)FE.L);
      r = write_rfilter(&fe->in, NULL, FE.L);
    } else {
      float complex *wptr = fe->in.input_write_pointer.c;            /* src/sig_gen.c:300-330, src/airspy.c: complex front ends */
      memcpy(wptr, FE.samples + (size_t)2 * b * FE.L, sizeof(float complex) * (size_t)FE.L);
      r = write_cfilter(&fe->in, NULL, FE.L);
    }
    fe->samples += FE.L;
    if (r != 1) { fprintf(stderr, "mini_radiod: write_rfilter returned %d at block %d\n", r, b); abort(); }
  }
  atomic_store(&FE.done, 1);
  return NULL;
}
static int fe_start(struct frontend *fe) {     /* (again whenever the channel count comes up from zero: the sample thread is started once) */
  static atomic_int started;
  if (atomic_exchange(&started, 1)) return 0;
  return pthread_create(&FE.thread, NULL, fe_thread, fe);
}   /* called by lookup_or_create_chan() for the first channel, src/radio.c:898-904 */
static atomic_int Shutdowns;
static int fe_shutdown(struct frontend *fe) { (void)fe; atomic_fetch_add(&Shutdowns, 1); return 0; }  /* called by close_chan() of the last channel, src/radio.c:1088-1091 */

/* what src/radio.c:502-623 setup_hardware() does after the driver's setup() has filled in the rates */
static void setup_hardware_like_radiod(void) {
  Frontend.start = fe_start; Frontend.shutdown = fe_shutdown;
  Frontend.samprate = FE.samprate; Frontend.isreal = FE.isreal != 0; Frontend.bitspersample = 16; Frontend.frequency = 0; Frontend.calibrate = 0;
  if (FE.isreal) { Frontend.min_IF = 0; Frontend.max_IF = 0.47 * FE.samprate; }                          /* src/rx888.c:343-345 */
  else { Frontend.min_IF = -0.47 * FE.samprate; Frontend.max_IF = 0.47 * FE.samprate; }                  /* src/sig_gen.c:170-172 */
  Frontend.rf_gain = 0; Frontend.rf_atten = 0; Frontend.rf_level_cal = 0;
  strlcpy(Frontend.description, "mini-radiod", sizeof Frontend.description);
  Frontend.L = FE.L; Frontend.M = FE.M;
  Blocktime = Frontend.L / Frontend.samprate;                       /* src/radio.c:584 */
  int const N = Frontend.M + Frontend.L - 1;
  if (create_filter_input(&Frontend.in, Frontend.L, Frontend.M, FE.isreal ? REAL : COMPLEX) != 0) { fprintf(stderr, "mini_radiod: create_filter_input failed\n"); exit(3); }
  Frontend.in.notches = calloc(NSPURS + 1, sizeof(struct notch_state));            /* src/radio.c:600-621 */
  struct notch_state *notch = Frontend.in.notches;
  for (int i = 0; i < NSPURS; i++) {
    int shift; double rem;
    if (compute_tuning(N, Frontend.M, Frontend.samprate, &shift, &rem, Frontend.spurs[i]) != 0) break;
    notch->state = 0; notch->bin = abs(shift); notch->alpha = .01;
    if (shift == 0) break;
    notch++;
  }
}

/* ---- the stand-in for loadpreset() (src/modes.c:315-570): same keys, same unit conversions ---- */
static bool truth(char const *v) { return v[0] == 'y' || v[0] == 'Y' || v[0] == 't' || v[0] == 'T' || v[0] == '1'; }
static int apply_kv(chan_t *chan, struct capture *c, char const *k, char const *v) {
  double const x = strtod(v, NULL);
  if (!strcmp(k, "demod")) { int t = demod_type_from_name(v); if (t < 0) return -1; chan->demod_type = t; }           /* :320-326 */
  else if (!strcmp(k, "samprate")) chan->output.samprate = round_samprate((unsigned)labs(lrint(x)));                  /* :331-338 */
  else if (!strcmp(k, "channels")) chan->output.channels = (int)x;
  else if (!strcmp(k, "mono")) { if (truth(v)) chan->output.channels = 1; }
  else if (!strcmp(k, "stereo")) { if (truth(v)) chan->output.channels = 2; }
  else if (!strcmp(k, "encoding")) { int e = parse_encoding(v); if (e == NO_ENCODING) return -1; chan->output.encoding = e; }
  else if (!strcmp(k, "kaiser-beta")) chan->filter.kaiser_beta = x;
  else if (!strcmp(k, "low")) chan->filter.min_IF = x;
  else if (!strcmp(k, "high")) chan->filter.max_IF = x;
  else if (!strcmp(k, "squelch-open")) chan->squelch.open = dB2power(x);
  else if (!strcmp(k, "squelch-close")) chan->squelch.close = dB2power(x);
  else if (!strcmp(k, "squelch-tail")) chan->squelch.tail = (int)x;
  else if (!strcmp(k, "headroom")) chan->output.headroom = dB2voltage(-fabs(x));
  else if (!strcmp(k, "shift")) chan->tune.shift = x;
  else if (!strcmp(k, "recovery-rate")) chan->linear.recovery_rate = dB2voltage(fabs(x));
  else if (!strcmp(k, "hang-time")) chan->linear.hangtime = fabs(x);
  else if (!strcmp(k, "threshold")) chan->linear.threshold = dB2voltage(-fabs(x));
  else if (!strcmp(k, "gain")) chan->output.gain = dB2voltage(x);
  else if (!strcmp(k, "envelope")) chan->linear.env = truth(v);
  else if (!strcmp(k, "pll")) chan->pll.enable = truth(v);
  else if (!strcmp(k, "square")) { chan->pll.square = truth(v); if (chan->pll.square) chan->pll.enable = true; }
  else if (!strcmp(k, "conj")) chan->filter2.out.isb = truth(v);
  else if (!strcmp(k, "pll-bw")) chan->pll.loop_bw = x;
  else if (!strcmp(k, "agc")) chan->linear.agc = truth(v);
  else if (!strcmp(k, "threshold-extend")) chan->fm.threshold = truth(v);
  else if (!strcmp(k, "snr-squelch")) chan->squelch.snr_enable = truth(v);
  else if (!strcmp(k, "dc-cut")) chan->linear.dc_alpha = -expm1(-2.0 * M_PI * x / chan->output.samprate);
  else if (!strcmp(k, "deemph-tc")) { double tc = fabs(x * 1e-6); chan->fm.rate = tc == 0 ? 0 : -expm1(-1.0 / (tc * chan->output.samprate)); }
  else if (!strcmp(k, "deemph-gain")) chan->fm.gain = dB2voltage(x);
  else if (!strcmp(k, "tone")) chan->fm.tone_freq = fabs(x);
  else if (!strcmp(k, "update")) chan->status.output_interval = abs((int)x);
  else if (!strcmp(k, "filter2")) chan->filter2.blocking = abs((int)x);
  else if (!strcmp(k, "beam")) chan->filter.beam = truth(v);                                                           /* :547-556: two antennas on I and Q */
  else if (!strcmp(k, "a-amp")) chan->filter.a_weight = x * (cabs(chan->filter.a_weight) > 0 ? chan->filter.a_weight / cabs(chan->filter.a_weight) : 1.0);
  else if (!strcmp(k, "a-phase")) chan->filter.a_weight = (cabs(chan->filter.a_weight) > 0 ? cabs(chan->filter.a_weight) : 1.0) * csincospi(x / 180.);
  else if (!strcmp(k, "b-amp")) chan->filter.b_weight = x * (cabs(chan->filter.b_weight) > 0 ? chan->filter.b_weight / cabs(chan->filter.b_weight) : 1.0);
  else if (!strcmp(k, "b-phase")) chan->filter.b_weight = (cabs(chan->filter.b_weight) > 0 ? cabs(chan->filter.b_weight) : 1.0) * csincospi(x / 180.);
  else if (!strcmp(k, "rbw")) chan->spectrum.rbw = x;                          /* RESOLUTION_BW / BIN_COUNT / SPECTRUM_AVG of a `control` command (src/radio_status.c:420-470) */
  else if (!strcmp(k, "bins")) chan->spectrum.bin_count = (int)x;
  else if (!strcmp(k, "fft-avg")) chan->spectrum.fft_avg = (int)x;
  else if (!strcmp(k, "poll")) c->poll = truth(v);
  else if (!strcmp(k, "start")) c->start = (int)x;         /* this test's own: the channel is created when `start` blocks have been written ... */
  else if (!strcmp(k, "life")) c->life = (int)x;           /* ... and its "lifetime" (src/modes.c:329-330) runs out after `life` blocks */
  /* this test's own keys: a command for the channel's own command queue at a given frame */
  else if (!strcmp(k, "retune") || !strcmp(k, "edges")) {
    if (c->nev >= MAXEV) return -1;
    struct event *e = &c->ev[c->nev++];
    e->kind = k[0] == 'r' ? 'F' : 'W';
    if (sscanf(v, "%d:%lf:%lf", &e->frame, &e->a, &e->b) < 2) return -1;
  } else if (!strcmp(k, "switch")) {                         /* switch=frame:key~value,key~value,... */
    if (c->nev >= MAXEV || c->nsw >= MAXEV) return -1;
    char const *colon = strchr(v, ':');
    if (colon == NULL) return -1;
    struct event *e = &c->ev[c->nev++];
    e->kind = 'M'; e->frame = atoi(v); e->a = c->nsw; e->b = 0;
    c->sw[c->nsw++] = strdup(colon + 1);
  } else return -1;
  return 0;
}

/* src/radio.c:807-834: what process_section() does per frequency (and what radio_status.c does for a dynamic channel: src/radio_status.c:60-96) */
static int create_channel(int ci) {
  struct capture *c = &Cap[ci];
  chan_t *tmpl = c->tmpl;
  if (c->life > 0) tmpl->lifestart = tmpl->lifetime = c->life + 1;
  else if (c->start > 0) tmpl->lifestart = tmpl->lifetime = FE.nblocks - c->start + 1;
  chan_t *chan = lookup_or_create_chan(c->ssrc, tmpl);
  if (chan == NULL || chan->state != CHANNEL_STARTING) { fprintf(stderr, "mini_radiod: ssrc %u\n", c->ssrc); return -1; }
  c->slot = (int)(chan - Channel_list);
  SlotCap[c->slot] = ci;
  snprintf(chan->name, sizeof chan->name, "%s %u", demod_name_from_type(chan->demod_type), chan->output.rtp.ssrc);
  set_freq(chan, c->freq);
  if (c->poll) {                                         /* the first poll is waiting when the thread starts; each one queues the next */
    struct event *e = calloc(1, sizeof *e);
    e->kind = 'P';
    chan->commands[0].buffer = (uint8_t *)e; chan->commands[0].length = sizeof *e;
  }
  pthread_mutex_unlock(&chan->status.lock);
  pthread_mutex_lock(&Channel_list_mutex);
  chan->state = CHANNEL_RUNNING;
  pthread_mutex_unlock(&Channel_list_mutex);
  Nchan++;
  start_demod(chan);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: mini_radiod <dir>\n"); return 2; }
  char path[1024];
  snprintf(path, sizeof path, "%s/cfg.txt", argv[1]);
  FILE *f = fopen(path, "r");
  if (!f) { perror(path); return 2; }
  int nchan = 0;
  if (fscanf(f, "%lf %d %d %d %d %d %d", &FE.samprate, &FE.L, &FE.M, &FE.nblocks, &nchan, &FE.paced, &FE.slack) != 7) { fprintf(stderr, "bad cfg\n"); return 2; }
  FE.isreal = 1;
  { int c = fgetc(f); while (c == ' ') c = fgetc(f); if (c == 'c') FE.isreal = 0; else if (c != EOF) ungetc(c, f); }      /* an 8th token "complex": a complex front end */
  if (nchan > 2 * Nchannels || FE.slack < 1 || FE.slack > 3) { fprintf(stderr, "bad cfg\n"); return 2; }
  size_t const per = FE.isreal ? 1 : 2;
  FE.samples = malloc(sizeof(float) * per * (size_t)FE.L * FE.nblocks);
  snprintf(path, sizeof path, "%s/in.f32", argv[1]);
  FILE *g = fopen(path, "rb");
  if (!g || fread(FE.samples, sizeof(float) * per * FE.L, FE.nblocks, g) != (size_t)FE.nblocks) { perror(path); return 2; }
  fclose(g);

  if (getenv("MINI_RADIOD_VERBOSE")) Verbose = atoi(getenv("MINI_RADIOD_VERBOSE"));
  if (getenv("MINI_RADIOD_FFT_F32") && oracle_fft_set_precision) oracle_fft_set_precision(1);
  /* src/osc.c:96-98: nco() fills its sine table on first use, and only the FIRST caller waits for that -- channel threads that start their
     PLLs at the same moment read a half-filled table (a start-up transient in radiod; here it made one run in eight of the checker link differ
     from the others on a coherent channel).  One call from the main thread before any channel exists makes every run the same. */
  { double s_, c_; nco(0, &s_, &c_); }
  setup_hardware_like_radiod();
  Overlap = 1 + FE.L / (FE.M - 1);                       /* "overlap" of [global], src/radio.c:283; set_channel_filter reads it */
  set_defaults(&Template);                               /* src/radio.c:439 */
  Template.lifestart = Template.lifetime = FE.nblocks + 1;   /* "lifetime" of a section (src/modes.c:329-330): the run's block budget */
  Template.status.output_interval = 5;
  Template.status.global_timer = 1;                      /* a "delayed status request" (src/radio.c:1543-1546): the first response() of the channel's loop answers it */

  /* one line per channel: ssrc freq key=value ... -- src/radio.c:807-834 process_section() per frequency */
  char line[4096];
  if (!fgets(line, sizeof line, f)) return 2;            /* rest of line 1 */
  for (int i = 0; i < nchan; i++) {
    if (!fgets(line, sizeof line, f)) { fprintf(stderr, "cfg: %d channel lines expected\n", nchan); return 2; }
    char *save = NULL;
    char *tok = strtok_r(line, " \t\n", &save);
    uint32_t ssrc = (uint32_t)strtoul(tok, NULL, 10);
    tok = strtok_r(NULL, " \t\n", &save);
    double const freq = strtod(tok, NULL);
    chan_t tmpl = Template;
    struct capture cfg; memset(&cfg, 0, sizeof cfg);
    while ((tok = strtok_r(NULL, " \t\n", &save)) != NULL) {
      char *eq = strchr(tok, '=');
      if (!eq) { fprintf(stderr, "cfg: %s\n", tok); return 2; }
      *eq = 0;
      if (apply_kv(&tmpl, &cfg, tok, eq + 1) != 0) { fprintf(stderr, "cfg: key %s value %s\n", tok, eq + 1); return 2; }
    }
    if (tmpl.filter.min_IF > tmpl.filter.max_IF) { double t = tmpl.filter.min_IF; tmpl.filter.min_IF = tmpl.filter.max_IF; tmpl.filter.max_IF = t; }   /* src/modes.c:372-377 */
    tmpl.output.rtp.type = pt_from_info(tmpl.output.samprate, tmpl.output.channels, tmpl.output.encoding);                                                /* :355-362 */
    int const ci = Ncfg++;
    Cap[ci] = cfg; Cap[ci].ssrc = ssrc; Cap[ci].freq = freq;
    Cap[ci].tmpl = malloc(sizeof tmpl); *Cap[ci].tmpl = tmpl;
    if (Cap[ci].start == 0 && create_channel(ci) != 0) return 3;
  }
  fclose(f);

  /* the front end starts when every channel thread is in its loop -- it has attached its slave to the master (create_filter_output in
   * demod_linear / demod_fm), set its filters, and answered the delayed status request above at the top of the loop.  radiod has no such
   * barrier (channels that join later start at the master's current block); the test wants every channel to see block 0 */
  for (int spins = 0;; spins++) {
    int ready = 0;
    int want = 0;
    for (int i = 0; i < Ncfg; i++) if (Cap[i].start == 0) { want++; ready += atomic_load(&Cap[i].status_calls) > 0; }
    if (ready == want) break;
    if (spins > 600000) { fprintf(stderr, "mini_radiod: %d of %d channels in their loops after 60 s\n", ready, want); return 4; }
    usleep(100);
  }
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  atomic_store(&FE.go, 1);
  /* channels run down their lifetime (downconvert() returns -1, src/radio.c:1424-1431), demod_thread() cleans up, close_chan() marks the entry idle */
  for (int spins = 0;; spins++) {
    int idle = 0;                    /* (a slot may have been taken over by a later channel: idle = created and its slot idle or no longer its own) */
    for (int i = 0; i < Ncfg; i++) idle += atomic_load(&Cap[i].status_calls) > 0 && (SlotCap[Cap[i].slot] != i || Channel_list[Cap[i].slot].state == CHANNEL_IDLE);
    if (idle == Ncfg && atomic_load(&FE.done)) break;
    if (spins > 6000000) { fprintf(stderr, "mini_radiod: %d of %d channels finished after 10 min\n", idle, Ncfg); return 5; }
    usleep(100);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  pthread_mutex_lock(&Channel_list_mutex);        /* close_chan() marks the entry idle under this mutex (src/radio.c:1084-1092): taking it once orders everything the */
  pthread_mutex_unlock(&Channel_list_mutex);      /* channel threads did before this thread's reads of their captures (and lets ThreadSanitizer see that) */
  pthread_join(FE.thread, NULL);
  unsigned const master_jobs = Frontend.in.next_jobnum;
  struct notch_state *const notch_list = Frontend.in.notches;        /* the caller's (src/radio.c:600; radiod never frees it): delete_filter_input() zeroes the struct */
  delete_filter_input(&Frontend.in);
  free(notch_list);

  snprintf(path, sizeof path, "%s/frames.bin", argv[1]);
  g = fopen(path, "wb");
  long total_calls = 0, commands = 0, status = 0;
  for (int i = 0; i < Ncfg; i++) {
    if (getenv("MINI_RADIOD_DEBUG")) fprintf(stderr, "chan %d ssrc %u slot %d start %d life %d calls %d status %d len %zu\n", i, Cap[i].ssrc, Cap[i].slot, Cap[i].start, Cap[i].life, Cap[i].calls, atomic_load(&Cap[i].status_calls), Cap[i].len);
    fwrite(Cap[i].buf, 1, Cap[i].len, g);
    total_calls += Cap[i].calls; commands += Cap[i].commands; status += atomic_load(&Cap[i].status_calls);
  }
  fclose(g);
  snprintf(path, sizeof path, "%s/meta.txt", argv[1]);
  g = fopen(path, "w");
  fprintf(g, "channels %d master_jobs %u frames %ld commands %ld status_calls %ld shutdowns %d seconds %.3f hdr_bytes %zu blocktime %.9f overlap %d\n",
          Nchan, master_jobs, total_calls, commands, status, atomic_load(&Shutdowns),
          (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), sizeof(struct frame_hdr), Blocktime, Overlap);
  fclose(g);
  return 0;
}
