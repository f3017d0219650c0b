/* oracle/ref_fm_wrap.c -- TEST INFRASTRUCTURE.  Pins the restated FM demodulator (chz_oracle.c:chzo_fmdemod_block, SURVEY 8f
 * rank 4) to the REFERENCE'S OWN CODE: this translation unit is the reference's src/fm.c, included unmodified from where it
 * lies, and demod_fm() (src/fm.c:19-345) is RUN, block after block, on caller-supplied baseband; downconvert() and
 * send_output() are stubs that feed it and capture its frames (PCM packed by the reference's own import.h).  fm_snr() and the
 * IIR / Goertzel helpers are the reference's misc.c and iir.c.  Never copied into the repo; built by oracle/Makefile only
 * where /root/reference exists, into oracle/_ref/. */
#include "fm.c"
#include "import.h"

#define EXPORT __attribute__((visibility("default")))

double Blocktime;
int Verbose;
struct frontend Frontend;

static struct {
  int nblocks, cur, N;
  const float *baseband; const double *bb_power, *n0;
  float complex *work;
  unsigned char *pcm; int pcm_stride; int *frame; int *mute; double *out_power, *gain, *fmsnr, *foffset, *pdev, *tonedev;
} B;

int downconvert(chan_t *chan) {
  if (B.cur >= B.nblocks) return -1;
  memcpy(B.work, B.baseband + (size_t)2 * B.cur * B.N, sizeof(float complex) * (size_t)B.N);
  chan->baseband = B.work; chan->sampcount = B.N;
  chan->sig.bb_power = B.bb_power[B.cur]; chan->sig.n0 = B.n0[B.cur];
  return 0;
}
int send_output(chan_t *restrict const chan, float const *restrict buffer, int frames, bool const mute) {
  int const b = B.cur++;
  B.mute[b] = mute; B.out_power[b] = chan->output.power; B.gain[b] = chan->output.gain;
  B.fmsnr[b] = chan->fm.snr; B.foffset[b] = chan->sig.foffset; B.pdev[b] = chan->fm.pdeviation; B.tonedev[b] = chan->fm.tone_deviation;
  if (buffer == NULL) { B.frame[b] = 1; return 0; }
  B.frame[b] = 0;
  int const samples = frames * chan->output.channels;
  uint8_t *dp = B.pcm + (size_t)b * B.pcm_stride;
  switch (chan->output.encoding) {
  case MULAW: export_mulaw(dp, buffer, samples); break;      /* float_to_mulaw / _alaw: the reference's rtp.c */
  case ALAW: export_alaw(dp, buffer, samples); break;
  case S16BE: export_s16_be(dp, buffer, samples); break;
  case S16LE: export_s16_le(dp, buffer, samples); break;
  case F32BE: export_f32_be(dp, buffer, samples); break;
  default: export_f32_le(dp, buffer, samples); break;
  }
  return 0;
}
void response(chan_t *chan, bool response_needed) { (void)chan; (void)response_needed; }
bool decode_radio_commands(chan_t *chan, uint8_t const *buffer, int length) { (void)chan; (void)buffer; (void)length; return false; }
int create_filter_output(struct filter_out *out, struct filter_in *master, int olen, enum filtertype out_type) { (void)out; (void)master; (void)olen; (void)out_type; return 0; }
int set_channel_filter(chan_t *chan) { (void)chan; return 0; }
void realtime(int prio) { (void)prio; }
size_t strlcpy(char *dst, const char *src, size_t size) {          /* libbsd's, for the reference's misc.c */
  size_t n = strlen(src);
  if (size) { size_t k = n < size - 1 ? n : size - 1; memcpy(dst, src, k); dst[k] = 0; }
  return n;
}

struct dm_params { int channels, env, agc, encoding, snr_squelch, squelch_tail, tuned, kind;
  double samprate, headroom, threshold, recovery_rate, hangtime, dc_alpha, bandwidth, shift, squelch_open, squelch_close, gain;
  double deemph_rate, deemph_gain, threshold_extend; int pll_enable, pll_square; double pll_loop_bw, tone_freq; };

EXPORT int reffm_run(const struct dm_params *p, double blocktime, int nblocks, int N, const float *baseband, const double *bb_power,
                     const double *n0, unsigned char *pcm, int pcm_stride, int *frame, int *mute, double *out_power, double *gain,
                     double *fmsnr, double *foffset, double *pdev, double *tonedev) {
  static chan_t chan;
  static struct frontend fe;
  memset(&chan, 0, sizeof chan);
  Blocktime = blocktime;
  chan.frontend = &fe;
  chan.output.samprate = (int)p->samprate; chan.output.headroom = p->headroom;
  chan.output.encoding = p->encoding == 0 ? S16BE : p->encoding == 1 ? S16LE : p->encoding == 3 ? F32BE : p->encoding == 4 ? MULAW : p->encoding == 5 ? ALAW : F32LE;
  chan.filter.min_IF = -p->bandwidth / 2; chan.filter.max_IF = p->bandwidth / 2;
  chan.squelch.snr_enable = p->snr_squelch; chan.squelch.open = p->squelch_open; chan.squelch.close = p->squelch_close; chan.squelch.tail = p->squelch_tail;
  chan.fm.threshold = p->threshold_extend != 0; chan.fm.rate = p->deemph_rate; chan.fm.gain = p->deemph_gain; chan.fm.tone_freq = p->tone_freq;
  chan.pll.enable = p->pll_enable != 0;
  chan.demod_type = FM_DEMOD;
  pthread_mutex_init(&chan.status.lock, NULL);
  B.nblocks = nblocks; B.cur = 0; B.N = N; B.baseband = baseband; B.bb_power = bb_power; B.n0 = n0;
  B.work = malloc(sizeof(float complex) * (size_t)N);
  B.pcm = pcm; B.pcm_stride = pcm_stride; B.frame = frame; B.mute = mute; B.out_power = out_power; B.gain = gain;
  B.fmsnr = fmsnr; B.foffset = foffset; B.pdev = pdev; B.tonedev = tonedev;
  int r = demod_fm(&chan);
  free(B.work);
  return r;
}
