/* oracle/ref_linear_wrap.c -- TEST INFRASTRUCTURE.  Pins the restated linear demodulator (chz_oracle.c:chzo_lindemod_block,
 * SURVEY 8f rank 4) to the REFERENCE'S OWN CODE: this translation unit is the reference's src/linear.c, included unmodified
 * from where it lies, and demod_linear() (src/linear.c:21-375) is RUN, block after block, on caller-supplied baseband:
 * the functions it calls out to are replaced by stubs that feed it (downconvert) and capture what it emits (send_output,
 * whose PCM packing is the reference's own export_* from src/import.h).  The oscillator is the reference's osc.c.
 * Never copied into the repo; built by oracle/Makefile only where /root/reference exists, into oracle/_ref/. */
#include "linear.c"
#include "import.h"

#define EXPORT __attribute__((visibility("default")))

double Blocktime;
int Verbose;
struct frontend Frontend;

/* ---- the test bench: blocks fed in, frames captured ---- */
static struct {
  int nblocks, cur, N;
  const float *baseband;        /* [nblocks][N] complex */
  const double *bb_power, *n0;  /* [nblocks]: chan->sig.bb_power and the SMOOTHED chan->sig.n0 downconvert() leaves */
  float complex *work;          /* demod_linear overwrites its input: a scratch copy per block */
  unsigned char *pcm; int pcm_stride; int *frame; int *mute; double *out_power; double *gain;
  double *pll;                  /* [nblocks][5]: chan->pll.snr, .lock, .cphase, .rotations, chan->sig.foffset */
} B;

int downconvert(chan_t *chan) {
  if (B.cur >= B.nblocks) return -1;                    /* terminate the demodulator loop */
  memcpy(B.work, B.baseband + (size_t)2 * B.cur * B.N, sizeof(float complex) * (size_t)B.N);
  chan->baseband = B.work; chan->sampcount = B.N;
  chan->sig.bb_power = B.bb_power[B.cur]; chan->sig.n0 = B.n0[B.cur];
  return 0;
}
int send_output(chan_t *restrict const chan, float const *restrict buffer, int frames, bool const mute) {
  int const b = B.cur++;
  B.mute[b] = mute; B.out_power[b] = chan->output.power; B.gain[b] = chan->output.gain;
  if (B.pll) { double *q = B.pll + 5 * b; q[0] = chan->pll.snr; q[1] = chan->pll.lock; q[2] = chan->pll.cphase; q[3] = chan->pll.rotations; q[4] = chan->sig.foffset; }
  if (buffer == NULL) { B.frame[b] = 1; return 0; }
  B.frame[b] = 0;
  int const samples = frames * chan->output.channels;
  uint8_t *dp = B.pcm + (size_t)b * B.pcm_stride;
  switch (chan->output.encoding) {                      /* src/audio.c:117-133 */
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

/* params: the chzo_lindemod_params layout (oracle/chz_oracle.h); encoding 0 S16BE, 1 S16LE, 2 F32LE, 3 F32BE */
struct lin_params { int channels, env, agc, encoding, snr_squelch, squelch_tail, tuned, pad;
  double samprate, headroom, threshold, recovery_rate, hangtime, dc_alpha, bandwidth, shift, squelch_open, squelch_close, gain;
  double deemph_rate, deemph_gain, threshold_extend; int pll_enable, pll_square; double pll_loop_bw, tone_freq; };

EXPORT int reflin_run(const struct lin_params *p, double blocktime, int nblocks, int N, const float *baseband, const double *bb_power,
                      const double *n0, unsigned char *pcm, int pcm_stride, int *frame, int *mute, double *out_power, double *gain, double *pll) {
  static chan_t chan;
  static struct frontend fe;
  memset(&chan, 0, sizeof chan);
  Blocktime = blocktime;
  chan.frontend = &fe;
  chan.output.samprate = (int)p->samprate; chan.output.channels = p->channels; chan.output.gain = p->gain; chan.output.headroom = p->headroom;
  chan.output.encoding = p->encoding == 0 ? S16BE : p->encoding == 1 ? S16LE : p->encoding == 3 ? F32BE : p->encoding == 4 ? MULAW : p->encoding == 5 ? ALAW : F32LE;
  chan.linear.env = p->env; chan.linear.agc = p->agc; chan.linear.threshold = p->threshold; chan.linear.recovery_rate = p->recovery_rate;
  chan.linear.hangtime = p->hangtime; chan.linear.dc_alpha = p->dc_alpha;
  chan.filter.min_IF = -p->bandwidth / 2; chan.filter.max_IF = p->bandwidth / 2;
  chan.tune.shift = p->shift; chan.tune.freq = p->tuned ? 7.0e6 : 0;
  chan.squelch.snr_enable = p->snr_squelch; chan.squelch.open = p->squelch_open; chan.squelch.close = p->squelch_close; chan.squelch.tail = p->squelch_tail;
  chan.pll.enable = p->pll_enable != 0; chan.pll.square = p->pll_square != 0; chan.pll.loop_bw = p->pll_loop_bw;
  chan.demod_type = LINEAR_DEMOD;
  pthread_mutex_init(&chan.status.lock, NULL);
  B.nblocks = nblocks; B.cur = 0; B.N = N; B.baseband = baseband; B.bb_power = bb_power; B.n0 = n0;
  B.work = malloc(sizeof(float complex) * (size_t)N);
  B.pcm = pcm; B.pcm_stride = pcm_stride; B.frame = frame; B.mute = mute; B.out_power = out_power; B.gain = gain; B.pll = pll;
  int r = demod_linear(&chan);
  free(B.work);
  return r;
}
