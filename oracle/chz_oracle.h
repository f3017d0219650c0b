/* oracle/chz_oracle.h -- TEST INFRASTRUCTURE (CPU oracle), not a product path.
 *
 * Plain-C restatement of ka9q-radio's overlap-save channelizer (the path behind
 * src/filter.h).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the shipped HIP path never does.
 *
 * Pinning: the reference has NO golden vectors or tests for this path
 * (SURVEY.md section 4), so this restatement is pinned against the reference
 * ITSELF: oracle/_ref/libka9q_ref.so is the reference's own filter.c, window.c,
 * misc.c, osc.c, gauss.c compiled unmodified (oracle/Makefile), and
 * tests/test_oracle_vs_reference.py checks every function below against it on
 * seeded inputs; tests/golden/ holds vectors generated from it.  The FFT
 * butterflies themselves come from FFTW3 in the reference (third-party, absent
 * from /root/reference and from this image); they are restated from the DFT
 * definition in oracle/dft.c and cross-checked against numpy's pocketfft.
 *
 * Type codes follow the reference's enum filtertype (src/filter.h:29-34):
 *   1 COMPLEX, 2 REAL, 3 SPECTRUM.
 */
#ifndef CHZ_ORACLE_H
#define CHZ_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { CHZO_COMPLEX = 1, CHZO_REAL = 2, CHZO_SPECTRUM = 3 };

/* ---- filter design (host side of the path; src/filter.c:968-1045) ---- */
double chzo_i0(double z);                                   /* src/misc.c:416-427 */
int chzo_make_kaiser(float *w, int M, double beta);         /* src/window.c:217-237 */
int chzo_normalize_window(float *w, int M);                 /* src/window.c:240-254 */
/* response[2*P] <- frequency response exactly as set_filter builds it.
   master_points = N of the master, master_real != 0 adds the +3 dB. */
int chzo_set_filter(int P, int olen, int master_points, int master_real, int out_type,
                    double low, double high, double beta, float *response);

/* ---- forward transform of one N-sample window (src/filter.c:505-508,573-582) ---- */
int chzo_forward(const float *window, int N, int in_type, float *spectrum);        /* float32 result */
int chzo_forward_f64(const float *window, int N, int in_type, double *spectrum);   /* unrounded */

/* ---- spur notches (src/filter.c:464-474); state = 2 doubles per notch; the
        list must end with bin 0 ---- */
void chzo_notch(double *state, const int *bins, int n, double alpha, float *spectrum);

/* ---- per-channel: gather x response (src/filter.c:728-911) ---- */
int chzo_gather(const float *spectrum, int m_bins, int in_type,
                int s_bins, int out_type, int shift, int isb,
                const float *response, float *fdomain);
/* gather + backward transform + keep the last olen samples (src/filter.c:357,914).
   out: 2*olen floats (COMPLEX) or olen floats (REAL). */
int chzo_channel(const float *spectrum, int m_bins, int in_type,
                 int P, int olen, int out_type, int shift, int isb,
                 const float *response, float *out);
/* the same carried in float64 from an unrounded spectrum: the "exact" answer
   used to measure both the GPU's and the reference's float32 error */
int chzo_channel_f64(const double *spectrum, int m_bins, int in_type,
                     int P, int olen, int out_type, int shift, int isb,
                     const float *response, double *out);

/* ---- deterministic sig_gen stream (src/sig_gen.c:291-296,321-326; src/osc.c:28-70;
        src/gauss.c:32-61,95-111) ---- */
typedef struct chzo_siggen chzo_siggen;
chzo_siggen *chzo_siggen_create(double cycles_per_sample, double amplitude, double noise,
                                double scale, int isreal, uint64_t seed);
void chzo_siggen_delete(chzo_siggen *s);
void chzo_siggen_generate(chzo_siggen *s, float *out, long n);
double chzo_scale_ad(double rf_gain_db, double rf_atten_db, double level_cal_db,
                     int isreal, int bitspersample);        /* src/radio.c:1630-1650 */
/* shift/remainder split of a tuning frequency (src/radio.c:1175-1199) */
int chzo_compute_tuning(int N, double samprate, double freq, int *shift, double *remainder);

/* slave->beam variant of the COMPLEX->COMPLEX gather (src/filter.c:756-775); alpha/beta as set_filter_weights leaves them */
int chzo_gather_beam(const float *spectrum, int m_bins, int s_bins, int shift, const float *response,
                     double ar, double ai, double br, double bi, float *fdomain);
int chzo_channel_beam(const float *spectrum, int m_bins, int P, int olen, int shift, const float *response,
                      double ar, double ai, double br, double bi, float *out);

/* estimate_noise() of src/radio.c:1783-1866: noise density (per Hz) around a channel from the master spectrum */
double chzo_estimate_noise(const float *spectrum, int m_bins, int in_type, int s_bins, int shift, double samprate);

/* rx888.c convert(): int16 A/D samples -> float32 * scale, energy += sum x^2, returns the clip count */
int chzo_convert_i16(const int16_t *samples, int n, float scale, int randomize, float *out, uint64_t *energy);

/* tail of downconvert() (src/radio.c:1476-1520): fine-tuning rotator stepped per output sample, per-block
   phase correction for bin shifts not divisible by the overlap factor, phase continuity across shift
   changes, and the baseband power average.  One object per channel. */
typedef struct chzo_downconv chzo_downconv;
chzo_downconv *chzo_downconv_create(void);
void chzo_downconv_delete(chzo_downconv *d);
double chzo_downconv_block(chzo_downconv *d, int shift, double remainder, double out_samprate,
                           double doppler_rate, int L, int M, float *buf, int olen);

/* ---- whole overlap-save stream driver (src/filter.c:186-269,558-651,1093-1134):
        keeps the M-1 sample history, first block is preceded by M-1 zeros ---- */
typedef struct chzo_stream chzo_stream;
chzo_stream *chzo_stream_create(int L, int M, int in_type);
void chzo_stream_delete(chzo_stream *s);
int chzo_stream_bins(const chzo_stream *s);
int chzo_stream_points(const chzo_stream *s);
/* push exactly L new samples; fills spectrum (2*bins floats) of the new block */
int chzo_stream_push(chzo_stream *s, const float *samples, float *spectrum);
int chzo_stream_push_f64(chzo_stream *s, const float *samples, double *spectrum);

/* ---- the linear demodulator's per-block work (src/linear.c:56-375, PLL of the coherent modes included) and PCM
   packing (src/import.h:88-118).  Field names follow the chan_t members linear.c reads. */
enum { CHZO_PCM_S16BE = 0, CHZO_PCM_S16LE = 1, CHZO_PCM_F32LE = 2, CHZO_PCM_F32BE = 3, CHZO_PCM_MULAW = 4, CHZO_PCM_ALAW = 5, CHZO_PCM_F16LE = 6, CHZO_PCM_F16BE = 7 };
/* float -> IEEE binary16 bits, round to nearest even (`float16_t temp_float = in[i]`, src/import.h:140-157) */
unsigned short chzo_f32_to_f16(float x);
/* send_output()'s packing of `count` float samples in the given encoding (src/audio.c:117-139, src/import.h) */
void chzo_pcm_pack(int encoding, const float *in, int count, unsigned char *out);
/* G.711 companding as send_output() applies it (float_to_mulaw / float_to_alaw, src/rtp.c:459-483,500-533) */
unsigned char chzo_float_to_mulaw(float x);
unsigned char chzo_float_to_alaw(float x);
enum { CHZO_FRAME_DATA = 0, CHZO_FRAME_SILENCE = 1 };
enum { CHZO_DEMOD_LINEAR = 0, CHZO_DEMOD_FM = 1 };
typedef struct chzo_lindemod_params {
  int channels, env, agc, encoding, snr_squelch, squelch_tail, tuned, kind;
  double samprate, headroom, threshold, recovery_rate, hangtime, dc_alpha, bandwidth, shift, squelch_open, squelch_close, gain;
  double deemph_rate, deemph_gain, threshold_extend;     /* FM only: chan->fm.rate, chan->fm.gain, chan->fm.threshold (0/1) */
  int pll_enable, pll_square;                            /* chan->pll.enable, chan->pll.square (linear: src/linear.c:83-153; FM: src/fm.c:176-203) */
  double pll_loop_bw;                                    /* chan->pll.loop_bw, Hz (linear only; FM uses 500 Hz, src/fm.c:181) */
  double tone_freq;                                      /* FM only: chan->fm.tone_freq, Hz; 0 = no PL/CTCSS tone squelch (src/fm.c:264-311) */
} chzo_lindemod_params;
typedef struct chzo_lindemod_status {
  int frame, mute, squelch_state, pll_lock;              /* pll_lock: chan->pll.lock (linear) */
  double output_power, gain, n0, snr;
  double foffset, pdeviation;                            /* chan->sig.foffset (FM; linear with PLL: src/linear.c:115), chan->fm.pdeviation */
  double pll_snr, pll_cphase, tone_deviation;            /* chan->pll.snr, chan->pll.cphase (linear), chan->fm.tone_deviation */
  int pll_rotations, tone_mute;                          /* chan->pll.rotations; FM: the tone squelch's mute decision */
} chzo_lindemod_status;
/* the PLL's oscillator (nco(), src/osc.c:91-126): sine table of 1024 entries per quadrant + second-order interpolation */
void chzo_nco(unsigned accum, double *s, double *c);
typedef struct chzo_lindemod chzo_lindemod;
chzo_lindemod *chzo_lindemod_create(const chzo_lindemod_params *p);
void chzo_lindemod_delete(chzo_lindemod *d);
void chzo_lindemod_set_params(chzo_lindemod *d, const chzo_lindemod_params *p);      /* everything but the running gain */
int chzo_lindemod_block(chzo_lindemod *d, float *buf, int N, double bb_power, double n0_est, double blocktime,
                        unsigned char *pcm, chzo_lindemod_status *st);
int chzo_pcm_bytes(int encoding, int nsamples);
/* the FM demodulator's per-block work (demod_fm, src/fm.c:19-345): both SNR estimators, the squelch sequencer, the
   discriminator with threshold extension or the PLL demodulator (:176-203), offset / deviation statistics, PM carrier removal,
   the PL-tone squelch (:264-311), de-emphasis, gain, PCM packing.  Same parameter / status records as the linear demodulator (kind = CHZO_DEMOD_FM). */
typedef struct chzo_fmdemod chzo_fmdemod;
chzo_fmdemod *chzo_fmdemod_create(const chzo_lindemod_params *p);
void chzo_fmdemod_delete(chzo_fmdemod *d);
int chzo_fmdemod_block(chzo_fmdemod *d, const float *buf, int N, double bb_power, double n0_est, double blocktime,
                       unsigned char *pcm, chzo_lindemod_status *st);

#ifdef __cplusplus
}
#endif
#endif
