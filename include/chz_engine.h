/* chz_engine.h -- C ABI of the MI355X overlap-save channelizer engine
 * (libchz_hip.so, built from ka9q-radio_amd/csrc/chz_engine.hip).
 *
 * Plain pointers and sizes only; no torch, no C++ types.  This is the layer a
 * foreign host (C, ctypes, cgo ...) binds.  Each entry point states which piece
 * of the reference's filter.c it replaces; the struct-compatible drop-in for the
 * reference's own filter.h (create_filter_input & co.) is layered on top of it in
 * ka9q-radio_amd/csrc/filter_hip.c and declared in include/ka9q_filter_abi.h.
 *
 * All functions return 0 on success and a negative value on error (the message
 * is available from chz_last_error()); there is NO CPU fallback: without a HIP
 * device chz_engine_create() fails.
 *
 * Sample/stream conventions are the reference's: blocks of L new samples, impulse
 * length M, N = L+M-1 point transform, window k = stream[kL-(M-1), kL+L) with
 * zeros before time 0 (src/filter.c:196,244,259); unnormalised transforms both ways
 * with all gain folded into the per-channel response (src/filter.c:1020-1028);
 * ND = 4 spectrum slots indexed job % 4 (src/filter.h:48).
 */
#ifndef CHZ_ENGINE_H
#define CHZ_ENGINE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CHZ_COMPLEX 1   /* enum filtertype COMPLEX, src/filter.h:31 */
#define CHZ_REAL    2   /* enum filtertype REAL,    src/filter.h:32 */
#define CHZ_ND      4   /* spectrum slots,          src/filter.h:48 */

typedef struct chz_engine chz_engine;

typedef struct chz_info {
  int L, M, N, in_type, bins;
  int ring_blocks;          /* device input ring holds ring_blocks*L samples */
  int Na, Nb, Nc;           /* axis lengths of the forward transform */
  int n_banks;
  int lanes;                /* HIP streams blocks are pipelined over */
  /* device storage order of a spectrum slot: bin k at [(k / spec_na) * spec_pitch + spec_off + k % spec_na];
   * natural order when spec_pitch == spec_na and spec_off == 0.  chz_spectrum_read returns natural
   * order; code that touches the device buffer directly (chz_spectrum_device / _attach, e.g. an RCCL
   * broadcast) must treat it as spec_elems opaque complex values. */
  long spec_elems;
  int spec_na, spec_pitch, spec_off;
  char plan[320];           /* human-readable plan description */
} chz_info;

typedef struct chz_timing {
  double total_ms;          /* HIP-event time of the whole run on the engine's stream */
  int blocks;
  /* per-kernel HIP-event time, accumulated over the run (only when instrumented) */
  double first_ms, cols_ms, rows_ms, notch_ms, chan_ms;   /* notch_ms/notch_n: the noise-estimate kernel (the spur notch is fused into fwd_rows) */
  int first_n, cols_n, rows_n, notch_n, chan_n;
  double enqueue_ms;        /* host wall time spent issuing the launches (close to total_ms = host-bound) */
  double fix_ms; int fix_n; /* the spur-notch kernel (notch_fix) */
  double demod_ms; int demod_n; /* the linear demodulator kernel */
} chz_timing;

const char *chz_last_error(void);
/* 1 once the process has begun to exit (exit() called: radiod's closedown(), src/main.c): the library has stopped issuing work to the HIP runtime, whose own exit
 * handlers are about to tear it down, and every call reports -98 from then on -- not a device failure, nothing to recover (round 6) */
int chz_process_exiting(void);
/* Options (round 6): dispatch thresholds and test hooks, process-wide, read when an engine / communicator is CREATED.  The shipped
 * library reads only the operator's environment variables; everything a test or an A/B script used to set
 * through CHZ_* variables is set here instead.  value NULL or "" = back to the default.  Names: chan_stage, noise_energy, demod_wave
 * (-1 auto / 0 / 1), enq_threads (1|2|4), graph_blocks, notch_fold (0|1), noise_hint (0|1), pll_lane0 (0|1), notch_wait_ms,
 * fault_ticket_skew + allow_fault_injection (recovery-path tests), launch_id (chz_comm_create_file).  < 0: unknown name. */
int chz_set_option(const char *name, const char *value);
int chz_device_count(void);

/* replaces create_filter_input (src/filter.c:186-269): allocates the device input
 * ring (zeroed, write position M-1 ahead of the read position), the ND spectrum
 * slots and the transform plan ("" / NULL = automatic, or "NaxNbxNc[:T1,T2,Ta]"). */
int chz_engine_create(chz_engine **out, int L, int M, int in_type, int device,
                      const char *plan_spec, int ring_blocks);
/* replaces delete_filter_input (src/filter.c:930-942) */
void chz_engine_destroy(chz_engine *e);
int chz_engine_info(const chz_engine *e, chz_info *info);
/* run all engine work on a caller-owned hipStream_t (e.g. torch's current stream) */
int chz_engine_set_stream(chz_engine *e, void *hip_stream);
int chz_sync(chz_engine *e);
/* non-blocking health check: < 0 (and chz_last_error) once a device-side consistency check has failed -- today that is
 * the spur-notch ticket (see chz_set_notches_alpha); chz_sync and chz_run_blocks report the same condition */
int chz_engine_check(const chz_engine *e);

/* replaces the sample hand-over of write_rfilter/write_cfilter
 * (src/filter.c:1093-1134): append n samples (floats; re,im pairs for COMPLEX) from
 * host memory at the ring's write position. */
int chz_input_write(chz_engine *e, const float *host_samples, long n);
/* same, but the samples already live in device memory */
int chz_input_write_device(chz_engine *e, const float *dev_samples, long n);
/* SURVEY 8(f) rank 3 -- raw A/D samples: replaces rx888.c's convert() + write_rfilter(.., NULL, n)
 * (src/rx888.c:753-767,800-829).  The int16 samples go to HBM as they are (half the PCIe bytes, half the
 * first pass's reads); x -> (float)x * scale, the LTC2208 de-randomiser (randomize != 0: if bit 0 is set,
 * flip bits 1..15, src/rx888.c:711-716) and the per-block statistics happen where the first transform
 * pass loads the samples.  REAL masters only; an engine takes either float or int16 input, not both. */
int chz_input_write_i16(chz_engine *e, const short *host_samples, long n, float scale, int randomize);
int chz_input_write_i16_device(chz_engine *e, const short *dev_samples, long n, float scale, int randomize);
/* sum of x*x and count of |x| > 32766 over the L new samples of the block last transformed into `slot`
 * (what rx_callback accumulates into frontend->if_power / overranges, src/rx888.c:780-795); synchronous */
int chz_input_stats(chz_engine *e, int slot, unsigned long long *energy, unsigned *clips);
/* direct access to the device ring for HBM-resident benchmarks */
int chz_input_ring(chz_engine *e, float **dev_ring, long *ring_len_floats);
/* Re-seat the input ring in front of block `job`: the next L samples written are that block's new samples and `history`
 * (M-1 samples from host memory; NULL = zeros, the state create_filter_input leaves, src/filter.c:244,259) is what came
 * before them.  A freshly created engine sits in front of job 0; a host that REPLACES an engine in mid-stream (the drop-in's
 * recovery from a failed device-side check: the reference exits and lets systemd restart it, src/radio.c:398, src/main.c:202)
 * continues its job numbering and its overlap history with this.  Drains the engine.  Float input only. */
int chz_input_seek(chz_engine *e, unsigned job, const float *history);
/* Fences for a producer that does not wait for blocks (the drop-in's KA9Q_HIP_INPUT_FULL=drop): chz_input_mark(e, k) records
 * marker k (0..7) behind everything chz_input_write has enqueued so far, chz_input_mark_wait(e, k) blocks until the copies in
 * front of marker k have read their host source (so that the host ring region may be overwritten); a marker never recorded
 * is complete. */
int chz_input_mark(chz_engine *e, int k);
int chz_input_mark_wait(chz_engine *e, int k);
/* by_event != 0: order the spur-notch recurrence by HIP events between the streams instead of the device ticket (what env
 * CHZ_NOTCH_ORDER=event selects at creation): no device-side wait that could run out.  Drains the engine. */
int chz_engine_notch_order(chz_engine *e, int by_event);

/* replaces the body of execute_filter_input / run_fft (src/filter.c:485-651):
 * forward transform of the window of job `job` (window start = job*L in ring
 * coordinates) into slot job % 4, then the spur notches. */
int chz_forward(chz_engine *e, unsigned job);
/* notch list as radio.c builds it (src/radio.c:601-620): last entry is bin 0;
 * replaces apply_notch_filters' state (src/filter.c:464-474).  n = 0 clears. */
int chz_set_notches(chz_engine *e, const int *bins, int n, double alpha);
/* the same with one averager gain per entry, as struct notch_state carries it (src/filter.h:42-46, src/filter.c:468);
 * an entry with alpha = 0 is the reference's no-op.  The recurrence over blocks is ordered by a ticket the tiny notch
 * kernels take in block order (bounded wait, loud failure, never a wrong state: chz_engine_check), or -- env
 * CHZ_NOTCH_ORDER=event, and always inside captured graphs -- by HIP events between the engine's streams. */
int chz_set_notches_alpha(chz_engine *e, const int *bins, const double *alpha, int n);
int chz_spectrum_read(chz_engine *e, int slot, float *host);     /* 2*bins floats, synchronous */
int chz_spectrum_device(chz_engine *e, int slot, float **dev);
/* the hipStream_t everything addressed by `slot` is enqueued on; lets a caller order foreign work
 * (an RCCL collective on the slot's spectrum) between chz_forward(job) and chz_bank_execute(.., slot) */
int chz_slot_stream(chz_engine *e, int slot, void **hip_stream);
/* point a slot at caller-owned device memory (2*spec_elems floats, see chz_info), e.g. a torch
 * tensor that RCCL broadcasts into */
int chz_spectrum_attach(chz_engine *e, int slot, float *dev);

/* A bank = all channels sharing one (P, olen); replaces create_filter_output's
 * allocations (src/filter.c:298-415) for COMPLEX output channels. */
int chz_bank_create(chz_engine *e, int P, int olen, int capacity);          /* returns bank id */
/* the same for REAL-output slaves (create_filter_output(.., REAL), src/filter.c:372-395; gather :794-809, c2r :914):
 * responses are still P complex values per channel (set_filter's array, of which bins 0..P/2 are used), outputs are
 * olen floats per channel; P must be even */
int chz_bank_create_real(chz_engine *e, int P, int olen, int capacity);
/* Channels with the same filter sharing its response: `nrows` response rows in the bank, every channel names the row it reads
 * (chz_bank_set_rows: takes effect like a retune, in stream order, no drain), rows are written by chz_bank_set_row_responses (between
 * blocks: drains).  The reference keeps one copy per slave (src/filter.c:1039-1043); the values are the same, the HBM traffic per
 * channel and block drops from 8P + 8*olen to 8*olen bytes.  chz_bank_set_responses is refused on such a bank. */
int chz_bank_create_shared(chz_engine *e, int P, int olen, int capacity, int nrows);
int chz_bank_set_rows(chz_engine *e, int bank, int ch0, int n, const int *rows);
int chz_bank_set_row_responses(chz_engine *e, int bank, int row0, int n, const float *resp);
/* None of the per-channel setters below waits for blocks in flight: shifts, tuning, ISB flags and beam weights live in
 * small descriptors kept once per spectrum slot and refreshed in stream order when the next block of that slot is
 * enqueued (blocks already enqueued keep what they were launched with); a response is written to a spare row and the
 * channel re-pointed, the old row being recycled once everything enqueued before the swap has drained.  Only edits of
 * more than 8192 channels at once (a bank being set up) and the first use of a feature that switches the kernel
 * variant (tuning, ISB, beam, noise) drain the engine. */
/* response[P] complex as set_filter leaves it (src/filter.c:968-1045) */
int chz_bank_set_responses(chz_engine *e, int bank, int ch0, int n, const float *resp);
/* `shift` of execute_filter_output(slave, shift) (src/filter.c:663).  A channel has NO gather descriptor until this (or
 * chz_bank_set_tuning) has been called for it -- shift 0 included -- and produces zeros until then. */
int chz_bank_set_shifts(chz_engine *e, int bank, int ch0, int n, const int *shifts);
/* slave->isb (src/filter.c:895-909, filter2 of the linear demodulator in ISB mode): LSB and USB are unpacked to I and Q
 * after the gather.  One flag byte per channel, non-zero = on.  COMPLEX-output banks only. */
int chz_bank_set_isb(chz_engine *e, int bank, int ch0, int n, const unsigned char *flags);
/* slave->beam (src/filter.c:756-775; set by radio.c:938-940): two antennas on I and Q of a COMPLEX master are selected or
 * combined with the weights set_filter_weights left in slave->alpha / ->beta.  ab = 4 doubles per channel (Re alpha,
 * Im alpha, Re beta, Im beta), on = one flag byte per channel.  Bins past the end of the master walk are zero (the
 * reference leaves them unwritten). */
int chz_bank_set_beam(chz_engine *e, int bank, int ch0, int n, const double *ab, const unsigned char *on);
int chz_bank_set_active(chz_engine *e, int bank, int n);                    /* channels [0,n) run */
/* replaces execute_filter_output's gather x response + backward transform
 * (src/filter.c:728-914) for every active channel of the bank at once */
/* the bank runs on the spectrum of block `job` (slot job % 4); for a bank without tuning only the slot
 * matters, so passing a slot number 0..3 is equivalent */
int chz_bank_execute(chz_engine *e, int bank, unsigned job);
/* the same for channels [ch0, ch0+n) only: the slow path of one retuned channel.  ALWAYS a partial re-run of a block the bank has
 * already seen -- whatever the range, including a bank of exactly one channel: the bank's demodulators are not stepped and the
 * block's PCM is left alone (only chz_bank_execute / chz_step / chz_run_blocks are "the block") */
int chz_bank_execute_range(chz_engine *e, int bank, unsigned job, int ch0, int n);

/* SURVEY 8(f) rank 1 -- the tail of radiod's downconvert() fused into the channel kernel's epilogue
 * (src/radio.c:1476-1520): every output sample is multiplied by the channel's fine-tuning rotator
 * (set_osc/step_osc, src/osc.c:28-70) including the per-block phase_adjust for bin shifts that are not a
 * multiple of the overlap factor V = 1 + L/(M-1) and the one-time phase kick on a shift change, and the
 * block's mean |sample|^2 (chan->sig.bb_power) is left in a per-slot array.
 *   shifts[i]  as for chz_bank_set_shifts (this call replaces it for tuned channels)
 *   freq[i]    cycles per OUTPUT sample, = -remainder / output samprate   (first argument of set_osc)
 *   rate[i]    cycles per sample^2,       = doppler_rate / samprate^2     (second argument; NULL = 0)
 * The update takes effect at block `job`, which must not have been enqueued yet; call it when shift,
 * remainder or sweep rate change (it does not wait for the engine), not every block.  The rotation is a closed
 * form in the block number, so tuned banks run eagerly (chz_run_blocks mode 0), any number in flight. */
int chz_bank_set_tuning(chz_engine *e, int bank, unsigned job, int ch0, int n, const int *shifts,
                        const double *freq, const double *rate);
int chz_bank_read_power(chz_engine *e, int bank, int slot, int ch0, int n, double *host);        /* synchronous */
int chz_bank_read_power_async(chz_engine *e, int bank, int slot, int ch0, int n, double *host);

/* SURVEY 8(f) rank 2 -- estimate_noise() (src/radio.c:1783-1866) on the device, run for every channel of
 * the bank right after its channel kernel: energies of max(P, 1000) master bins around |shift|, their 0.10
 * quantile, the mean of the energies below 1.5 x that, bias correction, per Hz.  It is the one reader of
 * the whole block spectrum outside filter.c (src/radio.c:1787-1836): with it on the device the 13 MB
 * spectrum no longer has to travel to the host every block.  samprate = front-end sample rate (Hz), 0 = off.
 * The exponential smoothing of chan->sig.n0 (src/radio.c:1466-1473) stays with the caller. */
int chz_bank_enable_noise(chz_engine *e, int bank, double samprate);
int chz_bank_read_noise(chz_engine *e, int bank, int slot, int ch0, int n, double *host);        /* synchronous */
int chz_bank_read_noise_async(chz_engine *e, int bank, int slot, int ch0, int n, double *host);
int chz_bank_destroy(chz_engine *e, int bank);                              /* frees the bank's device arrays */
int chz_bank_read(chz_engine *e, int bank, int ch0, int n, float *host);    /* n*olen complex (REAL banks: n*olen floats) of the most recent execute, synchronous */
/* Blocks are pipelined over 1, 2 or 4 HIP streams ("lanes", env CHZ_STREAMS, default 4): block j
 * runs on lane j % lanes with its own intermediate buffer, so block j+1's forward transform overlaps
 * block j's tail.  Bank outputs exist once per spectrum slot.  Everything addressed by `slot` below is
 * enqueued on the lane that owns that slot, in call order.
 * Asynchronous device-to-host copies (host memory should come from chz_host_alloc); completion is
 * observed through chz_host_callback or chz_sync */
int chz_bank_read_async(chz_engine *e, int bank, int slot, int ch0, int n, float *host);
int chz_spectrum_read_async(chz_engine *e, int slot, float *host);
/* run fn(arg) on a runtime thread once everything enqueued so far on `slot`'s lane has finished: replaces
 * run_fft's completion broadcast (src/filter.c:522-539) */
int chz_host_callback(chz_engine *e, int slot, void (*fn)(void *), void *arg);
/* wait for everything enqueued on `slot`'s lane only (the other lanes keep running) */
int chz_slot_sync(chz_engine *e, int slot);
/* page-locked host memory for the buffers the device copies into */
int chz_host_alloc(void **p, size_t bytes);
void chz_host_free(void *p);
/* page-lock memory the caller already owns (the mirrored host input ring) so chz_input_write DMAs
 * straight out of it; failure is not fatal, copies then go through the runtime's staging buffer */
int chz_host_register(void *p, size_t bytes);
void chz_host_unregister(void *p);
int chz_bank_output_device(chz_engine *e, int bank, int slot, float **dev);

/* one whole block: chz_forward(job) then every bank on slot job % 4 */
int chz_step(chz_engine *e, unsigned job);

/* Run `nblocks` consecutive blocks starting at `job0` and time them with HIP
 * events on the engine's stream.
 *   mode 0: eager launches          mode 1: one hipGraph per ring cycle, replayed
 *   instrument != 0 (eager only): HIP events around every kernel -> per-kernel ms */
int chz_run_blocks(chz_engine *e, unsigned job0, int nblocks, int mode, int instrument,
                   chz_timing *timing);

/* ---- multi-GPU (north_star: "the forward spectrum is RCCL-broadcast over xGMI and each GPU owns a disjoint
 * channel subset, overlapped with the next block's forward FFT on a second HIP stream").  One process per GPU and one
 * communicator per process; the reference has no counterpart (thread per channel in one process).  RCCL is bound at
 * run time.  Rank 0 obtains an id and ships its CHZ_COMM_ID_BYTES bytes to the peers by any means (or all ranks meet
 * through a file with chz_comm_create_file). */
#define CHZ_COMM_ID_BYTES 128
typedef struct chz_comm chz_comm;
int chz_comm_unique_id(void *id128);
int chz_comm_create(chz_comm **out, int rank, int world, const void *id128, int device);
int chz_comm_create_file(chz_comm **out, int rank, int world, const char *path, int device, double timeout_s);
/* ONE process, several devices (the filter.h drop-in's KA9Q_HIP_DEVICES with KA9Q_HIP_EXCHANGE=broadcast): a clique of n communicators
 * (ncclCommInitAll), out[i] on devices[i]; every device at most once.  chz_spectrum_broadcast_local sends slot `slot` of engines[root]
 * into the same slot of the other engines -- engines[i] created on devices[i], comms from one chz_comm_create_local call -- as one
 * grouped call, each part on its engine's slot stream.  Each communicator is released with chz_comm_destroy. */
int chz_comm_create_local(chz_comm **out, int n, const int *devices);
int chz_spectrum_broadcast_local(chz_engine *const *engines, chz_comm *const *comms, int n, int slot, int root);
void chz_comm_destroy(chz_comm *c);
int chz_comm_rank(const chz_comm *c);
int chz_comm_world(const chz_comm *c);
int chz_comm_barrier(chz_comm *c);
int chz_comm_allreduce_max(chz_comm *c, double *v, int n);           /* n <= 64, blocking: control plane only */
/* ncclBroadcast of spectrum slot `slot` (all spec_elems complex values) from `root`, enqueued on the slot's stream:
 * behind chz_forward(job) on the root, in front of chz_bank_execute(.., job) everywhere */
int chz_spectrum_broadcast(chz_engine *e, chz_comm *c, int slot, int root);
/* sub-band variant: rank r receives only spectrum rows [row_lo[r], row_hi[r]) (row = spec_pitch complex values),
 * one grouped ncclSend/ncclRecv */
int chz_spectrum_exchange_rows(chz_engine *e, chz_comm *c, int slot, int root, const int *row_lo, const int *row_hi);
/* BASELINE config 4 from a C host: per block the root transforms, the spectrum travels (mode 0 broadcast, 1 rows),
 * every rank runs its own banks; timed like chz_run_blocks.  mode 2 (SURVEY 8e's alternative): the block's L new samples
 * travel from the root's input ring into every rank's ring (ncclBroadcast on the communicator's own stream: 10.4 MB instead of
 * the 13.8 MB slot at 129.6 MS/s) and every rank runs the forward transform itself -- nobody waits for the root's transform; with the
 * run's first block the M-1 samples of overlap history in front of it travel too, so every rank's spectra (and notch states) equal the root's */
int chz_run_blocks_sharded(chz_engine *e, chz_comm *c, int root, int mode, const int *row_lo, const int *row_hi,
                           unsigned job0, int nblocks, chz_timing *timing);

/* ---- SURVEY 8(f) rank 4 -- the linear and FM demodulators behind the fine-tuned channel outputs (demod_linear,
 * src/linear.c:56-375; demod_fm, src/fm.c:19-345): the PLL of the coherent modes with its lock detector, noise smoothing
 * (src/radio.c:1466-1473), post-detection shift oscillator, block AGC, the final demodulation pass with its per-sample gain
 * ramp, the squelch sequencers; FM's SNR estimators, discriminator or PLL demodulator, PL-tone squelch, de-emphasis; and PCM
 * packing (src/import.h:88-118 and G.711, src/rtp.c:459-533, via send_output, src/audio.c:117-133).  Runs for every channel of a COMPLEX-output bank that has tuning (chz_bank_set_tuning:
 * bb_power) and the noise estimate (chz_bank_enable_noise) switched on, in block order on a stream of its own behind the
 * bank's channel kernel.  What leaves the device per channel and block is packed PCM plus a status record instead of the
 * complex baseband -- for a 12 kHz mono S16 channel 480 B instead of 1920 B.  RTP framing and the sockets stay with the host.
 * The fields are the chan_t members src/linear.c reads (linear amplitudes / power ratios, not dB). */
#define CHZ_PCM_S16BE 0
#define CHZ_PCM_S16LE 1
#define CHZ_PCM_F32LE 2
#define CHZ_PCM_F32BE 3
#define CHZ_PCM_MULAW 4   /* G.711 mu-law, one byte per sample (float_to_mulaw, src/rtp.c:459-483) */
#define CHZ_PCM_ALAW 5    /* G.711 A-law (float_to_alaw, src/rtp.c:500-533) */
#define CHZ_PCM_F16LE 6   /* IEEE binary16, round to nearest even (export_f16_le / _be, src/import.h:140-157,207-212; src/audio.c:135-139) */
#define CHZ_PCM_F16BE 7
typedef struct chz_demod_params {
  int channels;         /* chan->output.channels: 1 mono, 2 stereo; 0 switches the channel's demodulator off */
  int env;              /* chan->linear.env: envelope (AM) detection */
  int agc;              /* chan->linear.agc */
  int encoding;         /* CHZ_PCM_* (chan->output.encoding) */
  int snr_squelch;      /* chan->squelch.snr_enable */
  int squelch_tail;     /* chan->squelch.tail */
  int tuned;            /* chan->tune.freq != 0 */
  int kind;             /* CHZ_DEMOD_LINEAR (demod_linear, src/linear.c) or CHZ_DEMOD_FM (demod_fm, src/fm.c:19-345: both SNR
                           estimators, squelch sequencer, discriminator with threshold extension or PLL demodulator, offset /
                           deviation statistics, PM carrier removal, PL-tone squelch, de-emphasis, gain) */
  double samprate;      /* chan->output.samprate */
  double headroom;      /* chan->output.headroom */
  double threshold, recovery_rate, hangtime, dc_alpha;   /* chan->linear.* */
  double bandwidth;     /* |chan->filter.min_IF - chan->filter.max_IF| */
  double shift;         /* chan->tune.shift, Hz */
  double squelch_open, squelch_close;                     /* chan->squelch.open / .close, power ratios */
  double gain;          /* chan->output.gain when the demodulator starts (the AGC owns it afterwards) */
  double deemph_rate, deemph_gain;   /* FM: chan->fm.rate (0 = flat FM), chan->fm.gain */
  double threshold_extend;           /* FM: chan->fm.threshold, 0 or 1 */
  int pll_enable;       /* chan->pll.enable: linear -- carrier-tracking PLL in front of the detector, lock detector, PLL squelch
                           (src/linear.c:83-153); FM -- PLL demodulator instead of the discriminator (src/fm.c:176-203) */
  int pll_square;       /* chan->pll.square (linear): squaring loop for suppressed-carrier signals */
  double pll_loop_bw;   /* chan->pll.loop_bw, Hz (linear; the FM loop is 500 Hz wide, src/fm.c:181) */
  double tone_freq;     /* FM: chan->fm.tone_freq, Hz; non-zero = PL / CTCSS tone squelch (src/fm.c:264-311) */
} chz_demod_params;
#define CHZ_DEMOD_LINEAR 0
#define CHZ_DEMOD_FM 1
typedef struct chz_demod_status {
  int frame;            /* 0: PCM present (send_output(chan, samples, N, mute)); 1: no samples (send_output(chan, NULL, N, mute)) */
  int mute;
  int squelch_state;
  int pll_lock;         /* chan->pll.lock (linear) */
  double output_power;  /* chan->output.power */
  double gain;          /* chan->output.gain after the block */
  double n0;            /* chan->sig.n0 (smoothed) */
  double snr;           /* linear: the squelch's SNR (the SNR squelch's, else the PLL's); FM: chan->fm.snr */
  double foffset;       /* chan->sig.foffset (FM; linear with the PLL on) */
  double pdeviation;    /* FM: chan->fm.pdeviation */
  double pll_snr;       /* chan->pll.snr (linear) */
  double pll_cphase;    /* chan->pll.cphase, radians (linear) */
  double tone_deviation;/* FM: chan->fm.tone_deviation, Hz */
  int pll_rotations;    /* chan->pll.rotations (linear) */
  int tone_mute;        /* FM: 1 while the tone squelch keeps the channel muted */
} chz_demod_status;
/* parameters of channels [ch0, ch0+n) from block `job` on (it must not have been enqueued yet); blocktime = radiod's Blocktime.
 * Waits for the demodulator stream only, never for the transform lanes. */
int chz_bank_set_demod(chz_engine *e, int bank, unsigned job, int ch0, int n, const chz_demod_params *p, double blocktime);
/* Demodulating blocks the caller supplies (radiod puts a channel's private second filter, filter2 -- chz_mini_* -- between the
 * channelizer and the demodulator, src/radio.c:1572-1594): chz_bank_write_block stores n channels' olen complex samples
 * (+ bb_power, noise estimates; NULL leaves what is there) into `slot`'s output image, chz_bank_demod runs the bank's
 * demodulators over the slot as block `job` on the demodulator stream (results: chz_bank_read_pcm). */
int chz_bank_write_block(chz_engine *e, int bank, int slot, int ch0, int n, const float *samples, const double *bb_power, const double *n0);
int chz_bank_demod(chz_engine *e, int bank, unsigned job, int slot);
/* on = 0: the bank's demodulators run only when chz_bank_demod says so (its channels pass through filter2 on the way);
 * on = 1 (default): behind every whole-bank channel launch */
int chz_bank_demod_auto(chz_engine *e, int bank, int on);
/* bytes between two channels' PCM rows = the stride of chz_bank_read_pcm's buffer.  Default olen*8 (stereo float32 fits);
 * chz_bank_set_pcm_stride, before the first chz_bank_set_demod, shrinks the rows to what the bank's encodings need (olen*2 for
 * mono S16), so that a block's PCM is one contiguous device-to-host copy of only the bytes that matter */
int chz_bank_pcm_stride(chz_engine *e, int bank);
int chz_bank_set_pcm_stride(chz_engine *e, int bank, int bytes);
/* PCM (pcm_stride bytes per channel, the encoding's N*channels samples first) and status of the block last demodulated
 * on `slot`; synchronous / asynchronous on the demodulator stream (completion: chz_sync) */
int chz_bank_read_pcm(chz_engine *e, int bank, int slot, int ch0, int n, void *pcm, chz_demod_status *status);
int chz_bank_read_pcm_async(chz_engine *e, int bank, int slot, int ch0, int n, void *pcm, chz_demod_status *status);
/* the same with one byte per channel instead of the 96-byte record: what the caller of send_output() needs every block */
#define CHZ_FLAG_NO_SAMPLES 1   /* send_output(chan, NULL, N, mute) */
#define CHZ_FLAG_MUTE 2
#define CHZ_FLAG_PLL_LOCK 4
#define CHZ_FLAG_TONE_MUTE 8
int chz_bank_read_pcm_flags_async(chz_engine *e, int bank, int slot, int ch0, int n, void *pcm, unsigned char *flags);
/* returns once everything chz_bank_read_pcm_async / _flags_async has enqueued for `slot` so far has landed in the host buffers --
   without waiting for later blocks already handed to the demodulator stream (a double-buffered host loop waits for block j-1
   here while block j runs) */
int chz_bank_pcm_wait(chz_engine *e, int bank, int slot);

/* ---- small inline masters: radiod's filter2 (src/radio.c:1572-1594: a private COMPLEX master of N = round2(2*blocksize)
 * points with one same-size COMPLEX slave, run inline by the channel thread; share/presets.conf:204,223,297).  A pool holds
 * every instance of one geometry (8 <= N = L+M-1 <= 8192, no prime factor above 13); ONE kernel launch serves all instances that are
 * due (one workgroup each: forward transform, gather x response with the slave's shift, ISB unpacking, backward transform,
 * all in LDS).  The device keeps no overlap state: a request carries its whole N-sample window, exactly what the
 * reference's mirrored ring holds at input_read_pointer (src/filter.c:626-636). */
typedef struct chz_mini chz_mini;
int chz_mini_create(chz_mini **out, int L, int M, int capacity, int device);     /* replaces create_filter_input + _output */
void chz_mini_destroy(chz_mini *m);
int chz_mini_capacity(const chz_mini *m);
int chz_mini_add(chz_mini *m);                                                  /* -> instance index */
int chz_mini_release(chz_mini *m, int inst);
int chz_mini_set_response(chz_mini *m, int inst, const float *resp);            /* N complex, as set_filter leaves it */
/* n requests in one launch: instance inst[i], window win[i] (N complex on the host), shift[i] (NULL = 0), isb[i] (NULL = off),
 * L output samples to out[i]; synchronous, thread-safe */
int chz_mini_execute(chz_mini *m, int n, const int *inst, const float *const *win, const int *shift, const unsigned char *isb,
                     float *const *out);

/* host-side helper exposed for tests: the closed-form gather descriptor
 * {t0,cnt,src0,dir,conj,wrap} that restates src/filter.c:728-911 */
int chz_gather_descriptor(int in_type, int master_bins, int P, int shift, int out6[6]);

#ifdef __cplusplus
}
#endif
#endif
