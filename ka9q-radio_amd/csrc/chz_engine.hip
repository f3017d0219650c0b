// chz_engine.hip -- device-side engine behind include/chz_engine.h.
//
// Owns the HBM-resident state of one master (input ring, intermediate buffers,
// ND spectrum slots, twiddle tables) and of its channel banks, and launches the
// kernels of chz_kernels.h.  Blocks are pipelined over 1/2/4 HIP streams ("lanes");
// everything per block is enqueued asynchronously.  The one cross-block dependency of
// the path (the spur-notch recurrence) is ordered on the device: each block's notch_fix
// kernel takes a ticket in block order and waits -- bounded in time, failing loudly -- for
// its predecessor's (enqueue_notch below); CHZ_NOTCH_ORDER=event carries it by HIP events
// between the lanes instead.  No other kernel ever waits for another kernel.
// Retunes never drain the pipeline: the small per-channel descriptors exist once per
// spectrum slot and are refreshed in stream order, responses are swapped by row.
// gfx950 only, no fallback.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include <thread>
#include <cmath>
#include <unistd.h>
// ---- the process is exiting ---------------------------------------------------------------------------------------------------------------------
// radiod ends through exit() from a signal handler's closedown() (src/main.c) without deleting a single filter: its front-end thread is still handing over
// blocks, and channel threads still ask for re-runs, while the HIP runtime's own exit handlers tear it down -- a kernel launched into that dies inside
// libamdhip64 (round 6: a segmentation fault at every shutdown, tests/c/exit_midstream.c).  chz_exit::at_exit is registered (atexit) by the first engine, i.e.
// AFTER the runtime has registered whatever it runs at exit, so it runs BEFORE: it raises `exiting` and waits (bounded) for the runtime calls in flight to
// return; from then on every runtime call and kernel launch of this library is skipped and its API reports -98.
// (The two read-modify-writes around every runtime call do not show in the free-running loop: 14.1 / 14.6 / 14.6 us per block without them against
// 14.2 / 15.1 / 14.2 with, three alternating quick runs on one box -- inside the run-to-run scatter.)
namespace chz_exit {
static std::atomic<int> exiting{0};
static std::atomic<long> inflight{0};
struct Scope {
  bool ok;
  Scope() { inflight.fetch_add(1, std::memory_order_seq_cst); ok = exiting.load(std::memory_order_seq_cst) == 0; }
  ~Scope() { inflight.fetch_sub(1, std::memory_order_release); }
};
static void at_exit() {
  exiting.store(1, std::memory_order_seq_cst);
  for (int i = 0; i < 20000 && inflight.load(std::memory_order_acquire) > 0; i++) usleep(100);      // <= 2 s
}
static void arm() { static std::once_flag once; std::call_once(once, [] { std::atexit(at_exit); }); }
}  // namespace chz_exit
#define CHZ_EXIT_SCOPE(name) chz_exit::Scope name
#include "chz_launch.h"
#include "chz_finetune.h"
#include "../../include/chz_engine.h"

using namespace chz;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}
#define HIPOK(call) do { chz_exit::Scope _xs; if (!_xs.ok) return fail(-98, "the process is exiting"); hipError_t _e = (call); if (_e != hipSuccess) \
  return fail(-10, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

#define CHZ_MAX_LANES 4
#define CHZ_STAGE_CAP 8192          // channels per staged descriptor refresh; larger edits take the bulk path

// One retired batch of response rows: reusable once everything enqueued before the swap has drained
struct Retired {
  std::vector<int> rows;
  hipEvent_t ev[CHZ_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
};

struct Bank {
  int P = 0, olen = 0, cap = 0, active = 0;
  int out_real = 0;             // 1: REAL-output slaves (olen floats per channel, chan_c2r); 0: COMPLEX
  ChanGeom g;
  float2* resp = nullptr;       // [cap + spare][P]; a channel's row is desc.row
  int rows_total = 0;
  std::vector<int> free_rows;
  std::deque<Retired> retired;
  float2* out = nullptr;        // [ND][cap][olen]: one output image per spectrum slot
  float2* tw_sub = nullptr;
  int last_slot = 0;            // slot of the most recent execute (what chz_bank_read returns)
  // host master copies of the small per-channel state; the device holds one copy per spectrum slot
  std::vector<ChanDesc> desc_h;          // [cap]
  std::vector<FineHost> fine_h;          // [cap] fine-tuning bookkeeping (chz_finetune.h)
  std::vector<FineDesc> fine_dh;         // [cap]
  std::vector<unsigned char> isb_h;      // [cap]
  std::vector<BeamDesc> beam_h;          // [cap]
  ChanDesc* desc = nullptr;     // [ND][cap]
  FineDesc* fine = nullptr;     // [ND][cap], allocated by the first chz_bank_set_tuning
  unsigned char* isb = nullptr; // [ND][cap], allocated by chz_bank_set_isb
  BeamDesc* beam = nullptr;     // [ND][cap], allocated by chz_bank_set_beam
  // channels whose slot copy is older than the host copy: a short sorted list of disjoint ranges per slot (two retunes at the two
  // ends of a multi-million-channel bank must not re-upload everything in between); beyond CHZ_DIRTY_MAX ranges the closest two merge
  std::vector<std::pair<int, int>> dirty[CHZ_ND];
  char* stage[CHZ_ND] = {nullptr, nullptr, nullptr, nullptr};   // pinned staging of one refresh per slot
  // the staging area of a slot is two halves used in turn, each with its own event: filling one half never waits for the copy
  // that is still reading the other
  hipEvent_t stage_ev[CHZ_ND][2] = {};
  bool stage_busy[CHZ_ND][2] = {};
  int stage_next[CHZ_ND] = {0, 0, 0, 0};
  // the linear demodulator behind the channel outputs; allocated by the first chz_bank_set_demod
  float2* any_scratch = nullptr;         // [ND][cap][2][P or M]: transform buffers of channel sizes beyond the LDS (chan_any<true>)
  DemodChan* dm_chan = nullptr;          // [cap]
  DemodState* dm_state = nullptr;        // [cap]
  DemodExt* dm_ext = nullptr;            // [cap] PLL / tone-squelch state; allocated when the first channel asks for either
  float2* dm_mix = nullptr;              // [cap][olen] the coherent modes' blocks after their PLL (pll_lanes); allocated with dm_ext
  int dm_pll_lin = 0;                    // channels of the linear demodulator with a carrier PLL
  int dm_fm_pll = 0, dm_fm_tone = 0;     // FM channels with the PLL demodulator / a PL-tone squelch
  int dm_lin = 0, dm_fm = 0;             // channels of the linear / of the FM demodulator
  int dm_fm_nopll = 0;                   // FM channels on the discriminator (not the PLL demodulator)
  DemodStatus* dm_status = nullptr;      // [ND][cap]
  unsigned char* dm_flags = nullptr;     // [ND][cap] one status byte per channel and block
  unsigned char* dm_pcm = nullptr;       // [ND][cap][pcm_stride]
  std::vector<DemodChan> dm_chan_h;      // [cap]
  struct OscHost { bool init = false; double freq = 0.0, phase0 = 0.0; unsigned job0 = 0; };
  std::vector<OscHost> dm_osc;           // [cap] chan->shift with set_osc's phase continuity
  int shared_rows = 0;                   // > 0: the bank's channels share this many response rows (chz_bank_create_shared)
  int dm_on = 0;                         // channels with a demodulator
  bool dm_auto = true;                   // demodulate behind every channel launch (false: only on chz_bank_demod)
  int pcm_stride = 0;                    // bytes between two channels' PCM rows (default olen*8: stereo float32)
  double dm_blocktime = 0.02;
  hipEvent_t ev_bank[CHZ_ND] = {nullptr, nullptr, nullptr, nullptr};   // behind the slot's channel (+ noise) kernel
  hipEvent_t ev_tail[CHZ_ND] = {nullptr, nullptr, nullptr, nullptr};   // behind the slot's demodulator kernel
  hipEvent_t ev_pcm[CHZ_ND] = {nullptr, nullptr, nullptr, nullptr};    // behind the slot's latest PCM read (chz_bank_pcm_wait)
  hipEvent_t ev_pcmgo[CHZ_ND] = {nullptr, nullptr, nullptr, nullptr};  // on the demodulator stream where a PCM read on the copy stream may start
  bool pcm_copying[CHZ_ND] = {false, false, false, false};             // a read on the copy stream is (or was) in flight for the slot
  bool tail_used[CHZ_ND] = {false, false, false, false};
  double* power = nullptr;      // [ND][cap] (tail of downconvert(), src/radio.c:1516-1520)
  double* agc_peak = nullptr;   // [ND][cap] the block's largest 2 ms slice energy, left by chan_ifft for demod_lin_lanes (allocated with the demodulators of a large bank)
  bool agc_peak_valid[CHZ_ND] = {false, false, false, false};   // the slot's image was written by this slot's latest whole-bank channel launch
  double* n0 = nullptr;         // [ND][cap] estimate_noise() (src/radio.c:1783-1866)
  unsigned* noise_hint = nullptr;   // [cap] the binade the channel's quantile fell into last time (noise_est: a guess it verifies, never a result)
  double noise_samprate = 0.0;  // front-end sample rate; 0 = off
};

// A lane = one HIP stream + its own intermediate buffer.  Consecutive blocks go to
// consecutive lanes, so block j+1's forward transform overlaps block j's tail and channel
// kernel (the "second HIP stream" of the north star).
struct Lane {
  hipStream_t s = nullptr;
  float2* buf = nullptr;
};

// A persistent host thread that issues the launches of a subset of the lanes.  It spins for a short while
// after each task so that back-to-back chz_run_blocks calls do not pay a wake-up, then sleeps.
struct Issuer {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<int> state{0};      // 0 idle, 1 task posted, 2 task done, 3 quit
  bool sleeping = false;
  std::function<void()> fn;
  void loop() {
    for (;;) {
      int spins = 0;
      for (;;) {
        const int s = state.load(std::memory_order_acquire);
        if (s == 1 || s == 3) break;
        if (++spins < 20000) { __builtin_ia32_pause(); continue; }
        std::unique_lock<std::mutex> lk(m);
        sleeping = true;
        cv.wait(lk, [&] { const int t = state.load(std::memory_order_acquire); return t == 1 || t == 3; });
        sleeping = false;
      }
      if (state.load(std::memory_order_acquire) == 3) return;
      fn();
      state.store(2, std::memory_order_release);
    }
  }
  void post(std::function<void()> f) {
    fn = std::move(f);
    std::lock_guard<std::mutex> lk(m);
    state.store(1, std::memory_order_release);
    if (sleeping) cv.notify_one();
  }
  void wait() {
    while (state.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
    state.store(0, std::memory_order_release);
  }
  void quit() {
    { std::lock_guard<std::mutex> lk(m); state.store(3, std::memory_order_release); cv.notify_one(); }
    if (th.joinable()) th.join();
  }
};

// Hand-over between issuing threads: the notch section of block number `next` (counted from the start of
// the run) may be issued; see enqueue_forward().
struct NotchTurn {
  std::atomic<int> next{0};
  std::atomic<int> next_tail{0};   // the same hand-over for what is issued to the demodulator stream
  std::atomic<int> abort{0};
};

#define CHZ_NOTCH_EVENTS 8

struct chz_engine {
  int L = 0, M = 0, N = 0, in_type = 0, bins = 0, per = 1, device = 0, ring_blocks = 0;
  FwdPlan plan;
  float* energy[CHZ_ND] = {nullptr, nullptr, nullptr, nullptr};   // |X|^2 image of each slot for large banks' noise windows (spec_energy), made on first need
  int noise_energy = -1;            // -1: by launch size, 0 never, 1 always (env CHZ_NOISE_ENERGY)
  int demod_wave = -1;              // linear demodulators: -1 by bank size (one channel per lane from 65536 channels up: its ~120 us of
                                    // latency per launch is only worth paying for a bank that fills the chip), 1 always a wavefront per
                                    // channel, 0 always a lane per channel (env CHZ_DEMOD_WAVE)
  int chan_stage = -1;              // output staging of chan_ifft: -1 by launch size, 0 never, 1 always (env CHZ_CHAN_STAGE)
  hipStream_t stream = nullptr;     // == lanes[0].s: input copies and anything not tied to a block
  bool own_stream = false;
  // control-plane uploads (responses): never waits for a lane.  Created on first use, AFTER the lanes: the runtime deals
  // streams onto its (4) hardware queues round-robin in creation order, and two lanes sharing a queue serialise
  // (measured: a fifth stream created between the lanes cost 15 -> 20 us per block).
  hipStream_t upload = nullptr;
  hipStream_t tail = nullptr;       // the demodulators' stream: one in-order queue carries their block-to-block state (created on first use)
  // chz_bank_read_pcm_flags_async copies on a stream of its own, so that block j's PCM travels to the host while block j+1 is being
  // demodulated (on the in-order demodulator stream the copy and the next kernel would take turns); created on first use
  hipStream_t pcmcopy = nullptr;
  Lane lanes[CHZ_MAX_LANES];
  int nlanes = 1;
#define CHZ_INPUT_MARKS 8
  hipEvent_t input_mark[CHZ_INPUT_MARKS] = {};   // chz_input_mark / chz_input_mark_wait
  hipEvent_t input_ready = nullptr; // after the latest ring write
  bool input_pending = false;
  float* ring = nullptr; long ring_len = 0;   // floats
  // raw A/D input: the ring holds int16 and fwd_first_real converts on load
  short* ring16 = nullptr; float scale16 = 1.0f; int derand = 0;
  unsigned long long* energy_part = nullptr; unsigned* clip_part = nullptr; int stat_n = 0;   // [ND][stat_n] per-wave partials
  long wpos = 0;                              // write position (floats)
  float2* spec[CHZ_ND] = {nullptr, nullptr, nullptr, nullptr};
  bool spec_owned[CHZ_ND] = {false, false, false, false};
  // A master length outside the compiled axes runs as Bluestein's chirp-z over a planned complex transform of length Mz >= 2N - 1
  // (chz_kernels.h: blue_pre / blue_mul / blue_post).  The twiddle tables and lane buffers below then belong to THAT transform
  // (blue->zp); `plan` only carries the master's own numbers (N, bins, a natural-order spectrum layout).
  struct Blue {
    FwdPlan zp; long Mz = 0;
    float2* chirp = nullptr;              // [N]  exp(-i pi n^2 / N)
    float2* bf = nullptr;                 // [zp.spec_elems]  F(conj(chirp) wrapped), in zp's storage order
    float2* za[CHZ_MAX_LANES] = {};       // [Mz] per lane: the transform's input, natural order
    float2* zs[CHZ_MAX_LANES] = {};       // [zp.spec_elems] per lane: its output
  };
  Blue* blue = nullptr;
  float2 *tw_sub_a = nullptr, *tw_sub_b = nullptr, *tw_sub_c = nullptr;
  float2 *tw1_tile = nullptr, *tw1_col = nullptr, *tw2_tile = nullptr, *tw2_col = nullptr, *tw2_full = nullptr;
  // spur notches: device tables + the event chain that orders the recurrence across lanes
  int n_notch = 0;
  int *notch_addr = nullptr, *notch_next = nullptr, *notch_head = nullptr;
  double* notch_alpha = nullptr; double* notch_state = nullptr;
  hipEvent_t notch_ev[CHZ_NOTCH_EVENTS] = {};
  unsigned notch_seq = 0; bool notch_have = false;      // notch_ev[(notch_seq-1) % 8] is the latest recorded one
  bool opt_noise_hint = true, opt_pll_lane0 = false, opt_notch_fold = true;      // chz_set_option, read when the engine is created
  int notch_order = 0;                                  // 0: device ticket (default), 1: HIP events (env CHZ_NOTCH_ORDER=event)
  long long notch_max_wait = 0;                         // ticket wait budget, counter ticks (default 3 s; env CHZ_NOTCH_WAIT_MS)
  unsigned* notch_ver = nullptr;                        // device: tickets served so far
  unsigned notch_tickets = 0;                           // host: tickets handed out so far
  unsigned* notch_err = nullptr;                        // pinned host word the kernel raises when a ticket wait runs out
  NotchTables notch_tab; std::vector<double> notch_alpha_h;
  NotchOwn* notch_own = nullptr;                        // device: who stores each listed bin inside fwd_rows
  RowsNotch notch_fold{};                               // n > 0: the list rides inside fwd_rows (short lists, directly planned masters; env CHZ_NOTCH_FOLD=0 keeps the kernel)
  std::vector<Bank> banks;
  hipGraphExec_t graph = nullptr; unsigned graph_job0 = 0; int graph_blocks = 0;
  int capture_blocks = 0;           // blocks of the capture in progress (the last one moves the ticket base on)
  bool graph_notch_event = false;   // env CHZ_GRAPH_NOTCH=event: order the notches of captured blocks by HIP events (round 2's way)
  int graph_min_blocks = 32;        // env CHZ_GRAPH_BLOCKS: a replay covers at least this many blocks (drained once per replay)
  // chz_run_blocks: events and issuing threads live as long as the engine
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_fork = nullptr, ev_join[CHZ_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<Issuer*> issuers;
};

extern "C" { static int sync_all(chz_engine* e); }      // (defined inside the extern "C" block below: same linkage for every compiler)

template <class T> static int upload(T** dst, const std::vector<f2>& v) {
  *dst = nullptr;
  if (v.empty()) return 0;
  HIPOK(hipMalloc((void**)dst, v.size() * sizeof(f2)));
  HIPOK(hipMemcpy(*dst, v.data(), v.size() * sizeof(f2), hipMemcpyHostToDevice));
  return 0;
}

static void free_bank(Bank& b) {
  hipFree(b.resp); hipFree(b.desc); hipFree(b.out); hipFree(b.tw_sub); hipFree(b.any_scratch); b.any_scratch = nullptr;
  hipFree(b.fine); hipFree(b.power); hipFree(b.n0); hipFree(b.noise_hint); b.noise_hint = nullptr; hipFree(b.isb); hipFree(b.beam); hipFree(b.agc_peak); b.agc_peak = nullptr;
  hipFree(b.dm_chan); hipFree(b.dm_state); hipFree(b.dm_ext); hipFree(b.dm_status); hipFree(b.dm_flags); hipFree(b.dm_pcm); hipFree(b.dm_mix);
  b.dm_mix = nullptr; b.dm_pll_lin = 0; b.dm_fm_pll = 0; b.dm_fm_tone = 0; b.dm_lin = 0; b.dm_fm = 0; b.dm_fm_nopll = 0;
  b.dm_chan = nullptr; b.dm_state = nullptr; b.dm_ext = nullptr; b.dm_status = nullptr; b.dm_flags = nullptr; b.dm_pcm = nullptr; b.dm_on = 0; b.dm_chan_h.clear(); b.dm_osc.clear();
  for (int s = 0; s < CHZ_ND; s++) {
    if (b.ev_bank[s]) (void)hipEventDestroy(b.ev_bank[s]);
    if (b.ev_tail[s]) (void)hipEventDestroy(b.ev_tail[s]);
    if (b.ev_pcm[s]) (void)hipEventDestroy(b.ev_pcm[s]);
    if (b.ev_pcmgo[s]) (void)hipEventDestroy(b.ev_pcmgo[s]);
    b.ev_bank[s] = b.ev_tail[s] = b.ev_pcm[s] = b.ev_pcmgo[s] = nullptr; b.tail_used[s] = false; b.pcm_copying[s] = false;
  }
  for (int s = 0; s < CHZ_ND; s++) {
    if (b.stage[s]) (void)hipHostFree(b.stage[s]);
    for (int h = 0; h < 2; h++) { if (b.stage_ev[s][h]) (void)hipEventDestroy(b.stage_ev[s][h]); b.stage_ev[s][h] = nullptr; b.stage_busy[s][h] = false; }
    b.stage[s] = nullptr; b.stage_next[s] = 0; b.dirty[s].clear();
  }
  for (auto& r : b.retired) for (auto ev : r.ev) if (ev) (void)hipEventDestroy(ev);
  b.retired.clear(); b.free_rows.clear();
  b.resp = nullptr; b.desc = nullptr; b.out = nullptr; b.tw_sub = nullptr; b.fine = nullptr; b.power = nullptr;
  b.n0 = nullptr; b.isb = nullptr; b.beam = nullptr; b.noise_samprate = 0.0;
  b.desc_h.clear(); b.fine_h.clear(); b.fine_dh.clear(); b.isb_h.clear(); b.beam_h.clear();
  b.active = 0; b.cap = 0;
}

extern "C" {

int chz_process_exiting(void) { return chz_exit::exiting.load(std::memory_order_acquire); }
const char* chz_last_error(void) { return g_err; }

int chz_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int chz_gather_descriptor(int in_type, int master_bins, int P, int shift, int out6[6]) {
  ChanDescH d = make_chan_desc(in_type, master_bins, P, shift);
  out6[0] = d.t0; out6[1] = d.cnt; out6[2] = d.src0; out6[3] = d.dir; out6[4] = d.conj; out6[5] = d.wrap;
  return 0;
}

// The demodulator stream and the PCM copy stream get HARDWARE QUEUES OF THEIR OWN (round 5).  The runtime deals plain streams onto its
// GPU_MAX_HW_QUEUES (4) hardware queues round-robin, so the streams beyond the four transform lanes share a queue with a lane and take
// turns with it.  Measured at 1.5 M channels (profiles/r05_chain_queues.txt): the SURVEY 8f chain 4.40-4.61 ms per block with plain
// streams, 4.12-4.28 ms with the demodulator stream on its own queue; and with the host link in the loop (1.43 M channels of S16 PCM,
// double-buffered host loop) the time between two completions 13.25 ms mean / 16.9 worst with plain streams, 17.0 / 17.9 with only the
// demodulator stream moved (its copies then queue up behind a lane), 12.1 / 12.7 with the PCM copy stream moved as well.  A library
// cannot set GPU_MAX_HW_QUEUES (the runtime may be up already); what it can do is create these streams through
// hipExtStreamCreateWithCUMask with EVERY compute unit enabled -- such a stream is given a queue of its own.  The transform lanes keep
// plain streams: they are exactly the four the runtime has queues for (with every stream masked one driver-command bench run hung: a CU-masked
// stream is a BLOCKING stream, see chz_engine_create -- masked lanes are only to be had with the notch ordered by HIP events).
// EXPERIMENT knob on top: CHZ_TAIL_CUS=n gives the demodulator stream n of the compute units to itself (spread
// evenly over the XCDs) and the transform lanes the others -- the partition itself buys nothing (4.20-4.33 ms), see the decision record.
// ---- options (round 6).  Dispatch thresholds and test hooks are set through the C ABI -- chz_set_option(name, value) before the engine is
// created -- not through the environment: the shipped libraries read the operator's variables only (CHZ_PLAN,
// CHZ_STREAMS, CHZ_NOTCH_ORDER, CHZ_OWN_QUEUES, CHZ_RCCL_LIB, CHZ_COMM_TIMEOUT_S and the drop-in's KA9Q_HIP_*); the Python test mirror
// (ka9q-radio_amd/engine.py) and the C test drivers translate the CHZ_* names the tests use into these calls.  A/B EXPERIMENT hooks
// (CHZ_TAIL_CUS, CHZ_NO_TWFULL, CHZ_AGC_PEAK, CHZ_CHAN_WPB, CHZ_FWD_BATCH_N, CHZ_GRAPH_NOTCH ...) exist only in builds with -DCHZ_EXPERIMENTS
// (the A/B targets of the Makefile): CHZ_XENV() is getenv() there and a null pointer in the shipped library.
#ifdef CHZ_EXPERIMENTS
#define CHZ_XENV(name) getenv(name)
#else
#define CHZ_XENV(name) ((const char*)nullptr)
#endif
struct ChzOptions {
  int chan_stage = -1;        // chan_ifft rows staged through LDS: -1 auto (launches of >= 16384 channels), 0 never, 1 always
  int noise_energy = -1;      // |X|^2 image for the noise windows: -1 auto, 0 never, 1 always
  int demod_wave = -1;        // demodulators: -1 auto (lane-per-channel kernels from 65536 channels on), 0 lanes always, 1 wavefront-per-channel always
  int enq_threads = 2;        // host threads issuing the lanes of chz_run_blocks (1, 2 or 4)
  int graph_blocks = 32;      // least number of blocks a captured graph covers
  int notch_fold = 1;         // short notch lists applied inside fwd_rows (0: always the notch_fix kernel)
  int noise_hint = 1;         // the noise kernel's per-channel binade guess
  int pll_lane0 = 0;          // 1: the coherent modes' PLL as round 2's one-lane loop inside the demodulator kernel (no scratch block)
  double notch_wait_ms = 3000.0;  // budget of a device-side ticket wait
  int fault_ticket_skew = 0;  // fault injection for the hosts' recovery paths: the host's tickets start this far ahead of the device's counter ...
  int allow_fault_injection = 0;  // ... only with this set as well
  char launch_id[64] = {0};   // chz_comm_create_file: the id of this launch (ranks of another launch's rendezvous file are refused)
};
static ChzOptions g_opt;
static std::mutex g_opt_mu;
extern "C" int chz_set_option(const char* name, const char* value) {
  if (!name) return fail(-1, "null option name");
  std::lock_guard<std::mutex> lk(g_opt_mu);
  const ChzOptions def{};
  const bool d = value == nullptr || value[0] == 0;            // null / empty: back to the default
  const int iv = d ? 0 : atoi(value);
  const std::string n(name);
  if (n == "chan_stage") g_opt.chan_stage = d ? def.chan_stage : (iv != 0);
  else if (n == "noise_energy") g_opt.noise_energy = d ? def.noise_energy : (iv != 0);
  else if (n == "demod_wave") g_opt.demod_wave = d ? def.demod_wave : (iv != 0);
  else if (n == "enq_threads") g_opt.enq_threads = d ? def.enq_threads : ((iv == 2 || iv == 4) ? iv : 1);
  else if (n == "graph_blocks") g_opt.graph_blocks = (d || iv <= 0 || iv > 4096) ? def.graph_blocks : iv;
  else if (n == "notch_fold") g_opt.notch_fold = d ? def.notch_fold : (value[0] != '0');
  else if (n == "noise_hint") g_opt.noise_hint = d ? def.noise_hint : (value[0] != '0');
  else if (n == "pll_lane0") g_opt.pll_lane0 = d ? def.pll_lane0 : 1;
  else if (n == "notch_wait_ms") g_opt.notch_wait_ms = (d || !(atof(value) > 0)) ? def.notch_wait_ms : atof(value);
  else if (n == "fault_ticket_skew") g_opt.fault_ticket_skew = d ? 0 : iv;
  else if (n == "allow_fault_injection") g_opt.allow_fault_injection = d ? 0 : (value[0] == '1');
  else if (n == "launch_id") snprintf(g_opt.launch_id, sizeof g_opt.launch_id, "%s", d ? "" : value);
  else return fail(-1, "unknown option '%s'", name);
  return 0;
}
static ChzOptions options() { std::lock_guard<std::mutex> lk(g_opt_mu); return g_opt; }

static int tail_cus() { static const int n = [] { const char* v = CHZ_XENV("CHZ_TAIL_CUS"); const int k = v ? atoi(v) : 0; return (k > 0 && k < 256) ? k : 0; }(); return n; }
// CHZ_OWN_QUEUES: 0 plain streams everywhere (rounds 1-4), 1 (default) the demodulator stream and the PCM copy stream, 2 every stream the
// engine launches kernels on
static int own_queues() { static const int m = [] { const char* v = getenv("CHZ_OWN_QUEUES"); const int k = v ? atoi(v) : 1; return (k >= 0 && k <= 4) ? k : 1; }(); return m; }
static hipError_t stream_create_masked(hipStream_t* s, bool tail) {
#ifndef HIPEMU
  const int n = tail_cus();
  // (round 6 A/B) 3 / 4: the demodulator and PCM copy streams as NON-BLOCKING streams of another priority (3 = highest, 4 = lowest): the runtime keeps
  // a pool of hardware queues per priority, so such a stream does not share a queue with the four normal-priority lanes either -- and, unlike a
  // CU-masked stream, it IS a hipStreamNonBlocking stream (scripts/micro/masked_stream_blocking.hip, profiles/r06_masked_stream_blocking.txt)
  if (n == 0 && tail && own_queues() >= 3) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
      const hipError_t r = hipStreamCreateWithPriority(s, hipStreamNonBlocking, own_queues() == 3 ? hi : lo);
      if (r == hipSuccess) return r;
      (void)hipGetLastError();
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  }
  if (n > 0 || own_queues() == 2 || (own_queues() == 1 && tail)) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 && cus <= 1024) {
      uint32_t mask[32] = {};
      for (int i = 0; i < cus; i++) {
        const bool t = n > 0 && (((i + 1) * n) / cus > (i * n) / cus);      // n of the units, evenly spread
        if (n == 0 || t == tail) mask[i / 32] |= 1u << (i % 32);
      }
      const hipError_t r = hipExtStreamCreateWithCUMask(s, (uint32_t)((cus + 31) / 32), mask);
      if (r == hipSuccess) return r;
      (void)hipGetLastError();                       // a runtime without the extension: plain streams
    }
  }
#endif
  (void)tail;
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

static int blue_setup(chz_engine* e);
int chz_engine_create(chz_engine** out, int L, int M, int in_type, int device, const char* plan_spec, int ring_blocks) {
  if (!out) return fail(-1, "null out pointer");
  *out = nullptr;
  if (L < 1 || M < 1) return fail(-1, "bad L/M");
  if (in_type != CHZ_REAL && in_type != CHZ_COMPLEX) return fail(-1, "in_type must be REAL or COMPLEX");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(-2, "no HIP device: the channelizer engine has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(-2, "device %d out of range (%d devices)", device, ndev);
  HIPOK(hipSetDevice(device));
  { void* probe = nullptr; HIPOK(hipMalloc(&probe, 256)); HIPOK(hipFree(probe)); }      // (the runtime is fully up -- and has registered its own exit handlers -- ...)
  chz_exit::arm();                                          // ... before this library registers the one that must run ahead of them
  const int N = L + M - 1;                                  // src/filter.c:196
  const int bins = in_type == CHZ_COMPLEX ? N : N / 2 + 1;  // src/filter.c:197
  if (bins < 2) return fail(-1, "transform too small");     // src/filter.c:198-199
  if (in_type == CHZ_REAL && (L & 1)) return fail(-1, "real input needs an even block length L");
  chz_engine* e = new chz_engine();
  struct Guard { chz_engine* e; ~Guard() { if (e) chz_engine_destroy(e); } } guard{e};   // frees everything on an early return
  e->L = L; e->M = M; e->N = N; e->in_type = in_type; e->bins = bins; e->device = device;
  e->per = in_type == CHZ_REAL ? 1 : 2;
  const char* envspec = getenv("CHZ_PLAN");
  if ((!plan_spec || !*plan_spec) && envspec && *envspec) plan_spec = envspec;
  if (!build_fwd_plan(N, in_type, plan_spec, e->plan)) {
    if (plan_spec && *plan_spec)
      return fail(-3, "no transform plan for N=%d (spec '%s'): N must factor into the compiled axis lengths", N, plan_spec);
    // any other length: chirp-z over the next planned complex length
    chz_engine::Blue* bl = new chz_engine::Blue();
    e->blue = bl;
    // the cheapest planned length within 4 % above the first one that exists (the planner's cost model, scaled by the length)
    double best = 1e300;
    for (long m = 2L * N - 1; m <= 4L * N + 4096 && m < (1L << 30); m++) {
      if (bl->Mz && (double)m > 1.04 * (double)bl->Mz + 64) break;
      { long r = m;                                              // the compiled axes only hold the primes 2, 3, 5, 13, 19
        for (int q : {2, 3, 5, 13, 19}) while (r % q == 0) r /= q;
        if (r != 1) continue; }
      FwdPlan cand; double sc = 0.0;
      if (!build_fwd_plan((int)m, CHZ_COMPLEX, nullptr, cand, &sc)) continue;
      if (!bl->Mz) bl->Mz = m;                                   // (the first hit anchors the window)
      sc *= (double)m;
      if (sc < best) { best = sc; bl->zp = cand; }
    }
    if (bl->Mz) bl->Mz = bl->zp.N;
    if (!bl->Mz) return fail(-3, "no transform plan for N=%d, and no planned length above 2N-1 for the chirp-z form either", N);
    FwdPlan& q = e->plan;
    q = FwdPlan();
    q.N = N; q.in_type = in_type; q.bins = bins; q.Na = 128; q.Nb = 1; q.Nc = (bins + 127) / 128;
    q.spec_pitch = 128; q.spec_off = 0; q.spec_elems = (long)q.Nc * 128 + 16;
    char d[256];
    snprintf(d, sizeof d, "N=%d %s as chirp-z over [%s]", N, in_type == CHZ_REAL ? "real" : "complex", bl->zp.desc.c_str());
    q.desc = d;
  }
  const FwdPlan& tp = e->blue ? e->blue->zp : e->plan;         // the transform that is actually executed
  const int minblocks = (N + L - 1) / L + 1;
  if (ring_blocks < minblocks) ring_blocks = minblocks < 8 ? 8 : minblocks;
  e->ring_blocks = ring_blocks;
  e->ring_len = (long)ring_blocks * L * e->per;
  HIPOK(stream_create_masked(&e->stream, false));
  e->own_stream = true;
  const ChzOptions opt = options();
  e->opt_noise_hint = opt.noise_hint != 0; e->opt_pll_lane0 = opt.pll_lane0 != 0; e->opt_notch_fold = opt.notch_fold != 0;
  if (opt.chan_stage >= 0) e->chan_stage = opt.chan_stage;
  if (opt.noise_energy >= 0) e->noise_energy = opt.noise_energy;
  if (opt.demod_wave >= 0) e->demod_wave = opt.demod_wave;
  const char* envl = getenv("CHZ_STREAMS");
  int nl = envl ? atoi(envl) : 4;
  e->nlanes = (nl >= 4) ? 4 : (nl >= 2) ? 2 : 1;             // must divide ND so slot and lane stay aligned
  HIPOK(hipEventCreateWithFlags(&e->input_ready, hipEventDisableTiming));
  HIPOK(hipEventCreate(&e->ev_t0)); HIPOK(hipEventCreate(&e->ev_t1));
  HIPOK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  for (int i = 0; i < CHZ_MAX_LANES; i++) HIPOK(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));
  for (int i = 0; i < CHZ_NOTCH_EVENTS; i++) HIPOK(hipEventCreateWithFlags(&e->notch_ev[i], hipEventDisableTiming | hipEventReleaseToDevice));
  {
    int khz = 0;                                        // constant-rate counter, kHz (100 MHz on this family)
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->device) != hipSuccess || khz <= 0) khz = 100000;
    const double ms = opt.notch_wait_ms;
    e->notch_max_wait = (long long)(ms * (double)khz);
  }
  if (const char* gn = CHZ_XENV("CHZ_GRAPH_NOTCH")) e->graph_notch_event = strcmp(gn, "event") == 0;
  e->graph_min_blocks = opt.graph_blocks;
  if (const char* no = getenv("CHZ_NOTCH_ORDER")) e->notch_order = strcmp(no, "event") == 0 ? 1 : (strcmp(no, "unordered-timing-only") == 0 ? 2 : 0);
  // (round 6) CU-masked LANES are blocking streams -- hipExtStreamCreateWithCUMask takes no flags and makes hipStreamDefault streams, ordered with the
  // legacy null stream -- and a device-side ticket wait between two blocking streams deadlocks as soon as ANY null-stream operation (a synchronous
  // hipMemcpy / hipMemset of this engine, torch's default stream in bench.py) lands between the two launches: the waiter spins for a kernel that
  // queues behind the null-stream operation, which waits for the waiter.  That was round 5's bench run that never came back (reproducer:
  // scripts/micro/masked_stream_blocking.hip, profiles/r06_masked_stream_blocking.txt).  Masked lanes therefore order the notch by HIP events, always.
  if ((own_queues() == 2 || tail_cus() > 0) && e->notch_order == 0) e->notch_order = 1;
  HIPOK(hipHostMalloc((void**)&e->notch_err, sizeof(unsigned), hipHostMallocMapped));
  *e->notch_err = 0;
  for (int i = 0; i < e->nlanes; i++) {
    if (i == 0) e->lanes[i].s = e->stream;
    else HIPOK(stream_create_masked(&e->lanes[i].s, false));
    HIPOK(hipMalloc((void**)&e->lanes[i].buf, sizeof(float2) * (size_t)tp.Ra * tp.inner));
  }
  HIPOK(hipMalloc((void**)&e->ring, sizeof(float) * (size_t)e->ring_len));
  HIPOK(hipMemset(e->ring, 0, sizeof(float) * (size_t)e->ring_len));       // src/filter.c:242,257
  e->wpos = (long)(M - 1) * e->per;                                          // src/filter.c:244,259
  for (int i = 0; i < CHZ_ND; i++) {
    HIPOK(hipMalloc((void**)&e->spec[i], sizeof(float2) * (size_t)e->plan.spec_elems));
    HIPOK(hipMemset(e->spec[i], 0, sizeof(float2) * (size_t)e->plan.spec_elems));
    e->spec_owned[i] = true;
  }
  int r;
  if ((r = upload(&e->tw_sub_a, tp.tw_sub_a)) || (r = upload(&e->tw_sub_b, tp.tw_sub_b)) ||
      (r = upload(&e->tw_sub_c, tp.tw_sub_c)) || (r = upload(&e->tw1_tile, tp.tw1_tile)) ||
      (r = upload(&e->tw1_col, tp.tw1_col)) || (r = upload(&e->tw2_tile, tp.tw2_tile)) ||
      (r = upload(&e->tw2_col, tp.tw2_col)) || (r = upload(&e->tw2_full, tp.tw2_full)))
    return r;
  HIPOK(hipDeviceSynchronize());       // null-stream memsets vs the engine's non-blocking streams
  if (e->blue) { r = blue_setup(e); if (r) return r; }
  guard.e = nullptr;
  *out = e;
  return 0;
}

static void drop_graph(chz_engine* e) {
  if (e->graph) { hipGraphExecDestroy(e->graph); e->graph = nullptr; }
}

static void free_notches(chz_engine* e) {
  hipFree(e->notch_addr); hipFree(e->notch_next); hipFree(e->notch_head); hipFree(e->notch_alpha); hipFree(e->notch_state); hipFree(e->notch_ver);
  hipFree(e->notch_own); e->notch_own = nullptr;
  e->notch_addr = e->notch_next = e->notch_head = nullptr; e->notch_alpha = nullptr; e->notch_state = nullptr; e->notch_ver = nullptr;
  e->n_notch = 0; e->notch_have = false; e->notch_tickets = 0; e->notch_fold = RowsNotch{};
}

void chz_engine_destroy(chz_engine* e) {
  if (!e) return;
  hipSetDevice(e->device);
  for (Issuer* is : e->issuers) { is->quit(); delete is; }
  e->issuers.clear();
  for (int i = 0; i < e->nlanes; i++) if (e->lanes[i].s) hipStreamSynchronize(e->lanes[i].s);
  if (e->upload) hipStreamSynchronize(e->upload);
  if (e->tail) hipStreamSynchronize(e->tail);
  if (e->pcmcopy) hipStreamSynchronize(e->pcmcopy);
  drop_graph(e);
  for (int i = 0; i < e->nlanes; i++) {
    hipFree(e->lanes[i].buf);
    if (i > 0 && e->lanes[i].s) hipStreamDestroy(e->lanes[i].s);
  }
  if (e->input_ready) hipEventDestroy(e->input_ready);
  for (auto ev : e->input_mark) if (ev) hipEventDestroy(ev);
  if (e->ev_t0) hipEventDestroy(e->ev_t0);
  if (e->ev_t1) hipEventDestroy(e->ev_t1);
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  for (auto ev : e->ev_join) if (ev) hipEventDestroy(ev);
  for (auto ev : e->notch_ev) if (ev) hipEventDestroy(ev);
  for (auto& b : e->banks) free_bank(b);
  hipFree(e->ring); hipFree(e->ring16); hipFree(e->energy_part); hipFree(e->clip_part);
  for (int i = 0; i < CHZ_ND; i++) if (e->spec_owned[i]) hipFree(e->spec[i]);
  for (int i = 0; i < CHZ_ND; i++) if (e->energy[i]) hipFree(e->energy[i]);
  if (e->blue) {
    hipFree(e->blue->chirp); hipFree(e->blue->bf);
    for (int i = 0; i < CHZ_MAX_LANES; i++) { hipFree(e->blue->za[i]); hipFree(e->blue->zs[i]); }
    delete e->blue;
  }
  hipFree(e->tw_sub_a); hipFree(e->tw_sub_b); hipFree(e->tw_sub_c);
  hipFree(e->tw1_tile); hipFree(e->tw1_col); hipFree(e->tw2_tile); hipFree(e->tw2_col); hipFree(e->tw2_full);
  free_notches(e);
  if (e->notch_err) (void)hipHostFree(e->notch_err);
  if (e->upload) hipStreamDestroy(e->upload);
  if (e->tail) hipStreamDestroy(e->tail);
  if (e->pcmcopy) hipStreamDestroy(e->pcmcopy);
  if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
  delete e;
}

int chz_engine_info(const chz_engine* e, chz_info* info) {
  if (!e || !info) return fail(-1, "null argument");
  memset(info, 0, sizeof *info);
  info->L = e->L; info->M = e->M; info->N = e->N; info->in_type = e->in_type; info->bins = e->bins;
  info->ring_blocks = e->ring_blocks; info->Na = e->plan.Na; info->Nb = e->plan.Nb; info->Nc = e->plan.Nc;
  info->n_banks = (int)e->banks.size();
  info->spec_elems = e->plan.spec_elems; info->spec_na = e->plan.Na; info->spec_pitch = e->plan.spec_pitch; info->spec_off = e->plan.spec_off;
  info->lanes = e->nlanes;
  snprintf(info->plan, sizeof info->plan, "%s", e->plan.desc.c_str());
  return 0;
}

int chz_engine_set_stream(chz_engine* e, void* hip_stream) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e) return fail(-1, "null engine");
  int r = sync_all(e);
  if (r) return r;
  drop_graph(e);
  // a caller-owned stream means the caller does the ordering: collapse to one lane
  for (int i = 1; i < e->nlanes; i++) { hipFree(e->lanes[i].buf); hipStreamDestroy(e->lanes[i].s); e->lanes[i] = Lane(); }
  e->nlanes = 1;
  if (e->own_stream) { hipStreamDestroy(e->stream); e->own_stream = false; }
  e->stream = (hipStream_t)hip_stream;
  e->lanes[0].s = e->stream;
  e->notch_have = false;
  return 0;
}

// raised by notch_fix when its ticket never came up: the spectra of that block and of every later one are NOT notched
static int check_device_errors(const chz_engine* e) {
  const unsigned v = e->notch_err ? __atomic_load_n(e->notch_err, __ATOMIC_ACQUIRE) : 0u;
  if (v) return fail(-8, "spur-notch ordering failed at ticket %u: a block's notch kernel waited for its predecessor in vain; "
                         "the engine must be re-created (CHZ_NOTCH_ORDER=event orders by HIP events instead)", v - 1u);
  return 0;
}
static int sync_all(chz_engine* e) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  for (int i = 0; i < e->nlanes; i++) HIPOK(hipStreamSynchronize(e->lanes[i].s));
  if (e->tail) HIPOK(hipStreamSynchronize(e->tail));
  if (e->pcmcopy) HIPOK(hipStreamSynchronize(e->pcmcopy));
  // nothing is travelling any more: the demodulator launches that follow (a graph capture among them: an eagerly recorded event is no
  // business of a capture) need not wait for the copy stream's last read of their slot
  for (auto& b : e->banks) for (int s = 0; s < CHZ_ND; s++) b.pcm_copying[s] = false;
  return check_device_errors(e);
}
int chz_sync(chz_engine* e) {
  if (!e) return fail(-1, "null engine");
  return sync_all(e);
}
// non-blocking health check (the drop-in calls it from every execute_filter_input): < 0 once a device-side check failed
int chz_engine_check(const chz_engine* e) {
  if (!e) return fail(-1, "null engine");
  return check_device_errors(e);
}

static int ring_write(chz_engine* e, const float* src, long n, hipMemcpyKind kind) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (e->ring16) return fail(-1, "int16 and float input cannot be mixed on one engine");
  const long nf = n * e->per;
  if (nf < 0 || nf > e->ring_len) return fail(-1, "write of %ld samples does not fit the ring", n);
  const long first = (e->wpos + nf <= e->ring_len) ? nf : e->ring_len - e->wpos;
  if (first > 0) HIPOK(hipMemcpyAsync(e->ring + e->wpos, src, sizeof(float) * (size_t)first, kind, e->stream));
  if (nf > first) HIPOK(hipMemcpyAsync(e->ring, src + first, sizeof(float) * (size_t)(nf - first), kind, e->stream));
  e->wpos = (e->wpos + nf) % e->ring_len;
  if (e->nlanes > 1) { HIPOK(hipEventRecord(e->input_ready, e->stream)); e->input_pending = true; }
  return 0;
}
int chz_input_write(chz_engine* e, const float* host, long n) {
  if (!e || !host) return fail(-1, "null argument");
  return ring_write(e, host, n, hipMemcpyHostToDevice);
}
int chz_input_write_device(chz_engine* e, const float* dev, long n) {
  if (!e || !dev) return fail(-1, "null argument");
  return ring_write(e, dev, n, hipMemcpyDeviceToDevice);
}
// rx888.c's convert() + write_rfilter(.., NULL, n) (src/rx888.c:753-767,800-829): the samples stay int16
// all the way into HBM (half the PCIe and half the first pass's read bytes); scaling, de-randomising,
// the energy sum and the clip count happen where the first transform pass loads them.
static inline int blue_grid(long n);
static int ring16_write(chz_engine* e, const short* src, long n, float scale, int randomize, hipMemcpyKind kind) {
  if (e->in_type != CHZ_REAL) return fail(-1, "int16 input is a real A/D stream");
  if (n < 0 || n > e->ring_len) return fail(-1, "write of %ld samples does not fit the ring", n);
  HIPOK(hipSetDevice(e->device));
  if (!e->ring16) {
    if (e->wpos != (long)(e->M - 1)) return fail(-1, "int16 and float input cannot be mixed on one engine");
    HIPOK(hipMalloc((void**)&e->ring16, sizeof(short) * (size_t)e->ring_len));
    HIPOK(hipMemset(e->ring16, 0, sizeof(short) * (size_t)e->ring_len));
    e->stat_n = e->blue ? blue_grid(e->blue->Mz) * 4 : e->plan.grid1 * (e->plan.block1 / 64);     // per-wave partials of the pass that loads the samples
    HIPOK(hipMalloc((void**)&e->energy_part, sizeof(unsigned long long) * (size_t)CHZ_ND * e->stat_n));
    HIPOK(hipMalloc((void**)&e->clip_part, sizeof(unsigned) * (size_t)CHZ_ND * e->stat_n));
    HIPOK(hipMemset(e->energy_part, 0, sizeof(unsigned long long) * (size_t)CHZ_ND * e->stat_n));
    HIPOK(hipMemset(e->clip_part, 0, sizeof(unsigned) * (size_t)CHZ_ND * e->stat_n));
    HIPOK(hipDeviceSynchronize());     // the memsets run on the null stream, which the engine's non-blocking streams do not wait for
    drop_graph(e);
  }
  if (scale != e->scale16 || (randomize != 0) != (e->derand != 0)) drop_graph(e);   // baked into captured launches
  e->scale16 = scale; e->derand = randomize != 0;
  const long first = (e->wpos + n <= e->ring_len) ? n : e->ring_len - e->wpos;
  if (first > 0) HIPOK(hipMemcpyAsync(e->ring16 + e->wpos, src, sizeof(short) * (size_t)first, kind, e->stream));
  if (n > first) HIPOK(hipMemcpyAsync(e->ring16, src + first, sizeof(short) * (size_t)(n - first), kind, e->stream));
  e->wpos = (e->wpos + n) % e->ring_len;
  if (e->nlanes > 1) { HIPOK(hipEventRecord(e->input_ready, e->stream)); e->input_pending = true; }
  return 0;
}
int chz_input_write_i16(chz_engine* e, const short* host, long n, float scale, int randomize) {
  if (!e || !host) return fail(-1, "null argument");
  return ring16_write(e, host, n, scale, randomize, hipMemcpyHostToDevice);
}
int chz_input_write_i16_device(chz_engine* e, const short* dev, long n, float scale, int randomize) {
  if (!e || !dev) return fail(-1, "null argument");
  return ring16_write(e, dev, n, scale, randomize, hipMemcpyDeviceToDevice);
}
// sum of x^2 and number of clipped samples over the L new samples of the block last transformed into `slot`
int chz_input_stats(chz_engine* e, int slot, unsigned long long* energy, unsigned* clips) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  if (!e->ring16) return fail(-1, "input statistics exist for int16 input only");
  std::vector<unsigned long long> en((size_t)e->stat_n); std::vector<unsigned> cl((size_t)e->stat_n);
  hipStream_t st = e->lanes[slot % e->nlanes].s;
  HIPOK(hipMemcpyAsync(en.data(), e->energy_part + (size_t)slot * e->stat_n, sizeof(unsigned long long) * en.size(), hipMemcpyDeviceToHost, st));
  HIPOK(hipMemcpyAsync(cl.data(), e->clip_part + (size_t)slot * e->stat_n, sizeof(unsigned) * cl.size(), hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  unsigned long long se = 0; unsigned sc = 0;
  for (size_t i = 0; i < en.size(); i++) { se += en[i]; sc += cl[i]; }
  if (energy) *energy = se;
  if (clips) *clips = sc;
  return 0;
}
int chz_input_seek(chz_engine* e, unsigned job, const float* history) {
  if (!e) return fail(-1, "null engine");
  if (e->ring16) return fail(-1, "chz_input_seek re-seats a float ring (this engine takes int16 input)");
  HIPOK(hipSetDevice(e->device));
  { int r = sync_all(e); if (r) return r; }
  const long start = (long)(((unsigned long long)job * (unsigned long long)e->L) % (unsigned long long)((long)e->ring_blocks * e->L)) * e->per;
  const long nh = (long)(e->M - 1) * e->per;                      // floats of history in front of the block's new samples
  e->wpos = start;
  if (history) { int r = ring_write(e, history, e->M - 1, hipMemcpyHostToDevice); if (r) return r; }
  else {
    const long first = (start + nh <= e->ring_len) ? nh : e->ring_len - start;
    if (first > 0) HIPOK(hipMemsetAsync(e->ring + start, 0, sizeof(float) * (size_t)first, e->stream));
    if (nh > first) HIPOK(hipMemsetAsync(e->ring, 0, sizeof(float) * (size_t)(nh - first), e->stream));
    e->wpos = (start + nh) % e->ring_len;
  }
  HIPOK(hipStreamSynchronize(e->stream));                          // the caller's history buffer is free again
  e->input_pending = false;
  return 0;
}
int chz_input_mark(chz_engine* e, int k) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e || k < 0 || k >= CHZ_INPUT_MARKS) return fail(-1, "bad argument");
  if (!e->input_mark[k]) HIPOK(hipEventCreateWithFlags(&e->input_mark[k], hipEventDisableTiming));
  HIPOK(hipEventRecord(e->input_mark[k], e->stream));
  return 0;
}
int chz_input_mark_wait(chz_engine* e, int k) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e || k < 0 || k >= CHZ_INPUT_MARKS) return fail(-1, "bad argument");
  if (e->input_mark[k]) HIPOK(hipEventSynchronize(e->input_mark[k]));
  return 0;
}
int chz_engine_notch_order(chz_engine* e, int by_event) {
  if (!e) return fail(-1, "null engine");
  { int r = sync_all(e); if (r) return r; }
  drop_graph(e);
  e->notch_order = by_event ? 1 : 0;
  e->notch_have = false;
  return 0;
}
int chz_input_ring(chz_engine* e, float** dev_ring, long* ring_len_floats) {
  if (!e) return fail(-1, "null engine");
  if (dev_ring) *dev_ring = e->ring;
  if (ring_len_floats) *ring_len_floats = e->ring_len;
  return 0;
}

// ---- kernel launches ---------------------------------------------------------
struct Instr {      // optional per-kernel timing: one (begin, end) event pair per launch
  Instr() = default;
  Instr(const Instr&) = delete;
  ~Instr() { for (auto e : ev) hipEventDestroy(e); }
  bool on = false;
  std::vector<hipEvent_t> ev;   // 2 per launch: dispatch begin / end timestamps
  std::vector<int> kind;        // 0 first, 1 cols, 2 rows, 3 noise, 4 chan, 5 notch fix
  hipEvent_t e0 = nullptr, e1 = nullptr;   // pair for the launch being issued
};
// begin=true: allocate the pair the next launch will carry; begin=false: nothing (kept for symmetry)
static void mark(Instr* in, hipStream_t, int kind, bool begin) {
  if (!in || !in->on) return;
  if (begin) {
    hipEventCreate(&in->e0); hipEventCreate(&in->e1);
    in->ev.push_back(in->e0); in->ev.push_back(in->e1); in->kind.push_back(kind);
  } else { in->e0 = nullptr; in->e1 = nullptr; }
}
#define IN_E0(in) ((in) && (in)->on ? (in)->e0 : nullptr)
#define IN_E1(in) ((in) && (in)->on ? (in)->e1 : nullptr)

static inline int lane_of(const chz_engine* e, unsigned job, const Instr* in) {
  return (in && in->on) ? 0 : (int)(job % (unsigned)e->nlanes);
}

// K2 for one block.  apply_notch_filters (src/filter.c:464-474) is a recurrence over blocks, and consecutive
// blocks live on different streams.  Two ways to order it, both issued here in BLOCK ORDER (`turn` hands the notch
// section from issuing thread to issuing thread):
//   ticket (default)  notch_fix waits on the device for a counter to reach its own sequence number.  Because the
//                     kernels are issued in block order, block j-1's notch_fix -- and everything it depends on, which
//                     precedes it on its own stream -- sits IN FRONT of block j's on every hardware queue they might
//                     share (GPU_MAX_HW_QUEUES, several engines per process): the kernel being waited for can always
//                     start.  The wait is bounded anyway; if it runs out nothing is written, a host-visible error word
//                     is raised and every later entry point fails loudly (check_device_errors).
//   event             (CHZ_NOTCH_ORDER=event, and always inside a graph capture) a HIP event recorded behind block
//                     j-1's notch_fix is waited for by block j's stream.  No device-side waiting at all, but on this
//                     runtime a cross-stream event costs ~20 us of latency per link of the chain (measured: 26 us per
//                     block instead of 16), which makes the chain the bottleneck of the free-running pipeline.
// capture_first: first block of a graph capture -- the previous graph launch is ordered by the launching stream, and an
// event recorded outside the capture must not be waited on inside it.
static int enqueue_notch(chz_engine* e, int slot, hipStream_t st, Instr* in, NotchTurn* turn, int seq, bool capture_first, bool capturing) {
  if (e->n_notch <= 0) return 0;
  if (turn) {
    while (turn->next.load(std::memory_order_acquire) != seq) {
      if (turn->abort.load(std::memory_order_relaxed)) return fail(-6, "another issuing thread failed");
      __builtin_ia32_pause();
    }
  }
  int rc = 0;
  // inside a capture the ticket is relative to a device word (one captured node serves every replay); CHZ_GRAPH_NOTCH=event
  // keeps round 2's event chain inside graphs
  const bool by_event = e->notch_order == 1 || (capturing && e->graph_notch_event);
  do {
    if (by_event && e->notch_have && !capture_first && e->nlanes > 1) {
      hipError_t he = hipStreamWaitEvent(st, e->notch_ev[(e->notch_seq - 1u) % CHZ_NOTCH_EVENTS], 0);
      if (he != hipSuccess) { rc = fail(-10, "hipStreamWaitEvent failed: %s", hipGetErrorString(he)); break; }
    }
    NotchFixParams q{};
    q.spec = e->spec[slot]; q.addr = e->notch_addr; q.next = e->notch_next; q.head = e->notch_head;
    q.alpha = e->notch_alpha; q.state = e->notch_state; q.n = e->n_notch;
    e->notch_tab.fill_inline(q, e->notch_alpha_h.data());
    q.err = e->notch_err; q.max_wait = e->notch_max_wait;
    const bool ticket = !by_event && e->nlanes > 1 && e->notch_order != 2;      // 2: A/B timing of the bare kernel, WRONG results
    if (ticket && capturing) {
      q.ver = e->notch_ver; q.seq_base = e->notch_ver + 2; q.seq = (unsigned)seq;
      q.adv = seq == e->capture_blocks - 1 ? (unsigned)e->capture_blocks : 0u;
    } else if (ticket) { q.ver = e->notch_ver; q.seq = e->notch_tickets; }
    mark(in, st, 5, true);
    if (launch_notch_fix(st, q, IN_E0(in), IN_E1(in))) { rc = fail(-4, "notch list too long"); break; }
    mark(in, st, 5, false);
    if (ticket && !capturing) e->notch_tickets++;               // taken only by a launch that went out
    if (by_event && e->nlanes > 1) {
      hipError_t he = hipEventRecord(e->notch_ev[e->notch_seq % CHZ_NOTCH_EVENTS], st);
      if (he != hipSuccess) { rc = fail(-10, "hipEventRecord failed: %s", hipGetErrorString(he)); break; }
      e->notch_seq++; e->notch_have = true;
    }
  } while (0);
  if (turn) {
    if (rc) turn->abort.store(1, std::memory_order_relaxed);
    turn->next.store(seq + 1, std::memory_order_release);
  }
  return rc;
}

// the planned complex transform of a chirp-z master: za (natural order) -> zs (zp's storage order), all on `st`
static int blue_fft(chz_engine* e, const float2* za, float2* lbuf, float2* zs, hipStream_t st) {
  const FwdPlan& p = e->blue->zp;
  ColsParams a{};
  a.in = za; a.in_len = 0; a.in_start = 0; a.out = lbuf; a.rows = 1; a.inner = p.inner; a.T = p.T1; a.padk = p.padk1;
  a.tw_sub = e->tw_sub_a; a.tw_tile = e->tw1_tile; a.tw_col = e->tw1_col;
  if (launch_cols(p.ra, p.grid1, p.block1, p.lds1, st, a)) return fail(-4, "no kernel for axis a");
  if (p.Nb > 1) {
    ColsParams b{};
    b.in = lbuf; b.in_len = 0; b.in_start = 0; b.out = lbuf; b.rows = p.Ra; b.inner = p.Nc; b.T = p.T2;
    b.padk = p.padk2; b.tw_sub = e->tw_sub_b; b.tw_tile = e->tw2_tile; b.tw_col = e->tw2_col; b.tw_full = e->tw2_full;
    if (launch_cols(p.rb, p.grid2, p.block2, p.lds2, st, b)) return fail(-4, "no kernel for axis b");
  }
  RowsParams c{};
  c.lay = SpecLayout{p.Na, p.spec_pitch, p.spec_off}; c.ka_shift = p.ka_shift;
  c.buf = lbuf; c.spec = zs; c.Ra = p.Ra; c.Na = p.Na; c.Nb = p.Nb; c.Ta = p.Ta; c.ld = p.ld3;
  c.padg = p.padg3; c.N = p.N; c.mirror = false; c.tw_sub = e->tw_sub_c;
  if (launch_rows(p.rc, p.grid3, p.block3, p.lds3, st, c)) return fail(-4, "no kernel for axis c");
  return 0;
}
static inline int blue_grid(long n) { const long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }
// tables and buffers of a chirp-z master; Bf is made by the device's own transform
static int blue_setup(chz_engine* e) {
  chz_engine::Blue& b = *e->blue;
  const int N = e->N; const long Mz = b.Mz;
  std::vector<f2> c((size_t)N);
  for (long n = 0; n < N; n++) c[(size_t)n] = root_of_unity((long long)((unsigned long long)n * (unsigned long long)n % (unsigned long long)(2L * N)), 2LL * N, -1);
  int r = upload(&b.chirp, c);
  if (r) return r;
  for (int i = 0; i < e->nlanes; i++) {
    HIPOK(hipMalloc((void**)&b.za[i], sizeof(float2) * (size_t)Mz));
    HIPOK(hipMalloc((void**)&b.zs[i], sizeof(float2) * (size_t)b.zp.spec_elems));
    HIPOK(hipMemset(b.zs[i], 0, sizeof(float2) * (size_t)b.zp.spec_elems));
  }
  HIPOK(hipMalloc((void**)&b.bf, sizeof(float2) * (size_t)b.zp.spec_elems));
  std::vector<f2> bz((size_t)Mz, f2{0.f, 0.f});
  for (long m = 0; m < N; m++) {
    const f2 v{c[(size_t)m].x, -c[(size_t)m].y};
    bz[(size_t)m] = v;
    if (m) bz[(size_t)(Mz - m)] = v;
  }
  HIPOK(hipMemcpy(b.za[0], bz.data(), sizeof(f2) * (size_t)Mz, hipMemcpyHostToDevice));
  HIPOK(hipDeviceSynchronize());
  r = blue_fft(e, b.za[0], e->lanes[0].buf, b.zs[0], e->stream);
  if (r) return r;
  HIPOK(hipMemcpyAsync(b.bf, b.zs[0], sizeof(float2) * (size_t)b.zp.spec_elems, hipMemcpyDeviceToDevice, e->stream));
  HIPOK(hipStreamSynchronize(e->stream));
  HIPOK(hipGetLastError());
  return 0;
}
static int enqueue_forward_blue(chz_engine* e, unsigned job, int ln, hipStream_t st, long start, Instr* in) {
  chz_engine::Blue& b = *e->blue;
  const int slot = job % CHZ_ND;
  float2* lbuf = e->lanes[ln].buf;
  const SpecLayout zlay{b.zp.Na, b.zp.spec_pitch, b.zp.spec_off};
  mark(in, st, 0, true);
  BluePreParams pre{e->ring, e->ring_len, start, e->per, b.chirp, b.za[ln], e->N, b.Mz, nullptr, 1.0f, 0, 0, nullptr, nullptr};
  if (e->ring16) {
    pre.ring16 = e->ring16; pre.scale16 = e->scale16; pre.derand = e->derand; pre.new_from = e->M - 1;
    pre.energy_part = e->energy_part + (size_t)slot * e->stat_n; pre.clip_part = e->clip_part + (size_t)slot * e->stat_n;
  }
  CHZ_LAUNCH(blue_pre, blue_grid(b.Mz), 256, 0, st, IN_E0(in), IN_E1(in), pre);
  mark(in, st, 0, false);
  int r = blue_fft(e, b.za[ln], lbuf, b.zs[ln], st);
  if (r) return r;
  mark(in, st, 1, true);
  BlueMulParams mul{b.zs[ln], b.bf, b.za[ln], zlay, b.Mz};
  CHZ_LAUNCH(blue_mul, blue_grid(b.Mz), 256, 0, st, IN_E0(in), IN_E1(in), mul);
  mark(in, st, 1, false);
  r = blue_fft(e, b.za[ln], lbuf, b.zs[ln], st);
  if (r) return r;
  mark(in, st, 2, true);
  BluePostParams post{b.zs[ln], b.chirp, e->spec[slot], zlay, SpecLayout{e->plan.Na, e->plan.spec_pitch, e->plan.spec_off}, e->bins, (float)(1.0 / (double)b.Mz)};
  CHZ_LAUNCH(blue_post, blue_grid(e->bins), 256, 0, st, IN_E0(in), IN_E1(in), post);
  mark(in, st, 2, false);
  return 0;
}

static int enqueue_forward(chz_engine* e, unsigned job, Instr* in, NotchTurn* turn = nullptr, int seq = 0, bool capture_first = false, bool capturing = false) {
  const FwdPlan& p = e->plan;
  const int slot = job % CHZ_ND;
  const int ln = lane_of(e, job, in);
  hipStream_t st = e->lanes[ln].s;
  float2* lbuf = e->lanes[ln].buf;
  if (ln != 0 && e->input_pending) HIPOK(hipStreamWaitEvent(st, e->input_ready, 0));
  const long start = (long)(((unsigned long long)job * (unsigned long long)e->L) % (unsigned long long)((long)e->ring_blocks * e->L)) * e->per;
  if (e->blue) {
    const int r = enqueue_forward_blue(e, job, ln, st, start, in);
    return r ? r : enqueue_notch(e, slot, st, in, turn, seq, capture_first, capturing);
  }
  if (e->in_type == CHZ_REAL) {
    FirstRealParams a{};
    a.ring = e->ring; a.ring_len = e->ring_len; a.start = start; a.buf = lbuf; a.inner = p.inner;
    a.T = p.T1; a.Ra = p.Ra; a.padk = p.padk1; a.tw_sub = e->tw_sub_a; a.tw_tile = e->tw1_tile; a.tw_col = e->tw1_col;
#if CHZ_TW_SHUFFLE
    if (p.T1 != 16) return fail(-4, "this A/B build generates its twiddles across rows of 16 lanes: T1 must be 16");
#endif
    if (e->ring16) {
      a.ring16 = e->ring16; a.scale16 = e->scale16; a.derand = e->derand; a.new_from = e->M - 1;
      a.energy_part = e->energy_part + (size_t)slot * e->stat_n; a.clip_part = e->clip_part + (size_t)slot * e->stat_n;
    }
    mark(in, st, 0, true);
    if (launch_first_real(p.ra, p.grid1, p.block1, p.lds1, st, a, IN_E0(in), IN_E1(in))) return fail(-4, "no kernel for axis a");
    mark(in, st, 0, false);
  } else {
    ColsParams a{};
    a.in = reinterpret_cast<const float2*>(e->ring); a.in_len = e->ring_len / 2; a.in_start = start / 2;
    a.out = lbuf; a.rows = 1; a.inner = p.inner; a.T = p.T1; a.padk = p.padk1;
    a.tw_sub = e->tw_sub_a; a.tw_tile = e->tw1_tile; a.tw_col = e->tw1_col;
    mark(in, st, 0, true);
    if (launch_cols(p.ra, p.grid1, p.block1, p.lds1, st, a, IN_E0(in), IN_E1(in))) return fail(-4, "no kernel for axis a");
    mark(in, st, 0, false);
  }
  if (p.Nb > 1) {
    ColsParams b{};
    b.in = lbuf; b.in_len = 0; b.in_start = 0; b.out = lbuf; b.rows = p.Ra; b.inner = p.Nc; b.T = p.T2;
    b.padk = p.padk2; b.tw_sub = e->tw_sub_b; b.tw_tile = e->tw2_tile; b.tw_col = e->tw2_col;
    if (!CHZ_XENV("CHZ_NO_TWFULL")) b.tw_full = e->tw2_full;
    int grid2 = p.grid2;
#if CHZ_XCD_AFFINE
    // (experiment build: axis b of a three-axis plan only -- a complex master's first axis goes through launch_cols above, untouched... and
    //  would be remapped too: this build serves REAL three-axis masters, which is what the experiment measures)
    const int ncomp = (p.Ra + p.ka_shift + p.Ta - 1) / p.Ta;
    if (ncomp > 8) return fail(-4, "the XCD-affine experiment build places at most 8 components (plan has %d)", ncomp);
    b.xa = XcdAffine{CHZ_XCD_AFFINE, p.Ta, p.ka_shift, (int)((job * (unsigned)ncomp) & 7u), ncomp};
    grid2 = 8 * p.Ta * (p.Nc / p.T2);
#endif
    mark(in, st, 1, true);
    if (launch_cols(p.rb, grid2, p.block2, p.lds2, st, b, IN_E0(in), IN_E1(in))) return fail(-4, "no kernel for axis b");
    mark(in, st, 1, false);
  }
  RowsParams c{};
  c.lay = SpecLayout{p.Na, p.spec_pitch, p.spec_off}; c.ka_shift = p.ka_shift;
  c.buf = lbuf; c.spec = e->spec[slot]; c.Ra = p.Ra; c.Na = p.Na; c.Nb = p.Nb; c.Ta = p.Ta; c.ld = p.ld3;
  c.padg = p.padg3; c.N = p.N; c.mirror = e->in_type == CHZ_REAL; c.tw_sub = e->tw_sub_c;
  int grid3 = p.grid3;
#if CHZ_XCD_AFFINE
  {
    const int ncomp = (p.Ra + p.ka_shift + p.Ta - 1) / p.Ta;
    if (ncomp > 8 || p.Nb <= 1 || e->in_type != CHZ_REAL) return fail(-4, "the XCD-affine experiment build serves REAL three-axis masters with at most 8 a-tiles");
    c.xa = XcdAffine{CHZ_XCD_AFFINE, p.Ta, p.ka_shift, (int)((job * (unsigned)ncomp) & 7u), ncomp};
    grid3 = 8 * p.Nb;
  }
#endif
  // K2 inside this pass (short lists ordered by the device ticket): the launch is then what has to go out in block order
  const bool by_event = e->notch_order == 1 || (capturing && e->graph_notch_event);
  const bool fold = e->n_notch > 0 && e->notch_fold.n > 0 && !by_event && e->notch_order != 2;
  if (fold) {
    if (turn) {
      while (turn->next.load(std::memory_order_acquire) != seq) {
        if (turn->abort.load(std::memory_order_relaxed)) return fail(-6, "another issuing thread failed");
        __builtin_ia32_pause();
      }
    }
    c.nf = e->notch_fold;
    c.nf.state = e->notch_state; c.nf.err = e->notch_err; c.nf.max_wait = e->notch_max_wait;
    const bool ticket = e->nlanes > 1;
    if (ticket && capturing) {
      c.nf.ver = e->notch_ver; c.nf.seq_base = e->notch_ver + 2; c.nf.seq = (unsigned)seq;
      c.nf.adv = seq == e->capture_blocks - 1 ? (unsigned)e->capture_blocks : 0u;
    } else if (ticket) { c.nf.ver = e->notch_ver; c.nf.seq = e->notch_tickets; }
    mark(in, st, 2, true);
    const int lr = launch_rows(p.rc, grid3, p.block3, p.lds3, st, c, IN_E0(in), IN_E1(in));
    mark(in, st, 2, false);
    if (!lr && ticket && !capturing) e->notch_tickets++;           // taken only by a launch that went out
    if (turn) {
      if (lr) turn->abort.store(1, std::memory_order_relaxed);
      turn->next.store(seq + 1, std::memory_order_release);
    }
    if (lr) return fail(-4, "no kernel for axis c");
    return 0;
  }
  mark(in, st, 2, true);
  if (launch_rows(p.rc, grid3, p.block3, p.lds3, st, c, IN_E0(in), IN_E1(in))) return fail(-4, "no kernel for axis c");
  mark(in, st, 2, false);
  return enqueue_notch(e, slot, st, in, turn, seq, capture_first, capturing);
}

#if CHZ_FWD_BATCH
// EXPERIMENT (see chz_kernels.h): the forward transform of blocks job .. job+B-1 (B = 2 or 4, job a multiple of B) as three launches of
// B x the grid, on `st`; block y uses lane (job + y) % 4's intermediate buffer and spectrum slot (job + y) % 4.  REAL three-axis masters,
// float input, no notch list (timing only).
static int enqueue_forward_batch(chz_engine* e, unsigned job, int B, hipStream_t st) {
  const FwdPlan& p = e->plan;
  if (e->blue || e->in_type != CHZ_REAL || p.Nb <= 1 || e->ring16 || e->n_notch > 0 || e->nlanes != 4) return fail(-4, "the batched-pass experiment serves REAL three-axis masters on 4 lanes without a notch list");
  FirstRealParams a{};
  a.ring = e->ring; a.ring_len = e->ring_len; a.inner = p.inner; a.T = p.T1; a.Ra = p.Ra; a.padk = p.padk1;
  a.tw_sub = e->tw_sub_a; a.tw_tile = e->tw1_tile; a.tw_col = e->tw1_col; a.nbatch = B;
  ColsParams b{};
  b.in_len = 0; b.in_start = 0; b.rows = p.Ra; b.inner = p.Nc; b.T = p.T2; b.padk = p.padk2;
  b.tw_sub = e->tw_sub_b; b.tw_tile = e->tw2_tile; b.tw_col = e->tw2_col; b.tw_full = e->tw2_full; b.nbatch = B;
  RowsParams c{};
  c.lay = SpecLayout{p.Na, p.spec_pitch, p.spec_off}; c.ka_shift = p.ka_shift;
  c.Ra = p.Ra; c.Na = p.Na; c.Nb = p.Nb; c.Ta = p.Ta; c.ld = p.ld3; c.padg = p.padg3; c.N = p.N; c.mirror = 1; c.tw_sub = e->tw_sub_c; c.nbatch = B;
  for (int y = 0; y < B; y++) {
    const unsigned j = job + (unsigned)y;
    a.bstart[y] = (long)(((unsigned long long)j * (unsigned long long)e->L) % (unsigned long long)((long)e->ring_blocks * e->L)) * e->per;
    float2* lb = e->lanes[j % 4u].buf;
    a.bbuf[y] = lb; b.bbuf[y] = lb; c.bbuf[y] = lb; c.bspec[y] = e->spec[j % CHZ_ND];
  }
  a.start = a.bstart[0]; a.buf = a.bbuf[0]; b.in = b.bbuf[0]; b.out = b.bbuf[0]; c.buf = c.bbuf[0]; c.spec = c.bspec[0];
  if (launch_first_real(p.ra, dim3((unsigned)p.grid1, (unsigned)B), p.block1, p.lds1, st, a)) return fail(-4, "no kernel for axis a");
  if (launch_cols(p.rb, dim3((unsigned)p.grid2, (unsigned)B), p.block2, p.lds2, st, b)) return fail(-4, "no kernel for axis b");
  if (launch_rows(p.rc, dim3((unsigned)p.grid3, (unsigned)B), p.block3, p.lds3, st, c)) return fail(-4, "no kernel for axis c");
  return 0;
}
#endif

// output image of one slot; a sample is one float (REAL banks) or one float2
static inline size_t bank_sample_bytes(const Bank& b) { return b.out_real ? sizeof(float) : sizeof(float2); }
static inline float2* bank_out(const Bank& b, int slot) {
  return reinterpret_cast<float2*>(reinterpret_cast<char*>(b.out) + (size_t)slot * b.cap * b.olen * bank_sample_bytes(b));
}
static inline char* bank_out_at(const Bank& b, int slot, int ch) {
  return reinterpret_cast<char*>(bank_out(b, slot)) + (size_t)ch * b.olen * bank_sample_bytes(b);
}

// ---- per-slot descriptors ------------------------------------------------------
#define CHZ_DIRTY_MAX 16
static inline void mark_dirty(Bank& b, int ch0, int n) {
  if (n <= 0) return;
  for (int s = 0; s < CHZ_ND; s++) {
    auto& d = b.dirty[s];
    int lo = ch0, hi = ch0 + n;
    // absorb every range that touches [lo, hi), keep the list sorted
    size_t i = 0;
    while (i < d.size() && d[i].second < lo) i++;
    size_t j = i;
    while (j < d.size() && d[j].first <= hi) { if (d[j].first < lo) lo = d[j].first; if (d[j].second > hi) hi = d[j].second; j++; }
    d.erase(d.begin() + (long)i, d.begin() + (long)j);
    d.insert(d.begin() + (long)i, std::make_pair(lo, hi));
    while (d.size() > CHZ_DIRTY_MAX) {           // too many islands: merge the two with the smallest gap between them
      size_t best = 0; int gap = d[1].first - d[0].second;
      for (size_t k = 1; k + 1 < d.size(); k++) if (d[k + 1].first - d[k].second < gap) { gap = d[k + 1].first - d[k].second; best = k; }
      d[best].second = d[best + 1].second;
      d.erase(d.begin() + (long)best + 1);
    }
  }
}
static inline size_t stage_bytes_per_channel() { return sizeof(ChanDesc) + sizeof(FineDesc) + sizeof(BeamDesc) + 1; }

// Bring slot `slot`'s device copy of the descriptors up to the host copy, in stream order on `st`: blocks already
// enqueued on this slot keep what they were launched with, blocks of other slots are not touched at all.
static int refresh_slot(chz_engine* e, Bank& b, int slot, hipStream_t st) {
  if (b.dirty[slot].empty()) return 0;
  const int chunk = b.cap < CHZ_STAGE_CAP ? b.cap : CHZ_STAGE_CAP;
  const size_t half_bytes = (stage_bytes_per_channel() * (size_t)chunk + 64 + 63) & ~(size_t)63;      // (+ the 8-byte rounding of each segment)
  if (!b.stage[slot]) {
    HIPOK(hipHostMalloc((void**)&b.stage[slot], 2 * half_bytes, hipHostMallocDefault));
    for (int h = 0; h < 2; h++) HIPOK(hipEventCreateWithFlags(&b.stage_ev[slot][h], hipEventDisableTiming));
  }
  for (const auto& range : b.dirty[slot]) {
    const int lo = range.first, hi = range.second > b.cap ? b.cap : range.second;
    for (int c0 = lo; c0 < hi; c0 += chunk) {
      const int n = hi - c0 < chunk ? hi - c0 : chunk;
      const int h = b.stage_next[slot]; b.stage_next[slot] ^= 1;
      if (b.stage_busy[slot][h]) HIPOK(hipEventSynchronize(b.stage_ev[slot][h]));   // the copy before last out of this half: long done
      // (round 6) the staged descriptors travel by the desc_push kernel reading the pinned staging buffer, not by hipMemcpyAsync: see chz_kernels.h
      char* sp = b.stage[slot] + (size_t)h * half_bytes;
      PushParams pp{};
      auto seg = [&](void* dst, const void* src, size_t bytes) {
        memcpy(sp, src, bytes);
        pp.seg[pp.nseg].dst = dst; pp.seg[pp.nseg].src = sp; pp.seg[pp.nseg].bytes = (unsigned)bytes; pp.nseg++;
        sp += (bytes + 7) & ~(size_t)7;
      };
      seg(b.desc + (size_t)slot * b.cap + c0, b.desc_h.data() + c0, sizeof(ChanDesc) * (size_t)n);
      if (b.fine) seg(b.fine + (size_t)slot * b.cap + c0, b.fine_dh.data() + c0, sizeof(FineDesc) * (size_t)n);
      if (b.beam) seg(b.beam + (size_t)slot * b.cap + c0, b.beam_h.data() + c0, sizeof(BeamDesc) * (size_t)n);
      if (b.isb) seg(b.isb + (size_t)slot * b.cap + c0, b.isb_h.data() + c0, (size_t)n);
      launch_desc_push(pp, st);
      HIPOK(hipEventRecord(b.stage_ev[slot][h], st));
      b.stage_busy[slot][h] = true;
    }
  }
  b.dirty[slot].clear();
  return 0;
}
// Large edits (a whole bank being set up): drain once and write all four slot copies directly.
static int refresh_all_bulk(chz_engine* e, Bank& b) {
  int r = sync_all(e);
  if (r) return r;
  for (int s = 0; s < CHZ_ND; s++) {
    for (const auto& range : b.dirty[s]) {
      const int lo = range.first, hi = range.second > b.cap ? b.cap : range.second;
      if (lo >= hi) continue;
      const size_t n = (size_t)(hi - lo), off = (size_t)s * b.cap + lo;
      HIPOK(hipMemcpy(b.desc + off, b.desc_h.data() + lo, sizeof(ChanDesc) * n, hipMemcpyHostToDevice));
      if (b.fine) HIPOK(hipMemcpy(b.fine + off, b.fine_dh.data() + lo, sizeof(FineDesc) * n, hipMemcpyHostToDevice));
      if (b.beam) HIPOK(hipMemcpy(b.beam + off, b.beam_h.data() + lo, sizeof(BeamDesc) * n, hipMemcpyHostToDevice));
      if (b.isb) HIPOK(hipMemcpy(b.isb + off, b.isb_h.data() + lo, n, hipMemcpyHostToDevice));
    }
    b.dirty[s].clear();
  }
  return 0;
}
static int after_edit(chz_engine* e, Bank& b, int ch0, int n) {
  mark_dirty(b, ch0, n);
  if (n > CHZ_STAGE_CAP) return refresh_all_bulk(e, b);
  return 0;
}

// which kernels serve the bank's demodulators: the linear ones of a LARGE bank run one channel per lane (demod_lin_lanes: 1.2 against
// 2.2 ns per channel, but ~120 us per launch however small the bank -- profiles/r03_demod_crossover.jsonl), those of a small bank,
// FM, and a coherent-mode channel whose PLL has no scratch block the wavefront-per-channel kernel
static void demod_paths(const chz_engine* e, const Bank& b, DemodParams& d) {
  d.lin_lanes = (b.dm_lin > 0 && (e->demod_wave == 0 || (e->demod_wave < 0 && b.dm_lin >= 65536))) ? 1 : 0;
  // FM on the discriminator likewise (demod_fm_lanes keeps the baseband in the bank's scratch block)
  d.fm_lanes = (b.dm_fm_nopll > 0 && b.dm_mix != nullptr && (e->demod_wave == 0 || (e->demod_wave < 0 && b.dm_fm_nopll >= 65536))) ? 1 : 0;
  if (d.fm_lanes) d.mix = b.dm_mix;
  const bool fm_wave = b.dm_fm > 0 && (!d.fm_lanes || b.dm_fm > b.dm_fm_nopll);
  const bool lin_wave = b.dm_lin > 0 && (!d.lin_lanes || (b.dm_pll_lin > 0 && d.mix == nullptr));
  d.wave_any = (fm_wave || lin_wave) ? 1 : 0;
}
static int enqueue_bank(chz_engine* e, int bank, unsigned job, Instr* in, int ch0 = 0, int n = -1, bool partial = false) {
  Bank& b = e->banks[(size_t)bank];
  if (n < 0) n = b.active;
  if (n <= 0 || !b.resp) return 0;
  const int slot = (int)(job % CHZ_ND);
  hipStream_t st = e->lanes[lane_of(e, (unsigned)slot, in)].s;
  __atomic_store_n(&b.last_slot, slot, __ATOMIC_RELAXED);      // two issuing threads pass here; chz_run_blocks settles the final value
  { int r = refresh_slot(e, b, slot, st); if (r) return r; }
  // Only a launch over the WHOLE bank is "the block": a partial re-run (the drop-in's miss path re-runs single channels of a
  // block that has already been demodulated) must neither step anybody's AGC / squelch / PLL a second time nor overwrite PCM.
  const bool whole_bank = !partial && ch0 == 0 && n == b.active;      // (a range call over a one-channel bank is still a re-run: the caller says so)
  // the demodulator of the block that used this slot last (4 blocks ago) still reads the output image
  if (b.tail_used[slot] && !(in && in->on)) HIPOK(hipStreamWaitEvent(st, b.ev_tail[slot], 0));
  const size_t so = (size_t)slot * b.cap;
  ChanParams c{};
  if (!chan_layout(c, SpecLayout{e->plan.Na, e->plan.spec_pitch, e->plan.spec_off}, e->bins)) return fail(-4, "spectrum layout beyond the reach of the index reciprocal");
  c.spec = e->spec[slot]; c.resp = b.resp; c.desc = b.desc + so; c.out = bank_out(b, slot); c.ch0 = ch0; c.nch = n; c.olen = b.olen;
  c.tw_sub = b.tw_sub;
  c.stage = e->chan_stage >= 0 ? e->chan_stage : (n >= 16384);
  c.isb = b.isb ? b.isb + so : nullptr; c.beam = b.beam ? b.beam + so : nullptr;
  c.fine = b.fine ? b.fine + so : nullptr; c.power = b.power ? b.power + so : nullptr; c.job = job;
  if (b.fine) fine_launch(c, 1 + e->L / (e->M - 1), job);
  // the AGC's first look at the block rides in the channel kernel's epilogue when the rows pass through LDS anyway (large launches) and the
  // lane-per-channel demodulator follows (CHZ_AGC_PEAK=0: A/B knob, the demodulator then walks the block twice as before)
  // (only when somebody will use them: linear channels outside the coherent modes, served by the lane-per-channel demodulator)
  const bool peaks = b.agc_peak && whole_bank && c.stage && !b.g.any && !b.out_real && (c.fine || c.power) && b.dm_on > 0 && b.dm_auto &&
                     b.dm_lin > b.dm_pll_lin && (e->demod_wave == 0 || (e->demod_wave < 0 && b.dm_lin >= 65536));
  if (peaks) { c.agc_peak = b.agc_peak + so; int sps = (int)std::rint(b.olen * .002 / b.dm_blocktime); c.agc_sps = sps < 1 ? 1 : sps; }
  if (whole_bank) b.agc_peak_valid[slot] = peaks;
  const int per_block = b.g.any ? 1 : b.g.wpb * b.g.cpw;
  const int grid = (n + per_block - 1) / per_block;
  mark(in, st, 4, true);
  if (b.g.any) {                       // a size outside the register-tiled menu: one workgroup per channel
    c.m_bins = e->bins; c.m_real = e->in_type == CHZ_REAL;
    if (b.out_real) { c.fine = nullptr; c.power = nullptr; }
    if (launch_chan_any(b.g, n, st, c, b.tw_sub, b.out_real != 0, b.any_scratch ? b.any_scratch + (size_t)slot * b.cap * 2 * b.g.lb + (size_t)ch0 * 2 * b.g.lb : nullptr,
                        IN_E0(in), IN_E1(in))) return fail(-4, "no scratch for P=%d", b.P);
  } else if (b.out_real) {
    c.m_bins = e->bins; c.m_real = e->in_type == CHZ_REAL; c.fine = nullptr; c.power = nullptr; c.stage = 0;
    if (launch_chan_real(b.g.r, grid, b.g.wpb * 64, b.g.lds, st, c, IN_E0(in), IN_E1(in))) return fail(-4, "no real-output kernel for P=%d", b.P);
  } else if (launch_chan(b.g.r, grid, b.g.wpb * 64, b.g.lds, st, c, IN_E0(in), IN_E1(in))) return fail(-4, "no kernel for P=%d", b.P);
  mark(in, st, 4, false);
  if (b.n0 && b.noise_samprate > 0.0) {
    NoiseParams q = noise_params(e->bins, e->in_type == CHZ_REAL, b.out_real ? b.P / 2 + 1 : b.P, b.noise_samprate);   // slave->bins
    q.spec = e->spec[slot]; q.lay = c.lay; q.desc = b.desc + so; q.n0 = b.n0 + so; q.ch0 = ch0; q.nch = n; q.magic = c.magic; q.dpitch = c.dpitch; q.hint = b.noise_hint;
    // a launch that reads every bin many times over takes |X|^2 once per bin first (16384 channels x 1000 bins = 10 x the spectrum)
    const bool en = e->energy[slot] && (e->noise_energy >= 0 ? e->noise_energy != 0 : n >= 16384);
    mark(in, st, 3, true);
    if (en) {
      launch_spec_energy(e->spec[slot], e->energy[slot], e->bins, c.lay, c.magic, c.dpitch, st, IN_E0(in), nullptr);     // (timed together with the
      q.energy = e->energy[slot];                                                                                 //  windows when instrumented)
    }
    if (launch_noise(n, st, q, en ? nullptr : IN_E0(in), IN_E1(in))) return fail(-4, "no noise kernel for a %d-bin window", q.nbins);
    mark(in, st, 3, false);
  }
  // the linear demodulators of this bank, in block order on the demodulator stream.  Only whole-bank
  // launches feed them (a single-channel re-run of the drop-in's miss path does not advance anybody's AGC).
  if (b.dm_on > 0 && b.dm_auto && whole_bank) {
    hipStream_t ts = (in && in->on) ? st : e->tail;
    if (ts != st) {
      HIPOK(hipEventRecord(b.ev_bank[slot], st));
      HIPOK(hipStreamWaitEvent(ts, b.ev_bank[slot], 0));
    }
    if (b.pcm_copying[slot]) HIPOK(hipStreamWaitEvent(ts, b.ev_pcm[slot], 0));      // the slot's PCM of four blocks ago is still travelling
    DemodParams d{};
    d.in = bank_out(b, slot); d.power = b.power + so; d.n0 = b.n0 + so; d.chan = b.dm_chan; d.state = b.dm_state; d.ext = b.dm_ext;
    d.status = b.dm_status + so; d.flags = b.dm_flags + so; d.pcm = b.dm_pcm + so * (size_t)b.pcm_stride; d.ch0 = 0; d.nch = n; d.olen = b.olen;
    d.pcm_stride = b.pcm_stride; d.job = job; d.blocktime = b.dm_blocktime; d.power_alpha = 0.10;      // Power_alpha, src/radio.c:72
    d.lin_pll = b.dm_pll_lin > 0; d.fm_pll = b.dm_fm_pll > 0; d.fm_tone = b.dm_fm_tone > 0;
    d.mix = (d.lin_pll || d.fm_pll || d.fm_tone) ? b.dm_mix : nullptr;
    demod_paths(e, b, d);
    if (b.agc_peak && d.lin_lanes) {       // the AGC's first look: left by the channel kernel and, for coherent-mode channels, by pll_lanes
      d.agc_peak = b.agc_peak + so; d.peak_chan = b.agc_peak_valid[slot] ? 1 : 0; d.peak_pll = (d.lin_pll && d.mix != nullptr) ? 1 : 0;
    }
    mark(in, ts, 6, true);
    if (launch_demod(ts, d, IN_E0(in), IN_E1(in))) return fail(-4, "the demodulator kernel refuses blocks of %d samples", b.olen);
    mark(in, ts, 6, false);
    if (ts != st) { HIPOK(hipEventRecord(b.ev_tail[slot], ts)); b.tail_used[slot] = true; }
  }
  return 0;
}

int chz_forward(chz_engine* e, unsigned job) {
  if (!e) return fail(-1, "null engine");
  HIPOK(hipSetDevice(e->device));
  int r = enqueue_forward(e, job, nullptr);
  if (r) return r;
  HIPOK(hipGetLastError());
  return 0;
}

// notch list as radio.c builds it (src/radio.c:601-620), one averager gain per entry (src/filter.c:468)
int chz_set_notches_alpha(chz_engine* e, const int* bins, const double* alpha, int n) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e) return fail(-1, "null engine");
  { int r = sync_all(e); if (r) return r; }
  drop_graph(e);
  free_notches(e);
  if (n <= 0 || !bins || !alpha) return 0;
  if (n > 1024) return fail(-1, "at most 1024 notch entries (got %d): the list is applied by one workgroup", n);
  for (int i = 0; i < n; i++)
    if (bins[i] < 0 || bins[i] >= e->bins) return fail(-1, "notch bin %d out of range", bins[i]);
  NotchTables t = notch_tables(bins, n, SpecLayout{e->plan.Na, e->plan.spec_pitch, e->plan.spec_off});
  const size_t ib = sizeof(int) * (size_t)n, db = sizeof(double) * (size_t)n;
  HIPOK(hipMalloc((void**)&e->notch_addr, ib)); HIPOK(hipMalloc((void**)&e->notch_next, ib)); HIPOK(hipMalloc((void**)&e->notch_head, ib));
  HIPOK(hipMalloc((void**)&e->notch_alpha, db)); HIPOK(hipMalloc((void**)&e->notch_state, 2 * db));
  HIPOK(hipMemcpy(e->notch_addr, t.addr.data(), ib, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(e->notch_next, t.next.data(), ib, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(e->notch_head, t.head.data(), ib, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(e->notch_alpha, alpha, db, hipMemcpyHostToDevice));
  HIPOK(hipMemset(e->notch_state, 0, 2 * db));
  HIPOK(hipMalloc((void**)&e->notch_ver, 4 * sizeof(unsigned)));      // ticket counter, tombstone, base of captured tickets, spare
  HIPOK(hipMemset(e->notch_ver, 0, 4 * sizeof(unsigned)));
  HIPOK(hipDeviceSynchronize());
  e->notch_tab = t; e->notch_alpha_h.assign(alpha, alpha + n);
  e->notch_tickets = 0;
  // fault injection for the hosts' recovery paths (tests/test_dropin.py): the host's tickets start one ahead of the device's counter, so
  // the first notch waits for a turn that never comes, runs out of its budget and raises the error word -- what a wedged predecessor does
  // (a test hook in a production library: it takes TWO variables to arm, so one that leaks into a deployment's environment does nothing)
  { const ChzOptions o = options(); if (o.fault_ticket_skew && o.allow_fault_injection) e->notch_tickets = (unsigned)o.fault_ticket_skew; }
  e->n_notch = n;
  e->notch_fold = RowsNotch{};
  const char* nf = e->opt_notch_fold ? nullptr : "0";
#if CHZ_XCD_AFFINE
  nf = "0";        // the experiment builds remap blockIdx inside fwd_rows; the owner table of the folded notch is made for the default mapping
#endif
  if (!e->blue && !(nf && nf[0] == '0') && n <= CHZ_NOTCH_INLINE) {
    std::vector<NotchOwn> own;
    if (rows_notch_fill(e->notch_fold, own, e->plan, e->in_type == CHZ_REAL, bins, n)) {       // false leaves n = 0: the kernel serves the list
      HIPOK(hipMalloc((void**)&e->notch_own, sizeof(NotchOwn) * own.size()));
      HIPOK(hipMemcpy(e->notch_own, own.data(), sizeof(NotchOwn) * own.size(), hipMemcpyHostToDevice));
      e->notch_fold.own = e->notch_own; e->notch_fold.addr = e->notch_addr; e->notch_fold.next = e->notch_next;
      e->notch_fold.head = e->notch_head; e->notch_fold.alpha = e->notch_alpha;
    }
  }
  return 0;
}
int chz_set_notches(chz_engine* e, const int* bins, int n, double alpha) {
  std::vector<double> a((size_t)(n > 0 ? n : 0), alpha);
  return chz_set_notches_alpha(e, bins, a.data(), n);
}

static inline hipStream_t slot_stream(chz_engine* e, int slot) { return e->lanes[slot % e->nlanes].s; }

// device spectrum (SpecLayout order) -> host, natural bin order: whole rows as one 2-D copy, then
// the partial last row
static int copy_spectrum(chz_engine* e, int slot, float* host, hipStream_t st) {
  const FwdPlan& p = e->plan;
  const float2* src = e->spec[slot];
  if (p.spec_pitch == p.Na && p.spec_off == 0) {
    HIPOK(hipMemcpyAsync(host, src, sizeof(float2) * (size_t)e->bins, hipMemcpyDeviceToHost, st));
    return 0;
  }
  const long full = e->bins / p.Na, rest = e->bins - full * p.Na;
  if (full > 0)
    HIPOK(hipMemcpy2DAsync(host, sizeof(float2) * (size_t)p.Na, src + p.spec_off, sizeof(float2) * (size_t)p.spec_pitch,
                           sizeof(float2) * (size_t)p.Na, (size_t)full, hipMemcpyDeviceToHost, st));
  if (rest > 0)
    HIPOK(hipMemcpyAsync(host + 2 * full * p.Na, src + full * p.spec_pitch + p.spec_off, sizeof(float2) * (size_t)rest,
                         hipMemcpyDeviceToHost, st));
  return 0;
}

int chz_spectrum_read(chz_engine* e, int slot, float* host) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e || !host || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  hipStream_t st = slot_stream(e, slot);          // the lane that produced this slot
  int r = copy_spectrum(e, slot, host, st);
  if (r) return r;
  HIPOK(hipStreamSynchronize(st));
  return 0;
}
int chz_slot_stream(chz_engine* e, int slot, void** hip_stream) {
  if (!e || !hip_stream || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  *hip_stream = (void*)slot_stream(e, slot);
  return 0;
}
int chz_spectrum_device(chz_engine* e, int slot, float** dev) {
  if (!e || !dev || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  *dev = reinterpret_cast<float*>(e->spec[slot]);
  return 0;
}
int chz_spectrum_attach(chz_engine* e, int slot, float* dev) {
  if (!e || !dev || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  { int r = sync_all(e); if (r) return r; }
  drop_graph(e);
  if (e->spec_owned[slot]) hipFree(e->spec[slot]);
  e->spec[slot] = reinterpret_cast<float2*>(dev); e->spec_owned[slot] = false;
  return 0;
}

static int bank_create(chz_engine* e, int P, int olen, int capacity, int out_real, int shared_rows = 0) {
  if (!e) return fail(-1, "null engine");
  if (capacity < 1 || olen < 1 || olen > P) return fail(-1, "bad bank geometry");
  // P = olen*N/L must divide exactly (src/filter.c:312-316)
  if ((long long)olen * e->N % e->L != 0 || (long long)olen * e->N / e->L != P)
    return fail(-1, "P=%d is not olen*N/L for olen=%d N=%d L=%d", P, olen, e->N, e->L);
  Bank b;
  if (!build_chan_geom(P, b.g)) return fail(-3, "no channel kernel for P=%d (8 to %d points)", P, CHZ_ANY_MAX_P);
  HIPOK(hipSetDevice(e->device));
  // a failed allocation (the C_rt-sized banks take > 100 GB) must not leak the earlier ones
  struct Guard { Bank* b; ~Guard() { if (b) free_bank(*b); } } guard{&b};
  b.P = P; b.olen = olen; b.cap = capacity; b.active = 0; b.out_real = out_real;
  if (out_real && (P & 1)) return fail(-3, "real-output channels need an even P (got %d)", P);
  // spare response rows: set_filter writes a spare row and re-points the channel, so a swap never waits for the pipeline
  int spare = capacity / 16; if (spare < 16) spare = 16; if (spare > 4096) spare = 4096;
  b.rows_total = capacity + spare;
  if (shared_rows > 0) { b.shared_rows = shared_rows; b.rows_total = shared_rows; spare = 0; }     // rows are named by the caller, not swapped
  HIPOK(hipMalloc((void**)&b.resp, sizeof(float2) * (size_t)b.rows_total * P));
  HIPOK(hipMemset(b.resp, 0, sizeof(float2) * (size_t)b.rows_total * P));
  if (!b.shared_rows) for (int r = b.rows_total - 1; r >= capacity; r--) b.free_rows.push_back(r);
  b.desc_h.assign((size_t)capacity, ChanDesc{0, 0, 0, 1, 0, 0, 0, 0});
  for (int i = 0; i < capacity; i++) b.desc_h[(size_t)i].row = b.shared_rows ? 0 : i;
  HIPOK(hipMalloc((void**)&b.desc, sizeof(ChanDesc) * (size_t)CHZ_ND * capacity));
  for (int s = 0; s < CHZ_ND; s++)
    HIPOK(hipMemcpy(b.desc + (size_t)s * capacity, b.desc_h.data(), sizeof(ChanDesc) * (size_t)capacity, hipMemcpyHostToDevice));
  HIPOK(hipMalloc((void**)&b.out, bank_sample_bytes(b) * (size_t)CHZ_ND * capacity * olen));
  HIPOK(hipMemset(b.out, 0, bank_sample_bytes(b) * (size_t)CHZ_ND * capacity * olen));
  int r = upload(&b.tw_sub, b.g.any ? b.g.tw_any : b.g.tw_sub);
  if (r) return r;
  if (b.g.any && chan_any_prepare()) return fail(-3, "the runtime refuses %zu bytes of LDS per workgroup (P=%d)", b.g.lds, P);
  if (b.g.big) {
    const size_t bytes = sizeof(float2) * (size_t)CHZ_ND * capacity * 2 * (size_t)b.g.lb;
    if (bytes > ((size_t)16 << 30)) return fail(-2, "%d channels of %d points need %zu bytes of transform scratch", capacity, P, bytes);
    HIPOK(hipMalloc((void**)&b.any_scratch, bytes));
  }
  HIPOK(hipDeviceSynchronize());       // null-stream memsets vs the engine's non-blocking streams
  drop_graph(e);
  guard.b = nullptr;
  e->banks.push_back(std::move(b));
  return (int)e->banks.size() - 1;
}

int chz_bank_create(chz_engine* e, int P, int olen, int capacity) { return bank_create(e, P, olen, capacity, 0); }
// create_filter_output(.., REAL) (src/filter.c:372-395): olen real samples per channel and block
int chz_bank_create_real(chz_engine* e, int P, int olen, int capacity) { return bank_create(e, P, olen, capacity, 1); }
// Channels that use the same filter (every USB voice channel of a band, say) can share its response: the bank holds `nrows`
// response rows, a channel names the row it reads (chz_bank_set_rows), rows are written with chz_bank_set_row_responses.  The
// reference gives every slave its own copy (src/filter.c:1039-1043); the values are the same, the HBM traffic per channel and block
// drops from 8P + 8*olen to 8*olen bytes (the shared rows stay in the caches).
int chz_bank_create_shared(chz_engine* e, int P, int olen, int capacity, int nrows) {
  if (nrows < 1) return fail(-1, "a shared bank needs at least one response row");
  return bank_create(e, P, olen, capacity, 0, nrows);
}

#define BANK_CHECK(e, bank, ch0, n) \
  if (!(e) || (bank) < 0 || (bank) >= (int)(e)->banks.size()) return fail(-1, "bad bank"); \
  if ((ch0) < 0 || (n) < 0 || (ch0) + (n) > (e)->banks[(size_t)(bank)].cap) return fail(-1, "channel range out of bank capacity")

// Move retired response rows whose fences have passed back to the free list; with `need` > 0 wait (briefly: at most the
// blocks that were in flight when the rows were retired) until that many rows are free.
static int reclaim_rows(chz_engine* e, Bank& b, size_t need) {
  while (!b.retired.empty()) {
    Retired& r = b.retired.front();
    bool done = true;
    for (int l = 0; l < e->nlanes && done; l++) done = hipEventQuery(r.ev[l]) == hipSuccess;
    (void)hipGetLastError();      // hipEventQuery reports "not ready" through the sticky error
    if (!done) {
      if (b.free_rows.size() >= need) break;
      for (int l = 0; l < e->nlanes; l++) HIPOK(hipEventSynchronize(r.ev[l]));
    }
    for (int row : r.rows) b.free_rows.push_back(row);
    for (auto ev : r.ev) if (ev) (void)hipEventDestroy(ev);
    b.retired.pop_front();
  }
  return 0;
}

// set_filter's hot swap (src/filter.c:1039-1043): the new response goes to a spare row over the upload stream, the
// channel's descriptor is re-pointed (next block of every slot, in stream order), and the old row is recycled once
// everything enqueued before the swap has drained.  Nothing waits for the pipeline.
int chz_bank_set_responses(chz_engine* e, int bank, int ch0, int n, const float* resp) {
  BANK_CHECK(e, bank, ch0, n);
  if (!resp) return fail(-1, "null argument");
  Bank& b = e->banks[(size_t)bank];
  if (b.shared_rows) return fail(-1, "this bank's channels share %d response rows: use chz_bank_set_row_responses / chz_bank_set_rows", b.shared_rows);
  if (n == 0) return 0;
  HIPOK(hipSetDevice(e->device));
  const int spare_total = b.rows_total - b.cap;
  if (n > spare_total / 2) {
    // a whole bank being filled: drain once and write the channels' current rows in place
    { int r = sync_all(e); if (r) return r; }
    bool contiguous = true;
    for (int i = 0; i < n && contiguous; i++) contiguous = b.desc_h[(size_t)(ch0 + i)].row == b.desc_h[(size_t)ch0].row + i;
    if (contiguous) {
      HIPOK(hipMemcpy(b.resp + (size_t)b.desc_h[(size_t)ch0].row * b.P, resp, sizeof(float2) * (size_t)n * b.P, hipMemcpyHostToDevice));
    } else {
      for (int i = 0; i < n; i++)
        HIPOK(hipMemcpy(b.resp + (size_t)b.desc_h[(size_t)(ch0 + i)].row * b.P, resp + (size_t)2 * i * b.P, sizeof(float2) * (size_t)b.P, hipMemcpyHostToDevice));
    }
    return 0;
  }
  if (!e->upload) HIPOK(hipStreamCreateWithFlags(&e->upload, hipStreamNonBlocking));
  { int r = reclaim_rows(e, b, 0); if (r) return r; }
  if (b.free_rows.size() < (size_t)n) { int r = reclaim_rows(e, b, (size_t)n); if (r) return r; }
  if (b.free_rows.size() < (size_t)n) return fail(-7, "no spare response rows (%zu free, %d needed)", b.free_rows.size(), n);
  Retired old;
  for (int i = 0; i < n; i++) {
    const int row = b.free_rows.back(); b.free_rows.pop_back();
    HIPOK(hipMemcpyAsync(b.resp + (size_t)row * b.P, resp + (size_t)2 * i * b.P, sizeof(float2) * (size_t)b.P, hipMemcpyHostToDevice, e->upload));
    old.rows.push_back(b.desc_h[(size_t)(ch0 + i)].row);
    b.desc_h[(size_t)(ch0 + i)].row = row;
  }
  HIPOK(hipStreamSynchronize(e->upload));     // the caller's buffer may be pageable / reused; only the upload stream is waited for
  for (int l = 0; l < e->nlanes; l++) {
    HIPOK(hipEventCreateWithFlags(&old.ev[l], hipEventDisableTiming));
    HIPOK(hipEventRecord(old.ev[l], e->lanes[l].s));
  }
  b.retired.push_back(std::move(old));
  mark_dirty(b, ch0, n);
  return 0;
}
int chz_bank_set_rows(chz_engine* e, int bank, int ch0, int n, const int* rows) {
  BANK_CHECK(e, bank, ch0, n);
  if (!rows) return fail(-1, "null argument");
  Bank& b = e->banks[(size_t)bank];
  if (!b.shared_rows) return fail(-1, "bank was not created with chz_bank_create_shared");
  for (int i = 0; i < n; i++) if (rows[i] < 0 || rows[i] >= b.shared_rows) return fail(-1, "row %d out of range (bank has %d)", rows[i], b.shared_rows);
  for (int i = 0; i < n; i++) b.desc_h[(size_t)(ch0 + i)].row = rows[i];
  return after_edit(e, b, ch0, n);            // with the next block of each slot, in stream order, like a retune
}
int chz_bank_set_row_responses(chz_engine* e, int bank, int row0, int n, const float* resp) {
  BANK_CHECK(e, bank, 0, 0);
  if (!resp) return fail(-1, "null argument");
  Bank& b = e->banks[(size_t)bank];
  if (!b.shared_rows) return fail(-1, "bank was not created with chz_bank_create_shared");
  if (row0 < 0 || n < 0 || row0 + n > b.shared_rows) return fail(-1, "row range out of the bank's %d rows", b.shared_rows);
  HIPOK(hipSetDevice(e->device));
  { int r = sync_all(e); if (r) return r; }   // a shared row is read by many channels of the blocks in flight: written between blocks
  HIPOK(hipMemcpy(b.resp + (size_t)row0 * b.P, resp, sizeof(float2) * (size_t)n * b.P, hipMemcpyHostToDevice));
  return 0;
}
int chz_bank_set_shifts(chz_engine* e, int bank, int ch0, int n, const int* shifts) {
  BANK_CHECK(e, bank, ch0, n);
  if (!shifts) return fail(-1, "null argument");
  Bank& b = e->banks[(size_t)bank];
  for (int i = 0; i < n; i++) {
    ChanDescH h = make_chan_desc(e->in_type, e->bins, b.P, shifts[i]);
    ChanDesc& d = b.desc_h[(size_t)(ch0 + i)];
    d.t0 = h.t0; d.cnt = h.cnt; d.src0 = h.src0; d.dir = h.dir; d.conj = h.conj; d.wrap = h.wrap; d.shift = shifts[i];
  }
  return after_edit(e, b, ch0, n);
}
// The tuning half of downconvert() (src/radio.c:1440-1441,1479-1497): bin shift for the gather plus the fine
// oscillator.  Takes effect at block `job`; blocks before it must already have been enqueued.
int chz_bank_set_tuning(chz_engine* e, int bank, unsigned job, int ch0, int n, const int* shifts,
                        const double* freq, const double* rate) {
  BANK_CHECK(e, bank, ch0, n);
  if (!shifts || !freq) return fail(-1, "null argument");
  Bank& b = e->banks[(size_t)bank];
  if (b.out_real) return fail(-1, "fine tuning applies to COMPLEX-output banks");
  if (e->M < 2) return fail(-1, "impulse length %d has no overlap factor", e->M);
  const int V = 1 + e->L / (e->M - 1);
  if (V >= (1 << 26)) return fail(-1, "overlap factor %d: the block phase correction is exact up to 2^26", V);
  HIPOK(hipSetDevice(e->device));
  for (int i = 0; i < n; i++)
    if (!std::isfinite(freq[i]) || (rate && !std::isfinite(rate[i]))) return fail(-1, "non-finite tuning for channel %d", ch0 + i);
  if (!b.fine) {
    { int r = sync_all(e); if (r) return r; }     // the kernel variant changes: one-time switch
    b.fine_h.assign((size_t)b.cap, FineHost());
    b.fine_dh.assign((size_t)b.cap, FineDesc{0.0, 0.0, 0.0, 0u, 0, 1, 0, 0, 0, 1.0, 0.0});
    HIPOK(hipMalloc((void**)&b.fine, sizeof(FineDesc) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipMemset(b.fine, 0, sizeof(FineDesc) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipMalloc((void**)&b.power, sizeof(double) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipMemset(b.power, 0, sizeof(double) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipDeviceSynchronize());     // null-stream memsets vs the engine's non-blocking streams
    drop_graph(e);
  }
  for (int i = 0; i < n; i++) {
    ChanDescH h = make_chan_desc(e->in_type, e->bins, b.P, shifts[i]);
    ChanDesc& d = b.desc_h[(size_t)(ch0 + i)];
    d.t0 = h.t0; d.cnt = h.cnt; d.src0 = h.src0; d.dir = h.dir; d.conj = h.conj; d.wrap = h.wrap; d.shift = shifts[i];
    FineHost& fh = b.fine_h[(size_t)(ch0 + i)];
    fine_retune(fh, job, b.olen, V, shifts[i], freq[i], rate ? rate[i] : 0.0);
    b.fine_dh[(size_t)(ch0 + i)] = fine_desc(fh, V, b.g.any ? 0 : b.g.r.r1);
  }
  return after_edit(e, b, ch0, n);
}
// estimate_noise() (src/radio.c:1783-1866) for every channel of the bank right after its channel kernel;
// samprate = front-end sample rate in Hz (Frontend.samprate, :1865), 0 switches it off again
int chz_bank_enable_noise(chz_engine* e, int bank, double samprate) {
  BANK_CHECK(e, bank, 0, 0);
  Bank& b = e->banks[(size_t)bank];
  if (!(samprate >= 0.0)) return fail(-1, "bad sample rate");
  const int sb = b.out_real ? b.P / 2 + 1 : b.P;     // slave->bins (src/radio.c:1794)
  const int nb = sb < 1000 ? 1000 : sb;
  if (samprate > 0.0 && nb > e->bins) return fail(-1, "master has %d bins, fewer than the %d-bin noise window", e->bins, nb);
  if (samprate > 0.0 && nb > 2048) return fail(-3, "no noise kernel compiled for a %d-bin window", nb);
  HIPOK(hipSetDevice(e->device));
  { int r = sync_all(e); if (r) return r; }
  if (!b.n0 && samprate > 0.0) {
    HIPOK(hipMalloc((void**)&b.n0, sizeof(double) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipMemset(b.n0, 0, sizeof(double) * (size_t)CHZ_ND * b.cap));
    if (!b.noise_hint && e->opt_noise_hint) {
      HIPOK(hipMalloc((void**)&b.noise_hint, sizeof(unsigned) * (size_t)b.cap));
      HIPOK(hipMemset(b.noise_hint, 0, sizeof(unsigned) * (size_t)b.cap));
    }
    HIPOK(hipDeviceSynchronize());
  }
  if (samprate > 0.0 && !e->energy[0] && e->noise_energy != 0 && (b.cap >= 16384 || e->noise_energy == 1)) {
    for (int i = 0; i < CHZ_ND; i++) HIPOK(hipMalloc((void**)&e->energy[i], sizeof(float) * spec_energy_floats(e->bins)));
    HIPOK(hipDeviceSynchronize());
  }
  b.noise_samprate = samprate;
  drop_graph(e);
  return 0;
}
static int read_doubles(chz_engine* e, int bank, const double* base, int slot, int ch0, int n, double* host, bool wait, const char* what) {
  if (slot < 0 || slot >= CHZ_ND || !host) return fail(-1, "bad argument");
  Bank& b = e->banks[(size_t)bank];
  if (!base) return fail(-1, "%s", what);
  hipStream_t st = slot_stream(e, slot);
  HIPOK(hipMemcpyAsync(host, base + (size_t)slot * b.cap + ch0, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
  if (wait) HIPOK(hipStreamSynchronize(st));
  return 0;
}
int chz_bank_read_noise(chz_engine* e, int bank, int slot, int ch0, int n, double* host) {
  BANK_CHECK(e, bank, ch0, n);
  return read_doubles(e, bank, e->banks[(size_t)bank].n0, slot, ch0, n, host, true, "noise estimation is off: call chz_bank_enable_noise first");
}
int chz_bank_read_noise_async(chz_engine* e, int bank, int slot, int ch0, int n, double* host) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, ch0, n);
  return read_doubles(e, bank, e->banks[(size_t)bank].n0, slot, ch0, n, host, false, "noise estimation is off: call chz_bank_enable_noise first");
}
// chan->sig.bb_power of channels [ch0, ch0+n) for the block last executed on `slot` (src/radio.c:1516-1520)
int chz_bank_read_power(chz_engine* e, int bank, int slot, int ch0, int n, double* host) {
  BANK_CHECK(e, bank, ch0, n);
  return read_doubles(e, bank, e->banks[(size_t)bank].power, slot, ch0, n, host, true, "bank has no tuning: call chz_bank_set_tuning first");
}
int chz_bank_read_power_async(chz_engine* e, int bank, int slot, int ch0, int n, double* host) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, ch0, n);
  return read_doubles(e, bank, e->banks[(size_t)bank].power, slot, ch0, n, host, false, "bank has no tuning: call chz_bank_set_tuning first");
}
// slave->isb (src/filter.c:895-909): unpack LSB/USB to I/Q after the gather; flags != 0 switch it on per channel
int chz_bank_set_isb(chz_engine* e, int bank, int ch0, int n, const unsigned char* flags) {
  BANK_CHECK(e, bank, ch0, n);
  if (!flags) return fail(-1, "null argument");
  Bank& b = e->banks[(size_t)bank];
  if (b.out_real) return fail(-1, "ISB unpacking applies to COMPLEX-output banks");
  HIPOK(hipSetDevice(e->device));
  if (!b.isb) {
    bool any = false;
    for (int i = 0; i < n; i++) any = any || flags[i] != 0;
    if (!any) return 0;                          // nothing to switch on: keep the plain kernel variant
    { int r = sync_all(e); if (r) return r; }   // the kernel variant changes: one-time switch
    b.isb_h.assign((size_t)b.cap, 0);
    HIPOK(hipMalloc((void**)&b.isb, (size_t)CHZ_ND * b.cap));
    HIPOK(hipMemset(b.isb, 0, (size_t)CHZ_ND * b.cap));
    HIPOK(hipDeviceSynchronize());
    drop_graph(e);
  }
  for (int i = 0; i < n; i++) b.isb_h[(size_t)(ch0 + i)] = flags[i] ? 1 : 0;
  return after_edit(e, b, ch0, n);
}
// slave->beam with the weights set_filter_weights leaves in slave->alpha / ->beta (src/filter.c:756-775,922-929):
// ab = 4 doubles per channel (Re alpha, Im alpha, Re beta, Im beta), on = one flag byte per channel
int chz_bank_set_beam(chz_engine* e, int bank, int ch0, int n, const double* ab, const unsigned char* on) {
  BANK_CHECK(e, bank, ch0, n);
  if (!ab || !on) return fail(-1, "null argument");
  Bank& b = e->banks[(size_t)bank];
  if (b.out_real || e->in_type != CHZ_COMPLEX) return fail(-1, "beam mode combines I and Q of a COMPLEX master into COMPLEX-output channels");
  if (b.g.any) return fail(-3, "beam mode is compiled for the menu sizes only (P=%d runs through the one-workgroup-per-channel kernel)", b.P);
  HIPOK(hipSetDevice(e->device));
  if (!b.beam) {
    bool any = false;
    for (int i = 0; i < n; i++) any = any || on[i] != 0;
    if (!any) return 0;
    { int r = sync_all(e); if (r) return r; }
    b.beam_h.assign((size_t)b.cap, BeamDesc{0.0, 0.0, 0.0, 0.0, 0, 0});
    HIPOK(hipMalloc((void**)&b.beam, sizeof(BeamDesc) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipMemset(b.beam, 0, sizeof(BeamDesc) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipDeviceSynchronize());
    drop_graph(e);
  }
  for (int i = 0; i < n; i++) b.beam_h[(size_t)(ch0 + i)] = BeamDesc{ab[4 * i], ab[4 * i + 1], ab[4 * i + 2], ab[4 * i + 3], on[i] ? 1 : 0, 0};
  return after_edit(e, b, ch0, n);
}
// demod_linear()'s per-block work for channels [ch0, ch0+n) from block `job` on
int chz_bank_set_demod(chz_engine* e, int bank, unsigned job, int ch0, int n, const chz_demod_params* p, double blocktime) {
  BANK_CHECK(e, bank, ch0, n);
  if (!p || !(blocktime > 0)) return fail(-1, "bad argument");
  Bank& b = e->banks[(size_t)bank];
  if (b.out_real) return fail(-1, "the linear demodulator follows COMPLEX-output channels");
  if (!b.power || !b.n0 || b.noise_samprate <= 0.0)
    return fail(-1, "the demodulator needs the channel's bb_power and noise estimate: call chz_bank_set_tuning and chz_bank_enable_noise first");
  if (b.olen > 10240) return fail(-3, "the demodulator kernel keeps a block in LDS: at most 10240 samples per block (bank has %d)", b.olen);
  if (!b.pcm_stride) b.pcm_stride = b.olen * 8;
  bool need_ext = false, need_fm_mix = false;
  for (int i = 0; i < n; i++) {
    const chz_demod_params& q = p[i];
    if (q.channels < 0 || q.channels > 2 || q.encoding < CHZ_PCM_S16BE || q.encoding > CHZ_PCM_F16BE) return fail(-1, "bad demodulator parameters for channel %d", ch0 + i);
    if (q.channels * b.olen * ((q.encoding == CHZ_PCM_MULAW || q.encoding == CHZ_PCM_ALAW) ? 1 :
                               (q.encoding == CHZ_PCM_S16BE || q.encoding == CHZ_PCM_S16LE || q.encoding == CHZ_PCM_F16LE || q.encoding == CHZ_PCM_F16BE) ? 2 : 4) > b.pcm_stride)
      return fail(-1, "channel %d's PCM does not fit the bank's %d-byte rows (chz_bank_set_pcm_stride)", ch0 + i, b.pcm_stride);
    if (q.channels > 0 && !(q.samprate > 0 && q.headroom > 0 && q.bandwidth > 0 && std::isfinite(q.shift) && q.gain > 0))
      return fail(-1, "bad demodulator parameters for channel %d", ch0 + i);
    if (q.kind != CHZ_DEMOD_LINEAR && q.kind != CHZ_DEMOD_FM) return fail(-1, "unknown demodulator kind for channel %d", ch0 + i);
    if (q.channels > 1 && q.kind == CHZ_DEMOD_FM) return fail(-1, "the FM demodulator is mono (src/fm.c:37)");
    if (q.channels > 0 && q.pll_enable && q.kind == CHZ_DEMOD_LINEAR && !(q.pll_loop_bw > 0 && std::isfinite(q.pll_loop_bw)))
      return fail(-1, "channel %d: the PLL needs a loop bandwidth", ch0 + i);
    if (q.channels > 0 && !(q.tone_freq >= 0 && q.tone_freq < q.samprate / 2)) return fail(-1, "bad PL tone frequency for channel %d", ch0 + i);
    if (q.channels > 0 && (q.pll_enable || (q.kind == CHZ_DEMOD_FM && q.tone_freq != 0))) need_ext = true;
    if (q.channels > 0 && q.kind == CHZ_DEMOD_FM && !q.pll_enable) need_fm_mix = true;
  }
  // demod_fm_lanes keeps a block's baseband in the scratch block: only banks large enough to be served by it get one for that
  need_fm_mix = need_fm_mix && (e->demod_wave == 0 || (e->demod_wave < 0 && b.cap >= 65536));
  HIPOK(hipSetDevice(e->device));
  if (!e->tail) HIPOK(stream_create_masked(&e->tail, true));
  if (!b.dm_chan) {
    { int r = sync_all(e); if (r) return r; }        // one-time switch
    const size_t cap = (size_t)b.cap;
    HIPOK(hipMalloc((void**)&b.dm_chan, sizeof(DemodChan) * cap)); HIPOK(hipMemset(b.dm_chan, 0, sizeof(DemodChan) * cap));
    HIPOK(hipMalloc((void**)&b.dm_state, sizeof(DemodState) * cap)); HIPOK(hipMemset(b.dm_state, 0, sizeof(DemodState) * cap));
    HIPOK(hipMalloc((void**)&b.dm_status, sizeof(DemodStatus) * cap * CHZ_ND)); HIPOK(hipMemset(b.dm_status, 0, sizeof(DemodStatus) * cap * CHZ_ND));
    HIPOK(hipMalloc((void**)&b.dm_flags, cap * CHZ_ND)); HIPOK(hipMemset(b.dm_flags, 0, cap * CHZ_ND));
    HIPOK(hipMalloc((void**)&b.dm_pcm, (size_t)b.pcm_stride * cap * CHZ_ND)); HIPOK(hipMemset(b.dm_pcm, 0, (size_t)b.pcm_stride * cap * CHZ_ND));
    for (int s = 0; s < CHZ_ND; s++) {
      HIPOK(hipEventCreateWithFlags(&b.ev_bank[s], hipEventDisableTiming));
      HIPOK(hipEventCreateWithFlags(&b.ev_tail[s], hipEventDisableTiming));
      HIPOK(hipEventCreateWithFlags(&b.ev_pcm[s], hipEventDisableTiming));
      HIPOK(hipEventCreateWithFlags(&b.ev_pcmgo[s], hipEventDisableTiming));
    }
    HIPOK(hipDeviceSynchronize());
    DemodChan z; memset(&z, 0, sizeof z);
    b.dm_chan_h.assign(cap, z);
    b.dm_osc.assign(cap, Bank::OscHost());
    drop_graph(e);
  }
  HIPOK(hipStreamSynchronize(e->tail));      // the demodulator stream only: blocks already handed to it keep their parameters
  if (need_ext && !b.dm_ext) {
    // every channel gets the record init_pll() / the tone set-up would leave, whether or not it uses it yet
    std::vector<DemodExt> fresh((size_t)b.cap, demod_ext_init());
    HIPOK(hipMalloc((void**)&b.dm_ext, sizeof(DemodExt) * (size_t)b.cap));
    HIPOK(hipMemcpy(b.dm_ext, fresh.data(), sizeof(DemodExt) * (size_t)b.cap, hipMemcpyHostToDevice));
    drop_graph(e);
  }
  if (!b.agc_peak && (e->demod_wave == 0 || (e->demod_wave < 0 && b.cap >= 65536)) && !(CHZ_XENV("CHZ_AGC_PEAK") && CHZ_XENV("CHZ_AGC_PEAK")[0] == '0')) {
    HIPOK(hipMalloc((void**)&b.agc_peak, sizeof(double) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipMemset(b.agc_peak, 0, sizeof(double) * (size_t)CHZ_ND * b.cap));
    HIPOK(hipDeviceSynchronize());
    drop_graph(e);
  }
  if ((need_ext || need_fm_mix) && !b.dm_mix && !e->opt_pll_lane0) {      // (A/B knob: round 2's one-lane-per-channel loops inside the demodulator kernel)
    HIPOK(hipMalloc((void**)&b.dm_mix, sizeof(float2) * (size_t)b.cap * b.olen));
    HIPOK(hipMemset(b.dm_mix, 0, sizeof(float2) * (size_t)b.cap * b.olen));
    HIPOK(hipDeviceSynchronize());
    drop_graph(e);
  }
  std::vector<DemodState> init; std::vector<int> init_ch;
  for (int i = 0; i < n; i++) {
    const chz_demod_params& q = p[i];
    DemodChan& c = b.dm_chan_h[(size_t)(ch0 + i)];
    const bool was_on = c.on != 0;
    if (q.channels == 0) { if (was_on) b.dm_on--; c.on = 0; continue; }
    c.channels = q.channels; c.env = q.env != 0; c.agc = q.agc != 0; c.encoding = q.encoding; c.snr_squelch = q.snr_squelch != 0;
    c.squelch_tail = q.squelch_tail; c.tuned = q.tuned != 0; c.on = 1;
    c.samprate = q.samprate; c.headroom = q.headroom; c.threshold = q.threshold; c.recovery_rate = q.recovery_rate; c.hangtime = q.hangtime;
    c.recov_ps = std::pow(q.recovery_rate, 1.0 / q.samprate);
    c.dc_alpha = q.dc_alpha; c.bandwidth = q.bandwidth; c.squelch_open = q.squelch_open; c.squelch_close = q.squelch_close;
    if (q.kind == CHZ_DEMOD_FM) {                       // demod_fm()'s defaults for thresholds left unset (src/fm.c:38-41)
      if (!std::isfinite(c.squelch_open) || c.squelch_open == 0) c.squelch_open = 6.3;
      if (!std::isfinite(c.squelch_close) || c.squelch_close == 0) c.squelch_close = 4;
    }
    c.kind = q.kind; c.deemph_rate = q.deemph_rate; c.deemph_gain = q.deemph_gain; c.threshold_extend = q.threshold_extend;
    const bool pll_was = was_on && c.pll_enable != 0;
    const double tone_was = was_on ? c.tone_freq : 0.0;
    c.pll_enable = q.pll_enable != 0; c.pll_square = q.pll_square != 0; c.pll_loop_bw = q.pll_loop_bw;
    demod_tone_consts(q.kind == CHZ_DEMOD_FM ? q.tone_freq : 0.0, q.samprate, c);
    if (b.dm_ext) {
      DemodExt x;
      if (!was_on) {                                    // a new demodulator: init_pll(), tone squelch muted
        x = demod_ext_init();
        HIPOK(hipMemcpy(b.dm_ext + ch0 + i, &x, sizeof x, hipMemcpyHostToDevice));
      } else if ((c.pll_enable && !pll_was && c.kind == CHZ_DEMOD_LINEAR) || c.tone_freq != tone_was) {
        HIPOK(hipMemcpy(&x, b.dm_ext + ch0 + i, sizeof x, hipMemcpyDeviceToHost));
        if (c.pll_enable && !pll_was && c.kind == CHZ_DEMOD_LINEAR) {
          // what the blocks that ran with the PLL off left behind (src/linear.c:150-154); the loop itself keeps its state
          x.pll.lock = 0; x.pll.lock_count = -(int)std::lrint(0.5 * (int)q.samprate); x.pll_rotations = 0;
        }
        if (c.tone_freq != tone_was) { x.g_s0 = x.g_s1 = 0.0; x.pl_sample_count = 0; x.old_pl_phase = 0.0; x.tone_mute = 1; x.tone_deviation = 0.0; }
        HIPOK(hipMemcpy(b.dm_ext + ch0 + i, &x, sizeof x, hipMemcpyHostToDevice));
      }
    }
    // chan->shift: set_osc() keeps the phase when the frequency changes (src/osc.c:28-47); an oscillator at 0 Hz is not stepped (src/linear.c:170)
    Bank::OscHost& o = b.dm_osc[(size_t)(ch0 + i)];
    const double f = q.shift / q.samprate;
    if (!o.init) { o.init = true; o.freq = 0.0; o.phase0 = 0.0; o.job0 = job; }
    if (f != o.freq) {
      const double g = (double)(job - o.job0) * (double)b.olen;
      double hi = g * o.freq, lo = std::fma(g, o.freq, -hi);
      hi -= std::rint(hi);
      o.phase0 = frac1(o.phase0 + hi + lo); o.job0 = job; o.freq = f;
    }
    c.osc_phase0 = o.phase0; c.osc_freq = o.freq; c.osc_job0 = o.job0;
    if (!was_on) {
      b.dm_on++;
      DemodState st; memset(&st, 0, sizeof st);
      st.gain = q.gain; st.am_dc = 0.0; st.n0 = std::nan(""); st.hangcount = 0;
      st.squelch_state = c.kind == CHZ_DEMOD_FM ? 0 : ((!c.pll_enable && !c.snr_squelch) ? c.squelch_tail + 4 : 0);   // src/fm.c:58, src/linear.c:46
      st.squelch_open = 1;                                                              // src/linear.c:47
      init.push_back(st); init_ch.push_back(ch0 + i);
    }
  }
  b.dm_pll_lin = 0; b.dm_fm_pll = 0; b.dm_fm_tone = 0; b.dm_lin = 0; b.dm_fm = 0; b.dm_fm_nopll = 0;
  for (const DemodChan& dc : b.dm_chan_h) {
    if (!dc.on) continue;
    b.dm_lin += dc.kind == CHZ_DEMOD_LINEAR ? 1 : 0; b.dm_fm += dc.kind == CHZ_DEMOD_FM ? 1 : 0;
    b.dm_fm_nopll += (dc.kind == CHZ_DEMOD_FM && !dc.pll_enable) ? 1 : 0;
    b.dm_pll_lin += (dc.kind == CHZ_DEMOD_LINEAR && dc.pll_enable) ? 1 : 0;
    b.dm_fm_pll += (dc.kind == CHZ_DEMOD_FM && dc.pll_enable) ? 1 : 0;
    b.dm_fm_tone += (dc.kind == CHZ_DEMOD_FM && dc.tone_freq != 0) ? 1 : 0;
  }
  HIPOK(hipMemcpy(b.dm_chan + ch0, b.dm_chan_h.data() + ch0, sizeof(DemodChan) * (size_t)n, hipMemcpyHostToDevice));
  for (size_t k = 0; k < init.size(); k++)
    HIPOK(hipMemcpy(b.dm_state + init_ch[k], &init[k], sizeof(DemodState), hipMemcpyHostToDevice));
  b.dm_blocktime = blocktime;
  return 0;
}
// Demodulate blocks that did not come out of this bank's channel kernel: radiod runs a channel's samples through its private
// second filter (filter2, src/radio.c:1572-1594 -- chz_mini_* here) BEFORE the demodulator sees them, so the caller must be able
// to hand the demodulator stage its input.  chz_bank_write_block puts n channels' olen complex samples, bb_power and noise
// estimates into `slot`; chz_bank_demod runs the demodulators of the whole bank over what the slot holds, as block `job`.
int chz_bank_write_block(chz_engine* e, int bank, int slot, int ch0, int n, const float* samples, const double* bb_power, const double* n0) {
  BANK_CHECK(e, bank, ch0, n);
  if (slot < 0 || slot >= CHZ_ND || !samples) return fail(-1, "bad argument");
  Bank& b = e->banks[(size_t)bank];
  if (b.out_real) return fail(-1, "the demodulators follow COMPLEX-output channels");
  if (!b.dm_chan) return fail(-1, "bank has no demodulator: call chz_bank_set_demod first");
  HIPOK(hipSetDevice(e->device));
  const size_t so = (size_t)slot * b.cap + ch0;
  HIPOK(hipMemcpyAsync(bank_out(b, slot) + (size_t)ch0 * b.olen, samples, sizeof(float2) * (size_t)n * b.olen, hipMemcpyHostToDevice, e->tail));
  if (bb_power) HIPOK(hipMemcpyAsync(b.power + so, bb_power, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, e->tail));
  if (n0) HIPOK(hipMemcpyAsync(b.n0 + so, n0, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, e->tail));
  HIPOK(hipStreamSynchronize(e->tail));            // the caller's buffers are free again
  return 0;
}
// a bank whose channels go through filter2 first: the channel kernel's output is not what the demodulators should see
int chz_bank_demod_auto(chz_engine* e, int bank, int on) {
  BANK_CHECK(e, bank, 0, 0);
  { int r = sync_all(e); if (r) return r; }
  e->banks[(size_t)bank].dm_auto = on != 0;
  drop_graph(e);
  return 0;
}
int chz_bank_demod(chz_engine* e, int bank, unsigned job, int slot) {
  BANK_CHECK(e, bank, 0, 0);
  if (slot < 0 || slot >= CHZ_ND) return fail(-1, "bad slot");
  Bank& b = e->banks[(size_t)bank];
  if (!b.dm_chan || b.dm_on <= 0) return fail(-1, "bank has no demodulator: call chz_bank_set_demod first");
  HIPOK(hipSetDevice(e->device));
  const size_t so = (size_t)slot * b.cap;
  DemodParams d{};
  d.in = bank_out(b, slot); d.power = b.power + so; d.n0 = b.n0 + so; d.chan = b.dm_chan; d.state = b.dm_state; d.ext = b.dm_ext;
  d.status = b.dm_status + so; d.flags = b.dm_flags + so; d.pcm = b.dm_pcm + so * (size_t)b.pcm_stride; d.ch0 = 0; d.nch = b.active; d.olen = b.olen;
  d.pcm_stride = b.pcm_stride; d.job = job; d.blocktime = b.dm_blocktime; d.power_alpha = 0.10;
  d.lin_pll = b.dm_pll_lin > 0; d.fm_pll = b.dm_fm_pll > 0; d.fm_tone = b.dm_fm_tone > 0;
    d.mix = (d.lin_pll || d.fm_pll || d.fm_tone) ? b.dm_mix : nullptr;
    demod_paths(e, b, d);
  if (b.agc_peak && d.lin_lanes) { d.agc_peak = b.agc_peak + so; d.peak_chan = 0; d.peak_pll = (d.lin_pll && d.mix != nullptr) ? 1 : 0; }
  if (b.pcm_copying[slot]) HIPOK(hipStreamWaitEvent(e->tail, b.ev_pcm[slot], 0));
  if (launch_demod(e->tail, d)) return fail(-4, "the demodulator kernel refuses blocks of %d samples", b.olen);
  HIPOK(hipGetLastError());
  return 0;
}
int chz_bank_pcm_stride(chz_engine* e, int bank) {
  BANK_CHECK(e, bank, 0, 0);
  const Bank& b = e->banks[(size_t)bank];
  return b.pcm_stride ? b.pcm_stride : b.olen * 8;
}
// rows of exactly the size the bank's encodings need (480 B for 12 kHz mono S16) make the device-to-host copy of a block's
// PCM one contiguous transfer of only the bytes that matter; before the first chz_bank_set_demod
int chz_bank_set_pcm_stride(chz_engine* e, int bank, int bytes) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, 0, 0);
  Bank& b = e->banks[(size_t)bank];
  if (b.dm_chan) return fail(-1, "the PCM row size is fixed once demodulators exist");
  if (bytes < b.olen || bytes > 8 * b.olen || (bytes & 3)) return fail(-1, "PCM rows hold %d..%d bytes, a multiple of 4", b.olen, 8 * b.olen);
  b.pcm_stride = bytes;
  return 0;
}
static int read_pcm(chz_engine* e, int bank, int slot, int ch0, int n, void* pcm, chz_demod_status* status, bool wait) {
  BANK_CHECK(e, bank, ch0, n);
  if (slot < 0 || slot >= CHZ_ND) return fail(-1, "bad slot");
  Bank& b = e->banks[(size_t)bank];
  if (!b.dm_chan) return fail(-1, "bank has no demodulator: call chz_bank_set_demod first");
  static_assert(sizeof(chz_demod_status) == sizeof(DemodStatus), "status layouts must agree");
  const size_t so = (size_t)slot * b.cap + ch0, stride = (size_t)b.pcm_stride;
  // a read of the same slot on the copy stream (chz_bank_read_pcm_flags_async) may still be travelling: this record must not hide it
  // from the next demodulator launch and from chz_bank_pcm_wait, so the tail stream takes it in first
  if (b.pcm_copying[slot]) HIPOK(hipStreamWaitEvent(e->tail, b.ev_pcm[slot], 0));
  if (pcm) HIPOK(hipMemcpyAsync(pcm, b.dm_pcm + so * stride, stride * (size_t)n, hipMemcpyDeviceToHost, e->tail));
  if (status) HIPOK(hipMemcpyAsync(status, b.dm_status + so, sizeof(DemodStatus) * (size_t)n, hipMemcpyDeviceToHost, e->tail));
  HIPOK(hipEventRecord(b.ev_pcm[slot], e->tail));
  if (wait) HIPOK(hipStreamSynchronize(e->tail));
  return 0;
}
// the per-block essentials only: PCM + one flag byte per channel (the full status record is 96 bytes; a host that ships
// audio reads it when somebody asks, not 50 times a second for every channel)
int chz_bank_read_pcm_flags_async(chz_engine* e, int bank, int slot, int ch0, int n, void* pcm, unsigned char* flags) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, ch0, n);
  if (slot < 0 || slot >= CHZ_ND) return fail(-1, "bad slot");
  Bank& b = e->banks[(size_t)bank];
  if (!b.dm_chan) return fail(-1, "bank has no demodulator: call chz_bank_set_demod first");
  const size_t so = (size_t)slot * b.cap + ch0, stride = (size_t)b.pcm_stride;
  if (!e->pcmcopy) HIPOK(stream_create_masked(&e->pcmcopy, true));      // (own hardware queue unless CHZ_OWN_QUEUES=0: see stream_create_masked)
  HIPOK(hipEventRecord(b.ev_pcmgo[slot], e->tail));                 // behind the slot's demodulator kernel (and whatever else is queued there)
  HIPOK(hipStreamWaitEvent(e->pcmcopy, b.ev_pcmgo[slot], 0));
  if (pcm) HIPOK(hipMemcpyAsync(pcm, b.dm_pcm + so * stride, stride * (size_t)n, hipMemcpyDeviceToHost, e->pcmcopy));
  if (flags) HIPOK(hipMemcpyAsync(flags, b.dm_flags + so, (size_t)n, hipMemcpyDeviceToHost, e->pcmcopy));
  HIPOK(hipEventRecord(b.ev_pcm[slot], e->pcmcopy));
  b.pcm_copying[slot] = true;
  return 0;
}
int chz_bank_pcm_wait(chz_engine* e, int bank, int slot) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, 0, 0);
  if (slot < 0 || slot >= CHZ_ND) return fail(-1, "bad slot");
  Bank& b = e->banks[(size_t)bank];
  if (!b.dm_chan) return fail(-1, "bank has no demodulator: call chz_bank_set_demod first");
  HIPOK(hipEventSynchronize(b.ev_pcm[slot]));
  return check_device_errors(e);
}
int chz_bank_read_pcm(chz_engine* e, int bank, int slot, int ch0, int n, void* pcm, chz_demod_status* status) {
  return read_pcm(e, bank, slot, ch0, n, pcm, status, true);
}
int chz_bank_read_pcm_async(chz_engine* e, int bank, int slot, int ch0, int n, void* pcm, chz_demod_status* status) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  return read_pcm(e, bank, slot, ch0, n, pcm, status, false);
}
int chz_bank_set_active(chz_engine* e, int bank, int n) {
  BANK_CHECK(e, bank, 0, n);
  if (e->banks[(size_t)bank].active != n) drop_graph(e);
  e->banks[(size_t)bank].active = n;
  return 0;
}
int chz_bank_execute(chz_engine* e, int bank, unsigned job) {
  BANK_CHECK(e, bank, 0, 0);
  HIPOK(hipSetDevice(e->device));
  int r = enqueue_bank(e, bank, job, nullptr);
  if (r) return r;
  HIPOK(hipGetLastError());
  return 0;
}
int chz_bank_execute_range(chz_engine* e, int bank, unsigned job, int ch0, int n) {
  BANK_CHECK(e, bank, ch0, n);
  HIPOK(hipSetDevice(e->device));
  int r = enqueue_bank(e, bank, job, nullptr, ch0, n, true);
  if (r) return r;
  HIPOK(hipGetLastError());
  return 0;
}
int chz_bank_destroy(chz_engine* e, int bank) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, 0, 0);
  Bank& b = e->banks[(size_t)bank];
  { int r = sync_all(e); if (r) return r; }
  drop_graph(e);
  free_bank(b);
  return 0;
}
int chz_bank_read_async(chz_engine* e, int bank, int slot, int ch0, int n, float* host) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, ch0, n);
  if (slot < 0 || slot >= CHZ_ND) return fail(-1, "bad slot");
  Bank& b = e->banks[(size_t)bank];
  if (n == 0) return 0;
  HIPOK(hipMemcpyAsync(host, bank_out_at(b, slot, ch0), bank_sample_bytes(b) * (size_t)n * b.olen,
                       hipMemcpyDeviceToHost, slot_stream(e, slot)));
  return 0;
}
int chz_spectrum_read_async(chz_engine* e, int slot, float* host) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e || !host || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  return copy_spectrum(e, slot, host, slot_stream(e, slot));
}
int chz_host_callback(chz_engine* e, int slot, void (*fn)(void*), void* arg) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e || !fn || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  HIPOK(hipLaunchHostFunc(slot_stream(e, slot), fn, arg));
  return 0;
}
int chz_slot_sync(chz_engine* e, int slot) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  if (!e || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  HIPOK(hipStreamSynchronize(slot_stream(e, slot)));
  return 0;
}
int chz_host_alloc(void** p, size_t bytes) {
  if (!p) return fail(-1, "null pointer");
  HIPOK(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocPortable));      // every device of the process DMAs out of / into it
  return 0;
}
void chz_host_free(void* p) { if (p) (void)hipHostFree(p); }
int chz_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return fail(-1, "bad argument");
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(-10, "hipHostRegister failed: %s", hipGetErrorString(e)); }
  return 0;
}
void chz_host_unregister(void* p) { if (p) (void)hipHostUnregister(p); }
int chz_bank_read(chz_engine* e, int bank, int ch0, int n, float* host) {
  if (e) { chz_exit::Scope _xd; if (_xd.ok) (void)hipSetDevice(e->device); }       // several engines of one process may sit on different devices (KA9Q_HIP_DEVICES)
  BANK_CHECK(e, bank, ch0, n);
  Bank& b = e->banks[(size_t)bank];
  hipStream_t st = slot_stream(e, b.last_slot);
  HIPOK(hipMemcpyAsync(host, bank_out_at(b, b.last_slot, ch0), bank_sample_bytes(b) * (size_t)n * b.olen, hipMemcpyDeviceToHost, st));
  HIPOK(hipStreamSynchronize(st));
  return 0;
}
int chz_bank_output_device(chz_engine* e, int bank, int slot, float** dev) {
  BANK_CHECK(e, bank, 0, 0);
  if (slot < 0 || slot >= CHZ_ND) return fail(-1, "bad slot");
  *dev = reinterpret_cast<float*>(bank_out(e->banks[(size_t)bank], slot));
  return 0;
}

static int enqueue_step(chz_engine* e, unsigned job, Instr* in, NotchTurn* turn = nullptr, int seq = 0, bool capture_first = false, bool capturing = false) {
  int r = enqueue_forward(e, job, in, turn, seq, capture_first, capturing);
  if (r) return r;
  bool demod = false;
  for (const Bank& b : e->banks) demod = demod || b.dm_on > 0;
  if (demod && turn) {                // what goes to the demodulator stream is issued in block order
    while (turn->next_tail.load(std::memory_order_acquire) != seq) {
      if (turn->abort.load(std::memory_order_relaxed)) return fail(-6, "another issuing thread failed");
      __builtin_ia32_pause();
    }
  }
  for (int b = 0; b < (int)e->banks.size() && !r; b++) r = enqueue_bank(e, b, job, in);
  if (demod && turn) {
    if (r) turn->abort.store(1, std::memory_order_relaxed);
    turn->next_tail.store(seq + 1, std::memory_order_release);
  }
  return r;
}

int chz_step(chz_engine* e, unsigned job) {
  if (!e) return fail(-1, "null engine");
  HIPOK(hipSetDevice(e->device));
  int r = enqueue_step(e, job, nullptr);
  if (r) return r;
  HIPOK(hipGetLastError());
  return 0;
}

// fork: every other lane's stream waits on an event recorded on lane 0 (inside a capture this
// pulls the lane into the capture); join: lane 0 waits for every other lane's tail.
static int lanes_fork(chz_engine* e, hipEvent_t ev) {
  if (e->nlanes < 2) return 0;
  HIPOK(hipEventRecord(ev, e->lanes[0].s));
  for (int i = 1; i < e->nlanes; i++) HIPOK(hipStreamWaitEvent(e->lanes[i].s, ev, 0));
  return 0;
}
static int lanes_join(chz_engine* e, hipEvent_t* evs) {
  for (int i = 1; i < e->nlanes; i++) {
    HIPOK(hipEventRecord(evs[i], e->lanes[i].s));
    HIPOK(hipStreamWaitEvent(e->lanes[0].s, evs[i], 0));
  }
  return 0;
}

static int issue_threads() {
  return options().enq_threads;
}

int chz_run_blocks(chz_engine* e, unsigned job0, int nblocks, int mode, int instrument, chz_timing* timing) {
  if (!e || nblocks < 0) return fail(-1, "bad argument");
  if (mode == 1)
    for (const Bank& b : e->banks)
      if (b.fine || b.dm_on) return fail(-5, "graph replay bakes the block number into the captured launches; fine-tuned and demodulated banks need eager mode");
  HIPOK(hipSetDevice(e->device));
  { int r = sync_all(e); if (r) return r; }
  e->input_pending = false;                       // everything written so far is visible to every lane now
  e->notch_have = false;                          // the device is idle: nothing to order the first block behind
  hipEvent_t t0 = e->ev_t0, t1 = e->ev_t1, fork_ev = e->ev_fork;
  hipEvent_t* join_ev = e->ev_join;
  Instr in; in.on = instrument != 0 && mode == 0;
  hipStream_t s0 = e->lanes[0].s;
  int done = 0, rc = 0;
  const auto host_t0 = std::chrono::steady_clock::now();
  if (mode == 1) {
    // one graph = whole ring cycles of blocks (a multiple of ND so slots and lanes line up too), at least graph_min_blocks
    // of them: a replay is one stream-ordered unit, so the pipeline drains once per replay
    int unit = e->ring_blocks;
    while (unit % CHZ_ND) unit += e->ring_blocks;
    int cycle = unit;
    while (cycle < e->graph_min_blocks && cycle + unit <= nblocks) cycle += unit;
    const unsigned phase = job0 % (unsigned)unit;
    if (!e->graph || e->graph_blocks != cycle || e->graph_job0 != phase) {
      drop_graph(e);
      for (Bank& b : e->banks) { int r = refresh_all_bulk(e, b); if (r) return r; }   // no descriptor copies inside the capture
      hipGraph_t g = nullptr;
      e->capture_blocks = cycle;
      HIPOK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
      rc = lanes_fork(e, fork_ev);
      for (int i = 0; i < cycle && !rc; i++) rc = enqueue_step(e, phase + (unsigned)i, nullptr, nullptr, i, i == 0, true);
      if (!rc) rc = lanes_join(e, join_ev);
      hipError_t ce = hipStreamEndCapture(s0, &g);
      e->capture_blocks = 0;
      e->notch_have = false;                      // events recorded inside a capture are not waitable outside it
      if (rc) { if (g) hipGraphDestroy(g); return rc; }
      if (ce != hipSuccess) return fail(-10, "graph capture failed: %s", hipGetErrorString(ce));
      HIPOK(hipGraphInstantiate(&e->graph, g, nullptr, nullptr, 0));
      hipGraphDestroy(g);
      e->graph_blocks = cycle; e->graph_job0 = phase;
    }
    for (Bank& b : e->banks) { int r = refresh_all_bulk(e, b); if (r) return r; }
    const bool graph_ticket = e->n_notch > 0 && e->nlanes > 1 && e->notch_order == 0 && !e->graph_notch_event;
    if (graph_ticket)            // captured tickets count from here (the device is idle: sync_all above)
      HIPOK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->notch_ver + 2), (int)e->notch_tickets, 1, s0));
    HIPOK(hipEventRecord(t0, s0));
    while (nblocks - done >= cycle) {
      HIPOK(hipGraphLaunch(e->graph, s0)); done += cycle;
      if (graph_ticket) e->notch_tickets += (unsigned)cycle;
    }
    if (done < nblocks) {
      if ((rc = lanes_fork(e, fork_ev))) return rc;
      for (; done < nblocks; done++) if ((rc = enqueue_step(e, job0 + (unsigned)done, nullptr))) return rc;
      if ((rc = lanes_join(e, join_ev))) return rc;
    }
  } else {
    HIPOK(hipEventRecord(t0, s0));
    if (!in.on && (rc = lanes_fork(e, fork_ev))) return rc;
#if CHZ_FWD_BATCH
    {   // EXPERIMENT: CHZ_FWD_BATCH_N = 2 or 4 blocks per launch (forward transform only: every bank must be idle); 4 / N batches in flight
      static const int B = [] { const char* v = CHZ_XENV("CHZ_FWD_BATCH_N"); const int k = v ? atoi(v) : 0; return (k == 2 || k == 4) ? k : 0; }();
      bool idle = true;
      for (const Bank& bk : e->banks) idle = idle && bk.active == 0;
      if (B && idle && !in.on && job0 % (unsigned)B == 0) {
        for (; done + B <= nblocks; done += B) {
          const unsigned job = job0 + (unsigned)done;
          if ((rc = enqueue_forward_batch(e, job, B, e->lanes[(job / (unsigned)B) % (unsigned)(4 / B)].s))) return rc;
        }
      }
    }
#endif
    // Blocks of different lanes are independent launch sequences.  A single host thread issues ~5 launches per block at
    // ~3 us each, which bounds the small configurations -- so the lanes are split over CHZ_ENQ_THREADS host threads
    // (default 2; 1 = issue from the caller only): the caller takes its share, persistent issuers take the rest.
    const int T = issue_threads();
    if (T > 1 && !in.on && done == 0 && e->nlanes >= T && nblocks >= 2 * e->nlanes) {
      while ((int)e->issuers.size() < T - 1) {
        Issuer* is = new Issuer();
        const int dev = e->device;
        is->th = std::thread([is, dev] { (void)hipSetDevice(dev); is->loop(); });
        e->issuers.push_back(is);
      }
      NotchTurn turn;
      std::vector<int> rcs((size_t)T, 0);
      std::vector<std::string> errs((size_t)T);
      auto work = [&](int t) {
        for (int b = done; b < nblocks && !rcs[(size_t)t]; b++) {
          const unsigned job = job0 + (unsigned)b;
          if ((int)(job % (unsigned)e->nlanes) % T != t) continue;
          if ((rcs[(size_t)t] = enqueue_step(e, job, nullptr, &turn, b))) { errs[(size_t)t] = g_err; turn.abort.store(1); }
        }
      };
      for (int t = 1; t < T; t++) e->issuers[(size_t)(t - 1)]->post([&work, t] { work(t); });
      work(0);
      for (int t = 1; t < T; t++) e->issuers[(size_t)(t - 1)]->wait();
      for (size_t t = 0; t < rcs.size(); t++) if (rcs[t]) return fail(rcs[t], "%s", errs[t].c_str());   // the message lives in the worker's thread-local
      for (Bank& b : e->banks) b.last_slot = (int)((job0 + (unsigned)nblocks - 1u) % CHZ_ND);            // what a single issuer would have left
      done = nblocks;
    }
    for (; done < nblocks; done++) if ((rc = enqueue_step(e, job0 + (unsigned)done, &in))) return rc;
    if (!in.on && (rc = lanes_join(e, join_ev))) return rc;
  }
  if (e->tail && !in.on) {          // the demodulators of the last blocks belong to the run (ev_join[0] is free: lanes join from 1 up)
    HIPOK(hipEventRecord(join_ev[0], e->tail));
    HIPOK(hipStreamWaitEvent(s0, join_ev[0], 0));
  }
  HIPOK(hipEventRecord(t1, s0));
  const double enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  HIPOK(hipEventSynchronize(t1));
  HIPOK(hipGetLastError());
  if (timing) {
    memset(timing, 0, sizeof *timing);
    timing->enqueue_ms = enqueue_ms;
    float ms = 0; HIPOK(hipEventElapsedTime(&ms, t0, t1));
    timing->total_ms = ms; timing->blocks = nblocks;
    double* acc[7] = {&timing->first_ms, &timing->cols_ms, &timing->rows_ms, &timing->notch_ms, &timing->chan_ms, &timing->fix_ms, &timing->demod_ms};
    int* cnt[7] = {&timing->first_n, &timing->cols_n, &timing->rows_n, &timing->notch_n, &timing->chan_n, &timing->fix_n, &timing->demod_n};
    for (size_t i = 0; i < in.kind.size(); i++) {
      float k = 0; hipEventElapsedTime(&k, in.ev[2 * i], in.ev[2 * i + 1]);
      *acc[in.kind[i]] += k; *cnt[in.kind[i]] += 1;
    }
  }
  return 0;
}

}  // extern "C"

#include "chz_comm.inc"
#include "chz_mini.inc"
