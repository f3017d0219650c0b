/* filter_hip.c -- drop-in replacement for ka9q-radio's filter.o, host side in C.
 *
 * Exports exactly what src/filter.h:99-118 declares, on the struct layout callers
 * already compile against (include/ka9q_filter_abi.h pins it), and runs the hot path --
 * forward transform, notches, per-channel gather x response, backward transform -- on
 * the MI355X through the C ABI of libchz_hip.so (include/chz_engine.h).  radiod,
 * linear.c, fm.c and the front-end plugins stay untouched:
 *
 *   front end thread   writes floats through in.input_write_pointer, calls
 *                      write_rfilter(&in, NULL, n)             (src/rx888.c:800-826)
 *     -> execute_filter_input(): H2D of the L new samples, forward transform into slot
 *        job%ND, spectrum back to the host fdomain[] (estimate_noise reads it,
 *        src/radio.c:1801), then ONE batched launch per (P,olen) bank for every
 *        registered slave with its last-known shift (speculation), outputs staged in
 *        pinned memory; a stream callback publishes completed_jobs[] and broadcasts
 *        filter_cond exactly like run_fft did (src/filter.c:522-539).
 *   channel threads    execute_filter_output(slave, shift): same wait / lap / drop logic
 *                      as src/filter.c:680-707; if the staged result was computed with this
 *                      shift and this response it is copied out, otherwise that one channel is
 *                      re-run on the device (retunes are rare).
 *   set_filter()       Kaiser design on the host in float64 as before (src/filter.c:968-1045),
 *                      response uploaded to the bank.
 *
 *   KA9Q_HIP_DEVICES   "0,1,2,..." spreads the slaves of ONE master over the node's GPUs (north star: "channels shard
 *                      naturally across the 8 GPUs"; thread per channel inside one process, src/radio.c:996, every slave
 *                      reading one shared master, src/filter.c:704-712): one engine per listed device, every engine
 *                      takes the block's L new samples from the same pinned host ring and transforms them itself (no
 *                      collective in the data path; 518 MB/s of PCIe per device), slaves go to devices in creation
 *                      order, KA9Q_HIP_SHARD_CHANNELS (default 1024) at a time; a block is complete when every
 *                      device's completion callback has run.  filter.h callers see none of this.
 *
 * There is no CPU signal path here: REAL-output slaves and transform sizes the device
 * kernels are not compiled for fail loudly with -1.
 */
#define _GNU_SOURCE 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <time.h>
#include <unistd.h>
#include <errno.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sysexits.h>
#include <linux/futex.h>
#include "../../include/ka9q_filter_abi.h"
#include "../../include/chz_engine.h"

/* Environment: the shipped library reads the operator's variables only.  Tuning / A-B hooks of earlier rounds
   (KA9Q_HIP_WAKE, KA9Q_HIP_WAKE_SHARDS, KA9Q_HIP_BANK_CHANNELS, KA9Q_HIP_MINI, KA9Q_HIP_MINI_POOL) exist in -DCHZ_EXPERIMENTS builds only. */
#ifdef CHZ_EXPERIMENTS
#define XENV(name) getenv(name)
#else
#define XENV(name) ((const char *)NULL)
#endif

/* ---- globals the rest of radiod sets or reads (src/filter.c:40-48,476-479) ---- */
const char *Wisdom_file;
char const *System_wisdom_file = "/etc/fftw/wisdomf";
int N_worker_threads = 1;
int N_internal_threads = 1;
int FFTW_planning_level = 0;
int64_t Min_fft_time = INT64_MAX;
int64_t Max_fft_time = 0;
int64_t Avg_fft_time = 0;
int64_t Mean_dev = 0;

#define FREE(p) do { free(p); (p) = NULL; } while (0)

/* ------------------------------------------------------------------------- */
/* per-master / per-slave private state, hung off the opaque plan slots        */
/* ------------------------------------------------------------------------- */
struct hbank {
  int P, olen, id, cap, n;
  bool real;                        /* REAL-output slaves: olen floats per channel (else olen float complex) */
  struct filter_out **slaves;       /* [cap] */
  int *shift;                       /* [cap] shift the device descriptor currently holds */
  unsigned char *isb;               /* [cap] slave->isb as the device currently holds it */
  double *beam_ab;                  /* [cap][4] slave->alpha/beta as uploaded */
  unsigned char *beam_on;           /* [cap] slave->beam as uploaded */
  void *stage[ND];                  /* pinned [cap][olen] samples: staged outputs per job slot */
  int *stage_shift[ND];             /* [cap] shift each staged output was computed with */
  unsigned *stage_epoch[ND];        /* [cap] response epoch it was computed with (0 = invalid) */
  unsigned char *stage_isb[ND];     /* [cap] slave->isb flag it was computed with */
  unsigned stage_job[ND];
  int stage_n[ND];
  double *stage_n0[ND];             /* pinned [cap]: the device's estimate_noise() per channel (filter_hip_enable_noise), or NULL */
  bool noise_on;                    /* the device runs noise_est behind this bank's channel kernel */
};

struct done_note { struct mctx *ctx; unsigned job, seq; int shard; struct timespec t0; };

/* one device's share of a master (KA9Q_HIP_DEVICES): its engine and the banks of the slaves assigned to it */
#define MAX_SHARDS 16
struct shard {
  chz_engine *eng;
  int device;
  struct hbank *banks;
  int nbanks;
  int nslaves;                      /* slaves living on this device right now */
};

struct mctx {
  int kind;                         /* CTX_ENGINE; a small inline master carries a struct minictx instead (filter_hip_mini.h) */
  struct shard sh[MAX_SHARDS];      /* sh[0] is the primary: master->fdomain[] comes from it */
  int nsh;
  int shard_channels;               /* slaves per device before the next device is used (KA9Q_HIP_SHARD_CHANNELS) */
  /* how a block reaches the devices (KA9Q_HIP_EXCHANGE): "samples" (default) -- every device copies the L new samples out of the host
     ring and transforms them itself, no collective; "broadcast" -- the first device transforms, the spectrum slot travels to the others
     over xGMI as one in-process grouped ncclBroadcast on the slots' own streams (the north star's wording) */
  bool bcast;
  chz_comm *comm[MAX_SHARDS];
  struct filter_in *master;
  pthread_mutex_t lock;             /* serialises engine calls and bank bookkeeping */
  /* staged outputs: ~1000 channel threads read after every block, the launcher / bank edits write now and then.  One
     rwlock word shared by 1000 readers on 256 cores is a cache-line ping-pong, so the lock is split: a reader takes the
     shard its slave hashes to, a writer takes them all. */
#define STAGE_SHARDS 16
  struct { pthread_rwlock_t l; char pad[64 - sizeof(pthread_rwlock_t) % 64]; } stage_lock[STAGE_SHARDS];
  bool ring_pinned;                 /* host ring registered with the HIP runtime */
  bool host_spectrum;               /* copy every block's spectrum into master->fdomain[] (KA9Q_HIP_FDOMAIN, default on) */
  struct notch_state *notch_ptr;    /* list last uploaded to the device */
  int notch_n;
  int notch_bins[64];
  double notch_alpha[64];
  struct done_note note[ND][MAX_SHARDS];   /* one per job slot and device; execute_filter_input never has more than ND blocks in flight */
  unsigned pending[ND];             /* completion callbacks of the slot's block still to come (atomic): the LAST one publishes the block */
  /* KA9Q_HIP_INPUT_FULL=drop: the reference's producer never waits (src/filter.c:639-649 only queues a job); with ND blocks
     still in flight on the device this block is then NOT transformed -- its samples still travel so the overlap history stays
     whole -- and every slave gets zeros and a counted drop for it (src/filter.c:690-701), instead of the front-end thread
     (a USB callback, say) standing still.  Default: wait (no sample is ever lost to a momentary stall). */
  bool drop_when_full;
#define SKIP_RING 8
  uint64_t skipped[ND][SKIP_RING];  /* (1 << 32) | job of the blocks skipped in this slot lately, at [(job / ND) % SKIP_RING] */
  /* What the channel threads sleep on: a per-slot generation word, bumped after every publication in the slot (a completed
     block: completed_jobs[slot]; a skipped one: skipped[slot][..]).  A sleeper reads it BEFORE it looks at either, so a
     publication between its look and its sleep cannot be missed. */
#define WSHARDS_MAX 64
  /* Every slave sleeps on the word of its shard (sctx.shard).  1000+ sleepers on ONE futex word share one kernel hash bucket
     and one wake loop: measured on a 256-core host at 1024 channel threads (profiles/r03_dropin_wake.txt), one word 2.4 ms per block, 8 words 1.2, 32 words 0.98 (and the
     worst block 1.2 instead of 27 ms); helper threads that wake the shards side by side were slower again (an extra scheduling
     hop: 6 ms at 2000 threads against 2.5), as was passing the wake-up on from thread to thread in a tree (10 ms). */
  struct { unsigned v; char pad[60]; } gen[ND][WSHARDS_MAX];
  int wshards;                      /* words in use (KA9Q_HIP_WAKE_SHARDS, default 32) */
  int next_shard;
  /* what the DEVICE holds per slot: the last job enqueued there and the last one whose completion callback has run.  The slot's
     completion record, spectrum, staged outputs and host-ring window belong to the enqueued job until the two agree. */
  unsigned enq_seq[ND];             /* blocks enqueued in the slot so far (producer only) */
  unsigned dev_seq[ND];             /* of which the completion callback has run (futex word the producer sleeps on) */
  unsigned long n_skipped;          /* written under `lock`, read from other threads: atomic accesses */
  int consecutive_skips;            /* drop mode: skips in a row (bounded: the device ring holds 8 blocks, see execute_filter_input) */
  /* Failure policy (round 4).  The reference has none of its own -- a radiod whose front end or FFT dies exits and systemd
     restarts it (src/radio.c:398, src/main.c:202).  Here a failed device-side check (chz_engine_check: a notch ticket that ran
     out) or a HIP error while a block is being enqueued means the engine is gone: the blocks it still delivers are dropped
     (zeros + block_drops for every slave, as for a lapped block, src/filter.c:690-701), the engine is re-created ONCE with
     everything the slaves have registered (responses, shifts, ISB, beam weights, notch list, noise estimate) and the overlap
     history re-seated from the host ring, and the stream continues.  A second failure within RECOVERY_GRACE blocks of a
     recovery, or a re-creation that fails, ends the process with EX_SOFTWARE, like the reference, so the supervisor restarts it. */
#define RECOVERY_GRACE 500
  bool failed;                      /* set by the completion callback / the producer when the engine reports a failure (atomic) */
  unsigned recoveries, failed_blocks;   /* (atomic accesses: read lock-free by the channel threads) */
  unsigned last_recovery_job;
  unsigned engine_first_job;        /* the first block the CURRENT engine was handed: the spectra of earlier blocks died with its predecessor (atomic) */
  double noise_samprate;            /* > 0: banks run the device's estimate_noise() (filter_hip_enable_noise) */
  /* wake-up of the channel threads: the completion callback wakes wake_first of them (0 = all), every woken thread wakes
     wake_fan more (KA9Q_HIP_WAKE="first,fan"; default "0,2") */
  int wake_first, wake_fan;
  int wedged_ms;                    /* how long the producer waits for a device that completes nothing and reports nothing before it gives up (KA9Q_HIP_WEDGED_MS) */
  void *retired_mini;               /* an undecided small master became this engine in place: its old context (see create_input_impl) */
  int bank_cap0;                    /* channels a new bank starts with (KA9Q_HIP_BANK_CHANNELS, default 64; banks double as they fill) */
  /* KA9Q_HIP_PROFILE=1: where a block's host time goes, printed by delete_filter_input */
  bool profile;
  long long t_done_ns[ND];          /* when the slot's block was completed (CLOCK_MONOTONIC, ns; atomic) */
  unsigned long long prof_blocks, prof_input_ns, prof_wait_ns, prof_consume_sum_ns, prof_consume_n, prof_consume_max_ns, prof_hits, prof_misses, prof_dev_max_ns;
  unsigned long long prof_consume_max8_ns;   /* the same worst case over blocks 8.. only (the first blocks carry one-time costs: first touch of every
                                                slave's buffers, thread start-up, the runtime's first launches) */
  unsigned t_done_job[ND];
  unsigned warm_ran;                /* stream callbacks of create_filter_input's warm-up that have run (lives here, not on that call's stack: a late one must find it) */
  unsigned long long prof_first_dev_ns[8], prof_first_input_ns[8], prof_first_consume_ns[8];   /* blocks 0..7 one by one: enqueue -> callback; time inside execute_filter_input; the slowest slave's completion -> output in hand */
  unsigned prof_first_hits[8], prof_first_misses[8];
#define PROF_STAGES 8                /* lock, h2d, forward, spectrum read, bank edits, bank launch, bank reads, callback */
  unsigned long long prof_stage_ns[8][PROF_STAGES];
  /* channels whose staged result does not fit (retuned, new filter, just created) are re-run in batches: the first
     thread to miss becomes the leader and serves everybody who queued up meanwhile with one device round trip */
  pthread_mutex_t miss_lock;
  pthread_cond_t miss_cv;
  struct miss_req *miss_head, *miss_tail;
  bool miss_leader;
};

struct miss_req {
  struct filter_out *slave;
  int shift, slot;
  unsigned job;
  int rc;
  bool done, ranged;
  struct miss_req *next;
};

struct sctx {
  int kind;                         /* CTX_SLAVE: delete_filter_output must know what hangs off rev_plan when the master is already gone */
  int dev;                          /* index into mctx.sh: the device this slave lives on */
  int bank;                         /* index into that shard's banks */
  int idx;                          /* channel index inside the bank */
  unsigned epoch;                   /* bumped whenever the response changes */
  double n0;                        /* the device's noise estimate of the block this slave consumed last (NaN: none) */
  int shard;                        /* which of the master's wake words this slave sleeps on */
  /* The shift this slave will ask for, published by execute_filter_output BEFORE it waits for its block (atomic): the front-end
     thread launches the block's batch with it.  The API hands the shift over only with the request for a block, so without this
     the first block of every slave (and the first after every retune) was launched with the previous shift and had to be re-run:
     2000 re-runs in block 0 of a 2000-channel radiod (round 4: 70 ms and 279 lapped slave-blocks on a fresh box). */
  int want_shift;
  unsigned char want_valid;
};

struct hbank;
static void sync_notches(struct mctx *c, struct filter_in *f);
static void bank_free_host(struct hbank *b);
static void stage_wrlock(struct mctx *c);
static void stage_wrunlock(struct mctx *c);
static pthread_rwlock_t *stage_rdlock(struct mctx *c, const void *who);
static struct mctx *MCTX(struct filter_in *m) { return (struct mctx *)(void *)m->fwd_plan; }
static struct sctx *SCTX(struct filter_out *s) { return (struct sctx *)(void *)s->rev_plan; }
static void stage_wrlock(struct mctx *c) { for (int i = 0; i < STAGE_SHARDS; i++) pthread_rwlock_wrlock(&c->stage_lock[i].l); }
static void stage_wrunlock(struct mctx *c) { for (int i = STAGE_SHARDS - 1; i >= 0; i--) pthread_rwlock_unlock(&c->stage_lock[i].l); }
static pthread_rwlock_t *stage_rdlock(struct mctx *c, const void *who) {
  pthread_rwlock_t *l = &c->stage_lock[((uintptr_t)who >> 7) % STAGE_SHARDS].l;
  pthread_rwlock_rdlock(l);
  return l;
}

/* Block completion is published through completed_jobs[] and announced on a futex word (mctx.gen[]): the ~1000 channel
   threads of a big radiod all sleep on the same word and are released TOGETHER, instead of being
   handed filter_mutex one by one as pthread_cond_broadcast does (a 5-10 ms convoy per block at 1024
   threads).  Nothing outside filter.c touches filter_mutex / filter_cond / completed_jobs in the
   reference, so the waiting primitive is an implementation detail; the condvar is still signalled. */
static void futex_wait_u32(unsigned *addr, unsigned expected) {
  syscall(SYS_futex, addr, FUTEX_WAIT_PRIVATE, expected, NULL, NULL, 0);
}
static void futex_wake_all(unsigned *addr) { syscall(SYS_futex, addr, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0); }
static void futex_wake_n(unsigned *addr, int n) { syscall(SYS_futex, addr, FUTEX_WAKE_PRIVATE, n, NULL, NULL, 0); }

static void *lmalloc(size_t size) {           /* cache-line aligned, like src/filter.c:1163 */
  void *p = NULL;
  if (posix_memalign(&p, 64, size ? size : 64) != 0) return NULL;
  return p;
}

/* ------------------------------------------------------------------------- */
/* mirrored host ring: two adjacent mappings of one memfd, so an N-sample window */
/* is always contiguous (role of mirror_alloc, src/misc.c:635-682)              */
/* ------------------------------------------------------------------------- */
static size_t page_round(size_t n) {
  size_t pg = (size_t)sysconf(_SC_PAGESIZE);
  return (n + pg - 1) / pg * pg;
}
static void *ring_map(size_t size) {
  int fd = memfd_create("ka9q_hip_ring", 0);
  if (fd < 0) return NULL;
  if (ftruncate(fd, (off_t)size) != 0) { close(fd); return NULL; }
  unsigned char *base = mmap(NULL, 2 * size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (base == MAP_FAILED) { close(fd); return NULL; }
  if (mmap(base, size, PROT_READ | PROT_WRITE, MAP_FIXED | MAP_SHARED, fd, 0) != (void *)base ||
      mmap(base + size, size, PROT_READ | PROT_WRITE, MAP_FIXED | MAP_SHARED, fd, 0) != (void *)(base + size)) {
    munmap(base, 2 * size); close(fd); return NULL;
  }
  close(fd);
  return base;
}
static void ring_unmap(void **p, size_t size) { if (p && *p) { munmap(*p, 2 * size); *p = NULL; } }
static inline void ring_wrap(void **p, void *base, size_t size) {      /* src/misc.h:372-378 */
  if ((unsigned char *)*p >= (unsigned char *)base + size) *p = (unsigned char *)*p - size;
}

/* ------------------------------------------------------------------------- */
/* small host DFT for set_filter's P-point transform (P <= a few thousand)      */
/* ------------------------------------------------------------------------- */
static void host_dft_rec(int n, int stride, const double complex *in, double complex *out, const double complex *w, int wn) {
  if (n == 1) { out[0] = in[0]; return; }
  int p = 2;
  while (p * p <= n && n % p) p++;
  if (n % p) p = n;                       /* n itself is prime */
  int m = n / p;
  for (int r = 0; r < p; r++) host_dft_rec(m, stride * p, in + (size_t)r * stride, out + (size_t)r * m, w, wn);
  double complex t[64], *tt = p <= 64 ? t : malloc(sizeof(double complex) * (size_t)p);
  double complex u[64], *uu = p <= 64 ? u : malloc(sizeof(double complex) * (size_t)p);
  for (int k = 0; k < m; k++) {
    for (int r = 0; r < p; r++) tt[r] = out[k + (size_t)r * m] * w[((long)k * r * (wn / n)) % wn];
    for (int q = 0; q < p; q++) {
      double complex a = 0;
      for (int r = 0; r < p; r++) a += tt[r] * w[((long)q * r % p) * (wn / p)];
      uu[q] = a;
    }
    for (int q = 0; q < p; q++) out[k + (size_t)q * m] = uu[q];
  }
  if (p > 64) { free(tt); free(uu); }
}
/* forward unnormalised DFT, float complex in place via float64 */
static int host_dft_forward(int n, float complex *x) {
  double complex *w = malloc(sizeof(double complex) * (size_t)n * 3);
  if (!w) return -1;
  double complex *a = w + n, *b = a + n;
  for (int k = 0; k < n; k++) { double s, c; sincos(-2.0 * M_PI * k / n, &s, &c); w[k] = c + I * s; }
  for (int i = 0; i < n; i++) a[i] = x[i];
  host_dft_rec(n, 1, a, b, w, n);
  for (int i = 0; i < n; i++) x[i] = (float complex)b[i];
  free(w);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Kaiser window pieces (src/misc.c:416-427, src/window.c:217-254, src/misc.h:217) */
/* ------------------------------------------------------------------------- */
static double bessel_i0(double z) {
  double t = 0.25 * z * z, term = t, sum = 1 + t;
  for (int k = 2; k < 40; k++) { term *= t / ((double)k * k); sum += term; if (term < 1e-12 * sum) break; }
  return sum;
}
static void kaiser_f(float *w, int M, double beta) {
  double inv = 1.0 / bessel_i0(beta), pc = 2.0 / (M - 1);
  for (int n = 0; n < M / 2; n++) {
    double p = pc * n - 1;
    w[n] = w[M - 1 - n] = (float)(bessel_i0(beta * sqrt(1 - p * p)) * inv);
  }
  if (M & 1) w[(M - 1) / 2] = 1;
}
static double sinc_pi(double x) { return x == 0 ? 1.0 : sin(M_PI * x) / (M_PI * x); }
static double complex cis_pi(double x) {       /* e^{i pi x}, argument reduced in half-turns */
  double y = fmod(x, 2.0); if (y < 0) y += 2.0;
  double s, c; sincos(M_PI * (y > 1.0 ? y - 2.0 : y), &s, &c);
  return c + I * s;
}

#include "filter_hip_mini.h"

/* ------------------------------------------------------------------------- */
/* completion: runs on a HIP runtime thread after the block's work has drained   */
/* ------------------------------------------------------------------------- */
/* announce a publication in `slot` to every sleeper: bump the shard words, then wake them */
static void announce(struct mctx *c, int slot, bool everybody) {
  for (int k = 0; k < c->wshards; k++) __atomic_fetch_add(&c->gen[slot][k].v, 1u, __ATOMIC_RELEASE);
  for (int k = 0; k < c->wshards; k++) {
    if (c->wake_first > 0 && !everybody) futex_wake_n(&c->gen[slot][k].v, c->wake_first);
    else futex_wake_all(&c->gen[slot][k].v);
  }
}

static void block_done(void *arg) {
  struct done_note *n = arg;
  struct mctx *const c = n->ctx;
  struct filter_in *f = c->master;
  unsigned const job = n->job;
  int const slot = (int)(job % ND);
  /* a device of this master has reported a failed check: what the engines still deliver is not to be trusted (the notch recurrence
     of this block and of every later one was not applied) -- remembered here, acted on by whoever completes the block */
  if (chz_engine_check(c->sh[n->shard].eng) != 0) __atomic_store_n(&c->failed, true, __ATOMIC_RELEASE);
  /* one callback per device; the block is complete when the last of them has run (it alone goes on: the records of the slot are
     read in full BEFORE the job is published -- publishing releases the producer, which may reuse them) */
  if (__atomic_sub_fetch(&c->pending[slot], 1u, __ATOMIC_ACQ_REL) != 0) return;
  unsigned const seq = c->note[slot][0].seq;
  struct timespec const t0 = c->note[slot][0].t0;
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (__atomic_load_n(&c->failed, __ATOMIC_ACQUIRE)) {
    /* the block is announced like a skipped one: zeros and a counted drop for everybody */
    __atomic_fetch_add(&c->failed_blocks, 1u, __ATOMIC_RELAXED);
    __atomic_store_n(&c->skipped[slot][(job / ND) % SKIP_RING], ((uint64_t)1 << 32) | job, __ATOMIC_RELEASE);
    announce(c, slot, true);
    __atomic_store_n(&c->dev_seq[slot], seq, __ATOMIC_RELEASE);
    futex_wake_n(&c->dev_seq[slot], 1);
    return;
  }
  int64_t const ns = (t1.tv_nsec - t0.tv_nsec) + 1000000000LL * (t1.tv_sec - t0.tv_sec);
  if (c->profile) {
    /* (slaves of an older job on this slot may be reading these right now: tear-free, and before the block is announced) */
    __atomic_store_n(&c->t_done_ns[slot], (long long)t1.tv_sec * 1000000000LL + t1.tv_nsec, __ATOMIC_RELAXED);
    __atomic_store_n(&c->t_done_job[slot], job, __ATOMIC_RELAXED);
    if (job >= 8 && (unsigned long long)ns > c->prof_dev_max_ns) c->prof_dev_max_ns = (unsigned long long)ns;   /* the first blocks carry one-time set-up */
    if (job < 8) c->prof_first_dev_ns[job] = (unsigned long long)ns;
  }
  /* src/filter.c:544-552; before the block is announced: a caller that has seen its last block may read them (src/main.c:155-163) */
  if (ns > __atomic_load_n(&Max_fft_time, __ATOMIC_RELAXED)) __atomic_store_n(&Max_fft_time, ns, __ATOMIC_RELAXED);
  if (ns < __atomic_load_n(&Min_fft_time, __ATOMIC_RELAXED)) __atomic_store_n(&Min_fft_time, ns, __ATOMIC_RELAXED);
  int64_t const avg = __atomic_load_n(&Avg_fft_time, __ATOMIC_RELAXED), dev = ns - avg;
  __atomic_store_n(&Avg_fft_time, avg + (dev >> 4), __ATOMIC_RELAXED);
  int64_t const md = __atomic_load_n(&Mean_dev, __ATOMIC_RELAXED);
  __atomic_store_n(&Mean_dev, md + ((llabs(dev) - md) >> 4), __ATOMIC_RELAXED);
  pthread_mutex_lock(&f->filter_mutex);
  __atomic_store_n(&f->owner, pthread_self(), __ATOMIC_RELEASE);      /* read without the mutex by execute_filter_output */
  __atomic_store_n(&f->completed_jobs[slot], job, __ATOMIC_RELEASE);   /* src/filter.c:526-529 */
  pthread_cond_broadcast(&f->filter_cond);              /* src/filter.c:532-535 (kept; nobody in this build waits on it) */
  pthread_mutex_unlock(&f->filter_mutex);
  announce(c, slot, false);
  __atomic_store_n(&c->dev_seq[slot], seq, __ATOMIC_RELEASE);      /* the slot is the producer's again */
  futex_wake_n(&c->dev_seq[slot], 1);
}
static void warm_done(void *arg) { __atomic_fetch_add((unsigned *)arg, 1u, __ATOMIC_RELEASE); }

/* ------------------------------------------------------------------------- */
/* banks                                                                        */
/* ------------------------------------------------------------------------- */
static size_t bank_sample_bytes(const struct hbank *b) { return b->real ? sizeof(float) : sizeof(float complex); }
static int bank_create_dev(struct mctx *c, struct shard *sh, struct hbank *b, int cap);
static void bank_free_host(struct hbank *b) {
  for (int s = 0; s < ND; s++) { chz_host_free(b->stage[s]); b->stage[s] = NULL; chz_host_free(b->stage_n0[s]); b->stage_n0[s] = NULL; FREE(b->stage_shift[s]); FREE(b->stage_epoch[s]); FREE(b->stage_isb[s]); }
  FREE(b->slaves); FREE(b->shift); FREE(b->isb); FREE(b->beam_ab); FREE(b->beam_on);
}
static int bank_alloc_host(struct hbank *b, int cap) {
  b->cap = cap;
  b->slaves = calloc((size_t)cap, sizeof *b->slaves);
  b->shift = calloc((size_t)cap, sizeof *b->shift);
  b->isb = calloc((size_t)cap, 1);
  b->beam_ab = calloc((size_t)cap * 4, sizeof(double));
  b->beam_on = calloc((size_t)cap, 1);
  if (!b->slaves || !b->shift || !b->isb || !b->beam_ab || !b->beam_on) return -1;
  for (int s = 0; s < ND; s++) {
    void *p = NULL;
    if (chz_host_alloc(&p, bank_sample_bytes(b) * (size_t)cap * b->olen) != 0) return -1;
    b->stage[s] = p;
    b->stage_shift[s] = calloc((size_t)cap, sizeof(int));
    b->stage_epoch[s] = calloc((size_t)cap, sizeof(unsigned));
    b->stage_isb[s] = calloc((size_t)cap, 1);
    if (!b->stage_shift[s] || !b->stage_epoch[s] || !b->stage_isb[s]) return -1;
    if (chz_host_alloc(&p, sizeof(double) * (size_t)cap) != 0) return -1;
    b->stage_n0[s] = p;
    for (int k = 0; k < cap; k++) b->stage_n0[s][k] = NAN;
    b->stage_job[s] = UINT_MAX; b->stage_n[s] = 0;
  }
  return 0;
}
static int bank_create_dev(struct mctx *c, struct shard *sh, struct hbank *b, int cap) {
  int const id = b->real ? chz_bank_create_real(sh->eng, b->P, b->olen, cap) : chz_bank_create(sh->eng, b->P, b->olen, cap);
  b->noise_on = false;
  /* estimate_noise() on the device (filter_hip_enable_noise): sizes the noise kernel is not compiled for keep NaN */
  if (id >= 0 && c->noise_samprate > 0) b->noise_on = chz_bank_enable_noise(sh->eng, id, c->noise_samprate) == 0;
  return id;
}
/* A bank that has just been created (or re-created larger) runs its kernels once on every slot before a real block does: the
   first launch of the channel kernel of its size (and of the noise kernel), the descriptor staging of every slot, and one
   device-to-host copy into each of the staged-output images -- as the reference pays for planning inside create_filter_output
   (src/filter.c:359), so that the first block a new slave sees is an ordinary block.  Caller holds c->lock; `b` has its host side. */
static void bank_warm(struct shard *sh, struct hbank *b) {
  int const n = b->cap;              /* every row: the copies then walk the whole of each pinned image once (channels without a gather descriptor give zeros) */
  if (n < 1) return;
  for (int s = 0; s < ND; s++) {       /* (on an enqueue error: stop enqueueing, but never leave with copies into b->stage[] still in flight) */
    if (chz_bank_execute_range(sh->eng, b->id, (unsigned)s, 0, n) != 0) break;
    if (chz_bank_read_async(sh->eng, b->id, s, 0, n, (float *)b->stage[s]) != 0) break;
    if (b->noise_on && chz_bank_read_noise_async(sh->eng, b->id, s, 0, n, b->stage_n0[s]) != 0) break;
  }
  for (int s = 0; s < ND; s++) (void)chz_slot_sync(sh->eng, s);
  /* (a bank that already serves blocks -- filter_hip_enable_noise in mid-stream: what was staged is gone, its slaves re-run their block) */
  for (int s = 0; s < ND; s++) { memset(b->stage[s], 0, bank_sample_bytes(b) * (size_t)n * b->olen); for (int k = 0; k < n; k++) { b->stage_n0[s][k] = NAN; b->stage_epoch[s][k] = 0; } }
}
/* find (or create, or grow) the bank for (P, olen, output type) on one device; caller holds ctx->lock */
static int bank_for(struct mctx *c, struct shard *sh, int P, int olen, bool real) {
  for (int i = 0; i < sh->nbanks; i++) {
    struct hbank *b = &sh->banks[i];
    if (b->P != P || b->olen != olen || b->real != real) continue;
    if (b->n < b->cap) return i;
    /* grow: a new, larger device bank; move responses and shifts over */
    struct hbank nb = {.P = P, .olen = olen, .n = b->n, .real = real};
    nb.id = bank_create_dev(c, sh, &nb, b->cap * 2);
    if (nb.id < 0 || bank_alloc_host(&nb, b->cap * 2) != 0) {
      fprintf(stderr, "filter_hip: cannot grow bank: %s\n", chz_last_error());
      if (nb.id >= 0) chz_bank_destroy(sh->eng, nb.id);
      bank_free_host(&nb);
      return -1;
    }
    for (int k = 0; k < b->n; k++) {
      nb.slaves[k] = b->slaves[k]; nb.shift[k] = b->shift[k]; nb.isb[k] = b->isb[k];
      nb.beam_on[k] = 0;                                  /* re-uploaded by the next execute_filter_input if the slave is in beam mode */
      /* the slave's own thread may be inside set_filter right now: its swap frees the old response */
      pthread_mutex_lock(&nb.slaves[k]->response_mutex);
      if (nb.slaves[k]->response) chz_bank_set_responses(sh->eng, nb.id, k, 1, (const float *)nb.slaves[k]->response);
      pthread_mutex_unlock(&nb.slaves[k]->response_mutex);
    }
    chz_bank_set_shifts(sh->eng, nb.id, 0, nb.n, nb.shift);
    if (!real) chz_bank_set_isb(sh->eng, nb.id, 0, nb.n, nb.isb);
    bank_warm(sh, &nb);
    chz_bank_destroy(sh->eng, b->id);
    bank_free_host(b);
    *b = nb;
    return i;
  }
  struct hbank *nbanks = realloc(sh->banks, sizeof *nbanks * (size_t)(sh->nbanks + 1));
  if (!nbanks) return -1;
  sh->banks = nbanks;
  struct hbank *b = &sh->banks[sh->nbanks];
  memset(b, 0, sizeof *b);
  b->P = P; b->olen = olen; b->real = real;
  /* a master that will carry a thousand slaves doubles its way up from here; every doubling is a new device bank and new pinned images */
  int const cap0 = c->bank_cap0 > 0 ? c->bank_cap0 : 64;
  b->id = bank_create_dev(c, sh, b, cap0);
  if (b->id < 0) { fprintf(stderr, "filter_hip: %s\n", chz_last_error()); return -1; }
  if (bank_alloc_host(b, cap0) != 0) { chz_bank_destroy(sh->eng, b->id); bank_free_host(b); return -1; }
  { int const zero = 0; (void)chz_bank_set_shifts(sh->eng, b->id, 0, 1, &zero); }
  bank_warm(sh, b);
  return sh->nbanks++;
}

/* ------------------------------------------------------------------------- */
/* create / delete                                                               */
/* ------------------------------------------------------------------------- */
static void mctx_free(struct mctx *c) {          /* host side (and the communicators); the engines are destroyed by the caller first */
  for (int g = 0; g < c->nsh; g++) if (c->comm[g]) { chz_comm_destroy(c->comm[g]); c->comm[g] = NULL; }
  for (int g = 0; g < c->nsh; g++) {
    for (int i = 0; i < c->sh[g].nbanks; i++) bank_free_host(&c->sh[g].banks[i]);
    free(c->sh[g].banks);
  }
  free(c->retired_mini);
  pthread_mutex_destroy(&c->lock);
  for (int i = 0; i < STAGE_SHARDS; i++) pthread_rwlock_destroy(&c->stage_lock[i].l);
  pthread_mutex_destroy(&c->miss_lock);
  pthread_cond_destroy(&c->miss_cv);
  free(c);
}
/* KA9Q_HIP_DEVICES="0,1,2" (or the older KA9Q_HIP_DEVICE=n): the devices this master's slaves are spread over.  An entry may
   repeat ("0,0": two engines on one device -- how the sharded path is exercised on a one-GPU box). */
#define SHARD_MIN_POINTS 16384         /* a master of at most this many points -- a demodulator's private sub-filter (wfm's composite: 15,360), a small front end -- is not worth
                                          spreading: every listed device would copy and transform its every block for a handful of slaves.  It lives on the first device. */
static int device_list(struct mctx *c, int points) {
  const char *list = getenv("KA9Q_HIP_DEVICES");
  int n = 0;
  if (list && *list) {
    const char *q = list;
    while (*q && n < MAX_SHARDS) {
      char *end = NULL;
      long v = strtol(q, &end, 10);
      if (end == q || v < 0 || v > 1023) { fprintf(stderr, "create_filter_input: KA9Q_HIP_DEVICES=\"%s\" is not a list of device numbers\n", list); return -1; }
      c->sh[n++].device = (int)v;
      q = end;
      while (*q == ',' || *q == ' ') q++;
    }
    if (*q) { fprintf(stderr, "create_filter_input: KA9Q_HIP_DEVICES names more than %d devices\n", MAX_SHARDS); return -1; }
  }
  if (n == 0) { const char *dev = getenv("KA9Q_HIP_DEVICE"); c->sh[0].device = dev ? atoi(dev) : 0; n = 1; }
  if (n > 1 && points <= SHARD_MIN_POINTS) n = 1;
  return n;
}
/* Everything a block does, once, on zeros, before create_filter_input returns: the first launches of the plan's kernels on every
   lane, the first H2D copy out of the (pinned) host ring, the first D2H copies into fdomain[], the runtime's callback thread -- the
   reference pays for planning inside create_filter_input (src/filter.c:248,263), so block 0 of the stream is an ordinary block.
   The input ring is re-seated in front of job 0 afterwards (zero history, src/filter.c:244,259); the spectra of zeros are zeros. */
static void engines_warm(struct mctx *c, struct filter_in *f, const float *zeros) {
  unsigned want = 0;
  bool synced = true;
  __atomic_store_n(&c->warm_ran, 0u, __ATOMIC_RELEASE);
  for (int g = 0; g < c->nsh; g++) {
    chz_engine *e = c->sh[g].eng;
    const float *src = zeros ? zeros : (const float *)f->input_buffer;     /* zeros (a fresh ring is; a master promoted in place brings a block of them) */
    for (unsigned j = 0; j < ND; j++) {
      if ((g == 0 || !c->bcast) && (chz_input_write(e, src, f->ilen) != 0 || chz_forward(e, j) != 0)) break;
      if (g == 0 && c->host_spectrum && chz_spectrum_read_async(e, (int)j, (float *)f->fdomain[j]) != 0) break;
      if (chz_host_callback(e, (int)j, warm_done, &c->warm_ran) != 0) break;
      want++;
    }
  }
  if (c->bcast) {                     /* the collective's first use (RCCL sets its channels up then) is not block 0's business either */
    chz_engine *engs[MAX_SHARDS];
    for (int g = 0; g < c->nsh; g++) engs[g] = c->sh[g].eng;
    for (int j = 0; j < ND; j++)
      if (chz_spectrum_broadcast_local(engs, c->comm, c->nsh, j, 0) != 0) { fprintf(stderr, "create_filter_input: warm-up broadcast: %s\n", chz_last_error()); break; }
  }
  for (int g = 0; g < c->nsh; g++) {
    if (chz_sync(c->sh[g].eng) != 0 || chz_input_seek(c->sh[g].eng, 0, NULL) != 0) {
      fprintf(stderr, "create_filter_input: warm-up on device %d: %s\n", c->sh[g].device, chz_last_error());
      synced = false;
    }
  }
  /* (a stream callback may still be returning on its runtime thread after the stream has drained; after a FAILED sync nobody waits for
     callbacks that may never come -- the counter lives in the context, a late one does no harm) */
  for (int spin = 0; synced && spin < 2000 && __atomic_load_n(&c->warm_ran, __ATOMIC_ACQUIRE) != want; spin++) usleep(100);
}

static int create_input_impl(struct filter_in *master, int L, int M, enum filtertype in_type, bool force_engine);
int create_filter_input(struct filter_in *master, int const L, int const M, enum filtertype const in_type) {
  return create_input_impl(master, L, M, in_type, false);
}
/* force_engine: an UNDECIDED small COMPLEX master (filter_hip_mini.h: mini_wanted) becomes a full engine IN PLACE -- the caller's struct keeps its ring (the
   front end may already have written block 0 into it), its pointers, counters and flags; only what hangs off fwd_plan and the fdomain[] buffers change.
   Called with master->filter_mutex held, at job 0 (an undecided master has not run). */
static int create_input_impl(struct filter_in *master, int const L, int const M, enum filtertype const in_type, bool const force_engine) {
  if (master == NULL) return -1;
  if (!force_engine && master->init && master->ilen == L && master->impulse_length == M && in_type == master->in_type)
    return 0;                                                      /* src/filter.c:191-192 */
  if (in_type != REAL && in_type != COMPLEX) return -1;            /* src/filter.c:228-234 */
  if (L <= 0 || M <= 0) return -1;
  int const N = L + M - 1;
  int const bins = (in_type == COMPLEX) ? N : (N / 2 + 1);
  if (bins < 2) return -1;                                         /* src/filter.c:198-199 */

  if (!force_engine && master->init && master->fwd_plan && is_mini_master(master)) mini_free_input(master);
  if (!force_engine && master->init && master->fwd_plan) {         /* re-create with new geometry */
    struct mctx *old = MCTX(master);
    if (old->ring_pinned) chz_host_unregister(master->input_buffer);
    for (int g = 0; g < old->nsh; g++) chz_engine_destroy(old->sh[g].eng);
    mctx_free(old);
    master->fwd_plan = NULL;
    for (int i = 0; i < ND; i++) { chz_host_free(master->fdomain[i]); master->fdomain[i] = NULL; }
    ring_unmap(&master->input_buffer, master->input_buffer_size);
  }
  if (!force_engine && mini_wanted(L, M, in_type)) return mini_create_input(master, L, M);   /* radiod's filter2 and its like */
  struct mctx *c = calloc(1, sizeof *c);
  if (!c) return -1;
  c->kind = CTX_ENGINE;
  c->nsh = device_list(c, N);
  if (c->nsh < 1) { free(c); return -1; }
  for (int g = 0; g < c->nsh; g++)
    if (chz_engine_create(&c->sh[g].eng, L, M, in_type == REAL ? CHZ_REAL : CHZ_COMPLEX, c->sh[g].device, NULL, 0) != 0) {
      fprintf(stderr, "create_filter_input(L=%d M=%d, device %d): %s\n", L, M, c->sh[g].device, chz_last_error());
      for (int k = 0; k < g; k++) chz_engine_destroy(c->sh[k].eng);
      free(c);
      return -1;
    }
  c->master = master;
  { const char *ex = getenv("KA9Q_HIP_EXCHANGE");
    if (ex && strcmp(ex, "broadcast") == 0) c->bcast = true;
    else if (ex && *ex && strcmp(ex, "samples") != 0) fprintf(stderr, "create_filter_input: KA9Q_HIP_EXCHANGE=%s is neither \"samples\" nor \"broadcast\": using samples\n", ex); }
  if (c->bcast) {
    int devs[MAX_SHARDS];
    for (int g = 0; g < c->nsh; g++) devs[g] = c->sh[g].device;
    if (chz_comm_create_local(c->comm, c->nsh, devs) != 0) {
      /* the ladder (round 6): no in-process RCCL clique -- RCCL missing, a device listed twice, a time-out -- is no reason to come up
         without a front end: every device copies the samples and transforms them itself, as in the default mode (no collective) */
      fprintf(stderr, "create_filter_input: KA9Q_HIP_EXCHANGE=broadcast: %s -- falling back to KA9Q_HIP_EXCHANGE=samples\n", chz_last_error());
      c->bcast = false;
    }
  }
  c->shard_channels = 1024;                                       /* contiguous 1024-blocks */
  { const char *sc = getenv("KA9Q_HIP_SHARD_CHANNELS"); if (sc && atoi(sc) > 0) c->shard_channels = atoi(sc); }
  c->wedged_ms = 10000;
  { const char *wm = getenv("KA9Q_HIP_WEDGED_MS"); if (wm && atoi(wm) >= 500) c->wedged_ms = atoi(wm); }
  { const char *bc = XENV("KA9Q_HIP_BANK_CHANNELS"); if (bc && atoi(bc) > 0 && atoi(bc) <= 65536) c->bank_cap0 = atoi(bc); }
  /* master->fdomain[] is read by radiod's estimate_noise() (src/radio.c:1801) and by nothing else outside filter.c;
     a host that takes the noise estimate from the device (chz_bank_enable_noise) can switch the 13 MB per-block copy off */
  { const char *fd = getenv("KA9Q_HIP_FDOMAIN"); c->host_spectrum = !(fd && fd[0] == '0'); }
  { const char *fu = getenv("KA9Q_HIP_INPUT_FULL"); c->drop_when_full = fu && strcmp(fu, "drop") == 0; }
  c->wake_first = 0; c->wake_fan = 2;
  { const char *wk = XENV("KA9Q_HIP_WAKE"); int a = 0, b = 2; if (wk && sscanf(wk, "%d,%d", &a, &b) == 2 && a >= 0 && b >= 0) { c->wake_first = a; c->wake_fan = b; } }
  { const char *pf = getenv("KA9Q_HIP_PROFILE"); c->profile = pf && pf[0] == '1'; }
  c->wshards = 32;
  { const char *ws = XENV("KA9Q_HIP_WAKE_SHARDS"); if (ws) { int v = atoi(ws); if (v >= 1 && v <= WSHARDS_MAX) c->wshards = v; } }
  { const char *ns = getenv("KA9Q_HIP_NOISE_SAMPRATE"); if (ns && atof(ns) > 0) c->noise_samprate = atof(ns); }
  pthread_mutex_init(&c->lock, NULL);
  for (int i = 0; i < STAGE_SHARDS; i++) pthread_rwlock_init(&c->stage_lock[i].l, NULL);
  pthread_mutex_init(&c->miss_lock, NULL);
  pthread_cond_init(&c->miss_cv, NULL);
  size_t const ssz = in_type == COMPLEX ? sizeof(float complex) : sizeof(float);
  size_t const ring_bytes = page_round((size_t)ND * N * ssz);       /* src/filter.c:237,253 */
  void *ring = NULL;
  void *fd[ND] = {NULL, NULL, NULL, NULL};
  for (int i = 0; i < ND; i++)
    if (chz_host_alloc(&fd[i], sizeof(float complex) * (size_t)bins) != 0) { fprintf(stderr, "create_filter_input: %s\n", chz_last_error()); goto fail; }
  float *zeros = NULL;
  if (force_engine) {
    if (master->input_buffer == NULL || master->input_buffer_size != ring_bytes || (zeros = calloc((size_t)L, ssz)) == NULL) goto fail;
    ring = master->input_buffer;
  } else ring = ring_map(ring_bytes);
  if (!ring) { perror("create_filter_input: ring"); goto fail; }

  /* nothing below can fail: only now is the caller's struct touched */
  struct minictx *const old_mini = force_engine ? (struct minictx *)(void *)master->fwd_plan : NULL;
  void *old_fd[ND] = {NULL, NULL, NULL, NULL};
  if (!force_engine) { master->points = N; master->perform_inline = (N_worker_threads == 0); }   /* src/filter.c:205 */
  for (int i = 0; i < ND; i++) {
    if (force_engine) old_fd[i] = master->fdomain[i];
    memset(fd[i], 0, sizeof(float complex) * (size_t)bins);
    master->fdomain[i] = fd[i];                                    /* pinned: the device copies spectra here */
    if (!force_engine) master->completed_jobs[i] = UINT_MAX;      /* src/filter.c:214 */
  }
  if (!force_engine) { master->bins = bins; master->ilen = L; master->impulse_length = M; }     /* (in place: the same values, which other threads are reading) */
  if (!master->init) {
    pthread_mutex_init(&master->filter_mutex, NULL);
    pthread_cond_init(&master->filter_cond, NULL);
    master->init = true;
  }
  if (!force_engine) {
    master->owner = pthread_self();
    master->in_type = in_type;
    master->input_buffer_size = ring_bytes;
    master->input_buffer = ring;
    memset(master->input_buffer, 0, master->input_buffer_size);
  }
  /* both mappings of the ring, so a window that runs into the mirror is still DMA-able */
  c->ring_pinned = chz_host_register(master->input_buffer, 2 * master->input_buffer_size) == 0;
  if (force_engine) {
    /* (pointers, wcnt, next_jobnum = 0, sample_index, perform_inline, notches: the caller's, untouched) */
  } else if (in_type == COMPLEX) {                                 /* src/filter.c:243-246 */
    master->input_read_pointer.c = master->input_buffer;
    master->input_write_pointer.c = master->input_read_pointer.c + (M - 1);
    master->input_read_pointer.r = NULL; master->input_write_pointer.r = NULL;
  } else {                                                         /* src/filter.c:258-261 */
    master->input_read_pointer.r = master->input_buffer;
    master->input_write_pointer.r = master->input_read_pointer.r + (M - 1);
    master->input_read_pointer.c = NULL; master->input_write_pointer.c = NULL;
  }
  if (!force_engine) { master->wcnt = 0; master->next_jobnum = 0; }
  engines_warm(c, master, zeros);                                  /* (before the context is published: the warm-up blocks use the engines directly) */
  __atomic_store_n((void **)(void *)&master->fwd_plan, (void *)c, __ATOMIC_RELEASE);
  if (force_engine) {
    free(zeros);
    c->retired_mini = old_mini;            /* another thread may be looking at its kind word this instant (is_mini_master): it goes with the context, at delete */
    for (int i = 0; i < ND; i++) { free(old_fd[i]); futex_wake_all(&master->completed_jobs[i]); }      /* block clocks asleep on the old words look again */
  }
  return 0;

fail:
  for (int i = 0; i < ND; i++) chz_host_free(fd[i]);
  free(zeros);
  if (!force_engine) ring_unmap(&ring, ring_bytes);
  for (int g = 0; g < c->nsh; g++) chz_engine_destroy(c->sh[g].eng);
  mctx_free(c);
  return -1;
}

int delete_filter_input(struct filter_in *master) {
  if (master == NULL) return -1;
  if (is_mini_master(master)) {
    mini_free_input(master);
    if (master->init) { pthread_mutex_destroy(&master->filter_mutex); pthread_cond_destroy(&master->filter_cond); }
    memset(master, 0, sizeof *master);                             /* src/filter.c:940 */
    return 0;
  }
  if (master->fwd_plan) {
    struct mctx *c = MCTX(master);
    for (int g = 0; g < c->nsh; g++) chz_sync(c->sh[g].eng);
    if (c->profile && c->prof_blocks)
      fprintf(stderr, "filter_hip profile: blocks=%llu input_us=%.1f input_wait_us=%.1f consume_mean_us=%.1f consume_worst_us=%.1f reads=%llu "
              "hits=%llu misses=%llu skipped=%lu dev_block_max_us_after_8=%.1f consume_worst_after_8_us=%.1f recoveries=%u failed_blocks=%u points=%d devices=%d\n",
              c->prof_blocks, c->prof_input_ns / 1e3 / c->prof_blocks, c->prof_wait_ns / 1e3 / c->prof_blocks,
              c->prof_consume_n ? c->prof_consume_sum_ns / 1e3 / c->prof_consume_n : 0.0, c->prof_consume_max_ns / 1e3, c->prof_consume_n,
              c->prof_hits, c->prof_misses, c->n_skipped, c->prof_dev_max_ns / 1e3, c->prof_consume_max8_ns / 1e3, c->recoveries, c->failed_blocks, master->points, c->nsh);
    if (c->profile && c->prof_blocks) {
      fprintf(stderr, "filter_hip first blocks:");
      for (int j = 0; j < 8 && (unsigned long long)j < c->prof_blocks; j++)
        fprintf(stderr, " [%d dev_us=%.0f input_us=%.0f consume_worst_us=%.0f hits=%u misses=%u stages_us(lock,h2d,fwd,specread,edits,launch,reads,callback)=%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f]",
                j, c->prof_first_dev_ns[j] / 1e3, c->prof_first_input_ns[j] / 1e3, c->prof_first_consume_ns[j] / 1e3, c->prof_first_hits[j], c->prof_first_misses[j],
                c->prof_stage_ns[j][0] / 1e3, c->prof_stage_ns[j][1] / 1e3, c->prof_stage_ns[j][2] / 1e3, c->prof_stage_ns[j][3] / 1e3, c->prof_stage_ns[j][4] / 1e3,
                c->prof_stage_ns[j][5] / 1e3, c->prof_stage_ns[j][6] / 1e3, c->prof_stage_ns[j][7] / 1e3);
      fprintf(stderr, "\n");
    }
    if (c->ring_pinned) chz_host_unregister(master->input_buffer);
    for (int g = 0; g < c->nsh; g++) chz_engine_destroy(c->sh[g].eng);
    mctx_free(c);
  }
  if (master->init) { pthread_mutex_destroy(&master->filter_mutex); pthread_cond_destroy(&master->filter_cond); }
  ring_unmap(&master->input_buffer, master->input_buffer_size);
  for (int i = 0; i < ND; i++) chz_host_free(master->fdomain[i]);
  memset(master, 0, sizeof *master);                               /* src/filter.c:940 */
  return 0;
}

int create_filter_output(struct filter_out *slave, struct filter_in *master, int len, enum filtertype out_type) {
  if (master == NULL || slave == NULL || (out_type != SPECTRUM && len <= 0)) return -1;
  if (slave->master == master && slave->olen == len && slave->out_type == out_type && slave->init)
    goto done;                                                     /* src/filter.c:303-304 */
  if (out_type == SPECTRUM) len = 0;
  int const N = master->ilen + master->impulse_length - 1;
  int const L = master->ilen;
  if (((long)len * N % L) != 0) {                                  /* src/filter.c:313-316 */
    fprintf(stderr, "Invalid filter output length %d (fft size %d) for input N=%d, L=%d\n", len, (int)((long)len * N / L), N, L);
    return -1;
  }
  if (out_type == REAL && (((long)len * N / L) & 1)) {
    fprintf(stderr, "create_filter_output: REAL output needs an even block size (got %d)\n", (int)((long)len * N / L));
    return -1;
  }
  if (slave->init && slave->rev_plan) {                            /* geometry changed: start over ... */
    bool const isb = slave->isb, beam = slave->beam;               /* ... but create_filter_output (src/filter.c:298-415) */
    unsigned const drops = slave->block_drops;                     /* leaves these caller-owned fields alone */
    delete_filter_output(slave);
    slave->isb = isb; slave->beam = beam; slave->block_drops = drops;
  }
  slave->olen = len;
  slave->points = (int)((long)len * N / L);
  if (!slave->init) { pthread_mutex_init(&slave->response_mutex, NULL); slave->init = true; }
  slave->master = master;
  slave->out_type = out_type;
  set_filter_weights(slave, 1.0, 0.0);
  if (is_mini_master(master)) {
    pthread_mutex_lock(&master->filter_mutex);         /* what an undecided master becomes is settled under its mutex (see execute_filter_input) */
    int r = is_mini_master(master) ? mini_create_output(slave, master, len, out_type) : 0;
    if (r == 2)                            /* a slave no pooled instance serves, on a master still undecided: it becomes a full engine, in place */
      r = create_input_impl(master, master->ilen, master->impulse_length, master->in_type, true);
    pthread_mutex_unlock(&master->filter_mutex);
    if (r < 0) { slave->init = false; slave->master = NULL; return -1; }
  }
  if (!is_mini_master(master) && (out_type == COMPLEX || out_type == REAL)) {
    struct mctx *c = MCTX(master);
    bool const real = out_type == REAL;
    slave->bins = real ? slave->points / 2 + 1 : slave->points;    /* src/filter.c:346,374 */
    slave->fdomain = lmalloc(sizeof(float complex) * (size_t)slave->bins);
    struct sctx *sc = calloc(1, sizeof *sc);
    if (sc) sc->kind = CTX_SLAVE;
    if (real) {
      slave->output_buffer.r = lmalloc(sizeof(float) * (size_t)slave->points);
      if (slave->output_buffer.r) { memset(slave->output_buffer.r, 0, sizeof(float) * (size_t)slave->points); slave->output.r = slave->output_buffer.r + slave->points - len; }   /* src/filter.c:385 */
    } else {
      slave->output_buffer.c = lmalloc(sizeof(float complex) * (size_t)slave->points);
      if (slave->output_buffer.c) { memset(slave->output_buffer.c, 0, sizeof(float complex) * (size_t)slave->points); slave->output.c = slave->output_buffer.c + slave->bins - len; }  /* src/filter.c:357 */
    }
    if (!slave->fdomain || (!slave->output_buffer.c && !slave->output_buffer.r) || !sc) { FREE(slave->fdomain); FREE(slave->output_buffer.c); FREE(slave->output_buffer.r); free(sc); return -1; }
    pthread_mutex_lock(&c->lock);
    stage_wrlock(c);
    /* which device: the first one that still has room in its share of shard_channels slaves (creation order fills device 0, then 1, ...:
       contiguous blocks); with every share full, the one with the fewest slaves */
    int dv = 0;
    for (int g = 0; g < c->nsh; g++) if (c->sh[g].nslaves < c->shard_channels) { dv = g; goto chosen; }
    for (int g = 1; g < c->nsh; g++) if (c->sh[g].nslaves < c->sh[dv].nslaves) dv = g;
  chosen:;
    struct shard *const sh = &c->sh[dv];
    int bi = bank_for(c, sh, slave->points, len, real);
    if (bi < 0) {
      stage_wrunlock(c);
      pthread_mutex_unlock(&c->lock);
      fprintf(stderr, "create_filter_output: no device kernel for P=%d\n", slave->points);
      FREE(slave->fdomain); FREE(slave->output_buffer.c); FREE(slave->output_buffer.r); free(sc);
      return -1;
    }
    struct hbank *b = &sh->banks[bi];
    sc->dev = dv; sc->bank = bi; sc->idx = b->n; sc->epoch = 1; sc->n0 = NAN;
    sc->shard = c->next_shard++ % c->wshards;
    b->slaves[b->n] = slave; b->shift[b->n] = 0;
    /* the device row must hold the descriptor of the shift the host believes it holds: a channel that only ever asks for shift 0
       (the centre channel of a complex front end) would otherwise never send one and read an empty row */
    if (chz_bank_set_shifts(sh->eng, b->id, b->n, 1, &b->shift[b->n]) != 0) {
      stage_wrunlock(c);
      pthread_mutex_unlock(&c->lock);
      fprintf(stderr, "create_filter_output: %s\n", chz_last_error());
      FREE(slave->fdomain); FREE(slave->output_buffer.c); FREE(slave->output_buffer.r); free(sc);
      return -1;
    }
    if (!real && master->in_type == COMPLEX) {
      /* likewise the beam form at this index (a reused index may still hold an earlier occupant's; radio.c sets out.beam BEFORE the demodulator creates
         the filter output, src/radio.c:938-940, and create_filter_output has just reset the weights, src/filter.c:341) */
      unsigned char const on = slave->beam ? 1 : 0;
      if (b->beam_on[b->n] != on || on) {
        double const ab[4] = {creal(slave->alpha), cimag(slave->alpha), creal(slave->beta), cimag(slave->beta)};
        b->beam_on[b->n] = on; memcpy(b->beam_ab + 4 * b->n, ab, sizeof ab);
        chz_bank_set_beam(sh->eng, b->id, b->n, 1, ab, &on);
      }
    }
    for (int s = 0; s < ND; s++) b->stage_epoch[s][b->n] = 0;
    b->n++; sh->nslaves++;
    slave->rev_plan = (fftwf_plan)(void *)sc;
    sync_notches(c, master);               /* radio.c installs the list right after create_filter_input (src/radio.c:601-620): uploaded before block 0, not by it */
    stage_wrunlock(c);
    pthread_mutex_unlock(&c->lock);
  }
  /* SPECTRUM: no buffers, no plan: a block clock only (src/filter.c:368-371) */
done:;
  slave->next_jobnum = __atomic_load_n(&master->next_jobnum, __ATOMIC_RELAXED);   /* src/filter.c:413 */
  return 0;
}

int delete_filter_output(struct filter_out *slave) {
  if (slave == NULL) return -1;
  /* src/wfm.c:290-293 deletes its composite MASTER first and the three slaves after it (the reference's delete_filter_output never looks at the
     master, src/filter.c:943-957): what hangs off rev_plan says what it is by itself */
  int const kind = slave->rev_plan ? *(const int *)(const void *)slave->rev_plan : 0;
  if (kind == CTX_MSLAVE) mini_delete_output(slave);
  else if (kind == CTX_SLAVE && !(slave->master && slave->master->fwd_plan && !is_mini_master(slave->master))) {
    free(SCTX(slave));                     /* the master went first and took the banks' book-keeping (and the engines) with it */
    slave->rev_plan = NULL;
  }
  if (slave->rev_plan && slave->master && slave->master->fwd_plan) {
    struct mctx *c = MCTX(slave->master);
    struct sctx *sc = SCTX(slave);
    pthread_mutex_lock(&c->lock);
    stage_wrlock(c);
    struct shard *const sh = &c->sh[sc->dev];
    struct hbank *b = &sh->banks[sc->bank];
    int last = b->n - 1;
    if (sc->idx != last) {                 /* the last channel moves into the freed index */
      struct filter_out *mv = b->slaves[last];
      struct sctx *ms = SCTX(mv);
      b->slaves[sc->idx] = mv; ms->idx = sc->idx; ms->epoch++;
      b->shift[sc->idx] = b->shift[last];
      if (!b->real && b->isb[sc->idx] != b->isb[last]) { b->isb[sc->idx] = b->isb[last]; chz_bank_set_isb(sh->eng, b->id, sc->idx, 1, &b->isb[sc->idx]); }
      pthread_mutex_lock(&mv->response_mutex);                     /* (see bank_for) */
      if (mv->response) chz_bank_set_responses(sh->eng, b->id, ms->idx, 1, (const float *)mv->response);
      double const mv_ab[4] = {creal(mv->alpha), cimag(mv->alpha), creal(mv->beta), cimag(mv->beta)};
      pthread_mutex_unlock(&mv->response_mutex);
      if (!b->real && slave->master->in_type == COMPLEX) {
        /* ... and its beam form: the device holds the DELETED slave's flag and weights at this index, and the moved slave's block of this very moment may be
           re-run there by the miss path before the next execute_filter_input looks at the flags (round 6, found by the reference's own callers: a beam
           channel leaving handed its beam form to the plain channel that took its index) */
        unsigned char mv_on = __atomic_load_n((unsigned char const *)&mv->beam, __ATOMIC_RELAXED) ? 1 : 0;
        b->beam_on[sc->idx] = mv_on; memcpy(b->beam_ab + 4 * sc->idx, mv_ab, sizeof mv_ab);
        chz_bank_set_beam(sh->eng, b->id, sc->idx, 1, mv_ab, &mv_on);
      }
      chz_bank_set_shifts(sh->eng, b->id, ms->idx, 1, &b->shift[ms->idx]);
      for (int s = 0; s < ND; s++) b->stage_epoch[s][ms->idx] = 0;
    }
    b->beam_on[last] = 0xFF;               /* whatever the device holds at the vacated index: "unknown" makes the block that first sees a newcomer there upload its flag */
    b->slaves[last] = NULL; b->n--; sh->nslaves--;
    stage_wrunlock(c);
    pthread_mutex_unlock(&c->lock);
    free(sc);
  }
  if (slave->init) pthread_mutex_destroy(&slave->response_mutex);
  FREE(slave->output_buffer.c); FREE(slave->output_buffer.r);
  FREE(slave->response); FREE(slave->fdomain);
  memset(slave, 0, sizeof *slave);                                 /* src/filter.c:955 */
  return 0;
}

/* ------------------------------------------------------------------------- */
/* input side                                                                    */
/* ------------------------------------------------------------------------- */
static void sync_notches(struct mctx *c, struct filter_in *f) {
  /* radio.c installs f->notches after create_filter_input (src/radio.c:601-620); every entry carries its own
     averager gain (src/filter.c:468) -- the calloc'd DC sentinel of a full spur list has alpha 0, a no-op */
  struct notch_state *ns = f->notches;
  int n = 0;
  if (ns) { while (n < 63 && ns[n].bin != 0) n++; n++; }           /* list ends with the DC entry */
  bool same = (ns == c->notch_ptr && n == c->notch_n);
  for (int i = 0; same && i < n; i++) same = (ns[i].bin == c->notch_bins[i] && ns[i].alpha == c->notch_alpha[i]);
  if (same) return;
  for (int i = 0; i < n; i++) { c->notch_bins[i] = ns[i].bin; c->notch_alpha[i] = ns[i].alpha; }
  /* every device runs the recurrence itself on the same samples: the same states everywhere (src/filter.c:464-474) */
  for (int g = 0; g < c->nsh; g++)
    if (chz_set_notches_alpha(c->sh[g].eng, c->notch_bins, c->notch_alpha, n) != 0) fprintf(stderr, "filter_hip: notches: %s\n", chz_last_error());
  c->notch_ptr = ns; c->notch_n = n;
}

/* Replace failed engines (see struct mctx).  Caller holds c->lock; `job` is the block about to be enqueued, whose window starts at
   the master's read pointer.  Ends the process if an engine cannot be replaced or has just been. */
static void (*Exit_hook)(void);
void filter_hip_set_exit_hook(void (*hook)(void)) { __atomic_store_n(&Exit_hook, hook, __ATOMIC_RELEASE); }
static void die_for_the_supervisor(const char *why) {
  fprintf(stderr, "filter_hip: %s -- exiting (EX_SOFTWARE) so that the supervisor restarts the process, as the reference does on a fatal "
                  "front-end or FFT error (src/radio.c:398, src/main.c:202)\n", why);
  /* what the reference's fatal path does first (src/main.c:197-201): the host's hardware shut-down, if it registered one; then the host's
     buffered output (this is not exit(): nobody else will flush it) */
  void (*hook)(void) = __atomic_exchange_n(&Exit_hook, NULL, __ATOMIC_ACQ_REL);
  if (hook) hook();
  fflush(NULL);
  /* _exit, as the reference's fatal path (src/main.c:202): this thread holds the master's locks, a thousand channel threads and the
     runtime's callback thread are still running, and atexit handlers / static destructors of a runtime whose device has just been
     declared broken may block for ever */
  _exit(EX_SOFTWARE);
}
static void recover_engine(struct mctx *c, struct filter_in *f, unsigned job) {
  if (chz_process_exiting()) return;       /* not a device failure: the library has stopped working because the process is on its way out */
  char why[200] = "";
  for (int g = 0; g < c->nsh && !why[0]; g++)                      /* (chz_last_error is per thread: ask again from this one) */
    if (chz_engine_check(c->sh[g].eng) != 0) snprintf(why, sizeof why, "device %d: %s", c->sh[g].device, chz_last_error());
  if (!why[0]) snprintf(why, sizeof why, "%s", chz_last_error()[0] ? chz_last_error() : "a block could not be enqueued");
  unsigned const nrec = __atomic_load_n(&c->recoveries, __ATOMIC_RELAXED);
  if (nrec > 0 && job - c->last_recovery_job < RECOVERY_GRACE) {
    fprintf(stderr, "filter_hip: the device failed again %u blocks after a recovery (%s)\n", job - c->last_recovery_job, why);
    die_for_the_supervisor("second device failure");
  }
  fprintf(stderr, "filter_hip: device-side failure at block %u (%s): re-creating the engine, in-flight blocks are counted as drops\n", job, why);
  if (c->bcast) {
    /* second rung of the ladder: whatever failed, the replacement engines take the path without a collective (every device gets the samples) */
    fprintf(stderr, "filter_hip: KA9Q_HIP_EXCHANGE=broadcast was in use: the new engines exchange samples instead\n");
    c->bcast = false;
  }
  for (int g = 0; g < c->nsh; g++) (void)chz_sync(c->sh[g].eng);   /* every completion callback of the old engines has run after this (each one dropped its block) */
  stage_wrlock(c);
  /* the overlap history in front of this block's new samples: the first M-1 samples of its window in the host ring */
  const float *hist = f->in_type == COMPLEX ? (const float *)f->input_read_pointer.c : f->input_read_pointer.r;
  for (int g = 0; g < c->nsh; g++) {
    struct shard *sh = &c->sh[g];
    chz_engine_destroy(sh->eng);
    sh->eng = NULL;
    if (chz_engine_create(&sh->eng, f->ilen, f->impulse_length, f->in_type == REAL ? CHZ_REAL : CHZ_COMPLEX, sh->device, NULL, 0) != 0) {
      fprintf(stderr, "filter_hip: %s\n", chz_last_error());
      die_for_the_supervisor("the engine could not be re-created");
    }
    (void)chz_engine_notch_order(sh->eng, 1);   /* what failed was a device-side wait: the new engine orders its notches by HIP events, which cannot run out */
    for (int i = 0; i < sh->nbanks; i++) {
      struct hbank *b = &sh->banks[i];
      b->id = bank_create_dev(c, sh, b, b->cap);
      if (b->id < 0) { fprintf(stderr, "filter_hip: %s\n", chz_last_error()); die_for_the_supervisor("a channel bank could not be re-created"); }
      for (int k = 0; k < b->n; k++) {
        struct filter_out *sl = b->slaves[k];
        pthread_mutex_lock(&sl->response_mutex);                     /* (see bank_for) */
        if (sl->response) chz_bank_set_responses(sh->eng, b->id, k, 1, (const float *)sl->response);
        pthread_mutex_unlock(&sl->response_mutex);
        b->beam_on[k] = 0;                                           /* re-uploaded by the block below if the slave is in beam mode */
      }
      if (b->n > 0) {
        chz_bank_set_shifts(sh->eng, b->id, 0, b->n, b->shift);
        if (!b->real) chz_bank_set_isb(sh->eng, b->id, 0, b->n, b->isb);
      }
      for (int s2 = 0; s2 < ND; s2++) {                              /* nothing staged survives */
        b->stage_job[s2] = UINT_MAX; b->stage_n[s2] = 0;
        for (int k = 0; k < b->cap; k++) b->stage_epoch[s2][k] = 0;
      }
    }
    if (chz_input_seek(sh->eng, job, f->impulse_length > 1 ? hist : NULL) != 0) {
      fprintf(stderr, "filter_hip: %s\n", chz_last_error());
      die_for_the_supervisor("the input history could not be re-seated");
    }
  }
  c->notch_ptr = NULL; c->notch_n = -1;                            /* sync_notches uploads the list again */
  stage_wrunlock(c);
  __atomic_fetch_add(&c->recoveries, 1u, __ATOMIC_RELAXED); c->last_recovery_job = job;
  __atomic_store_n(&c->engine_first_job, job, __ATOMIC_RELEASE);
  __atomic_store_n(&c->failed, false, __ATOMIC_RELEASE);
}
static bool any_engine_failed(struct mctx *c) {
  if (__atomic_load_n(&c->failed, __ATOMIC_ACQUIRE)) return true;
  for (int g = 0; g < c->nsh; g++) if (chz_engine_check(c->sh[g].eng) != 0) return true;
  return false;
}
static int futex_wait_u32_ms(unsigned *addr, unsigned expected, long ms) {
  struct timespec ts = {.tv_sec = ms / 1000, .tv_nsec = (ms % 1000) * 1000000L};
  return (int)syscall(SYS_futex, addr, FUTEX_WAIT_PRIVATE, expected, &ts, NULL, 0);
}

/* KA9Q_HIP_PROFILE=1, blocks 0..7 only: host time of each stage of execute_filter_input */
#define PROF_STAGE(c, job, k, tref) do { if ((c)->profile && (job) < 8) { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); \
    (c)->prof_stage_ns[job][k] += (unsigned long long)((t_.tv_sec - (tref).tv_sec) * 1000000000LL + (t_.tv_nsec - (tref).tv_nsec)); (tref) = t_; } } while (0)
int execute_filter_input(struct filter_in *const f) {
  if (f == NULL || f->fwd_plan == NULL) return -1;
  if (is_mini_master(f)) {
    if (!__atomic_load_n(&((struct minictx *)(void *)f->fwd_plan)->decided, __ATOMIC_ACQUIRE)) {
      /* a block arrives at a small COMPLEX master nobody has created a same-size slave on: not a filter2 (which creates its slave right behind its
         master) but a small front end streaming before its channels exist -- a full engine from here on (filter_hip_mini.h: mini_wanted) */
      pthread_mutex_lock(&f->filter_mutex);
      int r = 0;
      if (is_mini_master(f) && !((struct minictx *)(void *)f->fwd_plan)->decided)
        r = create_input_impl(f, f->ilen, f->impulse_length, f->in_type, true);
      pthread_mutex_unlock(&f->filter_mutex);
      if (r != 0) return -1;
    }
    if (is_mini_master(f)) return mini_execute_input(f);
  }
  /* radiod ends through exit() with its front-end thread still running (src/main.c: closedown()): once the process has begun to exit the engine library issues
     nothing more to the runtime -- this block is nobody's any more (chz_process_exiting, include/chz_engine.h) */
  if (chz_process_exiting()) return 0;
  struct mctx *c = MCTX(f);
  /* Everything below is asynchronous, so the producer must not run more than ND blocks ahead of the device: block
     job-ND owns this job's completion record, spectrum slot, staged outputs and host-ring window until its callback has
     published it.  Default: wait for it.  KA9Q_HIP_INPUT_FULL=drop: do not wait -- this block is skipped (below), which is
     what a producer that only queues jobs amounts to once the slots have been lapped (src/filter.c:639-649,690-701). */
  struct timespec tp0 = {0, 0}, tp1 = {0, 0};
  if (c->profile) clock_gettime(CLOCK_MONOTONIC, &tp0);
  bool skip = false;
  {
    int const nslot = (int)(f->next_jobnum % ND);
    int waited_ms = 0;
    for (;;) {
      unsigned const done = __atomic_load_n(&c->dev_seq[nslot], __ATOMIC_ACQUIRE);
      if (done == c->enq_seq[nslot]) break;
      /* drop mode skips at most 3 blocks in a row: the samples of a skipped block still travel (below), and a 4th copy in a row
         would overwrite the device-ring window of a forward transform that may not have run yet (8 blocks of ring: jobs r-3..r in
         flight read regions r-4..r, the copy of job r+4 writes region r-4).  Then the producer waits like the default mode. */
      if (c->drop_when_full && c->consecutive_skips < 3) { skip = true; break; }
      /* a watchdog of a few block times: after a sticky device error the runtime delivers no more stream callbacks, and a producer
         asleep here for good would never reach the recovery below.  A failed check ends the wait (the recovery drains the old engines,
         whose callbacks, if they still come, drop their blocks); a device that reports nothing and completes nothing for
         wedged_ms (KA9Q_HIP_WEDGED_MS, default 10000) is beyond recovery from in here */
#define WATCHDOG_MS 250
      if (futex_wait_u32_ms(&c->dev_seq[nslot], done, WATCHDOG_MS) != 0 && errno == ETIMEDOUT) {
        waited_ms += WATCHDOG_MS;
        if (any_engine_failed(c)) { __atomic_store_n(&c->failed, true, __ATOMIC_RELEASE); break; }
        if (waited_ms >= c->wedged_ms) die_for_the_supervisor("the device has not completed a block for a long time (KA9Q_HIP_WEDGED_MS, default 10 s) and reports no error");
      }
    }
  }
  if (c->drop_when_full && f->next_jobnum >= ND) {
    /* ... and the H2D copy of block job-4 reads the region of the HOST ring the front end starts to overwrite as soon as this call
       returns (the ring holds ND windows): if the device's copy queue is THAT far behind, the producer has to wait for that copy --
       not for the block.  (In wait mode block job-4 has completed altogether by now.) */
    for (int g = 0; g < c->nsh; g++)
      if (chz_input_mark_wait(c->sh[g].eng, (int)((f->next_jobnum - ND) % 8)) != 0) fprintf(stderr, "execute_filter_input: %s\n", chz_last_error());
  }
  if (c->profile) clock_gettime(CLOCK_MONOTONIC, &tp1);
  pthread_mutex_lock(&c->lock);
  unsigned const job = __atomic_fetch_add(&f->next_jobnum, 1u, __ATOMIC_RELAXED);   /* src/filter.c:607; read lock-free by slaves being created */
  int const slot = (int)(job % ND);
  if (any_engine_failed(c)) {
    __atomic_store_n(&c->failed, true, __ATOMIC_RELEASE);
    recover_engine(c, f, job);
    skip = false;                                                   /* the new engines are idle */
    for (int s2 = 0; s2 < ND; s2++)
      if (__atomic_load_n(&c->dev_seq[s2], __ATOMIC_ACQUIRE) != c->enq_seq[s2]) {
        /* a block whose completion callback never came (after a sticky device error the runtime delivers none): the old engines are
           gone, nobody else will tell its slaves -- zeros and a counted drop, like every block the failure cost */
        unsigned const lost = c->note[s2][0].job;
        __atomic_fetch_add(&c->failed_blocks, 1u, __ATOMIC_RELAXED);
        __atomic_store_n(&c->skipped[s2][(lost / ND) % SKIP_RING], ((uint64_t)1 << 32) | lost, __ATOMIC_RELEASE);
        announce(c, s2, true);
        __atomic_store_n(&c->dev_seq[s2], c->enq_seq[s2], __ATOMIC_RELEASE);
      }
  }
  /* readers pick this up without a lock, possibly while a later lap overwrites it (as in the reference): tear-free accesses */
  __atomic_store_n(&f->samples_by_job[slot], f->sample_index, __ATOMIC_RELAXED);   /* src/filter.c:614-615 */
  f->sample_index += (uint64_t)f->ilen;
  /* the window is [read, read+N); its last L samples are the new ones (the mirror keeps them contiguous).  Advance the read
     pointer by L (src/filter.c:626-636). */
  const float *newsamples;
  if (f->in_type == COMPLEX) {
    newsamples = (const float *)(f->input_read_pointer.c + (f->impulse_length - 1));
    f->input_read_pointer.c += f->ilen;
    ring_wrap((void **)&f->input_read_pointer.c, f->input_buffer, f->input_buffer_size);
  } else {
    newsamples = f->input_read_pointer.r + (f->impulse_length - 1);
    f->input_read_pointer.r += f->ilen;
    ring_wrap((void **)&f->input_read_pointer.r, f->input_buffer, f->input_buffer_size);
  }
  if (skip) {
    /* the samples still go to the device rings (the next block's window starts with them); nothing is transformed, the
       slot's records stay with the block that is still in flight, and the slaves are told at once */
    int rc = 0;
    for (int g = 0; g < c->nsh && rc == 0; g++) {
      rc = chz_input_write(c->sh[g].eng, newsamples, f->ilen);
      if (rc == 0) rc = chz_input_mark(c->sh[g].eng, (int)(job % 8));
    }
    if (rc != 0) fprintf(stderr, "execute_filter_input: %s\n", chz_last_error());
    /* completed_jobs[slot] stays with the block still in flight there (publishing this one in it would lap that block away
       from every slave that has not fetched it yet): a skipped block is announced on its own */
    __atomic_store_n(&c->skipped[slot][(job / ND) % SKIP_RING], ((uint64_t)1 << 32) | job, __ATOMIC_RELEASE);
    __atomic_fetch_add(&c->n_skipped, 1ul, __ATOMIC_RELAXED);
    c->consecutive_skips++;
    pthread_mutex_unlock(&c->lock);
    announce(c, slot, true);
    return rc == 0 ? 0 : -1;
  }
  c->consecutive_skips = 0;
  unsigned const seq = ++c->enq_seq[slot];
  struct timespec tq;
  clock_gettime(CLOCK_MONOTONIC, &tq);
  struct timespec tst = tp1;
  sync_notches(c, f);
  PROF_STAGE(c, job, 0, tst);

  int rc = 0, callbacks = 0;
  __atomic_store_n(&c->pending[slot], (unsigned)c->nsh, __ATOMIC_RELEASE);
  /* the block reaches the devices: samples to everybody (each transforms), or samples to the first device and its spectrum to the rest */
  for (int g = 0; g < c->nsh && rc == 0; g++) {
    struct shard *const sh = &c->sh[g];
    struct done_note *note = &c->note[slot][g];
    note->ctx = c; note->job = job; note->seq = seq; note->shard = g; note->t0 = tq;
    if (c->bcast && g > 0) continue;
    rc = chz_input_write(sh->eng, newsamples, f->ilen);
    if (rc == 0 && c->drop_when_full) rc = chz_input_mark(sh->eng, (int)(job % 8));
    PROF_STAGE(c, job, 1, tst);
    if (rc == 0) rc = chz_forward(sh->eng, job);
    PROF_STAGE(c, job, 2, tst);
    if (rc == 0 && g == 0 && c->host_spectrum) rc = chz_spectrum_read_async(sh->eng, slot, (float *)f->fdomain[slot]);
    PROF_STAGE(c, job, 3, tst);
  }
  if (rc == 0 && c->bcast && c->nsh > 1) {
    chz_engine *engs[MAX_SHARDS];
    for (int g = 0; g < c->nsh; g++) engs[g] = c->sh[g].eng;
    rc = chz_spectrum_broadcast_local(engs, c->comm, c->nsh, slot, 0);
  }
  for (int g = 0; g < c->nsh && rc == 0; g++) {
    struct shard *const sh = &c->sh[g];
    struct done_note *note = &c->note[slot][g];
    /* batched channel launches: every slave with the shift it has asked for (or, failing that, asked for last) */
    for (int i = 0; rc == 0 && i < sh->nbanks; i++) {
      struct hbank *b = &sh->banks[i];
      stage_wrlock(c);
      {
        int lo = b->n, hi = 0, slo = b->n, shi = 0;
        for (int k = 0; k < b->n; k++) {
          struct filter_out *const sl = b->slaves[k];
          if (!b->real) {           /* callers flip slave->isb directly (src/radio.c:1586, src/radio_status.c:326) */
            /* a caller-owned bool its own thread flips at will: a relaxed one-byte load */
            unsigned char v = __atomic_load_n((unsigned char const *)&sl->isb, __ATOMIC_RELAXED) ? 1 : 0;
            if (v != b->isb[k]) { b->isb[k] = v; if (k < lo) lo = k; hi = k + 1; }
          }
          /* the shift the slave published when it came for a block (see struct sctx) */
          struct sctx *const sc = SCTX(sl);
          if (__atomic_load_n(&sc->want_valid, __ATOMIC_ACQUIRE)) {
            int const w = __atomic_load_n(&sc->want_shift, __ATOMIC_RELAXED);
            if (w != b->shift[k]) {
              b->shift[k] = w;
              if (shi > 0 && k > shi + 64) { chz_bank_set_shifts(sh->eng, b->id, slo, shi - slo, b->shift + slo); slo = k; }   /* islands far apart: separate edits */
              if (k < slo) slo = k;
              shi = k + 1;
            }
          }
        }
        if (hi > lo) chz_bank_set_isb(sh->eng, b->id, lo, hi - lo, b->isb + lo);
        if (shi > slo) chz_bank_set_shifts(sh->eng, b->id, slo, shi - slo, b->shift + slo);
      }
      b->stage_job[slot] = job; b->stage_n[slot] = b->n;
      for (int k = 0; k < b->n; k++) {
        b->stage_shift[slot][k] = b->shift[k];
        b->stage_epoch[slot][k] = __atomic_load_n(&b->slaves[k]->response, __ATOMIC_ACQUIRE) ? SCTX(b->slaves[k])->epoch : 0;
        b->stage_isb[slot][k] = b->isb[k];
      }
      stage_wrunlock(c);
      if (b->n == 0) continue;
      if (!b->real) {
        if (f->in_type == COMPLEX) {   /* slave->beam and its weights (src/radio.c:938-940) */
          for (int k = 0; k < b->n; k++) {
            struct filter_out *sl = b->slaves[k];
            /* set_filter_weights() runs in the slave's own thread and writes two complex doubles: read them whole */
            if (sl->init) pthread_mutex_lock(&sl->response_mutex);
            double ab[4] = {creal(sl->alpha), cimag(sl->alpha), creal(sl->beta), cimag(sl->beta)};
            if (sl->init) pthread_mutex_unlock(&sl->response_mutex);
            unsigned char on = __atomic_load_n((unsigned char const *)&sl->beam, __ATOMIC_RELAXED) ? 1 : 0;
            if (on != b->beam_on[k] || (on && memcmp(ab, b->beam_ab + 4 * k, sizeof ab) != 0)) {
              b->beam_on[k] = on; memcpy(b->beam_ab + 4 * k, ab, sizeof ab);
              chz_bank_set_beam(sh->eng, b->id, k, 1, ab, &on);
              stage_wrlock(c);
              for (int s2 = 0; s2 < ND; s2++) if (s2 != slot) b->stage_epoch[s2][k] = 0;   /* earlier staged results used other weights */
              stage_wrunlock(c);
            }
          }
        }
      }
      chz_bank_set_active(sh->eng, b->id, b->n);
      PROF_STAGE(c, job, 4, tst);
      rc = chz_bank_execute(sh->eng, b->id, slot);
      PROF_STAGE(c, job, 5, tst);
      if (rc == 0) rc = chz_bank_read_async(sh->eng, b->id, slot, 0, b->n, (float *)b->stage[slot]);
      if (rc == 0 && b->noise_on) rc = chz_bank_read_noise_async(sh->eng, b->id, slot, 0, b->n, b->stage_n0[slot]);
      PROF_STAGE(c, job, 6, tst);
    }
    if (rc == 0) rc = chz_host_callback(sh->eng, slot, block_done, note);
    PROF_STAGE(c, job, 7, tst);
    if (rc == 0) callbacks++;
  }
  if (rc != 0) {
    /* the block is dropped here (zeros + a counted drop for every slave), and the next call replaces the engines (ONE line of log,
       not one per block).  Devices that did take the block still call back: they see `failed` and the last of them announces the
       block as dropped; if none did, nobody will -- it is announced from here */
    fprintf(stderr, "execute_filter_input: block %u: %s\n", job, chz_last_error());
    __atomic_store_n(&c->failed, true, __ATOMIC_RELEASE);
    unsigned const missing = (unsigned)(c->nsh - callbacks);
    if (missing && __atomic_sub_fetch(&c->pending[slot], missing, __ATOMIC_ACQ_REL) == 0) {
      __atomic_fetch_add(&c->failed_blocks, 1u, __ATOMIC_RELAXED);
      __atomic_store_n(&c->skipped[slot][(job / ND) % SKIP_RING], ((uint64_t)1 << 32) | job, __ATOMIC_RELEASE);
      announce(c, slot, true);
      __atomic_store_n(&c->dev_seq[slot], seq, __ATOMIC_RELEASE);
    }
  }
  if (c->profile) {
    struct timespec tp2; clock_gettime(CLOCK_MONOTONIC, &tp2);
    if (c->prof_blocks < 8) c->prof_first_input_ns[c->prof_blocks] = (unsigned long long)((tp2.tv_sec - tp0.tv_sec) * 1000000000LL + (tp2.tv_nsec - tp0.tv_nsec));
    c->prof_blocks++;
    c->prof_input_ns += (unsigned long long)((tp2.tv_sec - tp0.tv_sec) * 1000000000LL + (tp2.tv_nsec - tp0.tv_nsec));
    c->prof_wait_ns += (unsigned long long)((tp1.tv_sec - tp0.tv_sec) * 1000000000LL + (tp1.tv_nsec - tp0.tv_nsec));
  }
  pthread_mutex_unlock(&c->lock);
  if (rc == 0 && f->perform_inline) {      /* inline masters hand the block over before returning (src/filter.c:562-600) */
    for (int g = 0; g < c->nsh; g++) chz_sync(c->sh[g].eng);
    pthread_mutex_lock(&f->filter_mutex);
    __atomic_store_n(&f->owner, pthread_self(), __ATOMIC_RELEASE);      /* read without the mutex by execute_filter_output */
    pthread_mutex_unlock(&f->filter_mutex);
  }
  return rc == 0 ? 0 : -1;
}

int write_cfilter(struct filter_in *f, float complex const *buffer, int size) {   /* src/filter.c:1093-1113 */
  if (f == NULL) return -1;
  if ((f->wcnt + size) * sizeof *buffer >= f->input_buffer_size) return -1;
  if (buffer != NULL) memcpy(f->input_write_pointer.c, buffer, (size_t)size * sizeof *buffer);
  f->input_write_pointer.c += size;
  ring_wrap((void **)&f->input_write_pointer.c, f->input_buffer, f->input_buffer_size);
  f->wcnt += size;
  bool executed = false;
  while (f->wcnt >= f->ilen) { f->wcnt -= f->ilen; execute_filter_input(f); executed = true; }
  return executed;
}
int write_rfilter(struct filter_in *f, float const *buffer, int size) {           /* src/filter.c:1114-1134 */
  if (f == NULL) return -1;
  if ((f->wcnt + size) * sizeof *buffer >= f->input_buffer_size) return -1;
  if (buffer != NULL) memcpy(f->input_write_pointer.r, buffer, (size_t)size * sizeof *buffer);
  f->input_write_pointer.r += size;
  ring_wrap((void **)&f->input_write_pointer.r, f->input_buffer, f->input_buffer_size);
  f->wcnt += size;
  bool executed = false;
  while (f->wcnt >= f->ilen) { f->wcnt -= f->ilen; execute_filter_input(f); executed = true; }
  return executed;
}

/* ------------------------------------------------------------------------- */
/* output side                                                                   */
/* ------------------------------------------------------------------------- */
/* Serve every queued miss with one device round trip per device: per request refresh the channel's shift / ISB flag (host-side
   edits, picked up in stream order); then, per bank and spectrum slot, a FEW requests are re-run and read back one channel at a time
   into their places in the staged image, MANY (a master whose slaves have all just been retuned, or the first block of slaves that
   arrived after it was launched) as ONE launch over the range of channels that covers them and one copy -- 2000 single-channel
   launches and copies cost tens of milliseconds, the whole bank again costs tens of microseconds; then ONE wait per slot touched.
   Caller is the batch leader. */
#define MISS_BATCH 8
static void serve_misses(struct mctx *c, struct miss_req *list) {
  pthread_mutex_lock(&c->lock);
  /* host-side edits first */
  for (struct miss_req *r = list; r; r = r->next) {
    struct sctx *sc = SCTX(r->slave);
    struct shard *sh = &c->sh[sc->dev];
    struct hbank *b = &sh->banks[sc->bank];
    int const k = sc->idx;
    unsigned char const isb = r->slave->isb ? 1 : 0;
    int rc = 0;
    if (b->shift[k] != r->shift) { b->shift[k] = r->shift; rc = chz_bank_set_shifts(sh->eng, b->id, k, 1, &b->shift[k]); }
    if (rc == 0 && !b->real && b->isb[k] != isb) { b->isb[k] = isb; rc = chz_bank_set_isb(sh->eng, b->id, k, 1, &b->isb[k]); }
    r->rc = rc; r->ranged = false;
  }
  bool wide = false;             /* a range re-run overwrites staged results of channels that did not ask: readers stay out meanwhile */
  for (int g = 0; g < c->nsh; g++) {
    struct shard *sh = &c->sh[g];
    bool touched[ND] = {false, false, false, false};
    for (int bi = 0; bi < sh->nbanks; bi++) for (int slot = 0; slot < ND; slot++) {
      struct hbank *b = &sh->banks[bi];
      int cnt = 0, lo = INT_MAX, hi = 0;
      bool same_job = true;      /* every requester wants the block the slot's image was staged for (a lapped slot holds another block's spectrum) */
      for (struct miss_req *r = list; r; r = r->next) {
        struct sctx *sc = SCTX(r->slave);
        if (sc->dev != g || sc->bank != bi || r->slot != slot || r->rc != 0) continue;
        cnt++; if (sc->idx < lo) lo = sc->idx; if (sc->idx + 1 > hi) hi = sc->idx + 1;
        if (r->job != b->stage_job[slot]) same_job = false;
      }
      if (cnt == 0) continue;
      int rc = 0;
      /* the range launch rewrites the staged image AND its shift / ISB / epoch records for every channel of [lo, hi): only for the block
         those records belong to (round 5's advisor: after a lap it relabelled another block's results; the single-channel path below
         checks the job before it records anything) */
      bool const ranged = cnt > MISS_BATCH && same_job && b->stage_job[slot] != UINT_MAX && hi <= b->stage_n[slot];
      if (ranged) {
        if (!wide) { stage_wrlock(c); wide = true; }
        rc = chz_bank_execute_range(sh->eng, b->id, (unsigned)slot, lo, hi - lo);
        if (rc == 0) rc = chz_bank_read_async(sh->eng, b->id, slot, lo, hi - lo, (float *)((char *)b->stage[slot] + (size_t)lo * b->olen * bank_sample_bytes(b)));
        if (rc == 0 && b->noise_on) rc = chz_bank_read_noise_async(sh->eng, b->id, slot, lo, hi - lo, b->stage_n0[slot] + lo);
        /* what the image now holds for EVERY channel of the range: computed with the bank's current shifts, ISB flags and responses */
        for (int k = lo; rc == 0 && k < hi; k++) {
          b->stage_shift[slot][k] = b->shift[k];
          b->stage_isb[slot][k] = b->isb[k];
          b->stage_epoch[slot][k] = __atomic_load_n(&b->slaves[k]->response, __ATOMIC_ACQUIRE) ? SCTX(b->slaves[k])->epoch : 0;
        }
      }
      for (struct miss_req *r = list; r; r = r->next) {
        struct sctx *sc = SCTX(r->slave);
        if (sc->dev != g || sc->bank != bi || r->slot != slot || r->rc != 0) continue;
        if (ranged) { r->rc = rc; r->ranged = true; }
        else {
          int const k = sc->idx;
          rc = chz_bank_execute_range(sh->eng, b->id, (unsigned)slot, k, 1);
          if (rc == 0) rc = chz_bank_read_async(sh->eng, b->id, slot, k, 1, (float *)((char *)b->stage[slot] + (size_t)k * b->olen * bank_sample_bytes(b)));
          if (rc == 0 && b->noise_on) rc = chz_bank_read_noise_async(sh->eng, b->id, slot, k, 1, b->stage_n0[slot] + k);
          r->rc = rc;
        }
        if (r->rc != 0) {
          /* a broken engine (the producer replaces it at its next block): this channel's block is lost with it, once, quietly */
          if (chz_engine_check(sh->eng) != 0 || __atomic_load_n(&c->failed, __ATOMIC_ACQUIRE)) __atomic_store_n(&c->failed, true, __ATOMIC_RELEASE);
          else fprintf(stderr, "execute_filter_output: %s\n", chz_last_error());
        }
      }
      touched[slot] = true;
    }
    for (int s2 = 0; s2 < ND; s2++)
      if (touched[s2] && chz_slot_sync(sh->eng, s2) != 0) {
        if (chz_engine_check(sh->eng) != 0) __atomic_store_n(&c->failed, true, __ATOMIC_RELEASE);
        else fprintf(stderr, "execute_filter_output: %s\n", chz_last_error());
        for (struct miss_req *r = list; r; r = r->next) if (r->slot == s2 && SCTX(r->slave)->dev == g) r->rc = -1;
      }
  }
  /* the staged image now holds exactly what each requester asked for */
  if (!wide) stage_wrlock(c);
  for (struct miss_req *r = list; r; r = r->next) {
    if (r->rc != 0 || r->ranged) continue;
    struct sctx *sc = SCTX(r->slave);
    struct hbank *b = &c->sh[sc->dev].banks[sc->bank];
    if (b->stage_job[r->slot] == r->job && sc->idx < b->stage_n[r->slot]) {
      b->stage_shift[r->slot][sc->idx] = r->shift;
      b->stage_epoch[r->slot][sc->idx] = sc->epoch;
      b->stage_isb[r->slot][sc->idx] = r->slave->isb ? 1 : 0;
    }
  }
  stage_wrunlock(c);
  pthread_mutex_unlock(&c->lock);
}

int execute_filter_output(struct filter_out *const slave, int const shift) {
  if (slave == NULL) return -1;
  struct filter_in *const master = slave->master;
  if (master == NULL) return -1;

  /* same wait / lap arithmetic as src/filter.c:680-702, on a futex instead of filter_cond */
  if (pthread_equal(__atomic_load_n(&master->owner, __ATOMIC_ACQUIRE), pthread_self()))
    slave->next_jobnum = __atomic_load_n(&master->next_jobnum, __ATOMIC_RELAXED) - 1;   /* src/filter.c:681-683 */
  unsigned const job = slave->next_jobnum;
  int const slot = (int)(job % ND);
  bool slept = false;
  /* (Measured and rejected: sleeping on 16 sharded wake words per slot with a small initial fan-out -- the tree's depth
     times the scheduler's wake latency cost 10 ms per block at 1024 threads, against 3 ms for one FUTEX_WAKE of everybody
     plus the pass-it-on below.) */
  if (master->fwd_plan && !is_mini_master(master) && slave->rev_plan) {   /* what this slave wants its next block computed with: read by the front end when it launches one */
    struct sctx *const sc0 = SCTX(slave);
    __atomic_store_n(&sc0->want_shift, shift, __ATOMIC_RELAXED);
    __atomic_store_n(&sc0->want_valid, 1, __ATOMIC_RELEASE);
  }
  bool skipped = false;
  for (;;) {
    /* (looked up every time round: an undecided small master may become an engine while a block clock sleeps on it -- it is woken then) */
    struct mctx *const mc = (master->fwd_plan && !is_mini_master(master)) ? MCTX(master) : NULL;
    int const shard = (mc && slave->rev_plan) ? SCTX(slave)->shard : 0;
    unsigned *const wake = mc ? &mc->gen[slot][shard].v : &master->completed_jobs[slot];
    unsigned const g = mc ? __atomic_load_n(wake, __ATOMIC_ACQUIRE) : 0u;          /* before looking at what it announces */
    unsigned done = __atomic_load_n(&master->completed_jobs[slot], __ATOMIC_ACQUIRE);
    if ((int)(job - done) <= 0) {
      /* Everybody asleep on this word waits for the same job, so a woken thread passes the wake-up on to a few more:
         the ~1000 channel threads are released in a tree (log depth, on many cores) instead of one after the other by
         the completion callback. */
      if (slept) { int const fan = mc ? mc->wake_fan : 2; if (fan > 0) futex_wake_n(wake, fan); }
      if ((int)(done - job) >= ND) {                               /* lapped: zeros + drop (src/filter.c:690-701) */
        slave->block_drops++;
        slave->next_jobnum++;
        if (slave->output_buffer.c != NULL) memset(slave->output_buffer.c, 0, (size_t)slave->points * sizeof *slave->output_buffer.c);
        if (slave->output_buffer.r != NULL) memset(slave->output_buffer.r, 0, (size_t)slave->points * sizeof *slave->output_buffer.r);
        return 0;
      }
      break;
    }
    if (mc && __atomic_load_n(&mc->skipped[slot][(job / ND) % SKIP_RING], __ATOMIC_ACQUIRE) == (((uint64_t)1 << 32) | job)) {
      if (slept && mc->wake_fan > 0) futex_wake_n(wake, mc->wake_fan);
      skipped = true;
      break;
    }
    futex_wait_u32(wake, mc ? g : done);                           /* src/filter.c:686-687 */
    slept = true;
  }
  slave->sample_index = __atomic_load_n(&master->samples_by_job[slot], __ATOMIC_RELAXED);   /* src/filter.c:705 */
  slave->next_jobnum++;
  if (skipped) {
    /* KA9Q_HIP_INPUT_FULL=drop: the device was ND blocks behind when this block arrived and it was never transformed:
       zeros and a counted drop, as for a lapped slave (src/filter.c:690-701) */
    slave->block_drops++;
    if (slave->output_buffer.c != NULL) memset(slave->output_buffer.c, 0, (size_t)slave->points * sizeof *slave->output_buffer.c);
    if (slave->output_buffer.r != NULL) memset(slave->output_buffer.r, 0, (size_t)slave->points * sizeof *slave->output_buffer.r);
    return 0;
  }

  if (slave->out_type == SPECTRUM || slave->rev_plan == NULL) return 0;   /* block clock only */
  if (chz_process_exiting()) return 0;                                    /* (see execute_filter_input) */
  pthread_mutex_lock(&slave->response_mutex);
  bool const real_out = slave->out_type == REAL;
  void *const dst = real_out ? (void *)slave->output.r : (void *)slave->output.c;
  bool const ready = slave->response != NULL && dst != NULL;
  pthread_mutex_unlock(&slave->response_mutex);
  if (!ready) return 0;                                            /* src/filter.c:715-718 */
  if (is_mini_master(master)) return mini_execute_output(slave, shift, slot);

  struct mctx *c = MCTX(master);
  struct sctx *sc = SCTX(slave);
  /* A served miss can still miss the re-test: another channel's delete_filter_output may move this slave to a new bank
     index between the two (its staged result then sits at the old index).  Re-run; a bounded number of times. */
  for (int attempt = 0; attempt < 32; attempt++) {
    bool hit = false;
    pthread_rwlock_t *const rl = stage_rdlock(c, slave);
    {
      struct hbank *b = &c->sh[sc->dev].banks[sc->bank];
      int const k = sc->idx;
      if (b->stage_job[slot] == job && k < b->stage_n[slot] && b->stage_shift[slot][k] == shift &&
          b->stage_epoch[slot][k] == sc->epoch && (b->real || b->stage_isb[slot][k] == (slave->isb ? 1 : 0))) {
        /* the batch (speculative, or the miss batch just served) computed exactly this */
        memcpy(dst, (char *)b->stage[slot] + (size_t)k * b->olen * bank_sample_bytes(b), bank_sample_bytes(b) * (size_t)b->olen);
        sc->n0 = b->noise_on ? b->stage_n0[slot][k] : NAN;
        hit = true;
      }
    }
    pthread_rwlock_unlock(rl);
    if (c->profile) {
      if (hit && attempt == 0) {
        struct timespec tn; clock_gettime(CLOCK_MONOTONIC, &tn);
        long long const ns = (long long)tn.tv_sec * 1000000000LL + tn.tv_nsec - __atomic_load_n(&c->t_done_ns[slot], __ATOMIC_RELAXED);
        if (ns >= 0 && ns < 1000000000LL) {
          __atomic_fetch_add(&c->prof_consume_sum_ns, (unsigned long long)ns, __ATOMIC_RELAXED);
          __atomic_fetch_add(&c->prof_consume_n, 1ull, __ATOMIC_RELAXED);
          unsigned long long mx = __atomic_load_n(&c->prof_consume_max_ns, __ATOMIC_RELAXED);
          while ((unsigned long long)ns > mx && !__atomic_compare_exchange_n(&c->prof_consume_max_ns, &mx, (unsigned long long)ns, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
          if (job < 8 && __atomic_load_n(&c->t_done_job[slot], __ATOMIC_RELAXED) == job) {
            mx = __atomic_load_n(&c->prof_first_consume_ns[job], __ATOMIC_RELAXED);
            while ((unsigned long long)ns > mx && !__atomic_compare_exchange_n(&c->prof_first_consume_ns[job], &mx, (unsigned long long)ns, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
          }
          if (job >= 8 && __atomic_load_n(&c->t_done_job[slot], __ATOMIC_RELAXED) == job) {
            mx = __atomic_load_n(&c->prof_consume_max8_ns, __ATOMIC_RELAXED);
            while ((unsigned long long)ns > mx && !__atomic_compare_exchange_n(&c->prof_consume_max8_ns, &mx, (unsigned long long)ns, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
          }
        }
      }
      __atomic_fetch_add(hit ? &c->prof_hits : &c->prof_misses, 1ull, __ATOMIC_RELAXED);
      if (job < 8) __atomic_fetch_add(hit ? &c->prof_first_hits[job] : &c->prof_first_misses[job], 1u, __ATOMIC_RELAXED);
    }
    if (hit) return 0;
    /* a block the PREVIOUS engine completed, asked for after that engine was replaced: its staged results were invalidated and its
       spectrum is gone -- re-running the channel on the new engine's slot would hand out garbage: zeros + a counted drop */
    if (__atomic_load_n(&c->recoveries, __ATOMIC_RELAXED) && (int)(job - __atomic_load_n(&c->engine_first_job, __ATOMIC_ACQUIRE)) < 0) {
      slave->block_drops++;
      memset(dst, 0, (real_out ? sizeof(float) : sizeof(float complex)) * (size_t)slave->olen);
      return 0;
    }

    /* retuned / new filter / newly created: queue this channel for a re-run on the block's spectrum */
    struct miss_req req = {.slave = slave, .shift = shift, .slot = slot, .job = job};
    pthread_mutex_lock(&c->miss_lock);
    if (c->miss_tail) c->miss_tail->next = &req; else c->miss_head = &req;
    c->miss_tail = &req;
    if (!c->miss_leader) {
      c->miss_leader = true;
      while (c->miss_head) {
        struct miss_req *list = c->miss_head;
        c->miss_head = c->miss_tail = NULL;
        pthread_mutex_unlock(&c->miss_lock);
        serve_misses(c, list);
        pthread_mutex_lock(&c->miss_lock);
        for (struct miss_req *r = list; r;) { struct miss_req *nx = r->next; r->done = true; r = nx; }   /* r may vanish once done */
        pthread_cond_broadcast(&c->miss_cv);
      }
      c->miss_leader = false;
    } else {
      while (!req.done) pthread_cond_wait(&c->miss_cv, &c->miss_lock);
    }
    pthread_mutex_unlock(&c->miss_lock);
    if (req.rc != 0) {
      if (__atomic_load_n(&c->failed, __ATOMIC_ACQUIRE)) {          /* the engine died under this block: zeros + a counted drop, like every block it lost */
        slave->block_drops++;
        memset(dst, 0, (real_out ? sizeof(float) : sizeof(float complex)) * (size_t)slave->olen);
        return 0;
      }
      return -1;
    }
  }
  /* the block's slot was re-used for a later block while this channel waited (it was being lapped) */
  fprintf(stderr, "execute_filter_output: block %u is gone from the device (the channel is more than %d blocks behind)\n", job, ND - 1);
  slave->block_drops++;
  /* (the size from the slave itself: c->banks may be growing under another thread's create_filter_output right now) */
  memset(dst, 0, (real_out ? sizeof(float) : sizeof(float complex)) * (size_t)slave->olen);
  return 0;
}

/* ---- beyond filter.h (include/ka9q_filter_hip_ext.h): estimate_noise() on the device -------------------------------
   radiod's estimate_noise() (src/radio.c:1783-1866) is the only reader of master->fdomain[] outside filter.c, and the
   reason every block's 13 MB spectrum travels back over PCIe.  A host that takes the estimate from the device instead --
   the same function run by the noise_est kernel right behind the channel kernel, pinned to radio.c's own code to 1e-12 --
   can switch that copy off (KA9Q_HIP_FDOMAIN=0).  samprate = Frontend.samprate (src/radio.c:1865). */
int filter_hip_enable_noise(struct filter_in *master, double samprate) {
  if (master == NULL || master->fwd_plan == NULL || is_mini_master(master) || !(samprate >= 0)) return -1;
  struct mctx *c = MCTX(master);
  int rc = 0;
  pthread_mutex_lock(&c->lock);
  stage_wrlock(c);
  c->noise_samprate = samprate;
  for (int g = 0; g < c->nsh; g++) for (int i = 0; i < c->sh[g].nbanks; i++) {
    struct hbank *b = &c->sh[g].banks[i];
    b->noise_on = chz_bank_enable_noise(c->sh[g].eng, b->id, samprate) == 0 && samprate > 0;
    if (samprate > 0 && !b->noise_on) rc = 1;                /* some channel size has no noise kernel: those slaves report NaN */
    for (int s = 0; s < ND; s++) for (int k = 0; k < b->cap; k++) b->stage_n0[s][k] = NAN;
    if (b->noise_on && b->n > 0) bank_warm(&c->sh[g], b);   /* the noise kernel's first launch is not block 0's business */
  }
  stage_wrunlock(c);
  pthread_mutex_unlock(&c->lock);
  return rc;
}
/* N0 (power per Hz, as estimate_noise() returns it) of the block this slave's last execute_filter_output delivered;
   NaN when the device estimate is off, the channel size has no kernel, or no block has been delivered yet */
double filter_hip_noise(struct filter_out const *slave) {
  if (slave == NULL || slave->rev_plan == NULL || slave->master == NULL || is_mini_master(slave->master)) return NAN;
  return ((struct sctx const *)(void const *)slave->rev_plan)->n0;
}
/* returns once the device has finished every block handed to it so far (an orderly shutdown reads its last results after this) */
int filter_hip_drain(struct filter_in *master) {
  if (master == NULL || master->fwd_plan == NULL || is_mini_master(master)) return -1;
  struct mctx *c = MCTX(master);
  pthread_mutex_lock(&c->lock);              /* the producer and the miss leaders enqueue (and a recovery swaps the engine) under it */
  int rc = 0;
  for (int g = 0; g < c->nsh; g++) if (chz_sync(c->sh[g].eng) != 0) rc = -1;
  pthread_mutex_unlock(&c->lock);
  return rc == 0 ? 0 : -1;
}
/* blocks the front end skipped because the device was ND blocks behind (KA9Q_HIP_INPUT_FULL=drop) */
unsigned long filter_hip_skipped_blocks(struct filter_in const *master) {
  if (master == NULL || master->fwd_plan == NULL || is_mini_master(master)) return 0;
  return __atomic_load_n(&((struct mctx const *)(void const *)master->fwd_plan)->n_skipped, __ATOMIC_RELAXED);
}
/* how often the engine had to be replaced, and how many blocks were lost to it (each a counted drop for every slave) */
unsigned filter_hip_recoveries(struct filter_in const *master, unsigned *blocks_lost) {
  if (master == NULL || master->fwd_plan == NULL || is_mini_master(master)) return 0;
  struct mctx const *c = (struct mctx const *)(void const *)master->fwd_plan;
  if (blocks_lost) *blocks_lost = __atomic_load_n(&c->failed_blocks, __ATOMIC_RELAXED);
  return __atomic_load_n(&c->recoveries, __ATOMIC_RELAXED);
}
/* how many devices the master's slaves are spread over (KA9Q_HIP_DEVICES), and -- counts != NULL -- how many slaves live on each */
int filter_hip_devices(struct filter_in const *master, int *counts, int max) {
  if (master == NULL || master->fwd_plan == NULL || is_mini_master(master)) return 0;
  struct mctx *c = (struct mctx *)(void *)master->fwd_plan;
  if (counts) {
    pthread_mutex_lock(&c->lock);
    for (int g = 0; g < c->nsh && g < max; g++) counts[g] = c->sh[g].nslaves;
    pthread_mutex_unlock(&c->lock);
  }
  return c->nsh;
}

int set_filter_weights(struct filter_out *out, double complex i_weight, double complex q_weight) {  /* src/filter.c:922-929 */
  if (out == NULL) return -1;
  /* the front-end thread reads both weights when it uploads them (execute_filter_input): written under the slave's mutex */
  if (out->init) pthread_mutex_lock(&out->response_mutex);
  out->alpha = 0.5 * i_weight - I * q_weight;
  out->beta = 0.5 * i_weight + I * q_weight;
  if (out->init) pthread_mutex_unlock(&out->response_mutex);
  return 0;
}

int set_filter(struct filter_out *const slave, double low, double high, double const kaiser_beta) {  /* src/filter.c:968-1045 */
  if (slave == NULL || isnan(low) || isnan(high) || isnan(kaiser_beta) || slave->master == NULL) return -1;
  if (slave->out_type == REAL) { low = fabs(low); high = fabs(high); }
  if (low > high) { double t = low; low = high; high = t; }
  low = low < -0.5 ? -0.5 : low > +0.5 ? +0.5 : low;
  high = high < -0.5 ? -0.5 : high > +0.5 ? +0.5 : high;
  int const N = slave->points, L = slave->olen, M = N - L + 1;
  if (M < 2) return -1;
  double const bw2 = (high == low) ? .0001 : fabs(high - low) / 2;
  double const center = (high + low) / 2;
  float *win = malloc(sizeof(float) * (size_t)M);
  float complex *response = lmalloc((size_t)N * sizeof *response);
  if (!win || !response) { free(win); free(response); return -1; }
  kaiser_f(win, M, kaiser_beta);
  double g = 0;
  for (int i = 0; i < M; i++) g += win[i];
  g = M / g;
  for (int i = 0; i < M; i++) win[i] *= (float)g;                  /* normalize_windowf */
  memset(response, 0, (size_t)N * sizeof *response);
  double window_gain = 0;
  for (int i = 0; i < M; i++) {
    double n = i - (double)(M - 1) / 2;
    double r = win[i] * 2 * bw2 * sinc_pi(2 * bw2 * n);
    window_gain += r;
    response[i] = (float complex)(cis_pi(2 * center * n) * r);
  }
  double const gain = (slave->master->in_type == REAL ? M_SQRT2 : 1.0) / (window_gain * slave->master->points);
  for (int i = 0; i < M; i++) response[i] *= gain;
  free(win);
  if (host_dft_forward(N, response) != 0) { free(response); return -1; }
  pthread_mutex_lock(&slave->response_mutex);                      /* hot swap, src/filter.c:1039-1043 */
  float complex *old = slave->response;
  __atomic_store_n(&slave->response, response, __ATOMIC_RELEASE);   /* execute_filter_input tests it for NULL without this mutex */
  pthread_mutex_unlock(&slave->response_mutex);
  free(old);
  if (slave->rev_plan && is_mini_master(slave->master)) {
    struct msctx *ms = (struct msctx *)(void *)slave->rev_plan;
    if (chz_mini_set_response(ms->pool->h, ms->inst, (const float *)response) != 0) { fprintf(stderr, "set_filter: %s\n", chz_last_error()); return -1; }
    return 0;
  }
  if (slave->rev_plan && slave->master->fwd_plan) {
    struct mctx *c = MCTX(slave->master);
    struct sctx *sc = SCTX(slave);
    pthread_mutex_lock(&c->lock);
    stage_wrlock(c);
    sc->epoch++;
    stage_wrunlock(c);
    int rc = chz_bank_set_responses(c->sh[sc->dev].eng, c->sh[sc->dev].banks[sc->bank].id, sc->idx, 1, (const float *)response);
    pthread_mutex_unlock(&c->lock);
    if (rc != 0) { fprintf(stderr, "set_filter: %s\n", chz_last_error()); return -1; }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* helpers other parts of ka9q-radio use                                         */
/* ------------------------------------------------------------------------- */
/* spectrum.c plans and runs its own analysis FFTs through these (src/spectrum.c:198,265);
   they stay FFTW-backed when the final link provides FFTW (radiod does), and return NULL
   otherwise -- never a silent substitute. */
extern fftwf_plan fftwf_plan_dft_1d(int, float complex *, float complex *, int, unsigned) __attribute__((weak));
extern fftwf_plan fftwf_plan_dft_r2c_1d(int, float *, float complex *, unsigned) __attribute__((weak));
extern fftwf_plan fftwf_plan_dft_c2r_1d(int, float complex *, float *, unsigned) __attribute__((weak));
extern void fftwf_destroy_plan(fftwf_plan) __attribute__((weak));
static pthread_mutex_t Planning_mutex = PTHREAD_MUTEX_INITIALIZER;
#define FFTW_ESTIMATE_FLAG (1U << 6)
fftwf_plan plan_complex(int N, float complex *in, float complex *out, int direction) {
  if (!fftwf_plan_dft_1d) return NULL;
  pthread_mutex_lock(&Planning_mutex);
  fftwf_plan p = fftwf_plan_dft_1d(N, in, out, direction, FFTW_ESTIMATE_FLAG);
  pthread_mutex_unlock(&Planning_mutex);
  return p;
}
fftwf_plan plan_r2c(int N, float *in, float complex *out) {
  if (!fftwf_plan_dft_r2c_1d) return NULL;
  pthread_mutex_lock(&Planning_mutex);
  fftwf_plan p = fftwf_plan_dft_r2c_1d(N, in, out, FFTW_ESTIMATE_FLAG);
  pthread_mutex_unlock(&Planning_mutex);
  return p;
}
fftwf_plan plan_c2r(int N, float complex *in, float *out) {
  if (!fftwf_plan_dft_c2r_1d) return NULL;
  pthread_mutex_lock(&Planning_mutex);
  fftwf_plan p = fftwf_plan_dft_c2r_1d(N, in, out, FFTW_ESTIMATE_FLAG);
  pthread_mutex_unlock(&Planning_mutex);
  return p;
}
void destroy_plan(fftwf_plan *plan) {
  if (plan == NULL || *plan == NULL) return;
  if (fftwf_destroy_plan) { pthread_mutex_lock(&Planning_mutex); fftwf_destroy_plan(*plan); pthread_mutex_unlock(&Planning_mutex); }
  *plan = NULL;
}
void *run_fft(void *p) { (void)p; return NULL; }     /* no CPU FFT workers exist in this build */
void suggest(int size, int dir, int clex) { (void)size; (void)dir; (void)clex; }   /* FFTW wisdom hint: nothing to log */
long gcd(long a, long b) { while (b != 0) { long t = b; b = a % b; a = t; } return a; }
long lcm(long a, long b) { if (a <= 0 || b <= 0) return 0; return (a / gcd(a, b)) * b; }
bool goodchoice(long n) {                              /* 2,3,5,7-smooth with at most one factor 11 or 13 (src/filter.c:444-451) */
  if (n <= 0) return false;
  static const int pr[6] = {2, 3, 5, 7, 11, 13};
  int big = 0;
  for (int i = 0; i < 6; i++) while (n % pr[i] == 0) { n /= pr[i]; if (i >= 4) big++; }
  return n == 1 && big <= 1;
}
int ceil_pow2(uint32_t x) {
  if (x <= 1) return 1;
  x--; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
  return (int)(x + 1);
}
