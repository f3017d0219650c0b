/* tests/c/dropin_harness.c -- drives the filter.h API the way radiod does (TEST CODE).
 *
 * One front-end thread writes samples in place through in.input_write_pointer and calls
 * write_rfilter/write_cfilter(&in, NULL, n) like the SDR drivers (src/rx888.c:800-826,
 * src/sig_gen.c:290-320); one thread per channel loops execute_filter_output(&out, shift)
 * like demod_linear/demod_fm via downconvert() (src/radio.c:1460); one SPECTRUM slave acts
 * as a block clock (src/spectrum.c).  Results go to files for tests/test_dropin.py.
 *
 * Built twice: against include/ka9q_filter_abi.h (default) and, where /root/reference
 * exists, against the reference's OWN src/filter.h (-DKA9Q_FILTER_HEADER='"filter.h"') to
 * prove the drop-in is source- and layout-compatible with unmodified callers.
 */
#define _GNU_SOURCE 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdatomic.h>
#include <pthread.h>
#include <complex.h>
#include <unistd.h>
#include <time.h>
#include <stdint.h>
#ifndef KA9Q_FILTER_HEADER
#define KA9Q_FILTER_HEADER "ka9q_filter_abi.h"
#endif
#include KA9Q_FILTER_HEADER

struct chanplan { int shift, shift2, retune_block, refilter_block; double low, high, beta, low2, high2; };

extern int64_t Avg_fft_time, Max_fft_time;   /* exported by filter.o (src/filter.c:476-479) */
/* beyond filter.h (include/ka9q_filter_hip_ext.h); weak, so the same source still links against the reference's own filter.o */
extern int filter_hip_enable_noise(struct filter_in *, double) __attribute__((weak));
extern double filter_hip_noise(struct filter_out const *) __attribute__((weak));
extern unsigned long filter_hip_skipped_blocks(struct filter_in const *) __attribute__((weak));
extern int filter_hip_drain(struct filter_in *) __attribute__((weak));
extern int filter_hip_devices(struct filter_in const *, int *, int) __attribute__((weak));
static int Devices = 1; static char Dev_counts[256] = "-";   /* KA9Q_HIP_DEVICES: devices behind the master and the slaves on each, once everybody has registered */
static double Noise_samprate; static double *Noise;   /* env HARNESS_NOISE=<front-end sample rate>: [Nblocks][Nch] device-side estimate_noise() */
static int Ahead = 2;                                  /* env HARNESS_AHEAD: blocks the front end may run ahead of the slowest channel (the filter keeps ND = 4 blocks: 3 loses nothing) */
static int Free_run;                                   /* env HARNESS_FREE_RUN=1: the front end does not wait for the channels (a real A/D never does); > 1: and takes that many microseconds per block */
static long Paced_us;                                  /* env HARNESS_PACED_US=20000: a front end on its own WALL CLOCK, as an A/D is (src/sig_gen.c:357-362 paces itself the same way):
                                                          every chunk goes in at the instant its last sample would have arrived (absolute deadlines, no drift), the
                                                          front end never waits for a channel, a late channel is a counted drop (src/filter.c:686-701) */
static long long *T_in;                                /* [Nblocks] when write_?filter returned for the chunk that completed the block */
static _Atomic long long *Last_out;                    /* [Nblocks] when the LAST slave left execute_filter_output with the block */
static _Atomic int *N_out;                             /* [Nblocks] slaves served */
static unsigned char *Skipflag; static unsigned long Skips_seen;   /* [Nblocks]: the front end's block was skipped (drop mode) */
static unsigned char *Dropped;                         /* [Nblocks][Nch]: the channel's block_drops went up in that call */
static struct filter_in Master;
static int L, M, In_type, Olen, Nch, Nblocks, Chunk;
static struct chanplan *Plan;
static float complex *Result;          /* [Nblocks][Nch][Olen] */
static unsigned *Drops;
static _Atomic int *Progress;          /* blocks consumed per channel */
static _Atomic int Done_count;         /* channel threads that have consumed their last block */
static float *Input;

struct chanarg { int idx; };
/* optional second filter per channel, created and driven exactly as radio.c does (src/radio.c:1503-1513,1572-1594):
   env HARNESS_FILTER2="blocking low high beta [isb_channel]" */
static int F2_blocking, F2_isb = -1; static double F2_low, F2_high, F2_beta;
/* env HARNESS_RETUNE_MOD=m: channel i flips between its two shifts at every block b with (b + i) % m == 0, i.e. 1/m of
   the channels retune EVERY block (a scanning / Doppler-tracking channel set) */
static int Retune_mod;
/* rate runs (bench.py's `dropin` object, scripts/dropin_rate.py): env HARNESS_INPUT_BLOCKS=k -- in.bin holds k blocks that are
   replayed cyclically; HARNESS_KEEP=0 -- results are not kept (one block's worth of memory, overwritten) */
static int Input_blocks, Keep = 1;
static long long Fe_copy_ns, Fe_call_ns, Fe_wait_ns;          /* front end: copying samples in, inside write_?filter, waiting for the slowest channel */
static long long Worst_gap_ns, Sum_gap_ns; static int N_gap;  /* block clock: time between consecutive blocks */
static long long Pace_t0, Fe_late_worst_ns, Fe_call_worst_ns; /* paced front end: start instant, worst wake-up lateness, longest single write_?filter call (after 8 blocks) */
static long long now_ns(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1000000000LL + t.tv_nsec; }
/* env HARNESS_CHURN_MOD=m: channel i with i % m == 1 leaves half way (delete_filter_output while everybody else runs: the bank
   closes the gap by moving its last slave) and comes back as a NEW slave a little later (create_filter_output + set_filter into
   a running master, then delete again), as radiod's dynamic channels do */
static int Churn_mod;

static void *channel_thread(void *a) {
  int const i = ((struct chanarg *)a)->idx;
  struct filter_out out;
  memset(&out, 0, sizeof out);
  if (create_filter_output(&out, &Master, Olen, COMPLEX) != 0) { fprintf(stderr, "create_filter_output failed\n"); exit(2); }
  if (set_filter(&out, Plan[i].low, Plan[i].high, Plan[i].beta) != 0) { fprintf(stderr, "set_filter failed\n"); exit(2); }
  if (getenv("HARNESS_ISB") && atoi(getenv("HARNESS_ISB")) == i) out.isb = true;   /* set by the caller after create (src/radio.c:1586) */
  struct filter_in f2in; struct filter_out f2out;
  memset(&f2in, 0, sizeof f2in); memset(&f2out, 0, sizeof f2out);
  if (F2_blocking > 0) {
    int const blocksize = F2_blocking * Olen;
    int const n = ceil_pow2((uint32_t)(2 * blocksize));        /* round2(2 * blocksize), src/radio.c:1578 */
    int const order = n - blocksize;
    if (create_filter_input(&f2in, blocksize, order + 1, COMPLEX) != 0) { fprintf(stderr, "filter2 create_filter_input failed\n"); exit(2); }
    f2in.perform_inline = true;
    if (create_filter_output(&f2out, &f2in, blocksize, COMPLEX) != 0) { fprintf(stderr, "filter2 create_filter_output failed\n"); exit(2); }
    f2out.isb = (F2_isb == i);
    if (set_filter(&f2out, F2_low, F2_high, F2_beta) != 0) { fprintf(stderr, "filter2 set_filter failed\n"); exit(2); }
  }
  atomic_store(&Progress[i], 0);       /* registered: the producer may start */
  for (int b = 0; b < Nblocks; b++) {
    if (b == Plan[i].refilter_block) set_filter(&out, Plan[i].low2, Plan[i].high2, Plan[i].beta);
    int shift = (b >= Plan[i].retune_block) ? Plan[i].shift2 : Plan[i].shift;
    if (Retune_mod > 0) shift = (((b + i) / Retune_mod) & 1) ? Plan[i].shift2 : Plan[i].shift;
    unsigned const drops_before = out.block_drops;
    if (execute_filter_output(&out, shift) != 0) { fprintf(stderr, "execute_filter_output failed\n"); exit(2); }
    if (Dropped) Dropped[(size_t)b * Nch + i] = out.block_drops != drops_before;
    if (Last_out) {                                     /* latency bookkeeping: the block just served is next_jobnum - 1 (slaves start at job 0) */
      unsigned const jb = out.next_jobnum - 1u;
      if (jb < (unsigned)Nblocks && out.block_drops == drops_before) {
        long long const t = now_ns();
        long long prev = atomic_load(&Last_out[jb]);
        while (t > prev && !atomic_compare_exchange_weak(&Last_out[jb], &prev, t)) { }
        atomic_fetch_add(&N_out[jb], 1);
      }
    }
    if (Noise && filter_hip_noise) Noise[(size_t)b * Nch + i] = filter_hip_noise(&out);
    if (F2_blocking > 0) {
      int r = write_cfilter(&f2in, out.output.c, Olen);          /* runs the input side once the block is full (src/radio.c:1508) */
      if (r < 0) { fprintf(stderr, "filter2 write_cfilter failed\n"); exit(2); }
      if (r > 0) {
        if (execute_filter_output(&f2out, 0) != 0) { fprintf(stderr, "filter2 execute_filter_output failed\n"); exit(2); }
        for (int q = 0; q < F2_blocking; q++)                     /* the F2_blocking*Olen samples continue the channel's stream */
          memcpy(Result + ((size_t)(b - F2_blocking + 1 + q) * Nch + i) * Olen, f2out.output.c + (size_t)q * Olen, sizeof(float complex) * (size_t)Olen);
      }
    } else
      memcpy(Result + ((size_t)(Keep ? b : 0) * Nch + i) * Olen, out.output.c, sizeof(float complex) * (size_t)Olen);
    atomic_store(&Progress[i], b + 1);
    if (Churn_mod > 0 && i % Churn_mod == 1 && b == Nblocks / 2 && F2_blocking == 0) {
      Drops[i] += out.block_drops;
      delete_filter_output(&out);
      memset(&out, 0, sizeof out);
      atomic_store(&Progress[i], Nblocks);                       /* the front end does not wait for somebody who is away */
      usleep(3000);
      if (create_filter_output(&out, &Master, Olen, COMPLEX) != 0) { fprintf(stderr, "create_filter_output (again) failed\n"); exit(2); }
      if (set_filter(&out, Plan[i].low, Plan[i].high, Plan[i].beta) != 0) { fprintf(stderr, "set_filter (again) failed\n"); exit(2); }
      /* (it does not wait for blocks again: the stream may end before it would get one) */
      break;
    }
  }
  Drops[i] += out.block_drops;
  /* nobody tears its slave down while others are still fetching the last block: delete_filter_output takes the master's write locks
     and talks to the device, and 1000 of those in front of the stragglers made the LAST block look 50 ms late (round 4, paced legs) */
  atomic_fetch_add(&Done_count, 1);
  while (atomic_load(&Done_count) < Nch) usleep(500);
  if (F2_blocking > 0) { delete_filter_output(&f2out); delete_filter_input(&f2in); }
  delete_filter_output(&out);
  return NULL;
}

/* optional REAL-output slave (wfm's composite filters use these, src/wfm.c): env HARNESS_REAL="shift low high beta" */
static int Real_on, Real_shift; static double Real_low, Real_high, Real_beta;
static _Atomic int Real_ready, Clock_ready;       /* the front end starts only when every consumer has its slave (they count Nblocks from there) */
static float *Real_result;             /* [Nblocks][Olen] */
static void *real_thread(void *a) {
  (void)a;
  struct filter_out out;
  memset(&out, 0, sizeof out);
  if (create_filter_output(&out, &Master, Olen, REAL) != 0) { fprintf(stderr, "create_filter_output(REAL) failed\n"); exit(2); }
  if (out.bins != out.points / 2 + 1 || out.output.r != out.output_buffer.r + out.points - Olen) { fprintf(stderr, "REAL slave geometry\n"); exit(2); }
  if (set_filter(&out, Real_low, Real_high, Real_beta) != 0) { fprintf(stderr, "set_filter(REAL) failed\n"); exit(2); }
  atomic_store(&Real_ready, 1);
  for (int b = 0; b < Nblocks; b++) {
    if (execute_filter_output(&out, Real_shift) != 0) { fprintf(stderr, "execute_filter_output(REAL) failed\n"); exit(2); }
    memcpy(Real_result + (size_t)b * Olen, out.output.r, sizeof(float) * (size_t)Olen);
  }
  delete_filter_output(&out);
  return NULL;
}

static _Atomic int Clock_blocks;
static void *clock_thread(void *a) {
  (void)a;
  struct filter_out sp;
  memset(&sp, 0, sizeof sp);
  if (create_filter_output(&sp, &Master, 0, SPECTRUM) != 0) { fprintf(stderr, "SPECTRUM slave failed\n"); exit(2); }
  atomic_store(&Clock_ready, 1);
  long long last = 0;
  for (int b = 0; b < Nblocks; b++) {
    execute_filter_output(&sp, 0); atomic_fetch_add(&Clock_blocks, 1);
    long long const t = now_ns();
    if (b >= 8) { long long const g = t - last; if (g > Worst_gap_ns) Worst_gap_ns = g; Sum_gap_ns += g; N_gap++; }   /* the first blocks fill the pipeline */
    last = t;
  }
  delete_filter_output(&sp);
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 1; }
  char path[512];
  snprintf(path, sizeof path, "%s/cfg.txt", argv[1]);
  FILE *f = fopen(path, "r");
  if (!f || fscanf(f, "%d %d %d %d %d %d %d", &L, &M, &In_type, &Olen, &Nch, &Nblocks, &Chunk) != 7) { perror("cfg"); return 1; }
  fclose(f);
  int const per = In_type == REAL ? 1 : 2;
  Plan = calloc((size_t)Nch, sizeof *Plan);
  snprintf(path, sizeof path, "%s/plan.bin", argv[1]);
  f = fopen(path, "rb");
  if (!f || fread(Plan, sizeof *Plan, (size_t)Nch, f) != (size_t)Nch) { perror("plan"); return 1; }
  fclose(f);
  /* the engine library does not read test variables from the environment (include/chz_engine.h: chz_set_option): the harness hands them over */
  {
    extern int chz_set_option(const char *, const char *);
    static const char *const map[][2] = {{"CHZ_NOTCH_WAIT_MS", "notch_wait_ms"}, {"CHZ_FAULT_TICKET_SKEW", "fault_ticket_skew"},
                                          {"CHZ_ALLOW_FAULT_INJECTION", "allow_fault_injection"}, {"CHZ_ENQ_THREADS", "enq_threads"},
                                          {"CHZ_NOTCH_FOLD", "notch_fold"}, {"CHZ_CHAN_STAGE", "chan_stage"}};
    for (size_t i = 0; i < sizeof map / sizeof map[0]; i++)
      if (getenv(map[i][0])) chz_set_option(map[i][1], getenv(map[i][0]));
  }
  if (getenv("HARNESS_INPUT_BLOCKS")) Input_blocks = atoi(getenv("HARNESS_INPUT_BLOCKS"));
  if (getenv("HARNESS_KEEP")) Keep = atoi(getenv("HARNESS_KEEP")) != 0;
  if (Input_blocks <= 0 || Input_blocks > Nblocks) Input_blocks = Nblocks;
  Input = malloc(sizeof(float) * (size_t)Input_blocks * L * per);
  snprintf(path, sizeof path, "%s/in.bin", argv[1]);
  f = fopen(path, "rb");
  if (!f || fread(Input, sizeof(float) * per, (size_t)Input_blocks * L, f) != (size_t)Input_blocks * L) { perror("input"); return 1; }
  fclose(f);
  Result = calloc((size_t)(Keep ? Nblocks : 1) * Nch * Olen, sizeof *Result);
  Drops = calloc((size_t)Nch, sizeof *Drops);
  Progress = calloc((size_t)Nch, sizeof *Progress);
  for (int i = 0; i < Nch; i++) atomic_store(&Progress[i], -1);

  memset(&Master, 0, sizeof Master);
  if (create_filter_input(&Master, L, M, (enum filtertype)In_type) != 0) { fprintf(stderr, "create_filter_input failed\n"); return 3; }
  if (create_filter_input(&Master, L, M, (enum filtertype)In_type) != 0) return 3;      /* idempotent (src/filter.c:191) */
  /* DC notch, installed after create exactly as radio.c does (src/radio.c:601-620) */
  struct notch_state *notch = calloc(1, sizeof *notch);
  notch[0].bin = 0; notch[0].alpha = 0.01;
  Master.notches = notch;

  if (getenv("HARNESS_FILTER2")) {
    int k = sscanf(getenv("HARNESS_FILTER2"), "%d %lf %lf %lf %d", &F2_blocking, &F2_low, &F2_high, &F2_beta, &F2_isb);
    if (k < 4) { fprintf(stderr, "HARNESS_FILTER2 needs: blocking low high beta [isb_channel]\n"); return 1; }
  }
  if (getenv("HARNESS_NOISE") && filter_hip_enable_noise) {
    Noise_samprate = atof(getenv("HARNESS_NOISE"));
    Noise = calloc((size_t)Nblocks * Nch, sizeof *Noise);
  }
  if (getenv("HARNESS_FREE_RUN")) { Free_run = atoi(getenv("HARNESS_FREE_RUN")); Dropped = calloc((size_t)Nblocks * Nch, 1); Skipflag = calloc((size_t)Nblocks, 1); }
  if (getenv("HARNESS_RECORD_DROPS") && !Dropped) Dropped = calloc((size_t)Nblocks * Nch, 1);   /* per-call drop flags without a free-running front end */
  if (getenv("HARNESS_PACED_US")) {
    Paced_us = atol(getenv("HARNESS_PACED_US"));
    if (Paced_us > 0) { Free_run = 1; if (!Dropped) Dropped = calloc((size_t)Nblocks * Nch, 1); if (!Skipflag) Skipflag = calloc((size_t)Nblocks, 1); }
  }
  if (Paced_us > 0 || getenv("HARNESS_LATENCY")) {
    T_in = calloc((size_t)Nblocks, sizeof *T_in);
    Last_out = calloc((size_t)Nblocks, sizeof *Last_out);
    N_out = calloc((size_t)Nblocks, sizeof *N_out);
  }
  if (getenv("HARNESS_AHEAD")) { Ahead = atoi(getenv("HARNESS_AHEAD")); if (Ahead < 1) Ahead = 1; }
  if (getenv("HARNESS_RETUNE_MOD")) Retune_mod = atoi(getenv("HARNESS_RETUNE_MOD"));
  if (getenv("HARNESS_CHURN_MOD")) Churn_mod = atoi(getenv("HARNESS_CHURN_MOD"));
  pthread_t *th = calloc((size_t)Nch, sizeof *th), clk;
  struct chanarg *args = calloc((size_t)Nch, sizeof *args);
  for (int i = 0; i < Nch; i++) { args[i].idx = i; pthread_create(&th[i], NULL, channel_thread, &args[i]); }
  pthread_create(&clk, NULL, clock_thread, NULL);
  pthread_t rth;
  if (getenv("HARNESS_REAL") && sscanf(getenv("HARNESS_REAL"), "%d %lf %lf %lf", &Real_shift, &Real_low, &Real_high, &Real_beta) == 4) {
    Real_on = 1;
    Real_result = calloc((size_t)Nblocks * Olen, sizeof *Real_result);
    pthread_create(&rth, NULL, real_thread, NULL);
  }
  for (int i = 0; i < Nch; i++) while (atomic_load(&Progress[i]) < 0) usleep(200);      /* all slaves registered */
  while (!atomic_load(&Clock_ready) || (Real_on && !atomic_load(&Real_ready))) usleep(200);

  if (Noise && filter_hip_enable_noise(&Master, Noise_samprate) < 0) { fprintf(stderr, "filter_hip_enable_noise failed\n"); return 5; }
  if (filter_hip_devices) {
    int cnt[64] = {0};
    Devices = filter_hip_devices(&Master, cnt, 64);
    size_t o = 0;
    for (int g = 0; g < Devices && g < 64 && o + 16 < sizeof Dev_counts; g++) o += (size_t)snprintf(Dev_counts + o, sizeof Dev_counts - o, "%s%d", g ? ":" : "", cnt[g]);
  }
  /* front end: write in place, then tell the filter how much arrived */
  struct timespec ts0, ts1;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  Pace_t0 = now_ns();
  long total = (long)Nblocks * L, pos = 0;
  long const cyc = (long)Input_blocks * L;
  while (pos < total) {
    int blk = (int)(pos / L);
    long long const w0 = now_ns();
    for (;;) {   /* never run more than 2 blocks ahead of the slowest channel: no drops wanted here */
      int slow = Nblocks;
      for (int i = 0; i < Nch; i++) { int p = atomic_load(&Progress[i]); if (p < slow) slow = p; }
      if (blk - slow < Ahead || Free_run) break;
      usleep(100);
    }
    int n = Chunk; if (pos + n > total) n = (int)(total - pos);
    long const src = pos % cyc;
    if (src + n > cyc) n = (int)(cyc - src);                    /* a chunk does not straddle the replay seam */
    if (pos / L != (pos + n - 1) / L) n = (int)(((pos / L) + 1) * L - pos);   /* ... nor a block boundary (so a block's completion has one instant) */
    if (Paced_us > 0) {                                         /* the chunk's last sample arrives at t0 + (pos + n) / fs */
      long long const due = Pace_t0 + (long long)((double)(pos + n) / (double)L * (double)Paced_us * 1000.0);
      struct timespec d = {.tv_sec = due / 1000000000LL, .tv_nsec = due % 1000000000LL};
      while (clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &d, NULL) != 0) { }
      long long const late = now_ns() - due;
      if (pos / L >= 8 && late > Fe_late_worst_ns) Fe_late_worst_ns = late;      /* (the first blocks: 1000 threads touching their buffers for the first time) */
    }
    long long const w1 = now_ns();
    long long w2;
#ifdef HARNESS_REF_HEADER
    /* the reference header's own inline producers (src/filter.h:119-145; ctcss.c, packetd.c, rdsd.c, stereod.c feed their filters this way): sample by sample through
       input_write_pointer / wcnt, mirror_wrap() and a direct call of execute_filter_input() -- none of it code of the library under test */
    if (getenv("HARNESS_PUT")) {
      w2 = now_ns();
      if (In_type == REAL) for (int i = 0; i < n; i++) put_rfilter(&Master, Input[src + i]);
      else for (int i = 0; i < n; i++) put_cfilter(&Master, Input[2 * (src + i)] + I * Input[2 * (src + i) + 1]);
    } else
#endif
    if (In_type == REAL) {
      memcpy(Master.input_write_pointer.r, Input + src, sizeof(float) * (size_t)n);
      w2 = now_ns();
      if (write_rfilter(&Master, NULL, n) < 0) { fprintf(stderr, "write_rfilter overrun\n"); return 4; }
    } else {
      memcpy(Master.input_write_pointer.c, Input + 2 * src, sizeof(float complex) * (size_t)n);
      w2 = now_ns();
      if (write_cfilter(&Master, NULL, n) < 0) { fprintf(stderr, "write_cfilter overrun\n"); return 4; }
    }
    long long const w3 = now_ns();
    Fe_wait_ns += w1 - w0; Fe_copy_ns += w2 - w1; Fe_call_ns += w3 - w2;
    { long long const c = w3 - w2; if (pos / L >= 8 && c > Fe_call_worst_ns) Fe_call_worst_ns = c; }
    pos += n;
    if (T_in && pos % L == 0) T_in[pos / L - 1] = w3;          /* the block is in: its clock starts */
    if (Skipflag && filter_hip_skipped_blocks && pos / L != (pos - n) / L) {      /* a block has just gone in: was it skipped? */
      unsigned long const sk = filter_hip_skipped_blocks(&Master);
      Skipflag[(pos - n) / L] = sk != Skips_seen; Skips_seen = sk;
    }
    if (Free_run > 1 && pos / L != (pos - n) / L) usleep((useconds_t)Free_run);   /* a front end with its own clock: this many microseconds per block */
  }
  for (int i = 0; i < Nch; i++) pthread_join(th[i], NULL);
  pthread_join(clk, NULL);
  if (Real_on) pthread_join(rth, NULL);
  if (filter_hip_drain) filter_hip_drain(&Master);   /* skipped blocks let the consumers finish before the device has; and the statistics below are the callback thread's */
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  double elapsed = (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec);

  snprintf(path, sizeof path, "%s/out.bin", argv[1]);
  f = fopen(path, "wb"); fwrite(Result, sizeof *Result, (size_t)(Keep ? Nblocks : 1) * Nch * Olen, f); fclose(f);
  if (Real_on) {
    snprintf(path, sizeof path, "%s/real.bin", argv[1]);
    f = fopen(path, "wb"); fwrite(Real_result, sizeof *Real_result, (size_t)Nblocks * Olen, f); fclose(f);
  }
  if (Skipflag) {
    snprintf(path, sizeof path, "%s/skipped.bin", argv[1]);
    f = fopen(path, "wb"); fwrite(Skipflag, 1, (size_t)Nblocks, f); fclose(f);
  }
  if (Dropped) {
    snprintf(path, sizeof path, "%s/dropped.bin", argv[1]);
    f = fopen(path, "wb"); fwrite(Dropped, 1, (size_t)Nblocks * Nch, f); fclose(f);
  }
  if (Noise) {
    snprintf(path, sizeof path, "%s/noise.bin", argv[1]);
    f = fopen(path, "wb"); fwrite(Noise, sizeof *Noise, (size_t)Nblocks * Nch, f); fclose(f);
  }
  if (T_in) {                                             /* per block: ns from 'the block is in' to 'the last slave has it', and how many slaves got it */
    snprintf(path, sizeof path, "%s/latency.bin", argv[1]);
    f = fopen(path, "wb");
    for (int b = 0; b < Nblocks; b++) {
      long long const lo = atomic_load(&Last_out[b]);
      long long rec[2] = {(lo && T_in[b]) ? lo - T_in[b] : -1, atomic_load(&N_out[b])};
      fwrite(rec, sizeof rec, 1, f);
    }
    fclose(f);
  }
  snprintf(path, sizeof path, "%s/spec.bin", argv[1]);   /* host-visible spectrum of the last block (estimate_noise reads it) */
  f = fopen(path, "wb"); fwrite(Master.fdomain[(Nblocks - 1) % ND], sizeof(float complex), (size_t)Master.bins, f); fclose(f);
  snprintf(path, sizeof path, "%s/meta.txt", argv[1]);
  f = fopen(path, "w");
  unsigned drops = 0; for (int i = 0; i < Nch; i++) drops += Drops[i];
  fprintf(f, "drops %u clock %d next_jobnum %u bins %d points %d sample_index %llu elapsed_s %.6f avg_block_ns %lld max_block_ns %lld "
             "worst_gap_ns %lld mean_gap_ns %lld fe_copy_ns %lld fe_call_ns %lld fe_wait_ns %lld skipped %lu fe_late_worst_ns %lld fe_call_worst_ns %lld "
             "devices %d dev_counts %s\n",
          drops, atomic_load(&Clock_blocks), Master.next_jobnum, Master.bins, Master.points,
          (unsigned long long)Master.sample_index, elapsed, (long long)Avg_fft_time, (long long)Max_fft_time,
          Worst_gap_ns, N_gap ? Sum_gap_ns / N_gap : 0, Fe_copy_ns, Fe_call_ns, Fe_wait_ns,
          filter_hip_skipped_blocks ? filter_hip_skipped_blocks(&Master) : 0ul, Fe_late_worst_ns, Fe_call_worst_ns, Devices, Dev_counts);
  fclose(f);
  delete_filter_input(&Master);
  free(th); free(args); free(notch);      /* a clean exit for the leak checker of the sanitizer runs (the caller owns the notch list) */
  return 0;
}
