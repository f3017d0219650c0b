/* ka9q_filter_abi.h -- ABI contract of the drop-in replacement for ka9q-radio's filter.o
 * (libka9q_filter_hip.so, built from ka9q-radio_amd/csrc/filter_hip.c).
 *
 * radiod, linear.c, fm.c, the front-end plugins etc. keep compiling against THEIR OWN
 * src/filter.h; this header is not meant to be included by them.  It restates, for this
 * project's own C sources and tests, exactly what that header promises to its callers --
 * the two caller-owned structs (field order, types and therefore offsets: callers read
 * and write many fields directly), the enum values, and the exported
 * functions and globals -- so that an object built from it is link- and layout-compatible.
 * The layout is pinned by the _Static_asserts at the bottom (x86-64 / LP64, offsets
 * measured on the reference's own header).
 *
 * Each declaration cites the reference interface it replaces (file:line in /root/reference).
 */
#ifndef KA9Q_FILTER_ABI_H
#define KA9Q_FILTER_ABI_H

#include <pthread.h>
#include <complex.h>
#include <stdbool.h>
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
#error "C interface: include from C (the reference is C11)"
#endif

/* src/filter.h:14 pulls in <fftw3.h> only for the plan handle type; an opaque pointer of the
 * same size keeps the layout without requiring FFTW headers. */
#ifndef KA9Q_HAVE_FFTW_PLAN_TYPE
typedef struct fftwf_plan_s *fftwf_plan;
#endif

/* src/filter.h:29-34 */
enum filtertype { NONE, COMPLEX, REAL, SPECTRUM };

/* src/filter.h:38-41: a buffer seen as real or as complex samples */
struct rc {
  float *r;
  float complex *c;
};

/* src/filter.h:42-46: one spur notch; a list ends with the bin-0 (DC) entry */
struct notch_state {
  int bin;
  double complex state;
  double alpha;
};

#define ND 4   /* src/filter.h:48: depth of the block-spectrum ring */

/* src/filter.h:49-74 -- the master (input half), owned by the caller */
struct filter_in {
  enum filtertype in_type;        /* REAL or COMPLEX */
  int points;                     /* N = L + M - 1 */
  int ilen;                       /* L: new samples per block */
  int bins;                       /* N (complex) or N/2+1 (real) */
  int impulse_length;             /* M */
  int wcnt;                       /* samples written since the last executed block */
  void *input_buffer;             /* mirrored host ring: front ends write samples here */
  size_t input_buffer_size;       /* bytes of ONE mapping of the ring */
  struct rc input_write_pointer;  /* where the front end writes next */
  struct rc input_read_pointer;   /* start of the next transform window */
  fftwf_plan fwd_plan;            /* opaque; the HIP drop-in keeps its per-master context here */
  pthread_mutex_t filter_mutex;   /* protects completed_jobs / next_jobnum hand-over */
  pthread_cond_t filter_cond;     /* broadcast when a block's spectrum is complete */
  struct notch_state *notches;    /* optional, set by the caller after create (src/radio.c:601) */
  float complex *fdomain[ND];     /* host-visible block spectra (estimate_noise reads them) */
  unsigned int next_jobnum;
  unsigned int completed_jobs[ND];
  bool perform_inline;            /* transform on the calling thread, no worker hand-off */
  uint64_t sample_index;
  uint64_t samples_by_job[ND];
  bool init;
  pthread_t owner;                /* thread that completed the newest block */
};

/* src/filter.h:76-97 -- one slave (output half / channel), owned by the caller */
struct filter_out {
  struct filter_in *master;
  enum filtertype out_type;       /* REAL, COMPLEX or SPECTRUM */
  int points;                     /* P = olen * N / L */
  int olen;                       /* output samples per block */
  int bins;                       /* P (complex) or P/2+1 (real) */
  double complex alpha;           /* beam weights (unused by the accelerated path) */
  double complex beta;
  float complex *fdomain;         /* per-channel frequency-domain scratch */
  float complex *response;        /* frequency response, hot-swapped by set_filter */
  pthread_mutex_t response_mutex;
  struct rc output_buffer;        /* P time-domain samples */
  struct rc output;               /* last olen of them: what the demodulator reads */
  fftwf_plan rev_plan;            /* opaque; the HIP drop-in keeps its per-slave context here */
  unsigned next_jobnum;
  unsigned block_drops;
  int rcnt;
  uint64_t sample_index;
  bool beam;
  bool isb;
  bool init;
};

/* ---- exported functions (src/filter.h:99-118) ---- */
int create_filter_input(struct filter_in *, int const L, int const M, enum filtertype const in_type);   /* src/filter.c:186 */
int create_filter_output(struct filter_out *restrict slave, struct filter_in *restrict master,
                         int olen, enum filtertype out_type);                                            /* src/filter.c:298 */
int execute_filter_input(struct filter_in *);                                                            /* src/filter.c:558 */
int execute_filter_output(struct filter_out *, int shift);                                               /* src/filter.c:663 */
int delete_filter_input(struct filter_in *);                                                             /* src/filter.c:930 */
int delete_filter_output(struct filter_out *);                                                           /* src/filter.c:943 */
int set_filter(struct filter_out *, double low, double high, double kaiser_beta);                        /* src/filter.c:968 */
void *run_fft(void *);                                                                                   /* src/filter.c:485 */
int write_cfilter(struct filter_in *restrict, float complex const *restrict, int size);                  /* src/filter.c:1093 */
int write_rfilter(struct filter_in *restrict, float const *restrict, int size);                          /* src/filter.c:1114 */
void suggest(int size, int dir, int clex);                                                               /* src/filter.c:1136 */
long gcd(long a, long b);                                                                                /* src/filter.c:1146 */
long lcm(long a, long b);                                                                                /* src/filter.c:1154 */
fftwf_plan plan_complex(int N, float complex *in, float complex *out, int direction);                    /* src/filter.c:101 */
fftwf_plan plan_r2c(int N, float *in, float complex *out);                                               /* src/filter.c:122 */
fftwf_plan plan_c2r(int N, float complex *in, float *out);                                               /* src/filter.c:143 */
void destroy_plan(fftwf_plan *plan);                                                                     /* src/filter.c:168 */
bool goodchoice(long);                                                                                   /* src/filter.c:444 */
int ceil_pow2(uint32_t x);                                                                               /* src/filter.c:452 */
int set_filter_weights(struct filter_out *out, double complex i_weight, double complex q_weight);        /* src/filter.c:922 */

/* ---- exported data (src/filter.c:40-48,476-479) ---- */
extern const char *Wisdom_file;
extern char const *System_wisdom_file;
extern int N_worker_threads;
extern int N_internal_threads;
extern int FFTW_planning_level;
extern int64_t Min_fft_time, Max_fft_time, Avg_fft_time, Mean_dev;

/* ---- layout pins (offsets of the reference's structs on x86-64) ---- */
_Static_assert(sizeof(struct filter_in) == 288, "struct filter_in size");
_Static_assert(offsetof(struct filter_in, input_buffer) == 24, "filter_in.input_buffer");
_Static_assert(offsetof(struct filter_in, input_buffer_size) == 32, "filter_in.input_buffer_size");
_Static_assert(offsetof(struct filter_in, input_write_pointer) == 40, "filter_in.input_write_pointer");
_Static_assert(offsetof(struct filter_in, input_read_pointer) == 56, "filter_in.input_read_pointer");
_Static_assert(offsetof(struct filter_in, fwd_plan) == 72, "filter_in.fwd_plan");
_Static_assert(offsetof(struct filter_in, filter_mutex) == 80, "filter_in.filter_mutex");
_Static_assert(offsetof(struct filter_in, filter_cond) == 120, "filter_in.filter_cond");
_Static_assert(offsetof(struct filter_in, notches) == 168, "filter_in.notches");
_Static_assert(offsetof(struct filter_in, fdomain) == 176, "filter_in.fdomain");
_Static_assert(offsetof(struct filter_in, next_jobnum) == 208, "filter_in.next_jobnum");
_Static_assert(offsetof(struct filter_in, completed_jobs) == 212, "filter_in.completed_jobs");
_Static_assert(offsetof(struct filter_in, perform_inline) == 228, "filter_in.perform_inline");
_Static_assert(offsetof(struct filter_in, sample_index) == 232, "filter_in.sample_index");
_Static_assert(offsetof(struct filter_in, samples_by_job) == 240, "filter_in.samples_by_job");
_Static_assert(offsetof(struct filter_in, init) == 272, "filter_in.init");
_Static_assert(offsetof(struct filter_in, owner) == 280, "filter_in.owner");
_Static_assert(sizeof(struct filter_out) == 184, "struct filter_out size");
_Static_assert(offsetof(struct filter_out, out_type) == 8, "filter_out.out_type");
_Static_assert(offsetof(struct filter_out, alpha) == 24, "filter_out.alpha");
_Static_assert(offsetof(struct filter_out, fdomain) == 56, "filter_out.fdomain");
_Static_assert(offsetof(struct filter_out, response) == 64, "filter_out.response");
_Static_assert(offsetof(struct filter_out, response_mutex) == 72, "filter_out.response_mutex");
_Static_assert(offsetof(struct filter_out, output_buffer) == 112, "filter_out.output_buffer");
_Static_assert(offsetof(struct filter_out, output) == 128, "filter_out.output");
_Static_assert(offsetof(struct filter_out, rev_plan) == 144, "filter_out.rev_plan");
_Static_assert(offsetof(struct filter_out, next_jobnum) == 152, "filter_out.next_jobnum");
_Static_assert(offsetof(struct filter_out, block_drops) == 156, "filter_out.block_drops");
_Static_assert(offsetof(struct filter_out, rcnt) == 160, "filter_out.rcnt");
_Static_assert(offsetof(struct filter_out, sample_index) == 168, "filter_out.sample_index");
_Static_assert(offsetof(struct filter_out, beam) == 176, "filter_out.beam");
_Static_assert(offsetof(struct filter_out, isb) == 177, "filter_out.isb");
_Static_assert(offsetof(struct filter_out, init) == 178, "filter_out.init");

#endif
