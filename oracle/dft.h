/* oracle/dft.h -- TEST INFRASTRUCTURE (CPU oracle), not a product path.
 *
 * Public face of the oracle's DFT provider.  All I/O is float32 (interleaved
 * re,im for complex) because that is what the reference exchanges with FFTW
 * (fftwf_*, src/filter.h:14); `precision` selects the internal arithmetic:
 *   ODFT_F64  compute in double, round once to float32  -> the parity oracle
 *   ODFT_F32  compute in float32                         -> CPU timing baseline
 * Semantics = FFTW's: unnormalised, forward sign -1, r2c returns n/2+1 bins,
 * c2r consumes n/2+1 bins (Hermitian), out-of-place.
 */
#ifndef ORACLE_DFT_H
#define ORACLE_DFT_H
#ifdef __cplusplus
extern "C" {
#endif

enum { ODFT_F64 = 0, ODFT_F32 = 1 };

typedef struct odft_plan odft_plan;

odft_plan *odft_create(int n, int precision);
void odft_destroy(odft_plan *p);
int odft_length(const odft_plan *p);
/* float32 timing path: run the long real transform on this many threads (what fftwf_plan_with_nthreads asks FFTW for) */
void odft_set_threads(odft_plan *p, int threads);
/* build the internal tables now, from one thread (real != 0: tables for r2c) */
void odft_warm(odft_plan *p, int real);

/* complex -> complex, sign = -1 forward / +1 backward; in may equal out */
void odft_c2c(odft_plan *p, const float *in, float *out, int sign);
/* n real -> n/2+1 complex (forward) */
void odft_r2c(odft_plan *p, const float *in, float *out);
/* n/2+1 complex -> n real (backward, unnormalised) */
void odft_c2r(odft_plan *p, const float *in, float *out);

/* float64 in/out variants used by the restated channelizer (chz_oracle.c) so
   that a whole channel can be carried in double without intermediate rounding */
void odft_c2c_f64(odft_plan *p, const double *in, double *out, int sign);
void odft_r2c_f64(odft_plan *p, const double *in, double *out);

#ifdef __cplusplus
}
#endif
#endif
