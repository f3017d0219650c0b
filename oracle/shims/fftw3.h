/* oracle/shims/fftw3.h -- TEST INFRASTRUCTURE, not a product path.
 *
 * Declarations-only stand-in for FFTW3's public single-precision C API, so the
 * reference's own src/filter.c can be compiled *unmodified, where it lies*, in
 * an image that has no FFTW3 (un-vendored third-party dependency; version
 * unpinned, docs/FFTW3.md:137-140).  Only the entry points and flag names that
 * src/filter.c, src/spectrum.c and src/radio.c mention are declared; they are
 * implemented by oracle/fftw_shim.c on top of oracle/dft.c.  The numeric flag
 * values are FFTW's documented public constants.
 */
#ifndef ORACLE_SHIM_FFTW3_H
#define ORACLE_SHIM_FFTW3_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* When <complex.h> precedes <fftw3.h>, FFTW's complex type is the C99 one. */
#if defined(_Complex_I) && defined(complex) && defined(I)
typedef float _Complex fftwf_complex;
#else
typedef float fftwf_complex[2];
#endif

typedef struct fftwf_plan_s *fftwf_plan;

#define FFTW_FORWARD  (-1)
#define FFTW_BACKWARD (+1)

#define FFTW_MEASURE      (0U)
#define FFTW_EXHAUSTIVE   (1U << 3)
#define FFTW_PATIENT      (1U << 5)
#define FFTW_ESTIMATE     (1U << 6)
#define FFTW_WISDOM_ONLY  (1U << 21)

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);
fftwf_plan fftwf_plan_dft_r2c_1d(int n, float *in, fftwf_complex *out, unsigned flags);
fftwf_plan fftwf_plan_dft_c2r_1d(int n, fftwf_complex *in, float *out, unsigned flags);
void fftwf_execute(const fftwf_plan p);
void fftwf_execute_dft(const fftwf_plan p, fftwf_complex *in, fftwf_complex *out);
void fftwf_execute_dft_r2c(const fftwf_plan p, float *in, fftwf_complex *out);
void fftwf_execute_dft_c2r(const fftwf_plan p, fftwf_complex *in, float *out);
void fftwf_destroy_plan(fftwf_plan p);
int  fftwf_init_threads(void);
void fftwf_plan_with_nthreads(int nthreads);
int  fftwf_import_system_wisdom(void);
int  fftwf_import_wisdom_from_filename(const char *filename);
void *fftwf_malloc(size_t n);
float *fftwf_alloc_real(size_t n);
fftwf_complex *fftwf_alloc_complex(size_t n);
void fftwf_free(void *p);
extern const char fftwf_version[];

#ifdef __cplusplus
}
#endif
#endif
