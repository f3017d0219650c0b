/* oracle/ref_rx888_wrap.c -- TEST INFRASTRUCTURE.  Pins the restated A/D conversion (chz_oracle.c:chzo_convert_i16,
 * SURVEY 8f rank 3) to the REFERENCE'S OWN CODE: this translation unit is the reference's src/rx888.c, included
 * unmodified from where it lies, plus exported wrappers around its two static conversion routines
 *     convert_avx2()  src/rx888.c:694-751   (what an x86-64 radiod with AVX2 runs)
 *     convert()       src/rx888.c:753-767   (portable C)
 * The rest of rx888.c (USB handling) is hidden and discarded at link time.  Never copied into the repo; built by
 * oracle/Makefile only where /root/reference exists, into oracle/_ref/. */
#include "rx888.c"

#define EXPORT __attribute__((visibility("default")))

EXPORT int refrx_convert(float *out, const int16_t *in, int n, float scale, uint64_t *energy, int randomize) {
  return convert(out, in, n, scale, energy, randomize != 0);
}
/* returns -1 when this host has no AVX2 (the caller skips the comparison); out must be 32-byte aligned, n % 16 == 0 */
EXPORT int refrx_convert_avx2(float *out, const int16_t *in, int n, float scale, uint64_t *energy, int randomize) {
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2")) return convert_avx2(out, in, n, scale, energy, randomize != 0);
#endif
  (void)out; (void)in; (void)n; (void)scale; (void)energy; (void)randomize;
  return -1;
}
