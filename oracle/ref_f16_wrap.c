/* oracle/ref_f16_wrap.c -- TEST INFRASTRUCTURE.  The reference's own 16-bit float PCM packing (src/import.h:140-157,
 * 207-212: export_f16_le / export_f16_be) compiled where it lies.  gcc 11 on x86-64 has no _Float16 -- with it the
 * reference builds WITHOUT these encodings (src/misc.h:52-63 leaves HAS_FLOAT16 undefined) -- so this one wrapper is
 * built with the image's clang (amdclang, host target), which has the type; float -> _Float16 is round-to-nearest-even
 * there as in hardware. */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
typedef _Float16 float16_t;          /* what src/misc.h:58-61 does where the compiler offers _Float16 */
#define HAS_FLOAT16 = 1
#include "import.h"
__attribute__((visibility("default"))) void ref_export_f16(uint8_t *out, float const *in, size_t count, int big_endian) {
  if (big_endian) export_f16_be(out, in, count); else export_f16_le(out, in, count);
}
__attribute__((visibility("default"))) void ref_import_f16(float *out, uint8_t const *in, size_t count, int big_endian) {
  if (big_endian) import_f16_be(out, in, count); else import_f16_le(out, in, count);
}
