/* oracle/shims/bsd/string.h -- TEST INFRASTRUCTURE.  The image ships libbsd's
 * runtime but not its headers; src/misc.h:16 includes <bsd/string.h> only for
 * these two prototypes. */
#ifndef ORACLE_SHIM_BSD_STRING_H
#define ORACLE_SHIM_BSD_STRING_H
#include <stddef.h>
size_t strlcpy(char *dst, const char *src, size_t siz);
size_t strlcat(char *dst, const char *src, size_t siz);
#endif
