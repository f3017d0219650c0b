/* oracle/shims/iniparser/iniparser.h -- TEST INFRASTRUCTURE: declaration-only stand-in so the reference's radio.c /
 * rx888.c compile where they lie (iniparser is an un-vendored dependency, debian/control).  Nothing here is ever
 * called by the oracle wrappers: the functions the wrappers export (estimate_noise, convert) never touch a config. */
#ifndef ORACLE_SHIM_INIPARSER_H
#define ORACLE_SHIM_INIPARSER_H
#include <stdio.h>
typedef struct _dictionary_ dictionary;
dictionary *iniparser_load(const char *ininame);
void iniparser_freedict(dictionary *d);
int iniparser_getnsec(const dictionary *d);
const char *iniparser_getsecname(const dictionary *d, int n);
const char *iniparser_getstring(const dictionary *d, const char *key, const char *def);
int iniparser_getint(const dictionary *d, const char *key, int notfound);
long int iniparser_getlongint(const dictionary *d, const char *key, long int notfound);
double iniparser_getdouble(const dictionary *d, const char *key, double notfound);
int iniparser_getboolean(const dictionary *d, const char *key, int notfound);
int iniparser_find_entry(const dictionary *ini, const char *entry);
int iniparser_getsecnkeys(const dictionary *d, const char *s);
const char **iniparser_getseckeys(const dictionary *d, const char *s, const char **keys);
void iniparser_dump(const dictionary *d, FILE *f);
void iniparser_dump_ini(const dictionary *d, FILE *f);
int iniparser_set(dictionary *ini, const char *entry, const char *val);
void iniparser_unset(dictionary *ini, const char *entry);
#endif
