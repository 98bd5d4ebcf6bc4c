/* kstr.h -- growable byte string with the memory layout of the reference's kstring_t
 * (kstring.h: size_t l, m; char *s) so that structs embedding it keep their ABI offsets. */
#ifndef BGT_KSTR_H
#define BGT_KSTR_H
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { size_t l, m; char *s; } kstring_t;

static inline void ks_need(kstring_t *k, size_t extra)
{
    if (k->l + extra + 1 > k->m) {
        size_t m = k->m ? k->m : 32;
        while (m < k->l + extra + 1) m <<= 1;
        k->s = (char*)realloc(k->s, m);
        k->m = m;
    }
}
static inline void ks_putn(kstring_t *k, const void *p, size_t n)
{
    ks_need(k, n);
    memcpy(k->s + k->l, p, n);
    k->l += n;
    k->s[k->l] = 0;
}
static inline void ks_puts(kstring_t *k, const char *p) { ks_putn(k, p, strlen(p)); }
static inline void ks_putc(kstring_t *k, int c) { ks_need(k, 1); k->s[k->l++] = (char)c; k->s[k->l] = 0; }
static inline void ks_puti(kstring_t *k, long long v)       /* decimal, as kputw/kputl print */
{
    char buf[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) buf[n++] = '-';
    ks_need(k, (size_t)n);
    while (n) k->s[k->l++] = buf[--n];
    k->s[k->l] = 0;
}
static inline void ks_printf(kstring_t *k, const char *fmt, ...)
{
    va_list ap;
    int n;
    va_start(ap, fmt);
    n = vsnprintf(NULL, 0, fmt, ap);
    va_end(ap);
    ks_need(k, (size_t)n);
    va_start(ap, fmt);
    vsnprintf(k->s + k->l, (size_t)n + 1, fmt, ap);
    va_end(ap);
    k->l += (size_t)n;
}
#endif
