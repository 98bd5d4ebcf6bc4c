/* metadata.h -- Flat Metadata Format (.spl sample files): "name<TAB>key:type:value..." per line, types
 * i (integer), f (real), anything else a string, no type = flag.  Struct layout as the reference's fmf.h
 * (sizeof(fmf_t) == 48 is relied upon by bgt-server.go). */
#ifndef BGT_METADATA_H
#define BGT_METADATA_H
#include <stdint.h>
#include "filter_expr.h"

#define FMF_FLAG 0
#define FMF_INT  1
#define FMF_REAL 2
#define FMF_STR  3

typedef struct { uint32_t key:28, type:4; union { int32_t i; float r; uint32_t s; } v; } fmf_meta_t;
typedef struct { char *name; int n_meta, m_meta; fmf_meta_t *meta; } fmf1_t;
typedef struct { int n_keys, m_keys; char **keys; int n_vals, m_vals; char **vals; int n_rows, m_rows; fmf1_t *rows; } fmf_t;

#ifdef __cplusplus
extern "C" {
#endif
fmf_t *fmf_read(const char *fn);                       /* plain or gzip text */
void   fmf_destroy(fmf_t *f);
int    fmf_test(const fmf_t *f, int row, kexpr_t *ke); /* does the row satisfy the expression? */
#ifdef __cplusplus
}
#endif
#endif
