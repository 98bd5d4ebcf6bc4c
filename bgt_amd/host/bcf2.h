/* bcf2.h -- the slice of BCF2/VCF the BGT read path needs: header dictionaries, the record container the
 * reader API fills (struct layouts are ABI: bgt-server.go and view.c reach into them),
 * typed-value encoding and VCF text formatting.  Written from the BCF2 specification; the reference's
 * vcf.c/vcf.h (an early htslib) define the same on-disk and in-memory layout. */
#ifndef BGT_BCF2_H
#define BGT_BCF2_H
#include <stdint.h>
#include "kstr.h"
#include "bgzf_io.h"

#define BCF_DT_ID     0
#define BCF_DT_CTG    1
#define BCF_DT_SAMPLE 2

#define BCF_BT_NULL  0
#define BCF_BT_INT8  1
#define BCF_BT_INT16 2
#define BCF_BT_INT32 3
#define BCF_BT_FLOAT 5
#define BCF_BT_CHAR  7

typedef struct { uint32_t info[3]; int id; } bcf_idinfo_t;          /* ref vcf.h:46-49; info[0] = contig length */
typedef struct { const char *key; const bcf_idinfo_t *val; } bcf_idpair_t;

typedef struct {                                                     /* 104 bytes, text at 72 (ref vcf.h:56-62) */
    int32_t l_text, m_text, n[3];
    bcf_idpair_t *id[3];
    void *dict[3];
    char *text;
    kstring_t mem;
} bcf_hdr_t;

typedef struct { int id, n, type, size; uint8_t *p; } bcf_fmt_t;
typedef struct { int key, type, len; union { int32_t i; float f; } v1; uint8_t *vptr; } bcf_info_t;
typedef struct {
    int m_fmt, m_info, m_str, m_allele, m_flt, n_flt;
    char *id, **allele;
    int *flt;
    bcf_info_t *info;
    bcf_fmt_t *fmt;
} bcf_dec_t;

typedef struct {                                                     /* 152 bytes, shared at 24, indiv at 48 */
    int32_t rid, pos, rlen;
    float qual;
    uint32_t n_info:16, n_allele:16;
    uint32_t n_fmt:8, n_sample:24;
    kstring_t shared, indiv;
    bcf_dec_t d;
    int unpacked;
    uint8_t *unpack_ptr;
} bcf1_t;

#ifdef __cplusplus
extern "C" {
#endif
bcf_hdr_t *bcf_hdr_init(void);
int        bcf_hdr_parse(bcf_hdr_t *h);              /* builds the dictionaries from h->text */
void       bcf_hdr_destroy(bcf_hdr_t *h);
bcf_hdr_t *bcf_hdr_read_stream(bgzr_t *fp);          /* "BCF\2\2", l_text, text */
int        bcf_id2int(const bcf_hdr_t *h, int which, const char *id);

bcf1_t *bcf_init1(void);
void    bcf_destroy1(bcf1_t *v);
int     bcf_read1_stream(bgzr_t *fp, bcf1_t *v);     /* 0, -1 at EOF, <-1 on error */

void bcf_enc_size(kstring_t *s, int size, int type);
void bcf_enc_int1(kstring_t *s, int32_t x);
void bcf_enc_vint(kstring_t *s, int n, const int32_t *a);
void bcf_enc_vchar(kstring_t *s, int l, const char *a);
int  bcf_append_info_ints(const bcf_hdr_t *h, bcf1_t *b, const char *key, int n, const int32_t *vals);
/* site-only record: empty ID, REF, first ALT, optional extra ALT, empty FILTER, QUAL 0 (ref vcf.c:1166-1182) */
void bcf_set_site(bcf1_t *b, int rid, int pos, int rlen, const char *ref, int l_ref, const char *alt, int l_alt,
                  const char *alt2);

int  bcf_dec_size(const uint8_t *p, const uint8_t **q, int *type);   /* typed value: size, type, payload */
int  vcf_format1(const bcf_hdr_t *h, const bcf1_t *v, kstring_t *s);
void vcf_hdr_write_text(FILE *fp, const bcf_hdr_t *h);
void bcf_hdr_write_stream(bgzw_t *fp, const bcf_hdr_t *h);
int  bcf_write1_stream(bgzw_t *fp, const bcf1_t *v);
#ifdef __cplusplus
}
#endif
#endif
