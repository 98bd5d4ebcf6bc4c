/* bgzf_io.h -- blocked-gzip streams (the container of .bcf): sequential reader and writer.
 * Only what the read path and the synthetic-cohort writer need; random access goes through the in-memory
 * site table instead of virtual file offsets (see sitetab.h). */
#ifndef BGT_BGZF_IO_H
#define BGT_BGZF_IO_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

typedef struct bgzr_s bgzr_t;
bgzr_t *bgzr_open(const char *path);                 /* NULL if unreadable or not gzip */
long    bgzr_read(bgzr_t *r, void *dst, size_t n);   /* bytes read (short at EOF), <0 on a corrupt stream */
int     bgzr_seek(bgzr_t *r, uint64_t voff);          /* to a virtual offset (block offset << 16 | offset in block) */
uint64_t bgzr_tell(bgzr_t *r);
void    bgzr_close(bgzr_t *r);

typedef struct bgzw_s bgzw_t;
void    bgzw_threads(bgzw_t *w, int n);                /* deflate batches of blocks on n threads (same bytes) */
bgzw_t *bgzw_open(FILE *fp, int level);              /* level -1 = zlib default, 0 = stored */
int     bgzw_write(bgzw_t *w, const void *src, size_t n);
uint64_t bgzw_tell(const bgzw_t *w);                 /* virtual offset: compressed<<16 | in-block */
int     bgzw_close(bgzw_t *w);                       /* flush, append the empty EOF block; does not fclose */
#endif
