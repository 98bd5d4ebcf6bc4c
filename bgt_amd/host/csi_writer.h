/* csi_writer.h -- coordinate-sorted index (CSI, min_shift / depth binning) of a BGZF-compressed BCF written record by
 * record, with the record-number trailer "RNI\1" the reference appends (an offset every 1 << 10 records; hts.h:71,
 * hts.c:541-547, used by bcf_seekn for `-i`).  One chunk per run of consecutive records in the same bin; bins carry the
 * offset of the first record overlapping their leftmost 1 << min_shift window.  The reference loads and queries it. */
#ifndef BGT_CSI_WRITER_H
#define BGT_CSI_WRITER_H
#include <stdint.h>

typedef struct csi_writer_s csi_writer_t;
csi_writer_t *csi_writer_init(int n_ref, int min_shift, int depth, uint64_t first_offset);
/* a record on contig rid covering [beg,end), stored at virtual offsets [off0,off1); records come sorted */
void csi_writer_push(csi_writer_t *w, int rid, int64_t beg, int64_t end, uint64_t off0, uint64_t off1);
int  csi_writer_save(csi_writer_t *w, const char *path);        /* 0, or -1 */
void csi_writer_destroy(csi_writer_t *w);
#endif
