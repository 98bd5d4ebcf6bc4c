/* INTEGRATION.md section A as running code: the reader half of the reference's codec interface (pbwt.h:35-88 --
 * pbf_open_r / pbf_subset / pbf_seek / pbf_read / pbf_close / pbf_get_*) on top of libbgt_hip.so.  The test links it with
 * the COMPILED REFERENCE's own objects (oracle/_ref/{view,bgt,vcf,hts,bgzf,fmf,kexpr,bedidx}.o, i.e. everything of
 * `bgt view` except pbwt.o) so that the reference's unmodified bgt.c / view.c run on the MI355X codec.
 * Test infrastructure: nothing under bgt_amd/ uses this file.  The prototypes are restated here (the reference's
 * header does not travel to the GPU box); they are the interface, not code. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "bgt_hip.h"

typedef struct pbf_s { bgth_pbf_t *img; bgth_reader_t *rd; } pbf_t;

pbf_t *pbf_open_r(const char *fn)                      /* pbwt.c:221 */
{
    pbf_t *pb = (pbf_t*)calloc(1, sizeof(*pb));
    if ((pb->img = bgth_pbf_open(fn, 0)) == 0 || (pb->rd = bgth_reader_create(pb->img)) == 0) {
        fprintf(stderr, "%s\n", bgth_last_error());    /* reference convention: NULL + message */
        if (pb->img) bgth_pbf_close(pb->img);
        free(pb);
        return 0;
    }
    return pb;
}
int pbf_subset(pbf_t *pb, int n_sub, int *sub)         /* pbwt.c:374; called by bgt_prepare, bgt.c:243 */
{   return bgth_reader_select(pb->rd, n_sub, sub, 0, 1); }
int pbf_seek(pbf_t *pb, uint64_t k)                    /* pbwt.c:349 */
{   return bgth_reader_seek(pb->rd, (int64_t)k); }
const uint8_t **pbf_read(pbf_t *pb)                    /* pbwt.c:313: g plane pointers, valid until the next call */
{   return bgth_reader_read(pb->rd); }
int pbf_close(pbf_t *pb)                               /* pbwt.c:264 */
{   if (pb) { bgth_reader_destroy(pb->rd); bgth_pbf_close(pb->img); free(pb); } return 0; }
int pbf_get_m(const pbf_t *pb) { return bgth_pbf_get_m(pb->img); }
int pbf_get_g(const pbf_t *pb) { return bgth_pbf_get_g(pb->img); }
int pbf_get_n(const pbf_t *pb) { return (int)bgth_pbf_get_n(pb->img); }
int pbf_get_shift(const pbf_t *pb) { return bgth_pbf_get_shift(pb->img); }

/* the front end of the reference, unchanged: view.o */
int main_view(int argc, char *argv[]);
int main(int argc, char *argv[]) { return main_view(argc - 1, argv + 1); }     /* as main.c:28-43 dispatches `bgt view ...` */
