/*
 * bgt_hip.h -- C ABI of the MI355X-native BGT genotype-matrix read path (libbgt_hip.so).
 *
 * This is the drop-in boundary one level below the reader API of the reference: the "codec seam"
 * of pbwt.h (pbf_open_r / pbf_subset / pbf_seek / pbf_read / pbf_close, reference pbwt.h:35-88,
 * pbwt.c:221-388) fused with the per-site reduction the reader runs on the decoded planes
 * (bgtm_cal_info, reference bgt.c:735-757) and the 2-bit genotype packing that feeds bgt_gen_gt
 * (reference bgt.c:290-313).  Plain pointers and sizes only; no C++/torch types.
 *
 * Objects
 *   bgth_pbf_t     a .pbf image resident in HBM: RLE row strings, row directory and the rank form of
 *                  every 'S' checkpoint.  Read-only after open; may be shared by many readers
 *                  (the way one bgt_file_t is shared by many bgt_t, reference bgt.h:27,92).
 *   bgth_reader_t  per-reader state: column selection, group table, HIP stream, result buffers
 *                  (the reader half of pbf_t + bgt_t::out/group, reference pbwt.c:189-196, bgt.h:33-34).
 *
 * Error convention mirrors the reference: NULL / negative int on failure, message retrievable with
 * bgth_last_error() (the reference prints "[E::func]" from the caller, view.c:103-137).
 * Nothing in this library falls back to the CPU: without a usable HIP device every entry point that
 * touches genotypes fails.
 */
#ifndef BGT_HIP_H
#define BGT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bgth_pbf_s bgth_pbf_t;
typedef struct bgth_reader_s bgth_reader_t;

/* ---- library ---- */
const char *bgth_last_error(void);               /* thread-local, never NULL                       */
int         bgth_device_count(void);
/* Start the HIP runtime and load the kernels on a background thread (returns at once): for a process that will open an
 * image a moment later and has host-only work to do first (bgt view: headers, sample tables, the site side-car). */
void        bgth_runtime_warmup_async(int device);
/* Wait for that thread (no-op if none was started): call before the process exits on a path that may not have opened an
 * image -- exit handlers must not run under a runtime that is still starting. */
void        bgth_runtime_warmup_wait(void);
const char *bgth_version(void);

/* ---- .pbf image in HBM  (replaces pbf_open_r / pbf_close / pbf_get_*, pbwt.c:221-286,390-393) ----
 * Widths: any int32 m the reference opens (pbwt.c:92-105, 221-262).  Up to 327,000 columns (haplotypes) a row's two bit-vectors
 * with their rank directories sit in the LDS together, up to 650,000 one at a time; beyond, the producer's toggle words and the
 * walk's directory entries live in memory (L2) -- slower, same results.  Planes: BGT's two (import.c:68); a whole ONE-plane file (prefix.pb1 of
 * `import -1`) opens too, held with an empty second plane (bgth_pbf_get_g says 1, bgth_pbf_save writes one plane back).
 * Files of MORE than two planes (pbf_open_w / pbf_read loop over any g, pbwt.c:211-213, 325-334; `pbfview` is g-agnostic):
 * every plane is a PBWT of its own, so such a file opens as a bundle of two-plane images and serves the CODEC interface --
 * bgth_reader_select / seek / read hand out g byte planes per row, bgth_pbf_save writes the file back -- while counts, genotype
 * codes, checkpoints, partial and sharded images, which are defined for BGT's two planes, fail with a message. */
bgth_pbf_t *bgth_pbf_open(const char *path, int device);
bgth_pbf_t *bgth_pbf_open_mem(const void *image, size_t len, int device);
/* Partial image: only the 1<<shift-row blocks of the file that cover rows [row0,row1) are read (through the
 * footer's block index, pbwt.c:268-276) and uploaded -- what a region query (pbf_seek to a checkpoint + a few
 * pbf_read, pbwt.c:349-372) needs of a large file.  Row arguments of the reader stay FILE rows; rows outside the
 * loaded blocks fail loudly.  bgth_pbf_get_n() still reports the rows of the file. */
bgth_pbf_t *bgth_pbf_open_rows(const char *path, int64_t row0, int64_t row1, int device);
/* ONE database over SEVERAL devices (the site-range sharding of SURVEY.md 8e behind this boundary): the file's 1<<shift-row
 * blocks are dealt out as n_shards contiguous block ranges (bgth_shard_ranges), shard i is a partial image on HIP device
 * devices[i] (a device may be listed more than once: several shards on one GPU).  Readers of a sharded image run every
 * bgth_reader_scan / refill of the pull interface on all shards concurrently -- one host thread and stream per shard --
 * and deliver rows in file order; the per-shard results meet in the caller's host arrays (each device copies its rows to
 * their place: no device-to-device hop).  bgth_reader_scan_device GATHERS the shards' counts on shard 0's device (where
 * d_counts must live; the stream, if any, belongs to that device): every shard scans on its own device and stream, the
 * counts of shards on other devices travel by ncclSend / ncclRecv in one group (RCCL over xGMI; librccl is bound at first
 * use), those of shards on shard 0's device by a device copy, and the caller's stream waits for all of it -- so a consumer
 * enqueued behind the call (bgth_filter_apply_device) sees the gathered counts.  Bit planes are not gathered (NULL).
 * The multi-process form (one rank per GPU, RCCL all-gather of the counts) is bench.py --gpus N. */
bgth_pbf_t *bgth_pbf_open_sharded(const char *path, int n_shards, const int *devices);
int         bgth_pbf_n_shards(const bgth_pbf_t *p);      /* 0 for a single-device image                    */
/* ranges[2 i], ranges[2 i + 1] = rows [row0,row1) of shard i: ceil(blocks / n_shards) whole blocks each          */
void        bgth_shard_ranges(int64_t n_rows, int shift, int n_shards, int64_t *ranges);
/* rows between two checkpoints the image can start decoding from (2048, or the file's 1 << shift when it keeps no
 * sub-checkpoints): the unit of work of a scan -- a caller that scans in pieces gives each piece >= 256 of them */
int64_t     bgth_pbf_unit_rows(const bgth_pbf_t *p);
int64_t     bgth_pbf_first_row(const bgth_pbf_t *p);     /* first loaded row (0 for a full image)        */
int64_t     bgth_pbf_loaded_rows(const bgth_pbf_t *p);   /* number of loaded rows                         */
/* Build an image from bare RLE strings (row-major, plane-minor: row0/plane0,row0/plane1,row1/plane0..)
 * and derive every checkpoint on the device: the device-side
 * equivalent of what pbf_write records while encoding (pbwt.c:292-301).  len[i] is the byte length
 * of string i; strings are concatenated in `rle`.  g must be 2. */
bgth_pbf_t *bgth_pbf_from_rle(int m, int g, int shift, int64_t n_rows, const uint8_t *rle,
                              const uint32_t *len, int device);
/* Images built by bgth_pbf_from_rle derive their checkpoints in parallel (every file block from the identity order at
 * once, then one composition of rank maps per block) and keep the ranks after their last row:
 *   bgth_pbf_final_ranks   out[2][m]: rank of every column after the last row, per plane
 *   bgth_pbf_rebase        make the image start from start_ranks[2][m] (rank of every column before its first row) instead
 *                          of the identity order: every checkpoint and the final ranks are re-based by one gather.
 * Together they open the site-range shards of ONE database side by side, one per GPU: every rank builds its
 * shard from the identity order, the final ranks are exchanged, and shard r is re-based onto final(r-1) o ... o final(0).
 * No reader of the image may be scanning during bgth_pbf_rebase.  Return 0, or -1 with bgth_last_error. */
int         bgth_pbf_final_ranks(const bgth_pbf_t *p, int32_t *out);
/* out[2][m]: the rank of every column BEFORE row `row` (a multiple of the image's checkpoint spacing, 2048 rows unless
 * BGTH_SUB_SHIFT says otherwise): the checkpoint in the form the kernels start from (the inverse of an 'S' record,
 * reference pbwt.c:343).  Any image. */
int         bgth_pbf_ranks_at(const bgth_pbf_t *p, int64_t row, int32_t *out);
int         bgth_pbf_rebase(bgth_pbf_t *p, const int32_t *start_ranks);
/* Serialise an image back to the on-disk format (header, 'S'/'B' records, 'I' footer; pbwt.c:199-311).
 * Returns bytes written or <0. */
int64_t     bgth_pbf_save(const bgth_pbf_t *p, const char *path);
void        bgth_pbf_close(bgth_pbf_t *p);
int         bgth_pbf_get_m(const bgth_pbf_t *p);       /* columns = 2 * samples                     */
int         bgth_pbf_get_g(const bgth_pbf_t *p);       /* bit planes (2 for BGT)                    */
int         bgth_pbf_get_shift(const bgth_pbf_t *p);   /* checkpoint every 1<<shift rows            */
int64_t     bgth_pbf_get_n(const bgth_pbf_t *p);       /* rows                                      */
int64_t     bgth_pbf_hbm_bytes(const bgth_pbf_t *p);   /* device footprint                          */
int64_t     bgth_pbf_rle_bytes(const bgth_pbf_t *p);   /* total RLE payload                         */

/* ---- writer: rows of 2-bit codes -> .pbf  (replaces pbf_open_w / pbf_write / pbf_close, pbwt.h:35,57,49;
 *      pbwt.c:199-219, :288-311, :264-277; the row encoder pbc_enc_core pbwt.c:57-66 and the run-length bytes of
 *      pbwt.c:24-36 run on the device) ----
 * The image is byte for byte the file the reference writer produces from the same rows.  `codes` is a HOST array
 * [n_rows][m], bit k of a byte = the bit of plane k (what import.c:96-97 hands to pbf_write as g byte arrays).
 * A call is cut into units of 4096 rows (1024 above 32768 columns) that are encoded in parallel, so hand over rows in
 * bulk; m <= 2,097,152 columns (beyond 262,144 the row directories live in memory instead of the LDS: slower, same bytes).
 * No CPU path. */
typedef struct bgth_encoder_s bgth_encoder_t;
bgth_encoder_t *bgth_encoder_open(int32_t m, int32_t g, int32_t shift, int device);      /* NULL on failure       */
int             bgth_encoder_write(bgth_encoder_t *e, const uint8_t *codes, int64_t n_rows);   /* <0 on failure   */
/* the same rows with four columns to a byte (column c in bits 2 (c & 3) of byte c >> 2, (m + 3) / 4 bytes per row):
 * the 2-bit rows bgth_reader_scan hands out; a quarter of the bytes cross PCIe (g <= 2) */
int             bgth_encoder_write_packed(bgth_encoder_t *e, const uint8_t *packed, int64_t n_rows);
/* Streaming: the bytes of the file produced so far (header, records), malloc'd, and forgotten by the encoder -- write
 * them out and the image never has to fit the host memory; bgth_encoder_finish then returns the rest and the footer. */
int64_t         bgth_encoder_take(bgth_encoder_t *e, uint8_t **chunk);
int64_t         bgth_encoder_finish(bgth_encoder_t *e, uint8_t **image);   /* footer; bytes of the malloc'd image */
void            bgth_encoder_free_image(uint8_t *image);
void            bgth_encoder_close(bgth_encoder_t *e);
double          bgth_encoder_kernel_ms(const bgth_encoder_t *e);           /* device time of the encode kernels   */
const char     *bgth_encoder_last_error(void);

/* ---- reader ----
 * A reader belongs to one image and one caller thread at a time (its own HIP stream, selection tables, result buffers);
 * any number of readers may work on one image concurrently (reference bgt.h:27: many bgt_t over one bgt_file_t).  Readers
 * are pooled per image: bgth_reader_destroy hands the reader -- stream, events, device and pinned buffers -- back to its
 * image, bgth_reader_create takes one from there and resets it to the state of a new reader (all columns, one group, row 0),
 * so a resident process pays for those allocations once.  Destroy every reader before closing its image. */
bgth_reader_t *bgth_reader_create(bgth_pbf_t *p);
void           bgth_reader_destroy(bgth_reader_t *r);

/* Column selection (replaces pbf_subset, pbwt.c:374-388, as called by bgt_prepare, bgt.c:239-243) and
 * group table (bgt_t::group / bgtm_t::group, bgt.c:612-621).
 *   n_sub / sub   columns to decode in output order; n_sub<=0 or sub==NULL selects all m columns.
 *   group         optional, one 1-based group id per PAIR of output columns (= per sample), n_sub/2
 *                 entries, as bgtm_t::group; NULL = one group.
 *   n_groups      number of groups (1..32, BGT_MAX_GROUPS bgt.h:13).
 * Returns 0 or <0. */
int bgth_reader_select(bgth_reader_t *r, int n_sub, const int32_t *sub, const uint32_t *group,
                       int n_groups);
int bgth_reader_width(const bgth_reader_t *r);         /* output columns per row (n_sub or m)       */

/* Decode rows [row0,row1) of the selection and reduce each to allele counts.
 *   counts  host buffer int32[(row1-row0)][1+Gx][3] with Gx = n_groups>1 ? n_groups : 0; per entry
 *           {AN, AC, AC<M>} exactly as bgt_info_t an/ac[0]/ac[1] and gan/gac (bgt.h:44-47): entry 0 is
 *           the total, entries 1..G the groups.  May be NULL.
 *   gt      optional host buffer uint8[(row1-row0)][(width+3)/4]: 2-bit codes a1<<1|a0 of output
 *           column i at bits 2*(i&3) of byte i>>2 (the pair bgt_gen_gt reads, bgt.c:306-311).
 * Returns rows decoded or <0. */
int64_t bgth_reader_scan(bgth_reader_t *r, int64_t row0, int64_t row1, int32_t *counts, uint8_t *gt);

/* Same, results left in HBM (for callers that own device memory, e.g. a collective over xGMI):
 *   d_counts  device int32[(row1-row0)][1+Gx][3]
 *   d_h0/d_h1 optional device uint64[(row1-row0)][slot_words]: bit planes in SLOT order
 *             (see bgth_reader_slot_map); pass NULL to skip.
 *   stream    hipStream_t to enqueue on (NULL = the reader's own stream); the call is asynchronous
 *             with respect to the host when a stream is given. */
int64_t bgth_reader_scan_device(bgth_reader_t *r, int64_t row0, int64_t row1, void *d_counts,
                                void *d_h0, void *d_h1, void *stream);
int     bgth_reader_slot_words(const bgth_reader_t *r);           /* uint64 words per row and plane  */
int     bgth_reader_slot_map(const bgth_reader_t *r, int32_t *slot_of_output); /* width entries      */

/* Pull interface with the semantics of pbf_seek + pbf_read (pbwt.c:349-372, 313-337): one row per
 * call, byte-per-column planes valid until the next call; rows are decoded on the device in batches
 * and served from a host ring.  NULL at end of file. */
int             bgth_reader_seek(bgth_reader_t *r, int64_t row);
const uint8_t **bgth_reader_read(bgth_reader_t *r);
/* What the pull interface delivers per row besides the counts: want_planes is a mask of BGTH_WANT_* bits.
 * 0 keeps only the counts (the `-G` path: the returned array then holds two NULL plane pointers) and lets one
 * refill cover millions of rows; max_rows_ahead bounds a refill (0 = automatic).
 *   BGTH_WANT_PLANES  the byte-per-column planes of pbf_read (default)
 *   BGTH_WANT_GT8     the genotype vector bgt_gen_gt builds from them (reference bgt.c:290-313, :250): one int8
 *                     per haplotype, (allele+1)<<1, in output order -- the payload of the BCF GT field
 *   BGTH_WANT_GTTEXT  the same as VCF text (reference vcf.c:940-969 with FORMAT = GT): "\tA/B" per sample,
 *                     A, B in {0, 1, ., 2}; 4 characters per sample, no terminator
 * GT8 / GTTEXT need an even number of selected columns (whole samples). */
#define BGTH_WANT_PLANES 1
#define BGTH_WANT_GT8    2
#define BGTH_WANT_GTTEXT 4
#define BGTH_WANT_BITS   8   /* keep the window's bit planes in HBM for bgth_reader_fold_last; nothing more is copied out */
int             bgth_reader_config(bgth_reader_t *r, int want_planes, int64_t max_rows_ahead);
/* genotype vector / text of the row returned by the last bgth_reader_read (NULL unless configured) */
const int8_t   *bgth_reader_last_gt8(const bgth_reader_t *r);
const char     *bgth_reader_last_gt_text(const bgth_reader_t *r);
/* counts of the row returned by the last bgth_reader_read: int32[1+Gx][3] */
const int32_t  *bgth_reader_last_counts(const bgth_reader_t *r);
/* Allele-set reductions on the device -- what bgtm_read_core does per matched site for `bgt view -a ... -S / -H`
 * (reference bgt.c:859-876) -- over the row returned by the last bgth_reader_read of a reader configured with
 * BGTH_WANT_BITS (whole samples selected: an even number of columns, haplotypes 2s and 2s+1 of a sample adjacent):
 *   code >= 0:  carriers[s] += 1 for every output sample s with a haplotype of that 2-bit code (1 = carries the
 *               allele, 0 = a reference-allele query; bgt.c:862-869)
 *   bit  >= 0:  hap[i] |= 1 << bit for every output haplotype i of code 1 (bgt.c:871-874), bit <= 63
 * The accumulators (int32[width/2], uint64[width]) live in HBM and start at zero after bgth_reader_select and after
 * bgth_reader_take_folds, which copies them out (either pointer may be NULL); with a sharded image every shard folds
 * the rows it decoded and take adds / ors the shards together.  Return 0, or -1 with bgth_last_error. */
int bgth_reader_fold_last(bgth_reader_t *r, int code, int bit);
int bgth_reader_take_folds(bgth_reader_t *r, int32_t *carriers, uint64_t *hap);

/* Timing of the last scan on the device (HIP events on the launch stream), milliseconds:
 * out[0] = decode kernel, out[1] = finalize kernel, out[2] = whole enqueue..done. */
int bgth_reader_last_timing(const bgth_reader_t *r, float out[3]);
/* ... of shard `shard` of a reader over a sharded image alone (last_timing reports the slowest shard); -1: no such shard */
int bgth_reader_shard_timing(const bgth_reader_t *r, int shard, float out[3]);
/* Launch geometry of the last scan: out = {threads, cols_per_thread, slices, rows_per_batch,
 * lds_bytes, workgroups}. */
int bgth_reader_last_geometry(const bgth_reader_t *r, int out[6]);
/* Which kernels the last scan ran: out[0] = 2 for the plane-split kernels (a sparse selection of a wide cohort: one
 * workgroup per bit plane, two per CU, counts from the bit planes), 1 if it took the directory path (wide cohorts whose columns span several
 * workgroups: every row's {bits, ones before} directory is built once into an HBM arena by a producer kernel and the
 * column slices only walk it, pulling rows into LDS by LDS-DMA), 0 for the kernels that rebuild the row per workgroup;
 * out[1] = passes over the arena, out[2] = producer launches (0: the arena still held the rows from the previous scan
 * of this reader), out[3] = milliseconds of the first producer launch.  BGTH_DIR_ARENA_MB bounds the arena
 * (default: 60 % of the HBM free at first use). */
int bgth_reader_last_path(const bgth_reader_t *r, float out[4]);
/* Override the automatic launch geometry (0 = automatic). For tuning and tests. */
int bgth_reader_tune(bgth_reader_t *r, int threads, int cols_per_thread, int rows_per_batch);
/* Test hook: force kernel families that the library otherwise chooses by the shape of the cohort and the selection.  Every
 * family gives the same results; tests use this to run each of them on shapes the CPU oracle decodes in seconds, bench.py to
 * make every timed step of the directory path rebuild its rows.  Process-wide, flags OR-ed, 0 = automatic (the default).
 * The library reads no environment variable for this. */
enum {
    BGTH_FORCE_NO_TOGGLE_ARRAY          = 1,     /* team mode toggles in place (the path of cohorts too wide for the toggle array) */
    BGTH_FORCE_NO_EMPTY_PLANE_SHORTCUT  = 2,     /* never / always the kernels that skip the lookups of an all-zero plane 1 */
    BGTH_FORCE_EMPTY_PLANE_SHORTCUT     = 4,     /*   (reference pbwt.c:135-138) */
    BGTH_FORCE_DIRECTORY_PATH           = 32,    /* always / never: rows built once into an HBM arena + walk-only workgroups */
    BGTH_FORCE_NO_DIRECTORY_PATH        = 64,
    BGTH_FORCE_COLUMN_ORDER             = 8,     /* whole-cohort counts take the general path: slots in column order (not in plane-0 rank
                                                  * order per sub-block), all three counts per column (not n(code 3) alone); sparse
                                                  * selections gather their start ranks per workgroup (no compact per-selection table) */
    BGTH_FORCE_REBUILD_ROWS             = 128,   /* a scan never walks the arena the previous scan of the reader left */
    BGTH_FORCE_SEQUENTIAL_CHECKPOINTS   = 512,   /* bgth_pbf_from_rle derives its checkpoints block after block */
    BGTH_FORCE_RCCL_TO_SELF             = 1024,  /* a sharded scan gathers through RCCL even between shards of ONE device */
    BGTH_FORCE_NO_PLANE_SPLIT           = 2048,  /* never / always one workgroup per bit plane (sparse selections of wide cohorts) */
    BGTH_FORCE_PLANE_SPLIT              = 4096,
    BGTH_FORCE_THREE_PLANE_BUFFERS      = 8192   /* directory path: the walk-only kernels of cohorts too wide for four plane-row buffers in
                                                  * LDS (m > 160,000: a row's planes walked one after the other) on narrower ones */
};
void bgth_force_kernels(unsigned flags);

/* ---- site filter on the device (bgtm_pass_site_flt, reference bgt.c:700-719, for counts that stay in HBM) ----
 * The `-f` expression is parsed on the host (ke_parse) and exported as a reverse-Polish program (ke_export of
 * libbgt.so): op[i] = 0 integer constant ival[i], 1 real constant rval[i], 2 variable = counts[slot[i]] of the
 * site (slot = entry*3 + field in the layout of bgth_reader_scan: AN 0, AC 1, AN1 3, AC1 4, AN2 6 ...; -1 =
 * unbound, which fails every site as in the reference), 16+k = operator k.  d_flags: uint8 per site,
 * d_n_pass: one uint64 the number of passing sites is ADDED to. */
typedef struct bgth_filter_s bgth_filter_t;
bgth_filter_t *bgth_filter_create(int device, int n_items, const int32_t *op, const int64_t *ival,
                                  const double *rval, const int32_t *slot);
void bgth_filter_destroy(bgth_filter_t *f);
int  bgth_filter_apply_device(const bgth_filter_t *f, const void *d_counts, int64_t n_rows, int ints_per_row,
                              void *d_flags, void *d_n_pass, void *stream);

/* Diagnostics: stream `bytes` of HBM `repeats` times with `width`-byte loads per lane (4 or 16); used under
 * rocprofv3 to calibrate the FETCH_SIZE counter against a known byte count. */
int bgth_debug_stream_read(int device, size_t bytes, int width, int repeats);
#ifdef __cplusplus
}
#endif
#endif
