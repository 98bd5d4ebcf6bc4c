/* bgt-server: the query server of the reference (bgt-server.go:129-427) over this repo's libbgt.so.
 *
 * The reference's server is a Go program that binds libbgt through cgo; there is no Go toolchain in this image, and the
 * program is nothing but a request handler around the reader API, so it is restated here in C: same command line
 * (-p PORT | $PORT, -m max genotypes per query, -d variant annotations, -g minimal group size), same parameters
 * (s r i n a f g C S H t), same call sequence per query (bgt-server.go:220-373: flags, then f r i n t a s, prepare,
 * test_mgs, header, read loop with the n / n_gt_read caps, haplotype counts and sample list, the trailing "*"), same
 * status codes and messages for its errors.  What the resident process buys here: the databases are opened once
 * (bgt_no_file = 1, :416) and their .pbf images stay in HBM, so a query pays neither the HIP start-up nor the image
 * build that dominate a `bgt view` process.
 *
 * One thread per connection; every query has its own bgtm_t over the shared bgt_file_t (the threading contract of the
 * library).  `--query STRING` answers one query on stdout without opening a socket: the differential
 * test links this same file with the compiled reference library and compares the two bodies.
 *
 * Only what the handler needs of HTTP is here: GET, the query string, `Connection: close`. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <errno.h>
#include <signal.h>
#include <time.h>
#include <unistd.h>
#include <pthread.h>
#include <strings.h>
#include <sys/time.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <sys/stat.h>
#include <limits.h>
#include <sched.h>
#include <netinet/in.h>
#include <arpa/inet.h>
#include "../../include/bgt_reader.h"

#ifndef BGS_REFERENCE_LIB
int bgt_file_preload(const bgt_file_t *bf);        /* extension of this repo's libbgt.so: image + site table now */
#endif

#define BGS_MAX_FILES 64

static bgt_file_t *g_files[BGS_MAX_FILES];
static const char *g_prefix[BGS_MAX_FILES];
static int g_n_files;
static fmf_t *g_vardb;
static uint64_t g_max_gt = 10000000;               /* bgt-server.go:127 */
static int g_min_group;

static long long now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------
 * the query string: key=value pairs, '+' and %XX decoded, a key may repeat (`s`), a bare key counts
 * as present (`S`, `H`, `g`, `C`)
 * ------------------------------------------------------------------------------------------------ */
typedef struct { char *key, *val; } pair_t;
typedef struct { int n; pair_t *a; char *buf; } form_t;

static int hexv(int c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }

static int unescape(char *s)                        /* in place; -1 on a bad escape */
{
    char *d = s;
    for (; *s; ++s) {
        if (*s == '+') *d++ = ' ';
        else if (*s == '%') {
            const int h = hexv((unsigned char)s[1]), l = h < 0 ? -1 : hexv((unsigned char)s[2]);
            if (l < 0) return -1;
            *d++ = (char)(h << 4 | l); s += 2;
        } else *d++ = *s;
    }
    *d = 0;
    return 0;
}

/* the server protects a literal "&&" of an expression from the parameter split (bgt-server.go:221) */
static char *protect_and(const char *q)
{
    const size_t n = strlen(q);
    char *out = (char*)malloc(n * 3 + 1), *d = out;
    size_t i;
    for (i = 0; i < n; ++i) {
        if (q[i] == '&' && q[i + 1] == '&') { memcpy(d, ".AND.", 5); d += 5; ++i; }
        else *d++ = q[i];
    }
    *d = 0;
    return out;
}

static void form_parse(form_t *f, const char *query)
{
    char *p;
    memset(f, 0, sizeof(*f));
    f->buf = protect_and(query);
    f->a = (pair_t*)calloc(strlen(f->buf) / 2 + 2, sizeof(pair_t));
    for (p = f->buf; p && *p;) {
        char *amp = strchr(p, '&'), *eq;
        if (amp) *amp = 0;
        if (*p) {
            eq = strchr(p, '=');
            if (eq) *eq = 0;
            if (unescape(p) == 0 && (eq == NULL || unescape(eq + 1) == 0) && *p) {   /* a pair with a bad escape is dropped */
                f->a[f->n].key = p;
                f->a[f->n++].val = eq ? eq + 1 : p + strlen(p);
            }
        }
        p = amp ? amp + 1 : NULL;
    }
}

static void form_free(form_t *f) { free(f->a); free(f->buf); }

static const char *form_first(const form_t *f, const char *key)
{
    int i;
    for (i = 0; i < f->n; ++i) if (strcmp(f->a[i].key, key) == 0) return f->a[i].val;
    return NULL;
}

/* .AND. .and. .OR. .or. -> && || (bgt-server.go:212-218) */
static char *replace_op(const char *t)
{
    static const char *from[4] = {".AND.", ".and.", ".OR.", ".or."};
    static const char *to[4] = {"&&", "&&", "||", "||"};
    char *out = (char*)malloc(strlen(t) + 1), *d = out;
    while (*t) {
        int k, hit = 0;
        for (k = 0; k < 4 && !hit; ++k) {
            const size_t n = strlen(from[k]);
            if (strncmp(t, from[k], n) == 0) { *d++ = to[k][0]; *d++ = to[k][1]; t += n; hit = 1; }
        }
        if (!hit) *d++ = *t++;
    }
    *d = 0;
    return out;
}

/* ------------------------------------------------------------------------------------------------
 * the handler
 * ------------------------------------------------------------------------------------------------ */
static void fmf_keys(FILE *w, const fmf_t *f)      /* as Go prints a []string */
{
    int i;
    fputc('[', w);
    for (i = 0; i < f->n_keys; ++i) fprintf(w, "%s%s", i ? " " : "", f->keys[i]);
    fputc(']', w);
}

static void help(FILE *w, const char *host)         /* bgt-server.go:160-210 */
{
    int i;
    fputs("Server Configuration\n====================\n\n", w);
    fputs("The following configurations were set when the server was launched. Clients can't override them.\n\n", w);
    fputs(" * BGT file prefix(es) and queryable sample annotations:\n", w);
    for (i = 0; i < g_n_files; ++i) { fprintf(w, "   - %s: ", g_prefix[i]); fmf_keys(w, g_files[i]->f); fputc('\n', w); }
    fputc('\n', w);
    if (g_vardb) { fputs(" * Queryable variant annotations: ", w); fmf_keys(w, g_vardb); fputs("\n\n", w); }
    else fputs(" * No variant annotations specified.\n\n", w);
    fputs(" * This server may report individual genotypes.\n\n", w);
    fprintf(w, " * Maximal genotypes processed internally per query: %llu\n\n", (unsigned long long)g_max_gt);
    fputs("Example Queries\n===============\n\n", w);
    fputs(" * Variants present in both FIN and CEU populations (.and. represents the logical AND operator):\n\n", w);
    fprintf(w, "   curl -s 'http://%s/?s=(population==\"FIN\")&s=(population==\"CEU\")&f=(AC1>0.and.AC2>0)'\n\n", host);
    if (g_vardb) {
        fputs(" * HIGH impact variants in the FIN population:\n\n", w);
        fprintf(w, "   curl -s 'http://%s/?a=(impact==\"HIGH\")&s=(population==\"FIN\")&f=(AC>0)'\n\n", host);
    }
    fputs(" * Tabular output: chromosome, 1-based start, end positions, REF, ALT alleles and ALT allele frequency:\n\n", w);
    fprintf(w, "   curl -s 'http://%s/?t=CHROM,POS,END,REF,ALT,AC/AN&f=(AN>0)&r=11:200,000-300,000'\n\n", host);
    fputs(" * Samples in FIN that have three specified alleles:\n\n", w);
    fprintf(w, "   curl -s 'http://%s/?a=,11:151344:1:G,11:110992:AACTT:A,11:160513::G&S&s=(population==\"FIN\")'\n\n", host);
    fputs("Accepted Parameters\n===================\n\n", w);
    fputs("Sample selection parameter:\n\n", w);
    fputs("  s EXPR  List of samples in a comma-leading comma-separate list (e.g. ,sample1,sample2) or an\n", w);
    fputs("          expression (e.g. s=population==\"FIN\"). There can be multiple 's' parameters. Each of\n", w);
    fputs("          them defines a sample group.\n\n", w);
    fputs("Site selection parameters:\n\n", w);
    fputs("  r STR   Region in a format like '11:200,000-300,000'\n\n", w);
    fputs("  i INT   Start from the i-th record; INT>0\n\n", w);
    fputs("  n INT   Read at most INT records\n\n", w);
    fputs("  a EXPR  List of alleles in a format similar to parameter 's'. An allele is specified by\n", w);
    fputs("          chr:1basedPos:refLen:alleleSeq. Conditions may not work unless the server is launched with\n", w);
    fputs("          a variant annotation database.\n\n", w);
    fputs("  f EXPR  Filters on per sample group allele counts. EXPR could include AC (primary allele count),\n", w);
    fputs("          AN (total called alleles), AC# (primary allele count of the #-th sample group) and AN#.\n\n", w);
    fputs("VCF output parameters:\n\n", w);
    fputs("  g       Output sample genotypes\n\n", w);
    fputs("  C       Output AC and AN VCF INFO fields. This parameter is automatically set if 's' is applied.\n\n", w);
    fputs("Non-VCF output parameters:\n\n", w);
    fputs("  S       Output samples having requested alleles (requiring parameter 'a')\n\n", w);
    fputs("  H       Output counts of haplotypes across requested alleles (requiring parameter 'a')\n\n", w);
    fputs("  t STR   Comma-separated list of fields in tabular output. Accepted variables:\n", w);
    fputs("          CHROM, POS, END, REF, ALT, AC, AN, AC#, AN# (# for a group number)\n\n", w);
}

typedef struct {
    bgtm_t *bm;
    int flag, max_read, vcf_out;
} query_t;

/* Everything of a query that can fail (bgt-server.go:226-326).  Returns the HTTP status; *msg = the error line. */
static int query_setup(const form_t *f, query_t *q, const char **msg)
{
    const char *v;
    char *t;
    int i, ret;
    bgtm_t *bm;
    q->flag = BGT_F_NO_GT; q->max_read = 2147483647; q->vcf_out = 1;
    q->bm = bm = bgtm_reader_init(g_n_files, g_files);
    bgtm_set_mgs(bm, g_min_group);
    if (form_first(f, "g")) q->flag &= ~BGT_F_NO_GT;
    if (form_first(f, "C") || form_first(f, "s")) q->flag |= BGT_F_SET_AC;
    if (form_first(f, "S")) q->flag |= BGT_F_CNT_AL;
    if (form_first(f, "H")) q->flag |= BGT_F_CNT_HAP;
    bgtm_set_flag(bm, q->flag);
    if (q->flag & (BGT_F_CNT_AL | BGT_F_CNT_HAP)) q->vcf_out = 0;
    if ((v = form_first(f, "f")) != NULL) {
        t = replace_op(v); ret = bgtm_set_flt_site(bm, t); free(t);
        if (ret != 0) { *msg = "400 Bad Request: failed to parse parameter 'f'"; return 400; }
    }
    if ((v = form_first(f, "r")) != NULL && bgtm_set_region(bm, v) < 0) {
        *msg = "400 Bad Request: failed to set region with parameter 'r'"; return 400;
    }
    if ((v = form_first(f, "i")) != NULL) {
        char *end;
        const long k = strtol(v, &end, 10);
        if (*v == 0 || *end || k < 1) { *msg = "400 Bad Request: failed to set start with parameter 'i'"; return 400; }
        bgtm_set_start(bm, k);
    }
    if ((v = form_first(f, "n")) != NULL) {             /* strconv.Atoi: anything but a number reads as 0 */
        char *end;
        const long k = strtol(v, &end, 10);
        q->max_read = (*v == 0 || *end || k > 2147483647L || k < -2147483647L) ? 0 : (int)k;
    }
    if ((v = form_first(f, "t")) != NULL) {
        q->vcf_out = 0;
        if (bgtm_set_table(bm, v) < 0) { *msg = "400 Bad Request: failed to parse tabular format with parameter 't'"; return 400; }
    }
    if ((v = form_first(f, "a")) != NULL) {
        t = replace_op(v); ret = bgtm_set_alleles(bm, t, g_vardb, NULL); free(t);
        if (ret < 0) { *msg = "400 Bad Request: failed to retrieve alleles with parameter 'a'"; return 400; }
        if (ret == 0) { *msg = "204 No Content: no alleles matching parameter 'a'"; return 204; }
    }
    for (i = 0; i < f->n; ++i) {
        if (strcmp(f->a[i].key, "s") != 0) continue;
        t = replace_op(f->a[i].val); ret = bgtm_add_group(bm, t); free(t);
        if (ret < 0) { *msg = "400 Bad Request: failed to set sample group with parameter 's'"; return 400; }
    }
    bgtm_prepare(bm);
    if (bgtm_test_mgs(bm) == 0) { *msg = "403 Forbidden: genotype summary can't be computed for small sample groups"; return 403; }
    return 200;
}

/* the body of a successful query (bgt-server.go:328-372) */
static void query_stream(query_t *q, FILE *w)
{
    bgtm_t *bm = q->bm;
    bcf1_t *b = bcf_init1();
    kstring_t s = {0, 0, 0};
    int n_read = 0;
    if (q->vcf_out) { fputs(bm->h_out->text, w); fputc('\n', w); }
    for (;;) {
        int rc;
        if (n_read > q->max_read || bm->n_gt_read > g_max_gt) break;
        if (ferror(w)) break;                                        /* the client is gone or stopped reading (send timeout): free the worker */
        if ((rc = bgtm_read(bm, b)) < 0) {
            /* -1 is the end of the data; anything below is a failure (a device error): the status line left long ago, so the
             * body says it -- a 200 that silently stops short would pass for a complete answer */
            if (rc < -1) { fprintf(w, "[E::bgt-server] reading stopped on an error (%d): the answer is incomplete\n", rc); fprintf(stderr, "[E::%s] bgtm_read returned %d\n", __func__, rc); }
            break;
        }
        if (q->vcf_out) { s.l = 0; vcf_format1(bm->h_out, b, &s); fwrite(s.s, 1, s.l, w); fputc('\n', w); }
        else if (bm->n_fields > 0) { fputs(bm->tbl_line.s, w); fputc('\n', w); }
        ++n_read;
    }
    if (!q->vcf_out && bm->n_aal > 0) {
        if (q->flag & BGT_F_CNT_HAP) {
            int n_hap;
            bgt_hapcnt_t *hc = bgtm_hapcnt(bm, &n_hap);
            char *t = bgtm_hapcnt_print_destroy(bm, n_hap, hc);
            if (t) { fputs(t, w); free(t); }
        }
        if (q->flag & BGT_F_CNT_AL) {
            char *t = bgtm_alcnt_print(bm);
            if (t) { fputs(t, w); free(t); }
        }
    }
    if (n_read > q->max_read || bm->n_gt_read > g_max_gt) fputs("*\n", w);
    free(s.s);
    bcf_destroy1(b);
}

/* one query to `w`; with_http: status line and headers first.  Returns the status. */
static int answer(const char *query, const char *host, FILE *w, int with_http)
{
    form_t f;
    query_t q;
    const char *msg = NULL;
    int status = 200;
    const long long t0 = now_ns();
    fprintf(stderr, "[%lld] got request: %s\n", t0, query);
    form_parse(&f, query);
    memset(&q, 0, sizeof(q));
    if (f.n > 0) status = query_setup(&f, &q, &msg);
    const long long t1 = now_ns();
    if (with_http) {
        fprintf(w, "HTTP/1.1 %d %s\r\nContent-Type: text/plain; charset=utf-8\r\n%sConnection: close\r\n\r\n", status,
                status == 200 ? "OK" : status == 204 ? "No Content" : status == 403 ? "Forbidden" : "Bad Request",
                status == 200 ? "" : "X-Content-Type-Options: nosniff\r\n");
    }
    if (f.n == 0) help(w, host);
    else if (status == 200) query_stream(&q, w);
    else if (status != 204) { fputs(msg, w); fputc('\n', w); }        /* (a 204 carries no body) */
    const long long t2 = now_ns();
    if (q.bm) bgtm_reader_destroy(q.bm);
    form_free(&f);
    fflush(w);
    if (getenv("BGS_TRACE")) fprintf(stderr, "[bgs trace] setup %.2f ms, body %.2f ms, destroy %.2f ms\n", (t1 - t0) * 1e-6, (t2 - t1) * 1e-6, (now_ns() - t2) * 1e-6);
    fprintf(stderr, "[%lld] responded %lld\n", now_ns(), t0);
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * HTTP, GET only.  A bounded pool of workers takes accepted connections from a bounded queue: the number of queries in
 * flight never exceeds the pooled device readers of an image (beyond it every query would create and free a reader with its
 * stream and HBM windows: 1,105 queries/s at 16 clients fell to 456 at 64 with a thread per connection), a full queue
 * answers 503 at once instead of piling threads up, and a client that stops reading its answer loses the connection after
 * the send timeout instead of holding a reader for ever.
 * ------------------------------------------------------------------------------------------------ */
#define BGS_QUEUE 1024
static int g_queue[BGS_QUEUE], g_q_head = 0, g_q_len = 0;
static pthread_mutex_t g_q_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_q_cond = PTHREAD_COND_INITIALIZER;

static void *serve_connection(void *arg)
{
    const int fd = (int)(intptr_t)arg;
    char *req = (char*)malloc(65536), *q, *sp, *host = NULL, *line;
    size_t n = 0;
    FILE *w;
    struct timeval tvs = {30, 0};                                    /* a client that stops reading its answer: see below */
    const long long deadline = now_ns() + 10LL * 1000000000LL;       /* the whole request within 10 s, however it trickles in */
    int complete = 0;
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tvs, sizeof(tvs));
    while (n < 65535) {
        const long long left = deadline - now_ns();
        struct timeval tv;
        ssize_t k;
        if (left <= 0) { complete = -1; break; }
        tv.tv_sec = (time_t)(left / 1000000000LL); tv.tv_usec = (suseconds_t)(left % 1000000000LL / 1000);
        if (tv.tv_sec == 0 && tv.tv_usec == 0) tv.tv_usec = 1;
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        k = read(fd, req + n, 65535 - n);
        if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) { complete = -1; break; }   /* the deadline passed */
        if (k <= 0) break;                                           /* the client closed its side: take what is there */
        n += (size_t)k; req[n] = 0;
        if (strstr(req, "\r\n\r\n") || strstr(req, "\n\n")) { complete = 1; break; }
    }
    req[n] = 0;
    if (complete < 0) {                 /* never finished its request: no query is run for it */
        static const char late[] = "HTTP/1.1 408 Request Timeout\r\nConnection: close\r\n\r\n";
        ssize_t wr = send(fd, late, sizeof(late) - 1, MSG_NOSIGNAL | MSG_DONTWAIT);
        (void)wr; close(fd); free(req);
        return NULL;
    }
    w = fdopen(fd, "w");
    if (w == NULL) { close(fd); free(req); return NULL; }
    if (strncmp(req, "GET ", 4) != 0) {
        fputs("HTTP/1.1 405 Method Not Allowed\r\nAllow: GET\r\nConnection: close\r\n\r\n", w);
    } else {
        for (line = req; (line = strstr(line, "\n")) != NULL;) {      /* Host: for the help text */
            ++line;
            if (strncasecmp(line, "Host:", 5) == 0) {
                host = line + 5;
                while (*host == ' ') ++host;
                host[strcspn(host, "\r\n")] = 0;
                break;
            }
        }
        q = req + 4;
        sp = q + strcspn(q, " \r\n"); *sp = 0;                        /* the request target */
        q = strchr(q, '?');
        answer(q ? q + 1 : "", host ? host : "localhost", w, 1);
    }
    /* a send that timed out leaves the stream in error: every later flush would block for the timeout again (glibc retries
     * the write), so the socket is shut down first -- the final flush inside fclose() then fails at once */
    if (ferror(w)) shutdown(fd, SHUT_RDWR);
    fclose(w);                                                        /* closes fd */
    free(req);
    return NULL;
}

static void *worker_main(void *arg)
{
    (void)arg;
    for (;;) {
        int fd;
        pthread_mutex_lock(&g_q_lock);
        while (g_q_len == 0) pthread_cond_wait(&g_q_cond, &g_q_lock);
        fd = g_queue[g_q_head]; g_q_head = (g_q_head + 1) % BGS_QUEUE; --g_q_len;
        pthread_mutex_unlock(&g_q_lock);
        serve_connection((void*)(intptr_t)fd);
    }
    return NULL;
}

#ifndef BGS_REFERENCE_LIB
/* ------------------------------------------------------------------------------------------------
 * -u SOCKET: a resident host for `bgt view` command lines (include/bgt_reader.h: view_run).  A `bgt view` process
 * started with BGT_SERVER=SOCKET sends its arguments, its working directory and its own stdout / stderr as file
 * descriptors; the query runs here, on images that are already in HBM, writes straight into the client's descriptors
 * and the answer on the socket is the exit status -- the bytes are those of a local run, the HIP start-up and the
 * image build (140-270 ms per process) are paid once per database.  Databases are opened on first
 * use (any prefix a client names, resolved in ITS working directory) and stay open; the ones on this command line are
 * opened at start.  This mode serves the user who started it (a unix socket with that user's permissions): arguments
 * may name files, so bgt_no_file stays 0 and no HTTP port is opened beside it.
 * ------------------------------------------------------------------------------------------------ */
#define BGS_CACHE_MAX 256
/* One resident image per database path.  Queries hold references (cache_open / cache_close); an entry whose files changed on
 * disk is re-pointed to a fresh bgt_open and the old image RETIRES: it stays with the queries that are reading it and is
 * closed (its HBM given back) when the last of them returns. */
typedef struct { bgt_file_t *bf; int refs; } image_t;
typedef struct { char *path; image_t *img; struct timespec mtime[3]; } cache_ent_t;
static cache_ent_t g_cache[BGS_CACHE_MAX];
static int g_n_cache;
static image_t **g_retired;
static int g_n_retired, g_m_retired;
static pthread_mutex_t g_cache_lock = PTHREAD_MUTEX_INITIALIZER;

/* The three files of a database are named by its PREFIX (prefix.pbf / .bcf / .spl, reference bgt.c:44-58) -- as the client
 * spelt it, made absolute: a `prefix.pbf` that is a symlink to a differently named file still has its .bcf / .spl beside
 * the LINK, so nothing is derived from a resolved name.  Modification times of the three; 0 on success. */
static int trio_mtimes(const char *abs_prefix, struct timespec mt[3])
{
    static const char *const ext[3] = {"pbf", "bcf", "spl"};
    char fn[2 * PATH_MAX + 8];
    int k;
    for (k = 0; k < 3; ++k) {
        struct stat st;
        snprintf(fn, sizeof(fn), "%s.%s", abs_prefix, ext[k]);
        memset(&mt[k], 0, sizeof(mt[k]));
        if (stat(fn, &st) != 0) { if (k == 0) return -1; continue; }
        mt[k] = st.st_mtim;
    }
    return 0;
}

/* the cache key of a database: where its three files really are (two spellings of one database share an image; two
 * prefixes that share a .pbf but not a .spl do not) */
static int trio_key(const char *abs_prefix, char *key, size_t cap)
{
    static const char *const ext[3] = {"pbf", "bcf", "spl"};
    char fn[2 * PATH_MAX + 8], real[PATH_MAX];
    size_t l = 0;
    int k;
    for (k = 0; k < 3; ++k) {
        snprintf(fn, sizeof(fn), "%s.%s", abs_prefix, ext[k]);
        if (realpath(fn, real) == NULL) { if (k == 0) return -1; real[0] = 0; }
        if (l + strlen(real) + 2 > cap) return -1;
        l += (size_t)sprintf(key + l, "%s\n", real);
    }
    return 0;
}

static bgt_file_t *cache_open(const char *prefix, void *ctx)
{
    char full[2 * PATH_MAX], key[3 * PATH_MAX + 8];
    struct timespec mt[3];
    image_t *img = NULL;
    int i;
    (void)ctx;
    /* (relative to the thread's own working directory: see unix_worker.  Absolute from here on: the image outlives the query
     *  and loads its site table lazily, maybe under another query's directory.) */
    if (prefix[0] == '/') { if (snprintf(full, sizeof(full), "%s", prefix) >= (int)sizeof(full)) return NULL; }
    else {
        char cwd[PATH_MAX];
        if (getcwd(cwd, sizeof(cwd)) == NULL || snprintf(full, sizeof(full), "%s/%s", cwd, prefix) >= (int)sizeof(full)) return NULL;
    }
    if (trio_key(full, key, sizeof(key)) != 0 || trio_mtimes(full, mt) != 0) return NULL;
    pthread_mutex_lock(&g_cache_lock);
    for (i = 0; i < g_n_cache; ++i)
        if (strcmp(g_cache[i].path, key) == 0) {
            if (memcmp(g_cache[i].mtime, mt, sizeof(mt)) == 0) img = g_cache[i].img;
            break;
        }
    if (img == NULL && (i < g_n_cache || g_n_cache < BGS_CACHE_MAX)) {
        bgt_file_t *bf;
        if ((bf = bgt_open(full)) != NULL) {
            if (bgt_file_preload(bf) < 0) fprintf(stderr, "[W::%s] '%s' is not resident yet; the first query will load it\n", __func__, full);
            img = (image_t*)calloc(1, sizeof(image_t));
            img->bf = bf;
            if (i == g_n_cache) { g_cache[i].path = strdup(key); g_cache[i].img = NULL; ++g_n_cache; }
            if (g_cache[i].img) {                                    /* the database was rewritten: the old image retires */
                image_t *old = g_cache[i].img;
                if (old->refs == 0) { bgt_close(old->bf); free(old); }
                else {
                    if (g_n_retired == g_m_retired) {
                        g_m_retired = g_m_retired ? 2 * g_m_retired : 8;
                        g_retired = (image_t**)realloc(g_retired, (size_t)g_m_retired * sizeof(image_t*));
                    }
                    g_retired[g_n_retired++] = old;
                }
            }
            g_cache[i].img = img; memcpy(g_cache[i].mtime, mt, sizeof(mt));
        }
    }
    if (img) ++img->refs;
    pthread_mutex_unlock(&g_cache_lock);
    return img ? img->bf : NULL;
}

static void cache_close(bgt_file_t *bf, void *ctx)
{
    int i;
    (void)ctx;
    pthread_mutex_lock(&g_cache_lock);
    for (i = 0; i < g_n_cache; ++i)
        if (g_cache[i].img && g_cache[i].img->bf == bf) { --g_cache[i].img->refs; break; }   /* (stays resident) */
    if (i == g_n_cache)
        for (i = 0; i < g_n_retired; ++i)
            if (g_retired[i]->bf == bf) {
                if (--g_retired[i]->refs == 0) { bgt_close(bf); free(g_retired[i]); g_retired[i] = g_retired[--g_n_retired]; }
                break;
            }
    pthread_mutex_unlock(&g_cache_lock);
}

static int recv_all(int fd, void *buf, size_t len)
{
    size_t k = 0;
    while (k < len) {
        const ssize_t n = recv(fd, (char*)buf + k, len - k, 0);
        if (n <= 0) { if (n < 0 && errno == EINTR) continue; return -1; }
        k += (size_t)n;
    }
    return 0;
}

/* the bytes of a spool file (written by a query that ran under g_cwd_lock) into the client's descriptor */
static void spool_to_fd(FILE *spool, int fd)
{
    char buf[1 << 16];
    size_t n;
    if (fd < 0 || fflush(spool) != 0) return;
    rewind(spool);
    while ((n = fread(buf, 1, sizeof(buf), spool)) > 0) {
        size_t k = 0;
        while (k < n) {
            const ssize_t w = write(fd, buf + k, n - k);
            if (w < 0) { if (errno == EINTR) continue; return; }      /* (EPIPE: the client left) */
            k += (size_t)w;
        }
    }
}

static __thread int t_private_cwd;                                  /* this thread's chdir() moves nobody else */
static pthread_mutex_t g_cwd_lock = PTHREAD_MUTEX_INITIALIZER;

/* only the user who runs the server may hand it command lines (they name files the server then reads) */
static int peer_is_me(int fd)
{
    struct ucred uc;
    socklen_t len = sizeof(uc);
    if (getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &uc, &len) != 0) return 0;
    return uc.uid == geteuid();
}

static void *unix_worker(void *arg)
{
    const int fd = (int)(intptr_t)arg;
    struct msghdr mh;
    struct iovec iov;
    union { struct cmsghdr h; char buf[CMSG_SPACE(2 * sizeof(int))]; } cm;
    struct cmsghdr *c;
    uint32_t body = 0;
    int fds[2] = {-1, -1}, argc = 0, i, rc = 1, cwd_locked = 0, spooled = 0;
    char *req = NULL, *p, *end, **argv = NULL;
    unsigned char status[2] = {'S', 1};
    FILE *out = NULL, *err = NULL;
    struct timeval tv = {10, 0};
    ssize_t n;
    static const bgt_view_host_t host = {cache_open, cache_close, NULL};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    if (!peer_is_me(fd)) { close(fd); return NULL; }
    memset(&mh, 0, sizeof(mh)); memset(&cm, 0, sizeof(cm));
    iov.iov_base = &body; iov.iov_len = 4;
    mh.msg_iov = &iov; mh.msg_iovlen = 1; mh.msg_control = cm.buf; mh.msg_controllen = sizeof(cm.buf);
    n = recvmsg(fd, &mh, MSG_CMSG_CLOEXEC);                           /* the descriptors arrive with the first bytes */
    if (n <= 0) goto done;
    for (c = CMSG_FIRSTHDR(&mh); c; c = CMSG_NXTHDR(&mh, c))
        if (c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS && c->cmsg_len == CMSG_LEN(sizeof(fds))) memcpy(fds, CMSG_DATA(c), sizeof(fds));
    if (n < 4 && recv_all(fd, (char*)&body + n, 4 - (size_t)n) < 0) goto done;
    if (fds[0] < 0 || fds[1] < 0 || body < 8 || body > (1u << 24)) goto done;
    req = (char*)malloc((size_t)body + 1);
    if (recv_all(fd, req, body) < 0) goto done;
    req[body] = 0; end = req + body;
    if (strcmp(req, "BGTV1") != 0) goto done;
    /* Without a private working directory (below) a query holds the process-wide g_cwd_lock while it runs: its answer then
     * goes to unlinked spool files and reaches the client AFTER the lock is released -- a client that stops reading its pipe
     * stalls its own thread, not every other query.  With a private directory: straight into the client's
     * descriptors. */
    if (!t_private_cwd) { out = tmpfile(); err = tmpfile(); spooled = 1; }
    else { out = fdopen(fds[0], "w"); err = fdopen(fds[1], "w"); }
    if (!out || !err) goto done;
    if (!spooled) setvbuf(err, NULL, _IONBF, 0);
    p = req + 6;
    /* this thread has a working directory of its own (unshare(CLONE_FS) when it started): relative paths in the
     * arguments -- databases, -B / -d / -s / -a files -- mean what they mean to the client.  Where the kernel refuses a
     * private one (a container without CAP_SYS_ADMIN: EPERM), chdir() moves the whole process: queries then run ONE AT A
     * TIME under g_cwd_lock, each in its client's directory -- slower, never in another client's directory. */
    if (!t_private_cwd) { pthread_mutex_lock(&g_cwd_lock); cwd_locked = 1; }
    if (chdir(p) != 0) { fprintf(err, "[E::main_view] the server cannot enter '%s'\n", p); goto done; }
    p += strlen(p) + 1;
    if (p >= end) goto done;
    argc = atoi(p); p += strlen(p) + 1;
    if (argc < 1 || argc > 4096) goto done;
    argv = (char**)calloc((size_t)argc + 1, sizeof(char*));
    for (i = 0; i < argc && p < end; ++i) { argv[i] = p; p += strlen(p) + 1; }
    if (i < argc) goto done;
    {
        const long long t0 = now_ns();
        rc = view_run(argc, argv, out, err, &host);
        if (getenv("BGS_TRACE")) fprintf(stderr, "[bgs trace] view query: %.2f ms, status %d\n", (now_ns() - t0) * 1e-6, rc);
    }
    status[1] = (unsigned char)rc;
done:
    if (cwd_locked) pthread_mutex_unlock(&g_cwd_lock);
    if (spooled) {                                                    /* deliver: stderr first (small), then the body */
        if (err) { spool_to_fd(err, fds[1]); fclose(err); err = NULL; }
        if (out) { spool_to_fd(out, fds[0]); fclose(out); out = NULL; }
    }
    if (out) fclose(out); else if (fds[0] >= 0) close(fds[0]);        /* everything is written before the status leaves */
    if (err) fclose(err); else if (fds[1] >= 0) close(fds[1]);
    n = send(fd, status, 2, MSG_NOSIGNAL);
    (void)n;
    close(fd);
    free(argv); free(req);
    return NULL;
}

static void *unix_thread_main(void *arg)
{
    /* a private working directory per thread: chdir() in one query must not move the others */
    static int warned;
    t_private_cwd = getenv("BGS_NO_PRIVATE_CWD") == NULL && unshare(CLONE_FS) == 0;   /* (the variable: tests of the fallback) */
    if (!t_private_cwd && !__sync_lock_test_and_set(&warned, 1))
        fprintf(stderr, "[W::%s] unshare(CLONE_FS) failed (%s): queries run one at a time, each in its client's directory\n", __func__, strerror(errno));
    return unix_worker(arg);
}

static int serve_unix(const char *path)
{
    struct sockaddr_un sa;
    int srv;
    if (strlen(path) >= sizeof(sa.sun_path)) { fprintf(stderr, "[E::%s] socket path too long\n", __func__); return 1; }
    signal(SIGPIPE, SIG_IGN);
    srv = socket(AF_UNIX, SOCK_STREAM, 0);
    memset(&sa, 0, sizeof(sa)); sa.sun_family = AF_UNIX; strcpy(sa.sun_path, path);
    unlink(path);
    {   /* the socket belongs to this user alone, whatever the umask (and every peer's uid is checked: peer_is_me) */
        const mode_t um = umask(077);
        const int rc_bind = srv < 0 ? -1 : bind(srv, (struct sockaddr*)&sa, sizeof(sa));
        umask(um);
        if (rc_bind == 0) chmod(path, 0600);
        if (rc_bind < 0) { fprintf(stderr, "[E::%s] cannot listen on '%s': %s\n", __func__, path, strerror(errno)); return 1; }
    }
    if (listen(srv, 128) < 0) {
        fprintf(stderr, "[E::%s] cannot listen on '%s': %s\n", __func__, path, strerror(errno));
        return 1;
    }
    fprintf(stderr, "[%lld] launched at socket %s\n", now_ns(), path);
    for (;;) {
        const int fd = accept(srv, NULL, NULL);
        pthread_t th;
        if (fd < 0) { if (errno == EINTR) continue; break; }
        if (pthread_create(&th, NULL, unix_thread_main, (void*)(intptr_t)fd) == 0) pthread_detach(th);
        else close(fd);
    }
    return 0;
}
#endif

static int usage(const char *port)
{
    fprintf(stderr, "Usage: bgt-server [options] <bgt.pre1> [...]\n");
    fprintf(stderr, "Options:\n");
    fprintf(stderr, "  -p INT    port number [%s or from $PORT env]\n", port);
    fprintf(stderr, "  -m INT    maximal genotypes processed per query [%llu]\n", (unsigned long long)g_max_gt);
    fprintf(stderr, "  -d FILE   variant annotations in the FMF format []\n");
    fprintf(stderr, "  -g INT    minimal sample group size (force -G if positive) [0]\n");
    fprintf(stderr, "  -q STR    answer this one query string on stdout and exit (no socket)\n");
#ifndef BGS_REFERENCE_LIB
    fprintf(stderr, "  -u PATH   serve `bgt view` command lines on this unix socket instead of HTTP (clients: BGT_SERVER=PATH bgt view ...)\n");
#endif
    return 1;
}

int main(int argc, char **argv)
{
    const char *port = getenv("PORT") && *getenv("PORT") ? getenv("PORT") : "8000", *one_query = NULL, *unix_path = NULL;
    int c, i, srv, on = 1;
    struct sockaddr_in addr;
    while ((c = getopt(argc, argv, "d:p:m:g:q:u:")) >= 0) {
        if (c == 'p') port = optarg;
        else if (c == 'm') g_max_gt = strtoull(optarg, NULL, 10);
        else if (c == 'd') g_vardb = fmf_read(optarg);
        else if (c == 'g') g_min_group = atoi(optarg);
        else if (c == 'q') one_query = optarg;
        else if (c == 'u') unix_path = optarg;
    }
#ifndef BGS_REFERENCE_LIB
    if (unix_path) {                                                  /* a host for `bgt view` clients; databases named here are made resident now */
        char cwd[PATH_MAX];
        if (getcwd(cwd, sizeof(cwd)) == NULL) cwd[0] = 0;
        for (i = optind; i < argc; ++i)
            if (cache_open(argv[i], NULL) == NULL) { fprintf(stderr, "[E::%s] failed to open '%s'\n", __func__, argv[i]); return 1; }
        return serve_unix(unix_path);
    }
#endif
    (void)unix_path;
    if (optind == argc) return usage(port);
    bgt_no_file = 1;                                                  /* bgt-server.go:416: arguments are never file names */
    if (argc - optind > BGS_MAX_FILES) { fprintf(stderr, "[E::%s] %d databases given, at most %d are served\n", __func__, argc - optind, BGS_MAX_FILES); return 1; }
    for (i = optind; i < argc && g_n_files < BGS_MAX_FILES; ++i) {
        const char *base = strrchr(argv[i], '/');
        if ((g_files[g_n_files] = bgt_open(argv[i])) == NULL) { fprintf(stderr, "[E::%s] failed to open '%s'\n", __func__, argv[i]); return 1; }
        g_prefix[g_n_files++] = base ? base + 1 : argv[i];
    }
#ifndef BGS_REFERENCE_LIB
    if (one_query == NULL)                                            /* resident images: no query pays for the load */
        for (i = 0; i < g_n_files; ++i)
            if (bgt_file_preload(g_files[i]) < 0) fprintf(stderr, "[W::%s] '%s' is not resident yet; the first query will load it\n", __func__, g_prefix[i]);
#endif
    if (one_query) {
        const int status = answer(one_query, "localhost", stdout, 0);
        for (i = 0; i < g_n_files; ++i) bgt_close(g_files[i]);
        return status == 200 ? 0 : status / 100;
    }
    signal(SIGPIPE, SIG_IGN);
    srv = socket(AF_INET, SOCK_STREAM, 0);
    setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &on, sizeof(on));
    memset(&addr, 0, sizeof(addr));
    addr.sin_family = AF_INET; addr.sin_addr.s_addr = htonl(INADDR_ANY); addr.sin_port = htons((uint16_t)atoi(port));
    if (srv < 0 || bind(srv, (struct sockaddr*)&addr, sizeof(addr)) < 0 || listen(srv, 128) < 0) {
        fprintf(stderr, "[E::%s] cannot listen on port %s: %s\n", __func__, port, strerror(errno));
        return 1;
    }
    {
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        int n_workers = getenv("BGS_WORKERS") ? atoi(getenv("BGS_WORKERS")) : (int)(ncpu < 4 ? 4 : ncpu > 32 ? 32 : ncpu), started = 0;
        if (n_workers < 1) n_workers = 1;
        if (n_workers > 256) n_workers = 256;
        for (i = 0; i < n_workers; ++i) {
            pthread_t th;
            if (pthread_create(&th, NULL, worker_main, NULL) == 0) { pthread_detach(th); ++started; }
        }
        if (started == 0) { fprintf(stderr, "[E::%s] cannot start a worker thread\n", __func__); return 1; }
        fprintf(stderr, "[%lld] launched at port %s (%d workers)\n", now_ns(), port, started);
    }
    for (;;) {
        const int fd = accept(srv, NULL, NULL);
        int full;
        if (fd < 0) { if (errno == EINTR) continue; break; }
        pthread_mutex_lock(&g_q_lock);
        full = g_q_len == BGS_QUEUE;
        if (!full) { g_queue[(g_q_head + g_q_len) % BGS_QUEUE] = fd; ++g_q_len; pthread_cond_signal(&g_q_cond); }
        pthread_mutex_unlock(&g_q_lock);
        if (full) {                                                   /* back-pressure: say so, do not queue without bound */
            static const char busy[] = "HTTP/1.1 503 Service Unavailable\r\nContent-Type: text/plain; charset=utf-8\r\nRetry-After: 1\r\nConnection: close\r\n\r\nserver busy\n";
            ssize_t wr = write(fd, busy, sizeof(busy) - 1);
            (void)wr;
            close(fd);
        }
    }
    return 0;
}
