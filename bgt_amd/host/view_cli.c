/* view_cli.c -- `bgt view`: option handling and the pull loop of reference view.c:14-183, on the MI355X
 * reader. */
#include <limits.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <signal.h>
#include <sys/wait.h>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <pthread.h>
#include "../../include/bgt_reader.h"
#include "../../include/bgt_hip.h"

static int usage(const char *cmd, FILE *e)
{
    fprintf(e, "Usage: bgt %s [options] <bgt-prefix> [...]\n", cmd);
    fprintf(e, "Options:\n");
    fprintf(e, "  -s EXPR   sample group: ,name1,name2 | file | expression on .spl metadata (repeatable)\n");
    fprintf(e, "  -r STR    region chr[:beg-end]\n");
    fprintf(e, "  -i INT    start from the INT-th site (1-based)      -n INT   emit at most INT sites\n");
    fprintf(e, "  -f STR    site filter on AC, AN, AC#, AN# (e.g. 'AC>0', 'AC1/AN1>=0.1&&AC2==0')\n");
    fprintf(e, "  -G        no sample genotypes     -C   write AC/AN (implied by -f or several -s)\n");
    fprintf(e, "  -b        BCF output   -l INT   compression level   -u   uncompressed BCF\n");
    fprintf(e, "  -B FILE   sites overlapping the BED intervals   -e   ... not overlapping\n");
    fprintf(e, "  -a EXPR   allele set: ,chr:pos:rlen:alt,... | ,chr:pos:REF:ALT | file   -S   samples carrying all of them\n");
    fprintf(e, "  -d FILE   variant annotations (FMF): -a EXPR then selects its rows by metadata   -M   load FILE in memory\n");
    fprintf(e, "  -H        haplotype counts over the allele set   -t STR   table of comma-separated expressions\n");
    return 1;
}

/* BGT_TRACE=1: wall time of the stages of a `bgt view` on stderr */
static double view_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void view_lap(double *t0, const char *what)
{
    const double t1 = view_now();
    if (getenv("BGT_TRACE")) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);                      /* (epoch ms: lines up with a `date +%s%N` around the process) */
        fprintf(stderr, "[bgt trace] %-34s %8.2f ms   @%lld\n", what, t1 - *t0, (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000);
    }
    *t0 = t1;
}

/* A query that uses the device leaves HBM allocations, queues and a runtime behind whose tear-down costs the process 70-180 ms
 * AFTER its last byte is written (measured: scripts/region_trace.sh, scripts/cli_time_c4shard.sh) -- time the caller's shell
 * would wait for nothing.  So the work runs in a child: the process the shell started only waits for one byte, the exit
 * status, which the child sends after it has flushed and closed stdout and stderr, and leaves with it; the child finishes its
 * tear-down on its own.  A child that ends any other way (an error return, BGT_CLEAN_EXIT=1, a signal) is waited for and its
 * status or signal handed on unchanged; a parent that is killed takes the child with it (PR_SET_PDEATHSIG).  Forked before the first thread and before the HIP runtime exists.  BGT_NO_FORK=1:
 * one process as before. */
static int g_done_fd = -1;
static void work_in_a_child(void)
{
    int fd[2];
    pid_t pid;
    if (getenv("BGT_NO_FORK") || pipe(fd) != 0) return;
    fflush(stdout); fflush(stderr);
    if ((pid = fork()) < 0) { close(fd[0]); close(fd[1]); return; }
    if (pid == 0) {
        /* a parent that is killed (a `timeout`, a closed terminal) takes the child with it; on the normal path the signal
         * arrives when the child has nothing left to do but release the device, which the kernel then does for it */
        const pid_t parent = getppid();
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        if (getppid() != parent) _exit(1);                         /* (it died in between) */
        close(fd[0]); g_done_fd = fd[1];
        return;
    }
    {
        unsigned char code = 0;
        ssize_t n;
        int st = 0;
        close(fd[1]);
        do n = read(fd[0], &code, 1); while (n < 0 && errno == EINTR);
        if (n == 1) _exit(code);                                   /* the answer is complete and out */
        while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
        if (WIFEXITED(st)) _exit(WEXITSTATUS(st));
        if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); kill(getpid(), WTERMSIG(st)); }
        _exit(1);
    }
}

#include "view_client.h"   /* view_via_server(): BGT_SERVER=<unix socket>, the query handed to a resident host */

static pthread_mutex_t g_getopt_lock = PTHREAD_MUTEX_INITIALIZER;   /* getopt's state is global; a resident host runs queries on threads */

int main_view(int argc, char *argv[])
{
    const char *srv = getenv("BGT_SERVER");
    if (srv && *srv) {
        const int rc = view_via_server(srv, argc, argv);
        if (rc >= 0) return rc;
        if (getenv("BGT_TRACE")) fprintf(stderr, "[bgt trace] no server at '%s': running locally\n", srv);
    }
    return view_run(argc, argv, stdout, stderr, NULL);
}

/* `bgt view` proper.  host == NULL: the command-line process (device work in a child, fast exit).  host != NULL: a resident
 * process runs the query on one of its threads -- databases come from (and go back to) the host's cache, out / err are the
 * client's, nothing process-wide is touched and everything is released in order. */
int view_run(int argc, char *argv[], FILE *out, FILE *err, const bgt_view_host_t *host)
{
    double t_lap = view_now();
    int i, c, n_files, out_bcf = 0, clevel = -1, flag = 0, u_set = 0, n_groups = 0, not_vcf = 0, excl = 0, in_mem = 0;
    fmf_t *vardb = NULL;
    void *bed = NULL;
    long seekn = -1, n_rec = LONG_MAX, n_read = 0;
    int rd_ret = -1;
    char *reg = NULL, *site_flt = NULL, *fmt = NULL, *aexpr = NULL, *dbfn = NULL, *gexpr[BGT_MAX_GROUPS];
    bgt_file_t **files = NULL;
    bgtm_t *bm = NULL;
    bcf1_t *b;
    int rc = 0, n_open = 0;
    bgzw_t *bz = NULL;
    kstring_t line = {0, 0, 0};

    int first_file;
    pthread_mutex_lock(&g_getopt_lock);
    optind = 0;                                                     /* glibc: 0 re-initialises the scanner -- it keeps a pointer INTO the argv of the call before (freed by a resident host) */
    while ((c = getopt(argc, argv, "ubs:r:l:CMGB:ef:g:a:i:n:SHt:d:")) >= 0) {
        switch (c) {
        case 'b': out_bcf = 1; break;
        case 'r': reg = optarg; break;
        case 'l': clevel = atoi(optarg); break;
        case 'u': u_set = 1; break;
        case 'C': flag |= BGT_F_SET_AC; break;
        case 'G': flag |= BGT_F_NO_GT; break;
        case 'i': seekn = atol(optarg) - 1; break;
        case 'n': n_rec = atol(optarg); break;
        case 'f': site_flt = optarg; break;
        case 's': if (n_groups < BGT_MAX_GROUPS) gexpr[n_groups++] = optarg; break;
        case 't': fmt = optarg; not_vcf = 1; break;                 /* tabular output instead of VCF (ref view.c:43) */
        case 'B': bed = bed_read(optarg); break;                     /* ref view.c:34 */
        case 'e': excl = 1; break;
        case 'a': aexpr = optarg; break;                            /* ref view.c:46 */
        case 'S': flag |= BGT_F_NO_GT | BGT_F_CNT_AL; not_vcf = 1; break;
        case 'H': flag |= BGT_F_NO_GT | BGT_F_CNT_HAP; not_vcf = 1; break;
        case 'd': dbfn = optarg; break;                             /* variant annotations (FMF) for -a EXPR */
        case 'M': in_mem = 1; break;
        default: break;
        }
    }
    first_file = optind;
    pthread_mutex_unlock(&g_getopt_lock);
    n_files = 0;
    if (n_rec < 0) { fprintf(err, "[E::%s] option -n must be at least 0.\n", "main_view"); VIEW_FAIL(1); }
    if (clevel > 9) clevel = 9;
    if (u_set) { clevel = 0; out_bcf = 1; }
    if (n_groups > 1) flag |= BGT_F_SET_AC;
    if (argc - first_file < 1) VIEW_FAIL(usage(argv[0], err));
    if ((flag & (BGT_F_CNT_AL | BGT_F_CNT_HAP)) && aexpr == NULL) {  /* ref view.c:93-96 */
        fprintf(err, "[E::%s] -a must be specified when -S/-H is in use.\n", "main_view");
        VIEW_FAIL(1);
    }

    /* a query that will touch genotypes needs the device: start the HIP runtime now, beside the host-only work below
     * (headers, sample tables, group expressions, the site side-car); `view -G` without counts never opens it */
    if (!host && (!(flag & BGT_F_NO_GT) || (flag & BGT_F_SET_AC) || site_flt)) {
        const char *gp = getenv("BGT_GPUS");
        work_in_a_child();
        bgth_runtime_warmup_async(gp && gp[0] >= '0' && gp[0] <= '9' && strchr(gp, ',') ? atoi(gp) : 0);
    }

    /* a plain walk of the whole file(s), counts only: the image may skip the sub-checkpoints a long-lived reader wants */
    if (!host && (flag & BGT_F_NO_GT) && !reg && !bed && !aexpr && seekn <= 0 && !fmt) setenv("BGTH_OPEN_HINT", "walk", 0);

    n_files = argc - first_file;
    files = (bgt_file_t**)calloc((size_t)n_files, sizeof(bgt_file_t*));
    for (i = 0; i < n_files; ++i)
        if ((files[i] = host ? host->open(argv[first_file + i], host->ctx) : bgt_open(argv[first_file + i])) == NULL) {
            fprintf(err, "[E::%s] failed to open BGT with prefix '%s'\n", "main_view", argv[first_file + i]);
            VIEW_FAIL(1);
        } else n_open = i + 1;
    view_lap(&t_lap, "open databases (header, samples)");
    bm = bgtm_reader_init(n_files, files);
    bgtm_set_flag(bm, flag);
    if (site_flt && bgtm_set_flt_site(bm, site_flt) != 0) {
        fprintf(err, "[E::%s] failed to set frequency filters. Syntax error?\n", "main_view");
        VIEW_FAIL(1);
    }
    if (reg && bgtm_set_region(bm, reg) < 0) {
        fprintf(err, "[E::%s] failed to set region. Region format error?\n", "main_view");
        VIEW_FAIL(1);
    }
    if (bed) bgtm_set_bed(bm, bed, excl);
    if (fmt && bgtm_set_table(bm, fmt) < 0) {
        fprintf(err, "[E::%s] failed to set tabular output.\n", "main_view");
        VIEW_FAIL(1);
    }
    if (seekn > 0) bgtm_set_start(bm, seekn);
    if (aexpr) {                                                    /* ref view.c:125-133 */
        int n_al;
        if (dbfn && in_mem) vardb = fmf_read(dbfn);                 /* ref view.c:76-84 */
        n_al = bgtm_set_alleles(bm, aexpr, vardb, in_mem ? NULL : dbfn);
        if (n_al < 0) { fprintf(err, "[E::%s] failed to set alleles.\n", "main_view"); VIEW_FAIL(1); }
        if (n_al == 0) fprintf(err, "[W::%s] no alleles selected.\n", "main_view");
    }
    for (i = 0; i < n_groups; ++i)
        if (bgtm_add_group(bm, gexpr[i]) < 0) {
            fprintf(err, "[E::%s] failed to add sample group '%s'.\n", "main_view", gexpr[i]);
            VIEW_FAIL(1);
        }
    if (!out_bcf && !not_vcf) bgtm_want_vcf_text(bm);
    if (bgtm_prepare(bm) < 0) { fprintf(err, "[E::%s] failed to prepare the readers.\n", "main_view"); VIEW_FAIL(1); }
    view_lap(&t_lap, "prepare (.pbf image -> HBM)");

    /* the reference builds the mode string "wb%d" and takes its first digit as the level, so the default
     * -1 compresses at level 1 (view.c:144-146, bgzf.c:138-146) */
    if (not_vcf) out_bcf = 0;
    else if (out_bcf) {
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        bz = bgzw_open(out, clevel < 0 ? 1 : clevel);
        bgzw_threads(bz, ncpu > 16 ? 16 : (int)ncpu);               /* blocks are independent: same bytes, deflated in parallel */
        bcf_hdr_write_stream(bz, bm->h_out);
    }
    else vcf_hdr_write_text(out, bm->h_out);

    b = bcf_init1();
    if (!bz && !not_vcf && !getenv("BGT_NO_BULK")) {                /* the whole walk at once where the query allows it */
        const long nb = bgtm_write_vcf_bulk(bm, out, n_rec);
        if (nb >= 0) n_read += nb;
        else if (nb < -1) rd_ret = -2;
    }
    while (rd_ret >= -1 && !ferror(out) && (rd_ret = ((bz || not_vcf) ? bgtm_read(bm, b) : bgtm_read_vcf(bm, b, &line))) >= 0 && n_read < n_rec) {
        if (bz) bcf_write1_stream(bz, b);
        else if (!not_vcf) { fwrite(line.s, 1, line.l, out); fputc('\n', out); }
        if (fmt && bm->n_fields > 0) { fputs(bm->tbl_line.s, out); fputc('\n', out); }
        ++n_read;
    }
    bcf_destroy1(b);
    view_lap(&t_lap, "sites: scan, filter, format, write");
    if (not_vcf && bm->n_aal > 0) {                                 /* ref view.c:158-173 */
        if (bm->flag & BGT_F_CNT_HAP) {
            int n_hap;
            bgt_hapcnt_t *hc = bgtm_hapcnt(bm, &n_hap);
            char *s = bgtm_hapcnt_print_destroy(bm, n_hap, hc);
            if (s) fputs(s, out);
            free(s);
        }
        if (bm->flag & BGT_F_CNT_AL) {
            char *s = bgtm_alcnt_print(bm);
            if (s) fputs(s, out);
            free(s);
        }
    }
    if (bz) bgzw_close(bz);
    fflush(out);
    if (!host && rd_ret >= -1 && !getenv("BGT_CLEAN_EXIT")) {
        /* everything is written: freeing the images in HBM one by one and tearing the HIP runtime down costs 20-60 ms
         * that nobody waits for (the driver reclaims the process's memory); BGT_CLEAN_EXIT=1 keeps the orderly path */
        view_lap(&t_lap, "done (fast exit)");
        fflush(err);
        if (g_done_fd >= 0) {                                       /* the waiting parent leaves with the status; see work_in_a_child */
            const unsigned char ok = 0;
            close(1); close(2);
            if (write(g_done_fd, &ok, 1) != 1) {}
        }
        _exit(0);
    }
    if (rd_ret < -1) {                                              /* -1 is the end of the data; anything below is a failure */
        fprintf(err, "[E::%s] reading stopped on an error (%d): the output is incomplete.\n", "main_view", rd_ret);
        rc = 1;
    }
done:
    free(line.s);
    if (bm) bgtm_reader_destroy(bm);
    if (bed) bed_destroy(bed);
    if (vardb) fmf_destroy(vardb);
    for (i = 0; i < n_open; ++i) { if (host) host->close(files[i], host->ctx); else bgt_close(files[i]); }
    free(files);
    view_lap(&t_lap, "close");
    return rc;
}
