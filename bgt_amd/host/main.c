/* main.c -- the `bgt` executable of this build: a LAUNCHER.  It links neither libbgt.so nor the HIP runtime: a `bgt view` that a
 * resident host answers (BGT_SERVER=<unix socket>, see view_client.h / server.c) needs neither, and loading them costs 13 ms per
 * process.  Everything else is dispatched into libbgt.so, loaded from ../lib next to the executable. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include "view_client.h"

static void *g_lib;
static void *sym(const char *name)
{
    if (!g_lib) {
        char exe[PATH_MAX], path[PATH_MAX + 32];
        ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
        char *sl;
        if (n <= 0) { fprintf(stderr, "[E::main] cannot find the executable's directory\n"); exit(1); }
        exe[n] = 0;
        if ((sl = strrchr(exe, '/')) != NULL) *sl = 0;
        snprintf(path, sizeof(path), "%s/../lib/libbgt.so", exe);
        if ((g_lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL)) == NULL) { fprintf(stderr, "[E::main] %s\n", dlerror()); exit(1); }
    }
    void *f = dlsym(g_lib, name);
    if (!f) { fprintf(stderr, "[E::main] %s\n", dlerror()); exit(1); }
    return f;
}

int main(int argc, char *argv[])
{
    if (argc < 2) {
        fprintf(stderr, "Usage: bgt <command> <arguments>\nCommands:\n  import   import VCF/BCF to BGT (PBWT encoder on MI355X)\n  view     extract from BGT (genotype-matrix read path on MI355X)\n  pbfview  decode / encode PBF <-> PIM text (the PBWT codec on MI355X)\n  version  show version\n");
        return 1;
    }
    if (strcmp(argv[1], "view") == 0 || strcmp(argv[1], "mview") == 0) {
        const char *srv = getenv("BGT_SERVER");
        int rc;
        if (srv && *srv && (rc = view_via_server(srv, argc - 1, argv + 1)) >= 0) return rc;   /* answered by the resident host */
        rc = ((int (*)(int, char**))sym("main_view"))(argc - 1, argv + 1);  /* (a complete answer leaves through _exit inside) */
        ((void (*)(void))sym("bgth_runtime_warmup_wait"))();        /* an early error: the HIP runtime may still be starting on its thread */
        return rc;
    }
    if (strcmp(argv[1], "import") == 0) return ((int (*)(int, char**))sym("main_import"))(argc - 1, argv + 1);
    if (strcmp(argv[1], "pbfview") == 0) return ((int (*)(int, char**))sym("main_pbfview"))(argc - 1, argv + 1);   /* the codec-level tool (reference pbfview.c) */
    if (strcmp(argv[1], "synth") == 0) {                       /* bgt synth <prefix> <samples> <sites> [seed] */
        if (argc < 5) { fprintf(stderr, "Usage: bgt synth <out-prefix> <n-samples> <n-sites> [seed]\n"); return 1; }
        return ((int (*)(const char*, int, long long, unsigned long long, int))sym("bgt_synth_trio"))(argv[2], atoi(argv[3]), atoll(argv[4]), argc > 5 ? strtoull(argv[5], 0, 10) : 1, 0) ? 1 : 0;
    }
    if (strcmp(argv[1], "version") == 0) { puts(((const char *(*)(void))sym("bgth_version"))()); return 0; }
    fprintf(stderr, "[E::%s] unrecognized command '%s' (this build provides: import, view, pbfview)\n", __func__, argv[1]);
    return 1;
}
