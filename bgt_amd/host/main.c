/* main.c -- the `bgt` executable of this build: dispatches `view` (alias `mview`) to the MI355X reader. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/bgt_synth.h"
#include "../../include/bgt_reader.h"
#include "../../include/bgt_hip.h"

int main(int argc, char *argv[])
{
    if (argc < 2) {
        fprintf(stderr, "Usage: bgt <command> <arguments>\nCommands:\n  import   import VCF/BCF to BGT (PBWT encoder on MI355X)\n  view     extract from BGT (genotype-matrix read path on MI355X)\n  pbfview  decode / encode PBF <-> PIM text (the PBWT codec on MI355X)\n  version  show version\n");
        return 1;
    }
    if (strcmp(argv[1], "view") == 0 || strcmp(argv[1], "mview") == 0) {
        const int rc = main_view(argc - 1, argv + 1);          /* (a complete answer leaves through _exit inside) */
        bgth_runtime_warmup_wait();                            /* an early error: the HIP runtime may still be starting on its thread */
        return rc;
    }
    if (strcmp(argv[1], "import") == 0) return main_import(argc - 1, argv + 1);
    if (strcmp(argv[1], "pbfview") == 0) return main_pbfview(argc - 1, argv + 1);   /* the codec-level tool (reference pbfview.c) */
    if (strcmp(argv[1], "synth") == 0) {                       /* bgt synth <prefix> <samples> <sites> [seed] */
        if (argc < 5) { fprintf(stderr, "Usage: bgt synth <out-prefix> <n-samples> <n-sites> [seed]\n"); return 1; }
        return bgt_synth_trio(argv[2], atoi(argv[3]), atoll(argv[4]), argc > 5 ? strtoull(argv[5], 0, 10) : 1, 0) ? 1 : 0;
    }
    if (strcmp(argv[1], "version") == 0) { puts(bgth_version()); return 0; }
    fprintf(stderr, "[E::%s] unrecognized command '%s' (this build provides: import, view, pbfview)\n", __func__, argv[1]);
    return 1;
}
