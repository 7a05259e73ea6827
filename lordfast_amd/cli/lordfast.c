/*
 * lordfast -- the reference's process interface over the MI355X path.
 *
 * Same options, defaults, validation and error texts as parseCommandLine (src/CommandLineParser.cpp:126-310) and the
 * same two modes as main (src/baseFAST.cpp:30-95):
 *     lordfast --index ref.fa
 *     lordfast --search ref.fa --seq reads.fa[.gz] [-o out.sam] [-t N] [-k 14] [-c 1000] [-n 10] [-l 1000] [-m 1000]
 *              [-a dp-n2|clasp] [-R '@RG\tID:x'] [--noSamHeader] [--chainReward r] [--chainPenalty p] [--gapPenalty g]
 * Everything below the option parser is the C ABI of liblfgpu.so (include/lordfast_amd.h): the index is built /
 * loaded into HBM, reads stream through lf_map_file (reader runs ahead of the GPU), SAM goes to --out or stdout.
 * Not in the reference: LF_DEVICE (HIP device, default 0) and LF_SAMPLED_SA=1 (keep the 1/32 sampled suffix array
 * instead of materialising the full one in HBM) are environment variables, so the command line stays the reference's.
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
#include "lordfast_amd.h"

#define PROG_VERSION "0.0.10"

static void help_short(void)
{   /* printHelp_short, src/CommandLineParser.cpp:78-83 */
    fprintf(stderr, "usage: lordfast --index ref.fa\n");
    fprintf(stderr, "       lordfast --search ref.fa --seq reads.fa [options]\n");
    fprintf(stderr, "For more details and command line options run \"lordfast --help\"\n");
}

static void help_long(void)
{   /* the reference prints a groff-rendered man page (HELP.man); this is the same option list as plain text */
    printf("lordfast %s (MI355X path)\n\n"
           "  -I, --index STR        build the index of the reference FASTA (next to it: .bwt .sa .pac .ann .amb .cache)\n"
           "  -S, --search STR       map against the indexed reference\n"
           "  -s, --seq STR          reads, FASTA/FASTQ, plain or gzip\n"
           "  -o, --out STR          SAM output [stdout]\n"
           "  -t, --threads INT      host threads (0 = all) [1]\n"
           "  -k, --minAnchorLen INT minimum anchor length, 12..20 [14]\n"
           "  -c, --anchorCount INT  anchoring positions per read [1000]\n"
           "  -n, --numMap INT       candidate windows per read [10]\n"
           "  -l, --minReadLen INT   minimum read length [1000]\n"
           "  -m, --maxRefHit INT    ignore anchors with more hits [1000]\n"
           "  -a, --chainAlg STR     dp-n2 or clasp [dp-n2]\n"
           "  -R, --readGroup STR    SAM read group line, e.g. '@RG\\tID:foo\\tSM:bar'\n"
           "      --noSamHeader      do not print the SAM header\n"
           "  -h, --help   -v, --version\n", PROG_VERSION);
}

int main(int argc, char *argv[])
{
    lf_params_t P; lf_params_default(&P);
    P.threads = 1;                                   /* THREAD_COUNT = 1, src/CommandLineParser.cpp:49 */
    int indexing = 0, searching = 0, no_header = 0, idx, ch;
    const char *ref_file = NULL, *seq_file = NULL, *out_file = NULL;
    static char cmdline[4000];
    if (argc < 2) { help_short(); return EXIT_FAILURE; }
    static struct option lo[] = {
        {"index", required_argument, 0, 'I'}, {"search", required_argument, 0, 'S'}, {"seq", required_argument, 0, 's'},
        {"out", required_argument, 0, 'o'}, {"threads", required_argument, 0, 't'}, {"minAnchorLen", required_argument, 0, 'k'},
        {"maxRefHit", required_argument, 0, 'm'}, {"minReadLen", required_argument, 0, 'l'}, {"anchorCount", required_argument, 0, 'c'},
        {"numMap", required_argument, 0, 'n'}, {"chainAlg", required_argument, 0, 'a'}, {"readGroup", required_argument, 0, 'R'},
        {"noSamHeader", no_argument, 0, 0}, {"chainReward", required_argument, 0, 'r'}, {"chainPenalty", required_argument, 0, 'p'},
        {"gapPenalty", required_argument, 0, 'g'}, {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {0, 0, 0, 0}
    };
    lo[12].flag = &no_header; lo[12].val = 1;
    while ((ch = getopt_long(argc, argv, "I:S:s:o:t:k:m:l:c:n:a:r:R:P:G:hv", lo, &idx)) != -1) {
        switch (ch) {
        case 0: fprintf(stderr, "[NOTE] option %s is set\n", lo[idx].name); break;
        case 'I': indexing = 1; ref_file = optarg; break;
        case 'S': searching = 1; ref_file = optarg; break;
        case 's': seq_file = optarg; break;
        case 'o': out_file = optarg; break;
        case 't':
            P.threads = atoi(optarg);
            if (P.threads <= 0 || P.threads > sysconf(_SC_NPROCESSORS_ONLN)) P.threads = 0;     /* all CPUs */
            break;
        case 'k': P.min_anchor_len = atoi(optarg); break;
        case 'm': P.max_ref_hits = atoi(optarg); break;
        case 'l': P.min_read_len = atoi(optarg); if (P.min_read_len < 100) P.min_read_len = 100; break;
        case 'c': P.sampling_count = atoi(optarg); break;
        case 'n': P.max_map = atoi(optarg); break;
        case 'a':
            if (strcmp(optarg, "clasp") == 0) P.chain_alg = 1;
            else if (strcmp(optarg, "dp-n2") == 0) P.chain_alg = 0;
            else { fprintf(stderr, "[WARNING] (parseCommandLine) unknown argument for -A/--chainAlg. Using dynamic programming (dp-n2)!\n"); P.chain_alg = 0; }
            break;
        case 'R': if (lf_params_set_read_group(&P, optarg)) { fprintf(stderr, "[ERROR] %s\n", lf_last_error()); return EXIT_FAILURE; } break;
        case 'r': P.chain_reward = atof(optarg); break;
        case 'p': P.chain_penalty = atof(optarg); break;
        case 'g': P.gap_penalty = atof(optarg); break;
        case 'h': help_long(); return EXIT_SUCCESS;
        case 'v': fprintf(stdout, "lordFAST %s\n", PROG_VERSION); return EXIT_SUCCESS;
        default: help_short(); return EXIT_FAILURE;
        }
    }
    if (indexing + searching != 1) { fprintf(stderr, "[ERROR] (parseCommandLine) indexing / searching mode should be selected\n"); help_short(); return EXIT_FAILURE; }
    if (searching && seq_file == NULL) { fprintf(stderr, "[ERROR] (parseCommandLine) please indicate a sequence file for searching.\n"); help_short(); return EXIT_FAILURE; }
    if (P.min_anchor_len < 12 || P.min_anchor_len > 20) { fprintf(stderr, "[ERROR] (parseCommandLine) -k/--minAnchorLen requires an argument in [12..20]\n"); return EXIT_FAILURE; }
    if (P.sampling_count <= 0) { fprintf(stderr, "[ERROR] (parseCommandLine) -c/--anchorCount requires a positive integer argument\n"); return EXIT_FAILURE; }
    if (P.max_map <= 0) { fprintf(stderr, "[ERROR] (parseCommandLine) -n/--numMap requires a positive integer argument\n"); return EXIT_FAILURE; }
    /* -n 1 divides by zero in the reference's MAPQ formula (src/LordFAST.cpp:326); refused here, at parse
     * time, instead of after the index has been loaded */
    if (searching && P.max_map < 2) { fprintf(stderr, "[ERROR] (parseCommandLine) -n/--numMap 1 is not supported (the reference's MAPQ formula divides by numMap - 1); use -n 2 or more\n"); return EXIT_FAILURE; }
    if (P.max_ref_hits <= 0) { fprintf(stderr, "[ERROR] (parseCommandLine) -m/--maxRefHit requires a positive integer argument\n"); return EXIT_FAILURE; }
    for (int i = 0; i < argc; i++) {               /* opt_commandAll: every argument followed by a blank (:303-307) */
        if (strlen(cmdline) + strlen(argv[i]) + 2 >= sizeof cmdline) break;
        strcat(cmdline, argv[i]); strcat(cmdline, " ");
    }
    /* LF_DEVICES="0-3" / "0,2,5": the read batches are spread over these GPUs (one index replica each); LF_DEVICE=d: one GPU */
    int devs[16], n_devs = 0;
    if (getenv("LF_DEVICES")) {
        const char *q = getenv("LF_DEVICES");
        while (*q && n_devs < 16) {
            char *e; long a = strtol(q, &e, 10), b = a;
            if (e == q) break;
            if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
            for (long d = a; d <= b && n_devs < 16; d++) devs[n_devs++] = (int)d;
            q = *e == ',' ? e + 1 : e;
        }
        if (n_devs == 0) { fprintf(stderr, "[ERROR] LF_DEVICES: expected a list like 0-3 or 0,2,5\n"); return EXIT_FAILURE; }
    } else devs[n_devs++] = getenv("LF_DEVICE") ? atoi(getenv("LF_DEVICE")) : 0;
    const int device = devs[0];
    for (int d = 0; d < n_devs; d++)
        if (devs[d] < 0 || lf_device_count() <= devs[d]) { fprintf(stderr, "[ERROR] no gfx950 device %d visible (this build has no CPU path)\n", devs[d]); return EXIT_FAILURE; }

    if (indexing) {
        if (lf_index_build(ref_file, device) != LF_OK) { fprintf(stderr, "[ERROR] (bwt_index) %s\n", lf_last_error()); return EXIT_FAILURE; }
        return EXIT_SUCCESS;
    }
    /* 0 / out of range = all CPUs (src/CommandLineParser.cpp:181-185); the NOTE prints the resolved count (:296) */
    fprintf(stderr, "[NOTE] number of threads: %d\n", P.threads > 0 ? P.threads : (int)sysconf(_SC_NPROCESSORS_ONLN));
    {   /* bwt_load builds the index first when <ref>.bwt is missing (src/BWT.cpp:203-208) */
        char path[4096]; snprintf(path, sizeof path, "%s.bwt", ref_file);
        if (access(path, R_OK) != 0) {
            fprintf(stderr, "[NOTE] (bwt_load) index not found, building it\n");
            if (lf_index_build(ref_file, device) != LF_OK) { fprintf(stderr, "[ERROR] (bwt_index) %s\n", lf_last_error()); return EXIT_FAILURE; }
        }
    }
    lf_index_t *ixs[16] = { NULL };
    const unsigned flags = (getenv("LF_SAMPLED_SA") && atoi(getenv("LF_SAMPLED_SA"))) ? 0u : LF_IDX_FULL_SA;
    for (int d = 0; d < n_devs; d++)
        if (lf_index_load(ref_file, devs[d], flags, &ixs[d]) != LF_OK) { fprintf(stderr, "[ERROR] (bwt_load) %s\n", lf_last_error()); return EXIT_FAILURE; }
    lf_stats_t st;
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = lf_map_file_multi((const lf_index_t *const *)ixs, n_devs, &P, seq_file, out_file, no_header, cmdline, 0, &st);
    if (rc != LF_OK) fprintf(stderr, "[ERROR] %s\n", lf_last_error());
    else {
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double wall = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        fprintf(stderr, "[NOTE] processed %llu reads in %.2f seconds\n", (unsigned long long)st.n_reads, st.ms_total / 1000.0);
        fprintf(stderr, "[NOTE] search wall time %.2f seconds (reading + mapping + writing, index resident): %.0f reads/s\n", wall, wall > 0 ? (double)st.n_reads / wall : 0.0);
    }
    for (int d = 0; d < n_devs; d++) lf_index_free(ixs[d]);
    return rc == LF_OK ? EXIT_SUCCESS : EXIT_FAILURE;
}
