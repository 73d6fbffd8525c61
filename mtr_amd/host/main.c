/* main.c — mTR [-acp] [-m ratio] <fasta file name>: the reference's command line (main.c:48-123) over the MI355X hot
 * path.  The file is cut at record boundaries, parsed on a few threads, pushed through two device contexts in turn
 * (pipeline.c) and every read's records are chained and printed in input order (chain.c, print.c). */
#define _POSIX_C_SOURCE 200809L
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/time.h>

static void usage(void)
{
    fprintf(stderr, "mTR [-acp] [-m ratio] <fasta file name> \n");
    fprintf(stderr, "-a: Output the alignment between the input sequence and predicted tandem repeat. \n");
    fprintf(stderr, "-c: Print the computation time of each step.\n");
    fprintf(stderr, "-m ratio: Give a minimum match ratio ranging from 0 to 1.\n");
    fprintf(stderr, "-p: Use Pearson's correlation coefficient distance in place of Manhattan distance.\n");
    fprintf(stderr, "-d n: (this build) use GPU n.  -B: (this build) results of a read depend on the longer reads before it in the file, as in reference mTR.\n");
}

static double now(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + t.tv_usec * 1.0E-6; }

int main(int argc, char **argv)
{
    int print_time = 0;
    mtrh_opts o; memset(&o, 0, sizeof o);
    o.manhattan = 1; o.min_match_ratio = 0.6f;    /* MIN_MATCH_RATIO, mTR.h:32 */
    o.world = 1;
    int opt;
    while ((opt = getopt(argc, argv, "acm:pd:B")) != -1) {
        switch (opt) {
        case 'a': o.print_alignment = 1; break;
        case 'c': print_time = 1; break;
        case 'm':
            o.min_match_ratio = (float)atof(optarg);
            if (!(0 <= o.min_match_ratio && o.min_match_ratio <= 1)) { fprintf(stderr, "The input minimum match ratio must range from 0 to 1.\n"); exit(EXIT_FAILURE); }
            break;
        case 'p': o.manhattan = 0; fprintf(stderr, "Pearson's correlation coefficient distance in place of Manhattan distance.\n"); break;
        case 'd': o.device = atoi(optarg); break;      /* extension: GPU ordinal */
        case 'B': o.file_order = 1; break;             /* extension: the reference's whole-file behaviour (mtr_hip.h, file-order mode) */
        default: usage(); exit(EXIT_FAILURE);
        }
    }
    if (optind >= argc) { fprintf(stderr, "The input file name is expected argument after options\n"); exit(EXIT_FAILURE); }

    const double t_all = now();
    const char *paths[1] = { argv[optind] };
    mtrh_run *run = mtrh_run_start(&o, paths, 1);
    if (!run) exit(EXIT_FAILURE);
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    mtrh_printer *pr = mtrh_printer_start(stdout, ncpu >= 8 ? 4 : (ncpu >= 4 ? 2 : 1));
    for (mtrh_result *x; (x = mtrh_run_next(run)) != NULL; ) mtrh_printer_push(pr, x);
    double t_chain = 0;
    const int status = mtrh_printer_finish(pr, &t_chain);
    mtrh_stamp("everything printed");
    double t_wait = 0, t_submit = 0, t_fetch = 0, t_kernel = 0; long long queries = 0;
    mtrh_run_timing(run, &t_wait, &t_submit, &t_fetch, &t_kernel, &queries);
    double t_create = 0, ph[MTR_N_KERNEL_TIMES];
    mtrh_run_phase_times(run, &t_create, ph, MTR_N_KERNEL_TIMES);
    char engine_path[4096];
    snprintf(engine_path, sizeof engine_path, "%s", mtrh_run_engine_path(run));
    mtrh_run_stop(run);
    mtrh_stamp("run stopped");
    if (getenv("MTR_HOST_TIMING"))                /* development aid: phase times on stderr */
        fprintf(stderr, "[host] waiting for the parser threads %.3f s, upload+launch %.3f s, waiting for the device + fetch %.3f s (kernels %.3f s), chain+print %.3f s, all %.3f s\n",
                t_wait, t_submit, t_fetch, t_kernel, t_chain, now() - t_all);
    if (print_time) {                             /* the reference's -c block (main.c:108-121), line for line.  Its timers sit around steps of ONE read at a
                                                   * time on the host; here the steps are kernels over a batch, so the lines carry the device time of the
                                                   * kernels that do that step (HIP events, mtr_get_kernel_times), summed over the batches:
                                                   *   allocating memory      <- creating the device contexts (malloc_global_variables, handle_one_file.c:71-136)
                                                   *   ranges                 <- the range kernels (fill_directional_index_with_end, handle_one_read.c:206-212)
                                                   *   Computing periods      <- every other kernel of the launches + chaining (handle_one_read.c:217-258)
                                                   *   Initialize the input   <- packing to 2 bit/base + upload + launch on the host (init_inputString, consensus.c:39-59)
                                                   *   count table generation <- the unit-search kernels: k-mer tables, seeds, walks (consensus.c:73-127)
                                                   *   wrap around            <- the alignment and revision kernels (wrap_around_DP_sub, wrap_around_DP.c:224-353)
                                                   * a batch the per-read kernel took has no phases: its whole launch counts as "Computing periods" */
        const double t_launches = ph[0] + ph[1], t_ranges = ph[2];
        fprintf(stderr, "Computation time\n");
        fprintf(stderr, "%f\tall\n", now() - t_all);
        fprintf(stderr, "%f\tallocating memory\n", t_create);
        fprintf(stderr, "%f\tranges\n", t_ranges);
        fprintf(stderr, "%f\tComputing periods\n", (t_launches > t_ranges ? t_launches - t_ranges : 0.0) + t_chain);
        fprintf(stderr, "\t%f\tInitialize the input\n", t_submit);
        fprintf(stderr, "\t%f\tcount table generation\n", ph[3]);
        fprintf(stderr, "\t%f\twrap around\n", ph[4] + ph[6]);
        fprintf(stderr, "\t%f\tchaining\n", t_chain);
        fprintf(stderr, "\t%i\tCount of queries\n", (int)queries);
        fprintf(stderr, "%s\tengine library%s\n", engine_path, getenv("MTR_LIB") ? " (from $MTR_LIB)" : "");   /* this build: what computed the records */
    }
    return status ? EXIT_FAILURE : EXIT_SUCCESS;
}
