/* main.c — mTR [-acp] [-m ratio] <fasta file name>: the reference's command line (main.c:48-123) over the MI355X hot
 * path.  The file is cut at record boundaries, parsed on a few threads, pushed through two device contexts in turn
 * (pipeline.c) and every read's records are chained and printed in input order (chain.c, print.c). */
#define _POSIX_C_SOURCE 200809L
#include "mtr_host.h"
#include <stdlib.h>
#include <malloc.h>
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
    fprintf(stderr, "-g N: (this build) shard the reads over N GPUs of this node (one process; the record tables are gathered to the first GPU over RCCL); several files may follow.\n");
}

static double now(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + t.tv_usec * 1.0E-6; }

int main(int argc, char **argv)
{
    int print_time = 0, n_gpus = 0;
    mtrh_opts o; memset(&o, 0, sizeof o);
    o.manhattan = 1; o.min_match_ratio = 0.6f;    /* MIN_MATCH_RATIO, mTR.h:32 */
    o.world = 1;
    int opt;
    while ((opt = getopt(argc, argv, "acm:pd:Bg:")) != -1) {
        switch (opt) {
        case 'a': o.print_alignment = 1; break;
        case 'c': print_time = 1; break;
        case 'm':
            o.min_match_ratio = (float)atof(optarg);
            if (!(0 <= o.min_match_ratio && o.min_match_ratio <= 1)) { fprintf(stderr, "The input minimum match ratio must range from 0 to 1.\n"); exit(EXIT_FAILURE); }
            break;
        case 'p': o.manhattan = 0; fprintf(stderr, "Pearson's correlation coefficient distance in place of Manhattan distance.\n"); break;
        case 'd': o.device = atoi(optarg); break;      /* extension: GPU ordinal */
        case 'g': n_gpus = atoi(optarg);               /* extension: the GPUs of one node (multi.c) */
            if (n_gpus < 1 || n_gpus > 64) { fprintf(stderr, "-g takes a number of GPUs from 1 to 64.\n"); exit(EXIT_FAILURE); }
            break;
        case 'B': o.file_order = 1; break;             /* extension: the reference's whole-file behaviour (mtr_hip.h, file-order mode) */
        default: usage(); exit(EXIT_FAILURE);
        }
    }
    /* The process leaves through _exit once the report is out and the contexts are destroyed (below): the HIP runtime's own teardown at exit() is 0.08 s of a
     * 0.72 s run on 100 000 reads [measured] and leaves nothing behind that the process's end does not.  (Leaving the CONTEXTS to the process's end as well was
     * measured and is wrong: the driver then reclaims 10+ GB of device memory of the dead process while the NEXT process allocates - its first batch
     * waited 0.4-1.2 s for its buffers.)  MTR_FULL_TEARDOWN=1: exit() as usual (leak checkers). */
    const int fast_exit = getenv("MTR_FULL_TEARDOWN") == NULL;
    { const char *cb = getenv("MTR_CHUNK_BYTES"); if (cb && atoll(cb) > 0) o.chunk_bytes = (size_t)atoll(cb); }     /* FASTA bytes per chunk (default 24 MiB = one device batch of 2 kb reads) */
    if (optind >= argc) { fprintf(stderr, "The input file name is expected argument after options\n"); exit(EXIT_FAILURE); }

    /* Batches, record tables and text buffers are tens of MB each and come and go at the rate of the GPUs: as mappings of their own (glibc's default
     * above 128 KB .. 32 MB, adjusted as it goes) every one of them is an mmap, a page fault per 4 KB and a munmap whose TLB shoot-down stops every thread
     * of the process - a quarter of the host's CPU time behind eight GPUs (tests/null_engine.c: sys 2.2 s of 6.9 s per million reads).  Kept in the heaps. */
    mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30);
    const double t_all = now();
    mtrh_run *run = NULL; mtrh_multi *multi = NULL;
    int status = 0; double t_chain = 0;
    if (n_gpus >= 1) {
        /* the GPUs of one node in this process: one run per GPU, the record tables gathered to the first GPU over RCCL (multi.c) */
        /* stdout belongs to the report: RCCL prints its version banner on descriptor 1 at the first collective, so the report keeps a descriptor
         * of its own and descriptor 1 points at stderr for the rest of the process */
        fflush(stdout);
        const int report_fd = dup(1);
        if (report_fd < 0 || dup2(2, 1) < 0) { fprintf(stderr, "fatal error: cannot set the report's descriptor aside\n"); exit(EXIT_FAILURE); }
        multi = mtrh_multi_start(&o, n_gpus, (const char *const *)(argv + optind), argc - optind);
        if (!multi) exit(EXIT_FAILURE);
        mtrh_printer *pr = mtrh_printer_start_fd(report_fd, mtrh_printer_default_threads());
        const int drained = mtrh_multi_drain(multi, pr);
        status = mtrh_printer_finish(pr, &t_chain);
        if (drained != 0) status = 1;
    } else {
        const char *paths[1] = { argv[optind] };
        run = mtrh_run_start(&o, paths, 1);
        if (!run) exit(EXIT_FAILURE);
        mtrh_printer *pr = mtrh_printer_start(stdout, mtrh_printer_default_threads());
        for (mtrh_result *x; (x = mtrh_run_next(run)) != NULL; ) mtrh_printer_push(pr, x);
        status = mtrh_printer_finish(pr, &t_chain);
    }
    mtrh_stamp("everything printed");
    double t_wait = 0, t_submit = 0, t_fetch = 0, t_kernel = 0; long long queries = 0;
    double t_create = 0, ph[MTR_N_KERNEL_TIMES];
    memset(ph, 0, sizeof ph);
    char engine_path[4096], gather_line[512] = "";
    for (int g = 0; g < (multi ? mtrh_multi_n(multi) : 1); g++) {        /* (several GPUs: the sums over the GPUs, which work side by side) */
        const mtrh_run *rg = multi ? mtrh_multi_run(multi, g) : run;
        double a = 0, b = 0, c = 0, d = 0, tc = 0, p8[MTR_N_KERNEL_TIMES]; long long q = 0;
        mtrh_run_timing(rg, &a, &b, &c, &d, &q);
        mtrh_run_phase_times(rg, &tc, p8, MTR_N_KERNEL_TIMES);
        t_wait += a; t_submit += b; t_fetch += c; t_kernel += d; queries += q; t_create += tc;
        for (int i = 0; i < MTR_N_KERNEL_TIMES; i++) ph[i] += p8[i];
        if (g == 0) snprintf(engine_path, sizeof engine_path, "%s", mtrh_run_engine_path(rg));
    }
    int leave_fast = fast_exit;
    if (multi) { mtrh_multi_gather_line(multi, gather_line, sizeof gather_line); if (mtrh_multi_stop(multi)) leave_fast = 1; }
    else mtrh_run_stop(run);
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
        if (gather_line[0]) fprintf(stderr, "%s\n", gather_line);                 /* this build, -g N: how the tables reached the printer */
    }
    if (leave_fast) {                             /* the fast way out (above); with MTR_FULL_TEARDOWN still taken when RCCL is coming up on its thread: no teardown under it */
        if (fflush(stdout) != 0) status = 1;        /* (the report itself: the printer flushed it and its status says whether that worked, mtrh_printer_finish) */
        fflush(stderr);
        _exit(status ? EXIT_FAILURE : EXIT_SUCCESS);
    }
    return status ? EXIT_FAILURE : EXIT_SUCCESS;
}
