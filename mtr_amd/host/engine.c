/* engine.c — binds the host to a library that implements include/mtr_hip.h.  The product library is libmtr_hip.so
 * (HIP kernels for gfx950; it fails with MTR_ERR_NO_DEVICE without a GPU — there is no CPU path).  The path can be
 * given explicitly (tests hand in a replay library that answers from recorded reference records, to exercise the host
 * side on machines without a GPU); by default it is $MTR_LIB, else libmtr_hip.so next to this code. */
#define _GNU_SOURCE
#include "mtr_host.h"
#include <dlfcn.h>
#include <libgen.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int default_path(char *out, size_t n)
{
    const char *env = getenv("MTR_LIB");
    if (env && *env) { snprintf(out, n, "%s", env); return 0; }
    /* next to the code that contains this function: libmtr_host.so and mTR live in mtr_amd/host/, the library in mtr_amd/ */
    Dl_info info;
    char self[PATH_MAX];
    if (dladdr((void *)default_path, &info) && info.dli_fname && realpath(info.dli_fname, self)) {
        char *d = dirname(self);
        snprintf(out, n, "%s/../libmtr_hip.so", d);
        if (access(out, R_OK) == 0) return 0;
        snprintf(out, n, "%s/libmtr_hip.so", d);
        if (access(out, R_OK) == 0) return 0;
    }
    snprintf(out, n, "libmtr_hip.so");
    return 0;
}

#define BIND(field, name) do { *(void **)(&e->field) = dlsym(e->dl, name); \
    if (!e->field) { snprintf(err, errlen, "%s does not export %s", path, name); dlclose(e->dl); memset(e, 0, sizeof *e); return 1; } } while (0)

int mtrh_engine_load(mtrh_engine *e, const char *lib_path, char *err, size_t errlen)
{
    char path[PATH_MAX];
    memset(e, 0, sizeof *e);
    if (lib_path && *lib_path) snprintf(path, sizeof path, "%s", lib_path); else default_path(path, sizeof path);
    e->dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!e->dl) { snprintf(err, errlen, "cannot load %s: %s", path, dlerror()); return 1; }
    BIND(create, "mtr_create"); BIND(destroy, "mtr_destroy"); BIND(last_error, "mtr_last_error");
    BIND(upload_packed, "mtr_upload_batch_packed"); BIND(upload_in_file, "mtr_upload_batch_in_file");
    BIND(run_async, "mtr_run_resident_async"); BIND(wait, "mtr_wait"); BIND(fetch_packed, "mtr_fetch_results_packed");
    BIND(first_failed, "mtr_get_first_failed_read"); BIND(alignments, "mtr_alignments"); BIND(bases_after, "mtr_get_bases_after_read");
    BIND(kernel_times, "mtr_get_kernel_times"); BIND(counters, "mtr_get_counters");
    BIND(fs_create, "mtr_file_state_create"); BIND(fs_destroy, "mtr_file_state_destroy"); BIND(fs_skip, "mtr_file_state_skip");
    BIND(device_count, "mtr_device_count"); BIND(gather_create, "mtr_gather_create"); BIND(gather_destroy, "mtr_gather_destroy");
    BIND(gather_last_error, "mtr_gather_last_error"); BIND(gather_stage, "mtr_gather_stage"); BIND(gather_exchange, "mtr_gather_exchange");
    BIND(gather_wait_ready, "mtr_gather_wait_ready"); BIND(gather_get_stats, "mtr_gather_get_stats");
    __typeof__(mtr_abi_version) *ver = NULL;
    *(void **)(&ver) = dlsym(e->dl, "mtr_abi_version");
    if (!ver || ver() != MTR_ABI_VERSION) {
        snprintf(err, errlen, "%s implements ABI version %d, this host was built for %d", path, ver ? ver() : -1, MTR_ABI_VERSION);
        dlclose(e->dl); memset(e, 0, sizeof *e); return 1;
    }
    if (!realpath(path, e->path)) snprintf(e->path, sizeof e->path, "%s", path);
    return 0;
}

void mtrh_engine_unload(mtrh_engine *e)
{
    /* the library stays mapped: unloading a HIP code object at exit races with the runtime's own teardown */
    memset(e, 0, sizeof *e);
}
