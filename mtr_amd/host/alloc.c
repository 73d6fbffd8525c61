/* alloc.c — every allocation of the host side goes through here (mtr_host.h redirects malloc / calloc / realloc / strdup).
 * The reference's answer to a failed malloc is fprintf + exit(EXIT_FAILURE) wherever it happens.  This host side runs parser, device
 * and printer threads, and under the multi-GPU launcher an exit() inside one of them would leave the other ranks waiting in a
 * collective: a worker thread that cannot allocate jumps back to its entry function instead (mtrh_oom_target), which reports the
 * failure the way a device-side error is reported — a fatal result that travels to the printer like any other, "everything before
 * it is printed, then the message, exit status 1".  Only the main thread (no target) still ends the process where it stands.
 * MTR_TEST_FAIL_ALLOC=<parser|device|printer>:<n> (tests): the n-th allocation made by threads of that kind fails. */
#define MTRH_NO_ALLOC_WRAP
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>

__thread jmp_buf *mtrh_oom_target = NULL;
__thread int mtrh_thread_kind = MTRH_THREAD_MAIN;

static int fail_kind = -2;
static long fail_at = 0, fail_count = 0;

static int injected(void)
{
    if (fail_kind == -2) {
        const char *e = getenv("MTR_TEST_FAIL_ALLOC");
        int k = -1;
        if (e) {
            if (!strncmp(e, "parser:", 7)) { k = MTRH_THREAD_PARSER; fail_at = atol(e + 7); }
            else if (!strncmp(e, "device:", 7)) { k = MTRH_THREAD_DEVICE; fail_at = atol(e + 7); }
            else if (!strncmp(e, "printer:", 8)) { k = MTRH_THREAD_PRINTER; fail_at = atol(e + 8); }
        }
        __atomic_store_n(&fail_kind, k, __ATOMIC_RELEASE);
    }
    if (fail_kind < 0 || fail_kind != mtrh_thread_kind) return 0;
    return __atomic_add_fetch(&fail_count, 1, __ATOMIC_RELAXED) == fail_at;
}

void mtrh_oom(size_t bytes)
{
    if (mtrh_oom_target) longjmp(*mtrh_oom_target, 1);
    fprintf(stderr, "cannot allocate %zu bytes\n", bytes);
    exit(EXIT_FAILURE);
}

void *mtrh_xmalloc(size_t n)
{
    void *p = injected() ? NULL : malloc(n ? n : 1);
    if (!p) mtrh_oom(n);
    return p;
}
void *mtrh_xcalloc(size_t n, size_t m)
{
    void *p = injected() ? NULL : calloc(n ? n : 1, m ? m : 1);
    if (!p) mtrh_oom(n * m);
    return p;
}
void *mtrh_xrealloc(void *q, size_t n)
{
    void *p = injected() ? NULL : realloc(q, n ? n : 1);
    if (!p) mtrh_oom(n);
    return p;
}
char *mtrh_xstrdup(const char *s)
{
    const size_t n = strlen(s) + 1;
    char *p = (char *)mtrh_xmalloc(n);
    memcpy(p, s, n);
    return p;
}
