/* fasta.c — batch FASTA reader keeping the reference's quirks (handle_one_file.c:169-269). */
#include "mtr_host.h"
#include <stdlib.h>
#include <string.h>

struct mtrh_fasta {
    FILE *fp;
    char  buf[MTRH_BLK];
    char *pending_id;       /* header already consumed for the next record */
    int   have_header, done;
};

static void *xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n);
    if (!q) { fprintf(stderr, "cannot allocate %zu bytes\n", n); exit(EXIT_FAILURE); }
    return q;
}

static char *header_id(const char *line)
{
    size_t i = 1;
    while (line[i] && line[i] != '\n' && line[i] != '\r') i++;
    char *id = (char *)xrealloc(NULL, i);
    memcpy(id, line + 1, i - 1);
    id[i - 1] = 0;
    return id;
}

mtrh_fasta *mtrh_fasta_open(const char *path)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { fprintf(stderr, "fatal error: cannot open %s\n", path); fflush(stderr); exit(EXIT_FAILURE); }
    mtrh_fasta *f = (mtrh_fasta *)calloc(1, sizeof(*f));
    f->fp = fp;
    return f;
}

void mtrh_fasta_close(mtrh_fasta *f) { if (f) { fclose(f->fp); free(f->pending_id); free(f); } }
void mtrh_read_free(mtrh_read *r) { free(r->id); free(r->codes); r->id = NULL; r->codes = NULL; r->len = 0; }

/* base code of a character: 0..3 for ACGT/acgt, 0xFE for the characters that end a line (NUL, LF, CR),
 * 0xFF for everything else (fatal, handle_one_file.c:169-188) */
static uint8_t code_of[256];
static void init_codes(void)
{
    if (code_of['C'] == 1) return;
    memset(code_of, 0xFF, sizeof code_of);
    code_of[0] = code_of['\n'] = code_of['\r'] = 0xFE;
    code_of['A'] = code_of['a'] = 0; code_of['C'] = code_of['c'] = 1; code_of['G'] = code_of['g'] = 2; code_of['T'] = code_of['t'] = 3;
}

/* one record; returns 0 at the end of input or at the first empty record (handle_one_file.c:283) */
static int next_read(mtrh_fasta *f, mtrh_read *out)
{
    if (f->done) return 0;
    init_codes();
    uint8_t *codes = NULL; size_t n = 0, cap = 0;
    char *id = f->pending_id; f->pending_id = NULL;
    while (fgets(f->buf, MTRH_BLK, f->fp)) {
        const unsigned char *s = (const unsigned char *)f->buf;
        if (s[0] == '>') {
            if (!f->have_header) { f->have_header = 1; id = header_id(f->buf); continue; }
            f->pending_id = header_id(f->buf);
            if (n == 0) { f->done = 1; free(id); free(codes); return 0; }
            out->id = id; out->codes = codes; out->len = (int32_t)n; out->after[0] = out->after[1] = 0;
            return 1;
        }
        if (n + MTRH_BLK > cap) { cap = cap ? cap * 2 : 4096; if (cap < n + MTRH_BLK) cap = n + MTRH_BLK; codes = (uint8_t *)xrealloc(codes, cap); }
        uint8_t *d = codes + n;
        uint8_t c;
        while ((c = code_of[*s]) <= 3) { *d++ = c; s++; }
        n = (size_t)(d - codes);
        if (MTR_MAX_INPUT_LENGTH <= (int64_t)n) {       /* the reference stops at the base that reaches the limit (before a later bad character) */
            fprintf(stderr, "fatal error: The length %d is tentatively at most %i.\nread ID = %s\nSet MAX_INPUT_LENGTH to a larger value", MTR_MAX_INPUT_LENGTH, MTR_MAX_INPUT_LENGTH, id ? id : "");
            exit(EXIT_FAILURE);
        }
        if (c == 0xFF) { fprintf(stderr, "Invalid character: %c \n", (char)*s); exit(EXIT_FAILURE); }
    }
    f->done = 1;
    if (n == 0) { free(id); free(codes); return 0; }
    if (!id) { id = (char *)xrealloc(NULL, 1); id[0] = 0; }
    out->id = id; out->codes = codes; out->len = (int32_t)n; out->after[0] = out->after[1] = 0;
    return 1;
}

int mtrh_fasta_next_batch(mtrh_fasta *f, mtrh_read *out, int max_reads, int64_t max_bases)
{
    int n = 0; int64_t bases = 0;
    while (n < max_reads && bases < max_bases) {
        if (!next_read(f, &out[n])) break;
        bases += out[n].len;
        n++;
    }
    return n;
}
