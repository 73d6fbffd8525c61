/* fasta.c — the reference's FASTA reader (handle_one_file.c:169-269) restated over a memory-mapped file, so that a file
 * can be cut at record boundaries and its pieces parsed on several threads / by several processes.
 *
 * What the reference does, and what is kept:
 *   - it reads with fgets(s, BLK = 4096, fp): a "window" is at most 4095 characters and ends after a newline; a line
 *     longer than that is seen as several windows (:208);
 *   - a window whose first character is '>' is a header: the ID is what follows up to NUL / LF / CR (:211-236);
 *   - any other window is sequence: characters up to the first NUL / LF / CR are bases, the rest of the window is
 *     ignored; ACGT/acgt only, anything else is fatal ("Invalid character", :169-188); a read that reaches
 *     MAX_INPUT_LENGTH bases is fatal (:241-246);
 *   - sequence before the first header joins the first record (:213-221); a record without bases ends the input (:283).
 * A header at a line start is always the start of a window, so cutting the file at "\n>" changes nothing.
 */
#define _GNU_SOURCE
#include "mtr_host.h"
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static void *xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n ? n : 1);
    if (!q) { fprintf(stderr, "cannot allocate %zu bytes\n", n); exit(EXIT_FAILURE); }
    return q;
}

int mtrh_file_open(mtrh_file *f, const char *path)
{
    memset(f, 0, sizeof *f);
    f->path = path; f->fd = open(path, O_RDONLY);
    struct stat st;
    if (f->fd < 0 || fstat(f->fd, &st) != 0) {
        fprintf(stderr, "fatal error: cannot open %s\n", path); fflush(stderr);     /* handle_one_file.c:192-196 */
        if (f->fd >= 0) close(f->fd);
        f->fd = -1;
        return 1;
    }
    f->size = (size_t)st.st_size;
    if (f->size > 0) {
        void *m = mmap(NULL, f->size, PROT_READ, MAP_PRIVATE, f->fd, 0);
        if (m == MAP_FAILED) { fprintf(stderr, "fatal error: cannot open %s\n", path); close(f->fd); f->fd = -1; return 1; }
        f->map = (const char *)m;
        (void)madvise(m, f->size, MADV_SEQUENTIAL);
    }
    return 0;
}

void mtrh_file_close(mtrh_file *f)
{
    if (f->map) munmap((void *)f->map, f->size);
    if (f->fd >= 0) close(f->fd);
    f->map = NULL; f->fd = -1; f->size = 0;
}

size_t *mtrh_plan_chunks(const mtrh_file *f, int n_target, int *n_chunks)
{
    if (n_target < 1) n_target = 1;
    size_t n = 0;
    size_t *off = (size_t *)xrealloc(NULL, sizeof(size_t) * ((size_t)n_target + 2));
    off[n++] = 0;
    /* bases before the first header join the first record (:213-221): that header stays in the first piece */
    size_t first_header = 0;
    if (f->size > 0 && f->map[0] != '>') {
        const char *p = f->map, *end = f->map + f->size;
        first_header = f->size;
        while (p < end) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
            if (!q || q + 1 >= end) break;
            if (q[1] == '>') { first_header = (size_t)(q + 1 - f->map); break; }
            p = q + 1;
        }
    }
    for (int k = 1; k < n_target; k++) {
        size_t target = (size_t)((unsigned __int128)f->size * (unsigned)k / (unsigned)n_target);
        if (target <= off[n - 1]) target = off[n - 1] + 1;
        if (target <= first_header) target = first_header + 1;
        if (target >= f->size) break;
        /* the first header at a line start at or after target (the '\n' before it may sit at target - 1) */
        const char *p = f->map + target - 1, *end = f->map + f->size;
        size_t cut = f->size;
        while (p < end) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
            if (!q || q + 1 >= end) break;
            if (q[1] == '>') { cut = (size_t)(q + 1 - f->map); break; }
            p = q + 1;
        }
        if (cut >= f->size) break;
        if (cut > off[n - 1]) off[n++] = cut;
    }
    off[n] = f->size;
    *n_chunks = (int)n;
    return off;
}

/* base code of a character: 0..3 for ACGT/acgt, 0xFE for what ends a window's sequence (NUL, LF, CR), 0xFF = fatal */
static uint8_t code_of[256];
static void init_codes(void)
{
    static int done;
    if (__atomic_load_n(&done, __ATOMIC_ACQUIRE)) return;       /* (two first callers both fill the table with the same bytes) */
    uint8_t t[256];
    memset(t, 0xFF, sizeof t);
    t[0] = t['\n'] = t['\r'] = 0xFE;
    t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
    memcpy(code_of, t, sizeof t);
    __atomic_store_n(&done, 1, __ATOMIC_RELEASE);
}

typedef struct {
    mtrh_batch *head, *cur;
    int64_t cap_reads, cap_codes, cap_words, n_codes;
    int max_reads; int64_t max_bases;
    int64_t hint_codes;                    /* bytes of the chunk still to parse when the batch was opened: its bases cannot be more */
    int pack_only;                         /* no code array: the bases go straight into the 2-bit image (a run without -a / -B never looks at the codes) */
    uint32_t pend; int n_pend;             /* pack_only: the bases of the read that do not fill a word yet (first base in the top bits) */
} builder;

static mtrh_batch *batch_new(void)
{
    mtrh_batch *b = (mtrh_batch *)calloc(1, sizeof *b);
    if (!b) { fprintf(stderr, "cannot allocate a batch\n"); exit(EXIT_FAILURE); }
    return b;
}

static void builder_open(builder *B)
{
    mtrh_batch *b = batch_new();
    if (B->cur) B->cur->next = b; else B->head = b;
    B->cur = b; B->cap_reads = B->cap_codes = B->cap_words = B->n_codes = 0;
}

/* room for one more read of up to `more` further bases in the current batch's code array */
static uint8_t *codes_room(builder *B, int64_t have, int64_t more)
{
    mtrh_batch *b = B->cur;
    if (B->n_codes + have + more > B->cap_codes) {
        /* the first allocation takes what the chunk can hold at most (doubling from 1 MB copied a 24 MB chunk's codes five times over: 31 MB of memcpy) */
        int64_t c = B->cap_codes ? B->cap_codes * 2 : (B->hint_codes > (1 << 20) ? B->hint_codes + MTRH_BLK : (1 << 20));
        while (c < B->n_codes + have + more) c *= 2;
        b->codes = (uint8_t *)xrealloc(b->codes, (size_t)c); B->cap_codes = c;
    }
    return b->codes + B->n_codes;
}

/* pack_only: room for `more` further bases of the read being parsed (its words so far: b->n_words .. the write position), plus the read's closing words */
static uint32_t *words_room(builder *B, int64_t wpos, int64_t more)
{
    mtrh_batch *b = B->cur;
    const int64_t need = wpos + more / 16 + 8;
    if (need > B->cap_words) {
        int64_t c = B->cap_words ? B->cap_words * 2 : (B->hint_codes / 12 > (1 << 18) ? B->hint_codes / 12 : (1 << 18));
        while (c < need) c *= 2;
        b->packed = (uint32_t *)xrealloc(b->packed, sizeof(uint32_t) * (size_t)c); B->cap_words = c;
    }
    return b->packed;
}

static void read_done(builder *B, int64_t len, const char *id, int32_t id_len)
{
    mtrh_batch *b = B->cur;
    if (b->n == B->cap_reads) {
        const int64_t c = B->cap_reads ? B->cap_reads * 2 : 4096;
        b->lens = (int32_t *)xrealloc(b->lens, sizeof(int32_t) * (size_t)c); b->offs = (int64_t *)xrealloc(b->offs, sizeof(int64_t) * (size_t)c);
        b->woff = (int64_t *)xrealloc(b->woff, sizeof(int64_t) * (size_t)c);
        b->ids = (const char **)xrealloc((void *)b->ids, sizeof(char *) * (size_t)c); b->id_lens = (int32_t *)xrealloc(b->id_lens, sizeof(int32_t) * (size_t)c);
        B->cap_reads = c;
    }
    const int64_t nw = mtr_packed_words((int32_t)len);
    if (B->pack_only) {
        /* the full words are written; the partial one (zero if the length is a multiple of 16) and three zero words close the read (mtr_hip.h: the image) */
        uint32_t *w = words_room(B, b->n_words + len / 16, 0) + b->n_words + len / 16;
        w[0] = B->n_pend ? B->pend : 0u; w[1] = 0; w[2] = 0; w[3] = 0;
        B->pend = 0; B->n_pend = 0;
        const int i = b->n++;
        b->lens[i] = (int32_t)len; b->offs[i] = B->n_codes; b->woff[i] = b->n_words; b->ids[i] = id; b->id_lens[i] = id_len;
        b->n_words += nw; B->n_codes += len;
        if (b->n >= B->max_reads || B->n_codes >= B->max_bases) { B->hint_codes = B->hint_codes > B->n_codes ? B->hint_codes - B->n_codes : 0; builder_open(B); }
        return;
    }
    if (b->n_words + nw > B->cap_words) {
        int64_t c = B->cap_words ? B->cap_words * 2 : (B->hint_codes / 12 > (1 << 18) ? B->hint_codes / 12 : (1 << 18));      /* (bases / 16 + 4 words per read: a twelfth of the bytes covers reads of 200 bases and more) */
        while (c < b->n_words + nw) c *= 2;
        b->packed = (uint32_t *)xrealloc(b->packed, sizeof(uint32_t) * (size_t)c); B->cap_words = c;
    }
    const int i = b->n++;
    b->lens[i] = (int32_t)len; b->offs[i] = B->n_codes; b->woff[i] = b->n_words; b->ids[i] = id; b->id_lens[i] = id_len;
    (void)mtr_pack_read(b->codes + B->n_codes, (int32_t)len, b->packed + b->n_words);    /* codes are 0..3 by construction */
    b->n_words += nw; B->n_codes += len;
    if (b->n >= B->max_reads || B->n_codes >= B->max_bases) { B->hint_codes = B->hint_codes > B->n_codes ? B->hint_codes - B->n_codes : 0; builder_open(B); }
}

/* ---- pack_only: a window's bases straight into the read's 2-bit words ---------------------------------------------------------------------------
 * Sixteen characters at a time where the CPU has SSSE3 (every x86-64 since 2006; anything else takes the byte loop): one table look-up by the low nibble of the
 * upper-cased character says which letter a base must be ('A' 1, 'C' 3, 'T' 4, 'G' 7 - the four low nibbles differ) and its code; a block with any other
 * character - the line's end included - goes to the byte loop, which stops exactly where the reader of handle_one_file.c:169-188 stops.  Two multiply-adds fold
 * the sixteen codes into one word, first base in the top bits (mtr_hip.h).  Parsing + packing was 2.7 of the host's 4.6 us of CPU per 2 kb read (bench.py:
 * host_ceiling): a byte at a time into a code array, then a second pass over it to pack. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("ssse3"))) static const unsigned char *pack16_ssse3(const unsigned char *s, const unsigned char *wend, uint32_t *out, int64_t *n_out)
{   /* whole blocks of 16 valid bases from s: their words to out[0..], their number to *n_out; returns where it stopped */
    const char X = (char)0xFF;                           /* no upper-cased character equals it (bit 5 is cleared): NUL and the space must not pass for nibble 0 */
    const __m128i lut_letter = _mm_setr_epi8(X, 'A', X, 'C', 'T', X, X, 'G', X, X, X, X, X, X, X, X);
    const __m128i lut_code = _mm_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m128i up = _mm_set1_epi8((char)0xDF), lo4 = _mm_set1_epi8(0x0F);
    const __m128i m1 = _mm_set1_epi16(0x0104);          /* (c0, c1) -> 4 c0 + c1 */
    const __m128i m2 = _mm_set1_epi32(0x00010010);      /* (p0, p1) -> 16 p0 + p1 */
    const __m128i gather = _mm_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    int64_t n = 0;
    while (wend - s >= 16) {
        const __m128i v = _mm_loadu_si128((const __m128i *)s);
        const __m128i u = _mm_and_si128(v, up);
        const __m128i nib = _mm_and_si128(u, lo4);
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_shuffle_epi8(lut_letter, nib), u)) != 0xFFFF) break;
        const __m128i c = _mm_shuffle_epi8(lut_code, nib);
        const __m128i q = _mm_madd_epi16(_mm_maddubs_epi16(c, m1), m2);
        out[n >> 4] = (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi8(q, gather));
        s += 16; n += 16;
    }
    *n_out = n;
    return s;
}
#endif

/* the bases of one window [s, wend) (up to the first character that is not a base) behind the read's `n` bases so far; returns where it stopped */
static const unsigned char *append_packed(builder *B, const unsigned char *s, const unsigned char *wend, int64_t n)
{
    mtrh_batch *b = B->cur;
    uint32_t *w = words_room(B, b->n_words + n / 16, wend - s) + b->n_words + n / 16;      /* the word the next base goes to */
    uint32_t pend = B->pend; int p = B->n_pend;
#if defined(__x86_64__)
    static int have_ssse3 = -1;                          /* (parser threads race to the same answer: relaxed atomics keep the sanitizer quiet) */
    int have = __atomic_load_n(&have_ssse3, __ATOMIC_RELAXED);
    if (have < 0) { have = __builtin_cpu_supports("ssse3") ? 1 : 0; __atomic_store_n(&have_ssse3, have, __ATOMIC_RELAXED); }
    if (have && wend - s >= 16) {
        if (p == 0) {                                     /* word-aligned: the blocks' words go where they belong */
            int64_t got = 0;
            s = pack16_ssse3(s, wend, w, &got);
            w += got >> 4;
        } else {
            /* behind p pending bases every word of 16 new bases straddles two words of the image */
            uint32_t tmp[64];
            while (wend - s >= 16) {
                const unsigned char *lim = wend - s > 1024 ? s + 1024 : wend;       /* tmp holds the words of 1 024 bases */
                int64_t got = 0;
                const unsigned char *s2 = pack16_ssse3(s, lim, tmp, &got);
                for (int64_t q = 0; q < (got >> 4); q++) { *w++ = pend | (tmp[q] >> (2 * p)); pend = tmp[q] << (32 - 2 * p); }
                const int stopped = lim - s2 >= 16;         /* a block that is not sixteen bases: the byte loop takes it from here */
                s = s2;
                if (stopped || got == 0) break;
            }
        }
    }
#endif
    uint8_t c;
    while (s < wend && (c = code_of[*s]) <= 3) {
        pend |= (uint32_t)c << (30 - 2 * p);
        if (++p == 16) { *w++ = pend; pend = 0; p = 0; }
        s++;
    }
    B->pend = pend; B->n_pend = p;
    return s;
}

static mtrh_batch *parse_chunk(const mtrh_file *f, size_t begin, size_t end, int max_reads, int64_t max_bases, int pack_only);
mtrh_batch *mtrh_parse_chunk(const mtrh_file *f, size_t begin, size_t end, int max_reads, int64_t max_bases) { return parse_chunk(f, begin, end, max_reads, max_bases, 0); }
/* the same without the code array (mtrh_batch.codes = NULL): for runs that only upload the 2-bit image (no -a rows, no -B state) */
mtrh_batch *mtrh_parse_chunk_packed(const mtrh_file *f, size_t begin, size_t end, int max_reads, int64_t max_bases) { return parse_chunk(f, begin, end, max_reads, max_bases, 1); }

static mtrh_batch *parse_chunk(const mtrh_file *f, size_t begin, size_t end, int max_reads, int64_t max_bases, int pack_only)
{
    init_codes();
    builder B; memset(&B, 0, sizeof B);
    B.pack_only = pack_only;
    B.max_reads = max_reads > 0 ? max_reads : 16384; B.max_bases = max_bases > 0 ? max_bases : ((int64_t)512 << 20);
    B.hint_codes = (int64_t)(end - begin) < B.max_bases ? (int64_t)(end - begin) : B.max_bases;
    builder_open(&B);
    const unsigned char *p = (const unsigned char *)f->map + begin, *e = (const unsigned char *)f->map + end;
    const char *id = NULL; int32_t id_len = 0;        /* ID of the record being read ("" until a header was seen) */
    int have_header = 0;                               /* the first header opens the record; at the start of the file bases may
                                                        * precede it and join that record (:213-221); a chunk that starts
                                                        * inside the file starts at a header */
    int64_t n = 0;                                     /* bases of the record being read */
    uint8_t *dst = pack_only ? NULL : codes_room(&B, 0, MTRH_BLK);
    int end_status = MTRH_END_NONE; char bad = 0;
    while (p < e) {
        /* one fgets window: up to 4095 characters, through the newline if it comes earlier */
        size_t wl = (size_t)(e - p) < (size_t)(MTRH_BLK - 1) ? (size_t)(e - p) : (size_t)(MTRH_BLK - 1);
        const unsigned char *nl = (const unsigned char *)memchr(p, '\n', wl);
        if (nl) wl = (size_t)(nl - p) + 1;
        const unsigned char *wend = p + wl;
        if (p[0] == '>') {
            const unsigned char *q = p + 1;
            while (q < wend && *q != 0 && *q != '\n' && *q != '\r') q++;
            const char *nid = (const char *)p + 1; const int32_t nid_len = (int32_t)(q - (p + 1));
            if (!have_header) { have_header = 1; id = nid; id_len = nid_len; p = wend; continue; }   /* the first header: keep feeding the same record */
            if (n == 0) { end_status = MTRH_END_EMPTY; break; }
            read_done(&B, n, id ? id : "", id ? id_len : 0);
            id = nid; id_len = nid_len; n = 0;
            if (!pack_only) dst = codes_room(&B, 0, MTRH_BLK);
            p = wend;
            continue;
        }
        const unsigned char *s = p;
        if (pack_only) {
            s = append_packed(&B, p, wend, n);
            n += (int64_t)(s - p);
        } else {
            dst = codes_room(&B, n, (int64_t)wl) + n;
            uint8_t c; uint8_t *d = dst;
            while (s < wend && (c = code_of[*s]) <= 3) { *d++ = c; s++; }
            n += (int64_t)(d - dst);
        }
        if (MTR_MAX_INPUT_LENGTH <= n) { end_status = MTRH_END_TOOLONG; break; }   /* the reference stops at the base that reaches the limit */
        if (s < wend && code_of[*s] == 0xFF) { end_status = MTRH_END_BADCHAR; bad = (char)*s; break; }
        p = wend;
    }
    if (end_status == MTRH_END_NONE) {
        /* the end of the chunk closes the record being read: the next chunk starts with a header (or the file ends,
         * :251-267); a record without bases ends the input (:283) */
        if (n > 0) read_done(&B, n, id ? id : "", id ? id_len : 0);
        else end_status = MTRH_END_EMPTY;
    }
    /* the status belongs to the last batch that holds reads (or the only, empty one) */
    mtrh_batch *tail = B.head;
    for (mtrh_batch *b = B.head; b; b = b->next) if (b->n > 0 || b == B.head) tail = b;
    for (mtrh_batch *b = tail->next; b; ) { mtrh_batch *nx = b->next; b->next = NULL; mtrh_batch_free(b); b = nx; }
    tail->next = NULL;
    tail->end = end_status; tail->bad_char = bad;
    if (end_status == MTRH_END_TOOLONG) {
        /* the reference prints currentRead->ID here, which still holds the ID of the record BEFORE the one being read
         * (:243; it is set when a record is returned, :226-229); nothing before the first record */
        tail->end_id = NULL; tail->end_id_len = 0;
        if (tail->n > 0) { tail->end_id = tail->ids[tail->n - 1]; tail->end_id_len = tail->id_lens[tail->n - 1]; }
        else if (begin > 0) {
            /* the over-long record is the first of a chunk inside the file: the record before it lives in the previous chunk - its
             * header is the last line start '>' before `begin` (chunks are cut at "\n>") */
            const char *m = (const char *)f->map;
            size_t q = begin - 1;
            while (q > 0 && !(m[q] == '>' && m[q - 1] == '\n')) q--;
            if (m[q] == '>') {
                size_t z = q + 1;
                while (z < begin && z - (q + 1) < (size_t)(MTRH_BLK - 2) && m[z] != 0 && m[z] != '\n' && m[z] != '\r') z++;
                tail->end_id = m + q + 1; tail->end_id_len = (int32_t)(z - (q + 1));
            }
        }
    }
    return B.head;
}

void mtrh_batch_free(mtrh_batch *b)
{
    while (b) {
        mtrh_batch *nx = b->next;
        free(b->lens); free(b->offs); free(b->codes); free(b->woff); free(b->packed); free((void *)b->ids); free(b->id_lens); free(b->id_store);
        free(b);
        b = nx;
    }
}
