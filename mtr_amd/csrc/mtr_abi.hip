// mtr_abi.hip — host side of libmtr_hip.so: the C-ABI of include/mtr_hip.h over the gfx950 kernels.
//
// Data layout in HBM for one resident batch
//   packed   2 bit/base, MSB first, every read word-aligned and followed by 3 zero words (so that the
//            reference's one-past-the-window accesses org[L], org[L+1] read 'A' = 0: isolated semantics)
//   woff/lens/order   per read: word offset, length, processing order (longest first: work balance)
//   ranges   per read a slice [r_off, r_off + L/2+64) of start/end/w/DI-bits written by the range phase, read by the unit phase
//   records  per read max_rec fixed slots written by the unit phase, compacted on the device before the copy back
//   scratch  one slice per resident wavefront, sized for the longest read of the batch (K1Layout/K2Layout)
// A batch runs as the staged chain (launch_staged, k3_staged.hip.inc); the per-read kernel (launch_reads: one wavefront per read,
// the reference's sequential range loop) takes a batch that outgrew one of the chain's buffers, reads that found more records than
// their slots, and MTR_STAGED=0.
// Batch buffers are kept between batches and only grow.
// There is no CPU path: every entry point needs the HIP device the context was created on.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <mutex>

#include "../../include/mtr_hip.h"
#include "../../include/mtr_hip_test.h"
#include "mtr_common.h"
#include "k1_ranges.hip.inc"
#include "k2_units.hip.inc"
#include "k3_staged.hip.inc"

static_assert(sizeof(DevRecord) == sizeof(mtr_record), "device and ABI record layouts must agree");
static_assert(MTR_N_COUNTERS == CNT_N, "counter count");
static_assert(MTR_MAX_READ_LENGTH == MTRC_MAX_SUPPORTED_LENGTH, "read length limit");

// ---- MT19937 (reference MT.h = stock mt19937ar), host-precomputed base stream -------------------------
static void mt_bases(std::vector<uint8_t> &out, size_t n)
{
    uint32_t s[624];
    s[0] = 0u;                                                   // init_genrand(0), fill_directional_index.c:140
    for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
    int idx = 624;
    out.resize(n);
    for (size_t t = 0; t < n; t++) {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
                s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = s[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        out[t] = (uint8_t)(y % 4u);                              // random_base(), fill_directional_index.c:131
    }
}

#define MTR_N_PHASE_TIMES 8                         // ids 0..7 of mtr_get_kernel_times: the launch and the phases of the chain; 8, 9: the two dominant kernels
struct mtr_ctx {
    int device = 0, manhattan = 1; float min_ratio = 0.6f;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t ev_ph[MTR_N_PHASE_TIMES] = {};     // phase boundaries of the staged chain (ev_ph[p] = end of phase p, p = 2..7)
    hipEvent_t ev_ph2[MTR_N_PHASE_TIMES] = {};
    bool last_quads = false;                         // the last chain ran its alignments / revisions four per wavefront (ev_dom recorded)
    hipEvent_t ev_dom[2][2][2] = {};               // [0 = mtr_k_revise_quads, 1 = mtr_k_dp2_quads][pass][start, end]: the two dominant kernels by themselves (ids 8, 9)    // the same for the chain's second pass (ev_ph2[2] = its start); last_two_pass: they were recorded
    bool last_two_pass = false;
    std::string err;
    int n_cu = 256;
    uint8_t *d_mt = nullptr;
    // resident batch
    int n_reads = 0, Lmax = 0;
    std::vector<int32_t> lens; std::vector<int64_t> roff;
    long long packed_words = 0;
    long long st_last_arena_cap = 0;
    int32_t st_cand_cap = 0;
    int st_last_nsub = 1;              // sub-lists of the chain's last launch (the arena's cursors are read after an overflow)
    int64_t st_arena_per_base = 256;   // bytes of candidate arena per read base reserved by the staged chain (x 4 after an overflow)
    uint32_t *d_packed = nullptr; int64_t *d_woff = nullptr; int32_t *d_lens = nullptr, *d_order = nullptr;
    int64_t *d_roff = nullptr; int32_t *d_rcount = nullptr, *d_rstart = nullptr, *d_rend = nullptr, *d_rw = nullptr; uint64_t *d_rdi = nullptr;
    int64_t total_rcap = 0;
    DevRecord *d_records = nullptr; int32_t *d_reccount = nullptr; int max_rec = 0;
    int64_t *d_recoff = nullptr;     // [n_reads+1] exclusive prefix of the record counts (compaction)
    DevRecord *d_out = nullptr;      // compacted records on their way to the host
    int32_t *d_status = nullptr; unsigned int *d_work = nullptr; unsigned long long *d_counters = nullptr;
    uint8_t *d_scratch = nullptr; size_t scratch_bytes = 0;
    int32_t *d_trace = nullptr; unsigned *d_trace_n = nullptr; int trace_cap = 0;
    // range-parallel mode (small batches): work items = (read, range), parked candidate records
    int32_t *d_item_read = nullptr, *d_item_idx = nullptr; int64_t *d_item_off = nullptr;
    int64_t item_cap = 0; bool last_staged = false;
    // staged mode (k3_staged.hip.inc): fixed-capacity buffers of one batch
    uint8_t *d_st_arena = nullptr; int64_t *d_st_kc = nullptr; StDpItem *d_st_dp = nullptr;
    unsigned *d_st_bincnt = nullptr; int32_t *d_st_dpbin = nullptr, *d_st_dprank = nullptr, *d_st_binstart = nullptr, *d_st_sorted = nullptr, *d_st_classwave = nullptr;
    DevRecord *d_st_cand = nullptr; int32_t *d_st_flag = nullptr; unsigned long long *d_st_scalars = nullptr;
    int32_t *d_st_wv = nullptr, *d_st_res = nullptr; int4 *d_st_items = nullptr, *d_st_cont = nullptr; int64_t *d_st_rev = nullptr;
    int8_t *d_st_ipass = nullptr; int32_t *d_st_plist0 = nullptr, *d_st_plist1 = nullptr, *d_st_re2 = nullptr;      // the chain's two passes
    // cost-ordered queue of the per-read unit kernel
    unsigned *d_lpt_count = nullptr; int32_t *d_lpt_start = nullptr, *d_lpt_bin = nullptr, *d_lpt_rank = nullptr, *d_lpt_order = nullptr;
    std::vector<std::pair<void *, size_t>> caps;       // (address of the pointer member, bytes allocated)
    // reads that found more records than their max_rec slots are run again with room for all of them (resolve_overflow)
    std::vector<int32_t> ovf_reads; int ovf_cap = 0;
    DevRecord *d_ovf_records = nullptr; int64_t *d_rec_base = nullptr; int32_t *d_ovf_order = nullptr;
    const DevRecord **d_src = nullptr;                 // per read: where its records are (compaction)
    bool sub_active = false;                           // launch_reads works on ovf_reads with the overflow buffers
    // file-order mode (mtr_upload_batch_in_file): per read the stale tail of the reference's inputString_w_rand
    bool file_order = false; uint16_t *d_tail = nullptr; int64_t *d_tail_off = nullptr;
    std::vector<uint8_t> after;                        // 2 per read: orgInputString[L], [L+1] (zeros unless file-order)
    mtr_kernel_time kt[MTR_N_KERNEL_TIMES] = {};
    unsigned long long counters[CNT_N] = { 0 };
    bool ran = false, pending = false;
    mtr_status run_status = MTR_OK;                    // of the last run: latched until the next upload
    int32_t first_failed = -1; int32_t *d_fail_read = nullptr;   // first read (input order) whose DP exceeded WrapDPsize
    // wire-form fetch: per-read byte sizes / offsets on the device, pinned staging on the host (grow-only)
    int64_t *d_wire_bytes = nullptr, *d_wire_off = nullptr; uint8_t *d_wire = nullptr;
    void *h_counts = nullptr, *h_sizes = nullptr, *h_blob = nullptr; size_t h_counts_cap = 0, h_sizes_cap = 0, h_blob_cap = 0;
    // -a: task buffers of mtr_alignments (grow-only, like the batch buffers)
    int32_t *d_al_i32 = nullptr, *d_al_len = nullptr, *d_al_ends = nullptr; uint8_t *d_al_units = nullptr, *d_al_ops = nullptr; int64_t *d_al_off = nullptr;
    // test entry points
    int32_t *d_t_i32 = nullptr, *d_t_out = nullptr; uint8_t *d_t_units = nullptr;
};

// The MT19937 base stream is the same for every read and every context: one device copy per GPU, shared by the
// contexts on it (a second context costs no generation and no upload).
static std::mutex g_mt_mu;
static struct { uint8_t *d = nullptr; int refs = 0; } g_mt[64];
// launches the host has not waited for, per device (the mode policy asks whether launches overlap, use_staged)
static std::vector<uint8_t> g_mt_host;
static const std::vector<uint8_t> &mt_host()
{   // callers hold g_mt_mu
    if (g_mt_host.empty()) mt_bases(g_mt_host, (size_t)MTRC_MAX_INPUT_LENGTH + 2 * 100000 + 64);
    return g_mt_host;
}

// ---- file-order mode: the host shadow of the reference's process-wide buffers ---------------------------------------
// The reference keeps inputString_w_rand and orgInputString for the whole file (handle_one_file.c:85, mTR.h:65-67).  A
// read rewrites [0, E) of the first (E = max(L + 2r, min(L + 4r, 1e6)), fill_directional_index.c:137-169, three times:
// k = 1, 3, 5, so it LEAVES the k = 5 encoding) and [0, L) of the second; the passes of the next read look up to
// L + r + 2w - k (:232) and its DPs up to org[L + 1], i.e. into what the most recent LONGER read left there.
// That state is a staircase: of all earlier reads only those longer than every read after them still show.
struct mtr_file_state {
    struct Entry { int32_t L = 0, r = 0; int64_t N = 0, n = 0, E = 0; std::vector<uint8_t> codes; };
    std::vector<Entry> stairs;               // E (and L) strictly increasing from back() = most recent to front()
    std::vector<uint8_t> mt;                 // the MT19937 base stream (same as the device's)
    int64_t reads_seen = 0;
    int raw(const Entry &e, int64_t q) const
    {   // the buffer before the rolling encode, as k1_raw (k1_ranges.hip.inc)
        if (q < e.r) return mt[(size_t)(e.N + q)];
        if (q < e.r + e.L) return e.codes[(size_t)(q - e.r)];
        if (q < e.n) return mt[(size_t)(e.N + e.r + (q - e.r - e.L))];
        return mt[(size_t)q];                // q < N
    }
    int left_at(const Entry &e, int64_t p) const
    {   // what the read left at p < E: the 5-mer code where one was formed (:162-168), else the raw entry
        if (p < e.n - 4) { int v = 0; for (int t = 0; t < 5; t++) v = 4 * v + raw(e, p + t); return v; }
        return raw(e, p);
    }
    static void geometry(int32_t L, Entry &e)
    {
        e.L = L; e.r = mtrc_rand_len(L); e.n = (int64_t)L + 2 * e.r;
        e.N = std::min<int64_t>((int64_t)L + 4 * (int64_t)e.r, MTRC_MAX_INPUT_LENGTH);
        e.E = std::max(e.N, e.n);
    }
    // the entries [E, reach) of inputString_w_rand as the NEXT read of length L finds them
    void tail_for(int32_t L, std::vector<uint16_t> &out) const
    {
        Entry me; geometry(L, me);
        int wtop = 0;
        for (int w = MTRC_MIN_WINDOW; w <= MTRC_MAX_WINDOW && w < L / 2; w *= 2) wtop = w;
        const int64_t reach = std::max<int64_t>((int64_t)L + me.r + 2 * wtop + 8, me.E);      // = ncode of k1_read
        int64_t cur = me.E;
        for (size_t k = stairs.size(); k-- > 0 && cur < reach; ) {
            const Entry &e = stairs[k];
            if (e.E <= cur) continue;
            const int64_t end = std::min(e.E, reach);
            for (int64_t p = cur; p < end; p++) out.push_back((uint16_t)left_at(e, p));
            cur = end;
        }
    }
    int org_at(int64_t p) const
    {   // orgInputString[p] for p >= the length of the next read
        for (size_t k = stairs.size(); k-- > 0; ) if (stairs[k].L > p) return stairs[k].codes[(size_t)p];
        return 0;
    }
    void push(const uint8_t *codes, int32_t L)
    {
        Entry e; geometry(L, e);
        while (!stairs.empty() && stairs.back().L <= L) stairs.pop_back();
        e.codes.assign(codes, codes + L);
        stairs.push_back(std::move(e));
        reads_seen++;
    }
};

extern "C" mtr_status mtr_file_state_create(mtr_file_state **out)
{
    if (!out) return MTR_ERR_BAD_ARG;
    mtr_file_state *fs = new (std::nothrow) mtr_file_state();
    if (!fs) { *out = nullptr; return MTR_ERR_OOM; }
    { std::lock_guard<std::mutex> lk(g_mt_mu); fs->mt = mt_host(); }
    *out = fs;
    return MTR_OK;
}
extern "C" void mtr_file_state_destroy(mtr_file_state *fs) { delete fs; }
extern "C" mtr_status mtr_file_state_skip(mtr_file_state *fs, const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n)
{
    if (!fs || !bases || !offsets || !lens || n < 0) return MTR_ERR_BAD_ARG;
    for (int i = 0; i < n; i++) {
        if (lens[i] <= 0 || lens[i] > MTRC_MAX_SUPPORTED_LENGTH) return MTR_ERR_BAD_ARG;
        fs->push(bases + offsets[i], lens[i]);
    }
    return MTR_OK;
}

// test knob: MTR_DP16_MAX_ROWS=0 sends every DP through the 32-bit kernels (the fallbacks of reads > 64 kb)
static int dp16_max_rows() { const char *e = getenv("MTR_DP16_MAX_ROWS"); return e ? atoi(e) : 0x7fffffff; }
static bool dbg() { static int v = -1; if (v < 0) v = getenv("MTR_DEBUG") ? 1 : 0; return v == 1; }
static double dbg_ms() { static const auto t0 = std::chrono::steady_clock::now(); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
#define DBG(...) do { if (dbg()) { fprintf(stderr, "[mtr +%.1f ms] ", dbg_ms()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return MTR_ERR_HIP; } } while (0)

// Blocking copy on the context's own stream.  The stream is created non-blocking and nothing on the run path
// touches the null stream, so two contexts on one device overlap their kernels (a null-stream hipMemcpy would
// wait for every other context's work).
static hipError_t copy_sync(mtr_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}

template <typename T> static void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
// Batch buffers are kept between batches and only grow (hipMalloc / hipFree cost ~0.1-0.5 ms each; a dozen of them per
// batch were a third of a single read's latency).  The capacity of each buffer is remembered by its slot's address.
template <typename T> static hipError_t ensure_dev(mtr_ctx *ctx, T *&p, size_t bytes);

template <typename T> static hipError_t ensure_dev(mtr_ctx *ctx, T *&p, size_t bytes)
{
    void *slot = (void *)&p;
    size_t *cap = nullptr;
    for (auto &c : ctx->caps) if (c.first == slot) cap = &c.second;
    if (!cap) { ctx->caps.emplace_back(slot, (size_t)0); cap = &ctx->caps.back().second; }
    if (p && bytes <= *cap) return hipSuccess;
    dfree(p); *cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 256);          // head room: batches of similar size reuse the buffer
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) *cap = want;
    return e;
}

// forget the resident batch (its buffers stay allocated for the next one)
static void free_batch(mtr_ctx *ctx) { ctx->n_reads = 0; ctx->ran = false; ctx->run_status = MTR_OK; ctx->first_failed = -1; ctx->ovf_reads.clear(); }
// release every batch buffer (mtr_destroy)
static void release_batch_buffers(mtr_ctx *ctx)
{
    dfree(ctx->d_packed); dfree(ctx->d_woff); dfree(ctx->d_lens); dfree(ctx->d_order);
    dfree(ctx->d_roff); dfree(ctx->d_rcount); dfree(ctx->d_rstart); dfree(ctx->d_rend); dfree(ctx->d_rw); dfree(ctx->d_rdi);
    dfree(ctx->d_records); dfree(ctx->d_reccount); dfree(ctx->d_recoff); dfree(ctx->d_out);
    dfree(ctx->d_ovf_records); dfree(ctx->d_rec_base); dfree(ctx->d_ovf_order); dfree(ctx->d_src);
    dfree(ctx->d_item_read); dfree(ctx->d_item_idx); dfree(ctx->d_item_off); ctx->item_cap = 0;
    dfree(ctx->d_tail); dfree(ctx->d_tail_off);
    dfree(ctx->d_st_arena); dfree(ctx->d_st_kc); dfree(ctx->d_st_dp); dfree(ctx->d_st_bincnt); dfree(ctx->d_st_dpbin); dfree(ctx->d_st_dprank);
    dfree(ctx->d_st_binstart); dfree(ctx->d_st_sorted); dfree(ctx->d_st_classwave); dfree(ctx->d_st_cand); dfree(ctx->d_st_flag); dfree(ctx->d_st_scalars); dfree(ctx->d_st_wv); dfree(ctx->d_st_res); dfree(ctx->d_st_items); dfree(ctx->d_st_cont); dfree(ctx->d_st_rev);
    dfree(ctx->d_st_ipass); dfree(ctx->d_st_plist0); dfree(ctx->d_st_plist1); dfree(ctx->d_st_re2);
    dfree(ctx->d_lpt_count); dfree(ctx->d_lpt_start); dfree(ctx->d_lpt_bin); dfree(ctx->d_lpt_rank); dfree(ctx->d_lpt_order);
    dfree(ctx->d_wire_bytes); dfree(ctx->d_wire_off); dfree(ctx->d_wire);
    dfree(ctx->d_al_i32); dfree(ctx->d_al_len); dfree(ctx->d_al_ends); dfree(ctx->d_al_units); dfree(ctx->d_al_ops); dfree(ctx->d_al_off);
    dfree(ctx->d_t_i32); dfree(ctx->d_t_out); dfree(ctx->d_t_units);
    ctx->caps.clear();
    if (ctx->h_counts) { (void)hipHostFree(ctx->h_counts); ctx->h_counts = nullptr; ctx->h_counts_cap = 0; }
    if (ctx->h_sizes) { (void)hipHostFree(ctx->h_sizes); ctx->h_sizes = nullptr; ctx->h_sizes_cap = 0; }
    if (ctx->h_blob) { (void)hipHostFree(ctx->h_blob); ctx->h_blob = nullptr; ctx->h_blob_cap = 0; }
}

// pinned host staging, grow-only
static hipError_t ensure_pinned(void *&p, size_t &cap, size_t bytes)
{
    if (p && bytes <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
}

extern "C" int mtr_abi_version(void) { return MTR_ABI_VERSION; }

extern "C" const char *mtr_last_error(const mtr_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

extern "C" mtr_status mtr_create(int device, int manhattan, float min_match_ratio, mtr_ctx **out)
{
    if (!out) return MTR_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    DBG("mtr_create: begin");
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MTR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MTR_ERR_NO_DEVICE;
    mtr_ctx *ctx = new (std::nothrow) mtr_ctx();
    if (!ctx) return MTR_ERR_OOM;
    ctx->device = device; ctx->manhattan = manhattan ? 1 : 0; ctx->min_ratio = min_match_ratio;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    DBG("mtr_create: runtime up, device selected");
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 4; i++) ok = hipEventCreate(&ctx->ev[i]) == hipSuccess;
    for (int i = 2; ok && i < MTR_N_PHASE_TIMES; i++) ok = hipEventCreate(&ctx->ev_ph[i]) == hipSuccess && hipEventCreate(&ctx->ev_ph2[i]) == hipSuccess;
    for (int i = 0; ok && i < 8; i++) ok = hipEventCreate(&ctx->ev_dom[i >> 2][(i >> 1) & 1][i & 1]) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_status, sizeof(int32_t)) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_work, sizeof(unsigned)) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_counters, sizeof(unsigned long long) * CNT_N) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_trace_n, sizeof(unsigned)) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_fail_read, sizeof(int32_t)) == hipSuccess;
    if (ok && device < 64) {
        // the read-independent MT19937 base stream: first min(L+4r,1e6) draws + two flanks of r <= 1e5
        std::lock_guard<std::mutex> lk(g_mt_mu);
        if (!g_mt[device].d) {
            const std::vector<uint8_t> &mt = mt_host();
            uint8_t *d = nullptr;
            ok = hipMalloc(&d, mt.size()) == hipSuccess && copy_sync(ctx, d, mt.data(), mt.size(), hipMemcpyHostToDevice) == hipSuccess;
            if (ok) g_mt[device].d = d; else if (d) (void)hipFree(d);
        }
        if (ok) { ctx->d_mt = g_mt[device].d; g_mt[device].refs++; }
    } else ok = false;
    if (ok) {
        // test knob (tests/test_gpu_parity.py: the WrapDPsize failure on the GPU): the limit the kernels test, per device; untouched unless asked for
        static long long wrap_set[64];                           // what this process last wrote on the device (0 = the built-in 2e8)
        const char *e = getenv("MTR_TEST_WRAP_DP_SIZE");
        const long long want = e && atoll(e) > 0 ? atoll(e) : 0;
        std::lock_guard<std::mutex> lk(g_mt_mu);
        if (want != wrap_set[device]) {
            const long long v = want ? want : (long long)MTRC_WRAP_DP_SIZE;
            ok = hipMemcpyToSymbol(HIP_SYMBOL(mtr_dev_wrap_dp_size), &v, sizeof v) == hipSuccess;
            if (ok) wrap_set[device] = want;
        }
    }
    if (!ok) { mtr_destroy(ctx); return MTR_ERR_NO_DEVICE; }
    *out = ctx;
    DBG("mtr_create: stream, events, small buffers, MT stream ready");
    return MTR_OK;
}

extern "C" void mtr_destroy(mtr_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->pending) { (void)hipStreamSynchronize(ctx->stream); ctx->pending = false; }
    release_batch_buffers(ctx);
    if (ctx->d_mt) {
        std::lock_guard<std::mutex> lk(g_mt_mu);
        if (--g_mt[ctx->device].refs == 0) { (void)hipFree(g_mt[ctx->device].d); g_mt[ctx->device].d = nullptr; }
        ctx->d_mt = nullptr;
    }
    dfree(ctx->d_status); dfree(ctx->d_work); dfree(ctx->d_counters); dfree(ctx->d_scratch);
    dfree(ctx->d_trace); dfree(ctx->d_trace_n); dfree(ctx->d_fail_read);
    for (int i = 0; i < 4; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < MTR_N_PHASE_TIMES; i++) { if (ctx->ev_ph[i]) (void)hipEventDestroy(ctx->ev_ph[i]); if (ctx->ev_ph2[i]) (void)hipEventDestroy(ctx->ev_ph2[i]); }
    for (int i = 0; i < 8; i++) if (ctx->ev_dom[i >> 2][(i >> 1) & 1][i & 1]) (void)hipEventDestroy(ctx->ev_dom[i >> 2][(i >> 1) & 1][i & 1]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static mtr_status ensure_scratch(mtr_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return MTR_OK;
    // [measured, round 6: BASELINE config 5 through mTR -p -g 1, 15 files of one read each] 14 batches ran on 16.2-16.5 GB of scratch per context, the 15th (the
    // 140 kb read) asked for 16.9 GB: hipFree + hipMalloc of that buffer took 2.6 s of a 3.0 s job - memory the driver has just taken back is slow to hand out
    // again (pick_waves, above).  So (i) a request beyond half the cap takes the whole cap (pick_waves never asks for more: the next batch of any shape fits),
    // and (ii) the new buffer is allocated BEFORE the old one is freed, so that it does not come out of the pages just given back.
    size_t want = bytes;
    {
        const char *e = getenv("MTR_SCRATCH_MAX_GB");
        const size_t cap = (size_t)(e ? atof(e) : 16.0) << 30;
        if (bytes > cap / 2 && bytes < cap) want = cap;
    }
    uint8_t *fresh = nullptr;
    if (hipMalloc(&fresh, want) != hipSuccess) {
        (void)hipGetLastError();
        fresh = nullptr; want = bytes;
        dfree(ctx->d_scratch); ctx->scratch_bytes = 0;                        // (not both at once, then: the old way)
        if (hipMalloc(&fresh, want) != hipSuccess) { ctx->err = "scratch allocation of " + std::to_string(bytes) + " bytes failed"; return MTR_ERR_OOM; }
    }
    dfree(ctx->d_scratch);
    ctx->d_scratch = fresh;
    ctx->scratch_bytes = want;
    return MTR_OK;
}

// number of resident wavefronts for a kernel: as many as the reads, the per-CU occupancy and the memory allow
static int pick_waves(mtr_ctx *ctx, int n_items, int per_cu, size_t per_wave, size_t *total)
{
    size_t free_b = 0, tot_b = 0;
    if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess) free_b = (size_t)8 << 30;
    size_t budget = (size_t)((double)(free_b + ctx->scratch_bytes) * 0.7);
    // Scratch is sized for the worst DP of the longest read (L x 499 cells per wavefront): 4096 wavefronts of 42 kb reads
    // would take 86 GB, and an allocation of that size right after another process freed the memory takes seconds
    // ([measured] 2-6 s for 86 GB, 0.5-1.3 s for 32 GB, none for 8 GB).  Past MTR_SCRATCH_MAX_GB (default 16) fewer
    // wavefronts run; 10 000 reads of 2 kb need 6 GB.
    {
        const char *e = getenv("MTR_SCRATCH_MAX_GB");
        const size_t cap = (size_t)(e ? atof(e) : 16.0) << 30;
        if (budget > cap) budget = cap;
    }
    long waves = (long)ctx->n_cu * per_cu;
    if (waves > n_items) waves = n_items;
    if (waves < 1) waves = 1;
    while (waves > 1 && (size_t)waves * per_wave > budget) waves = waves * 3 / 4;
    *total = (size_t)waves * per_wave;
    return (int)waves;
}

static mtr_status upload_batch(mtr_ctx *ctx, mtr_file_state *fs, const uint32_t *packed_in, int64_t n_words_in, const int64_t *woff_in,
                               const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n);
extern "C" mtr_status mtr_upload_batch(mtr_ctx *ctx, const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n)
{
    if (ctx && (!bases || !offsets)) { ctx->err = "null input"; return MTR_ERR_BAD_ARG; }
    return upload_batch(ctx, nullptr, nullptr, 0, nullptr, bases, offsets, lens, n);
}
extern "C" mtr_status mtr_upload_batch_in_file(mtr_ctx *ctx, mtr_file_state *fs, const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n)
{
    if (!fs) { if (ctx) ctx->err = "null file state"; return MTR_ERR_BAD_ARG; }
    if (ctx && (!bases || !offsets)) { ctx->err = "null input"; return MTR_ERR_BAD_ARG; }
    return upload_batch(ctx, fs, nullptr, 0, nullptr, bases, offsets, lens, n);
}
extern "C" mtr_status mtr_upload_batch_packed(mtr_ctx *ctx, const uint32_t *packed, int64_t n_words, const int64_t *woff, const int32_t *lens, int32_t n)
{
    if (ctx && (!packed || !woff || n_words <= 0)) { ctx->err = "null input"; return MTR_ERR_BAD_ARG; }
    return upload_batch(ctx, nullptr, packed, n_words, woff, nullptr, nullptr, lens, n);
}
static mtr_status upload_batch(mtr_ctx *ctx, mtr_file_state *fs, const uint32_t *packed_in, int64_t n_words_in, const int64_t *woff_in,
                               const uint8_t *bases, const int64_t *offsets, const int32_t *lens, int32_t n)
{
    if (!ctx) return MTR_ERR_BAD_ARG;
    if (!lens || n <= 0) { ctx->err = "null input or n_reads <= 0"; return MTR_ERR_BAD_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    { mtr_status w = mtr_wait(ctx); if (w != MTR_OK && w != MTR_ERR_OVERFLOW && w != MTR_ERR_DP_TOO_LARGE) return w; }
    free_batch(ctx);
    std::vector<int64_t> woff_own;
    std::vector<uint32_t> packed_own;
    const int64_t *woff = woff_in; const uint32_t *packed = packed_in;
    int64_t words = 0; int Lmax = 0;
    for (int i = 0; i < n; i++) {
        if (lens[i] <= 0 || lens[i] > MTRC_MAX_SUPPORTED_LENGTH) { ctx->err = "read " + std::to_string(i) + ": length " + std::to_string(lens[i]) + " outside 1.." + std::to_string(MTRC_MAX_SUPPORTED_LENGTH); return MTR_ERR_BAD_ARG; }
        Lmax = std::max(Lmax, (int)lens[i]);
    }
    if (packed_in) {
        for (int i = 0; i < n; i++)
            if (woff_in[i] < 0 || woff_in[i] + mtr_packed_words(lens[i]) > n_words_in) { ctx->err = "read " + std::to_string(i) + ": words outside the packed image"; return MTR_ERR_BAD_ARG; }
        words = n_words_in;
    } else {
        woff_own.resize((size_t)n);
        for (int i = 0; i < n; i++) { woff_own[(size_t)i] = words; words += mtr_packed_words(lens[i]); }
        packed_own.assign((size_t)words, 0u);
        for (int i = 0; i < n; i++)
            if (mtr_pack_read(bases + offsets[i], lens[i], packed_own.data() + woff_own[(size_t)i]) != MTR_OK) { ctx->err = "read " + std::to_string(i) + ": base code > 3"; return MTR_ERR_BAD_ARG; }
        woff = woff_own.data(); packed = packed_own.data();
    }
    // file-order mode: in file order, what each read finds beyond its own part of the reference's two buffers
    std::vector<uint16_t> tail; std::vector<int64_t> tail_off;
    ctx->after.assign((size_t)n * 2, 0);
    if (fs) {
        tail_off.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; i++) {
            fs->tail_for(lens[i], tail);
            tail_off[(size_t)i + 1] = (int64_t)tail.size();
            uint32_t *w = packed_own.data() + woff_own[(size_t)i];
            for (int64_t p = lens[i]; p < (int64_t)lens[i] + 2; p++) {
                const int b = fs->org_at(p);
                ctx->after[(size_t)i * 2 + (size_t)(p - lens[i])] = (uint8_t)b;
                w[p >> 4] |= (uint32_t)b << (30 - 2 * (int)(p & 15));
            }
            fs->push(bases + offsets[i], lens[i]);
        }
    }
    ctx->file_order = fs != nullptr;
    std::vector<int32_t> order((size_t)n); std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    ctx->roff.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) ctx->roff[(size_t)i + 1] = ctx->roff[(size_t)i] + mtrc_range_cap(lens[i]);
    ctx->total_rcap = ctx->roff[(size_t)n];
    ctx->lens.assign(lens, lens + n);
    ctx->n_reads = n; ctx->Lmax = Lmax;
    ctx->max_rec = 16 + Lmax / 100;
    DBG("upload_batch: %d reads checked and ordered", n);
    HIPCHK(ensure_dev(ctx, ctx->d_packed, ((size_t)words + 80) * 4));      // + 80: the DP stages 64-word blocks, the last block must stay readable
    HIPCHK(ensure_dev(ctx, ctx->d_woff, (size_t)n * 8)); HIPCHK(ensure_dev(ctx, ctx->d_lens, (size_t)n * 4)); HIPCHK(ensure_dev(ctx, ctx->d_order, (size_t)n * 4));
    HIPCHK(ensure_dev(ctx, ctx->d_roff, ((size_t)n + 1) * 8)); HIPCHK(ensure_dev(ctx, ctx->d_rcount, (size_t)n * 4));
    HIPCHK(ensure_dev(ctx, ctx->d_rstart, (size_t)ctx->total_rcap * 4)); HIPCHK(ensure_dev(ctx, ctx->d_rend, (size_t)ctx->total_rcap * 4));
    HIPCHK(ensure_dev(ctx, ctx->d_rw, (size_t)ctx->total_rcap * 4)); HIPCHK(ensure_dev(ctx, ctx->d_rdi, (size_t)ctx->total_rcap * 8));
    HIPCHK(ensure_dev(ctx, ctx->d_records, (size_t)n * (size_t)ctx->max_rec * sizeof(DevRecord)));
    HIPCHK(ensure_dev(ctx, ctx->d_reccount, (size_t)n * 4));
    HIPCHK(ensure_dev(ctx, ctx->d_recoff, ((size_t)n + 1) * 8));
    HIPCHK(ensure_dev(ctx, ctx->d_item_off, ((size_t)n + 1) * 8));
    DBG("upload_batch: batch buffers ready");
    HIPCHK(hipMemcpyAsync(ctx->d_packed, packed, (size_t)words * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_packed + words, 0, 80 * 4, ctx->stream));
    ctx->packed_words = (long long)words + 80;
    HIPCHK(hipMemcpyAsync(ctx->d_woff, woff, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_lens, lens, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_roff, ctx->roff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (fs) {
        HIPCHK(ensure_dev(ctx, ctx->d_tail, std::max<size_t>(tail.size(), 1) * 2)); HIPCHK(ensure_dev(ctx, ctx->d_tail_off, ((size_t)n + 1) * 8));
        if (!tail.empty()) HIPCHK(hipMemcpyAsync(ctx->d_tail, tail.data(), tail.size() * 2, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_tail_off, tail_off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    DBG("upload_batch: copies done");
    return MTR_OK;
}

static BatchView view(const mtr_ctx *ctx)
{
    BatchView b; b.packed = ctx->d_packed; b.woff = ctx->d_woff; b.lens = ctx->d_lens; b.order = ctx->d_order; b.n_reads = ctx->n_reads;
    if (ctx->sub_active) { b.order = ctx->d_ovf_order; b.n_reads = (int32_t)ctx->ovf_reads.size(); }
    return b;
}

static mtr_status check_status(mtr_ctx *ctx)
{
    int32_t st = 0;
    HIPCHK(copy_sync(ctx, &st, ctx->d_status, 4, hipMemcpyDeviceToHost));
    switch (st) {
    case DEV_OK: return MTR_OK;
    case DEV_ERR_RANGE_OVERFLOW: ctx->err = "a read produced more candidate ranges than L/2+64"; return MTR_ERR_OVERFLOW;
    case DEV_ERR_RECORD_OVERFLOW: ctx->err = "a read produced more records than the per-read capacity"; return MTR_ERR_OVERFLOW;
    case DEV_ERR_DP_TOO_LARGE: ctx->err = "You need to increse the value of WrapDPsize. (a DP exceeded the reference's 2e8 cells)"; return MTR_ERR_DP_TOO_LARGE;
    case DEV_ERR_STAGED_OVERFLOW: ctx->err = "a buffer of the staged mode was too small for this batch"; return MTR_ERR_OVERFLOW;
    default: ctx->err = "internal device error " + std::to_string(st); return MTR_ERR_HIP;
    }
}

static int waves_per_cu()
{   // 16 = the LDS / VGPR limit of both kernels; MTR_K2_WAVES_PER_CU is a tuning knob
    int v = getenv("MTR_K2_WAVES_PER_CU") ? atoi(getenv("MTR_K2_WAVES_PER_CU")) : 16;
    return v < 1 ? 1 : (v > 16 ? 16 : v);
}

static void k1_args(mtr_ctx *ctx, K1Args &a, size_t per_wave)
{
    a.b = view(ctx); a.mt = ctx->d_mt; a.manhattan = ctx->manhattan; a.Lmax = ctx->Lmax;
    a.scratch = ctx->d_scratch; a.scratch_per_wave = per_wave;
    a.r_count = ctx->d_rcount; a.r_off = ctx->d_roff; a.r_start = ctx->d_rstart; a.r_end = ctx->d_rend; a.r_w = ctx->d_rw; a.r_di = ctx->d_rdi;
    a.status = ctx->d_status; a.work_counter = ctx->d_work; a.counters = ctx->d_counters;
    a.tail = ctx->file_order ? ctx->d_tail : nullptr; a.tail_off = ctx->file_order ? ctx->d_tail_off : nullptr;
}

// K1 alone (mtr_test_ranges)
static mtr_status launch_k1(mtr_ctx *ctx)
{
    K1Layout y = k1_layout(ctx->Lmax);
    size_t total = 0;
    int waves = pick_waves(ctx, ctx->n_reads, waves_per_cu(), y.total, &total);
    mtr_status s = ensure_scratch(ctx, total); if (s != MTR_OK) return s;
    K1Args a{}; k1_args(ctx, a, y.total);
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(mtr_k1_ranges, dim3((unsigned)waves), dim3(64), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    return MTR_OK;
}

// number of (k,w) passes of a read of length L (fill_directional_index.c:559-582; same enumeration as k1_read)
static int k1_num_passes(int L)
{
    int np = 0;
    for (int k = 1; k <= 5; k += 2) {
        const int max_w = (k == 1) ? 20 : (k == 3 ? 80 : MTRC_MAX_WINDOW);
        for (int w = MTRC_MIN_WINDOW; w <= max_w && w < L / 2; w *= 2) np++;
    }
    return np;
}

// The range finder as three kernels over per-READ scratch: code arrays, one wavefront per (read, pass), extraction +
// de-duplication.  For batches of few reads (the passes of a read otherwise take turns inside one wavefront).
static mtr_status launch_k1_parts(mtr_ctx *ctx)
{
    const int n = ctx->n_reads;
    K1Layout y = k1_layout(ctx->Lmax);
    mtr_status s = ensure_scratch(ctx, (size_t)n * y.total); if (s != MTR_OK) return s;
    std::vector<int32_t> iread, ipass;
    // a pass is cut into position segments (k1_pass_pearson_seg / k1_passes_manhattan; same segment length as k1_seg_len), item = pass | (segment + 1) << 8
    const bool segs = !ctx->file_order;
    for (int i = 0; i < n; i++) {
        const int L = ctx->lens[(size_t)i], np = k1_num_passes(L);
        const int nn = L + 2 * mtrc_rand_len(L);
        int sl = ((nn + 63) / 64 + 63) & ~63; if (sl < 512) sl = 512;
        const int ns = segs ? (nn + sl - 1) / sl : 1;
        for (int p = 0; p < np; p++) for (int sg = 0; sg < ns; sg++) { iread.push_back(i); ipass.push_back(segs ? (p | ((sg + 1) << 8)) : p); }
    }
    const size_t items = iread.size();
    HIPCHK(ensure_dev(ctx, ctx->d_item_read, std::max<size_t>(items, 1) * 4)); HIPCHK(ensure_dev(ctx, ctx->d_item_idx, std::max<size_t>(items, 1) * 4));
    if (items > 0) {
        HIPCHK(hipMemcpyAsync(ctx->d_item_read, iread.data(), items * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->d_item_idx, ipass.data(), items * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    K1Args a{}; k1_args(ctx, a, y.total);
    const int slots = ctx->n_cu * waves_per_cu();
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(mtr_k1_part, dim3((unsigned)std::min(n, slots)), dim3(64), 0, ctx->stream, a, (int)K1_CODES, (const int32_t *)nullptr, (const int32_t *)nullptr, n);
    HIPCHK(hipGetLastError());
    if (items > 0) {
        HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(mtr_k1_part, dim3((unsigned)std::min<size_t>(items, (size_t)slots)), dim3(64), 0, ctx->stream, a, (int)K1_PASSES,
                           (const int32_t *)ctx->d_item_read, (const int32_t *)ctx->d_item_idx, (int)items);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    // extraction + de-duplication: the passes of a read as a pipeline of 16 wavefronts (the file-order mode keeps one wavefront per
    // read: its passes look at the stale tail)
    if (ctx->file_order) hipLaunchKernelGGL(mtr_k1_part, dim3((unsigned)std::min(n, slots)), dim3(64), 0, ctx->stream, a, (int)K1_FINISH, (const int32_t *)nullptr, (const int32_t *)nullptr, n);
    else hipLaunchKernelGGL(mtr_k1_finish_pipe, dim3((unsigned)std::min(n, ctx->n_cu)), dim3(64 * K1_PIPE_WAVES), 0, ctx->stream, a, n, segs ? 1 : 0);
    HIPCHK(hipGetLastError());
    return MTR_OK;
}

static void k2_args(mtr_ctx *ctx, K2Args &a, size_t per_wave)
{
    a.b = view(ctx); a.min_match_ratio = ctx->min_ratio; a.Lmax = ctx->Lmax;
    a.scratch = ctx->d_scratch; a.scratch_per_wave = per_wave;
    a.r_count = ctx->d_rcount; a.r_off = ctx->d_roff; a.r_start = ctx->d_rstart; a.r_end = ctx->d_rend; a.r_w = ctx->d_rw;
    a.records = ctx->d_records; a.max_rec_per_read = ctx->max_rec; a.rec_count = ctx->d_reccount; a.rec_base = nullptr;
    if (ctx->sub_active) { a.records = ctx->d_ovf_records; a.max_rec_per_read = ctx->ovf_cap; a.rec_base = ctx->d_rec_base; }
    a.status = ctx->d_status; a.fail_read = ctx->d_fail_read; a.work_counter = ctx->d_work; a.counters = ctx->d_counters;
    a.trace = ctx->d_trace; a.trace_cap = ctx->trace_cap; a.trace_n = ctx->d_trace_n;
    a.trace_mask = getenv("MTR_TRACE_MASK") ? (int32_t)strtol(getenv("MTR_TRACE_MASK"), nullptr, 0) : -1;
    a.dp16_max_rows = dp16_max_rows();
    a.packed_words = ctx->packed_words;
}

// One wavefront per read as TWO kernels: the range kernel, then the unit kernel (mtr_k_units) with its work queue ordered by
// the predicted cost of the reads (k3_staged.hip.inc: mtr_k_cost_*), each with its own scratch layout over the same
// allocation.  The file-order mode always runs this way (its range kernel reads the stale tails).
static mtr_status launch_two_kernels(mtr_ctx *ctx)
{
    const int n = ctx->n_reads;
    K1Layout y1 = k1_layout(ctx->Lmax); K2Layout y2 = k2_layout(ctx->Lmax);
    size_t t1 = 0, t2 = 0;
    const int w1 = pick_waves(ctx, n, waves_per_cu(), y1.total, &t1);
    const int w2 = pick_waves(ctx, n, waves_per_cu(), y2.total, &t2);
    mtr_status s = ensure_scratch(ctx, std::max(t1, t2)); if (s != MTR_OK) return s;
    HIPCHK(ensure_dev(ctx, ctx->d_lpt_count, (size_t)LPT_BINS * 4)); HIPCHK(ensure_dev(ctx, ctx->d_lpt_start, (size_t)LPT_BINS * 4));
    HIPCHK(ensure_dev(ctx, ctx->d_lpt_bin, (size_t)n * 4)); HIPCHK(ensure_dev(ctx, ctx->d_lpt_rank, (size_t)n * 4)); HIPCHK(ensure_dev(ctx, ctx->d_lpt_order, (size_t)n * 4));
    K1Args a1{}; k1_args(ctx, a1, y1.total);
    K2Args a{}; k2_args(ctx, a, y2.total);
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_trace_n, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_lpt_count, 0, (size_t)LPT_BINS * 4, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    hipLaunchKernelGGL(mtr_k1_ranges, dim3((unsigned)w1), dim3(64), 0, ctx->stream, a1);
    HIPCHK(hipGetLastError());
    if (!ctx->sub_active) {                              // (the few reads of an overflow rerun keep their order)
        const unsigned g = (unsigned)std::min(1024, (n + 255) / 256);
        hipLaunchKernelGGL(mtr_k_cost_bins, dim3(g), dim3(256), 0, ctx->stream, a, ctx->d_lpt_count, ctx->d_lpt_bin, ctx->d_lpt_rank);
        hipLaunchKernelGGL(mtr_k_cost_scan, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned *)ctx->d_lpt_count, ctx->d_lpt_start);
        hipLaunchKernelGGL(mtr_k_cost_order, dim3(g), dim3(256), 0, ctx->stream, (int32_t)n, (const int32_t *)ctx->d_lpt_start, (const int32_t *)ctx->d_lpt_bin,
                           (const int32_t *)ctx->d_lpt_rank, ctx->d_lpt_order);
        HIPCHK(hipGetLastError());
        a.b.order = ctx->d_lpt_order;
    }
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(mtr_k_units, dim3((unsigned)w2), dim3(64), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    return MTR_OK;
}

// the per-read kernel: ranges (K1 code) and unit search / DP (K2 code) of a read by the same wavefront
static mtr_status launch_reads(mtr_ctx *ctx)
{
    if (ctx->file_order) return launch_two_kernels(ctx);
    K1Layout y1 = k1_layout(ctx->Lmax); K2Layout y2 = k2_layout(ctx->Lmax);
    const size_t per_wave = std::max(y1.total, y2.total);       // the two phases of a read use the arena one after the other
    size_t total = 0;
    int waves = pick_waves(ctx, ctx->n_reads, waves_per_cu(), per_wave, &total);
    mtr_status s = ensure_scratch(ctx, total); if (s != MTR_OK) return s;
    K1Args a1{}; k1_args(ctx, a1, per_wave);
    K2Args a{}; k2_args(ctx, a, per_wave);
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_trace_n, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    if (ctx->manhattan) hipLaunchKernelGGL(mtr_k_reads<true>, dim3((unsigned)waves), dim3(64), 0, ctx->stream, a1, a);
    else hipLaunchKernelGGL(mtr_k_reads<false>, dim3((unsigned)waves), dim3(64), 0, ctx->stream, a1, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    return MTR_OK;
}

// Staged mode (k3_staged.hip.inc): ranges of every read, then every (read, range) through phase 1, then ALL two-parameter DPs
// of the batch one per lane, then tracebacks + selection + revision per range, then the replay of the sequential loop.
// One stream, no host round trip; every buffer has a capacity fixed here, and a batch that outgrows one is run again
// by the per-read kernel (mtr_wait).  MTR_STAGED=0/1 forces it off/on.
static int staged_quad_min(int64_t bases)
{   // Alignments of units of up to 128 bases go four per wavefront (mtr_k_dp2_quads) in batches of at least 8 M bases (4 000 reads
    // of 2 kb): smaller ones do not give every SIMD a wavefront that way, and then more wavefronts count for more than fewer
    // instructions per row.  The value is the number of such alignments the batch must have (checked on the device).
    // MTR_QUAD_MIN forces it (tests: 1 = always, 0 = never).
    const char *e = getenv("MTR_QUAD_MIN");
    if (e) { const long v = atol(e); return (int)(v < 0 ? 0 : (v > 0x7fffffff ? 0x7fffffff : v)); }
    return bases >= 8000000 ? 1024 : 0;
}
static int staged_two_pass(int64_t bases, int n_reads)
{   // the chain in two passes - the ranges of wide windows first, then what their records leave (k3_staged.hip.inc): 1 = the wide ranges that no earlier
    // wide range contains first, 2 = every wide range first.  MTR_TWO_PASS forces it.
    // [measured, MI355X, host call on one resident batch, ms: one pass / two passes / every wide range first]  10 000 reads of 2 kb 53.6 / 61.4 / 56.2
    // (pipelined over two contexts 46.7 / 47.6 / 47.0 a step); 20 000 config-4 reads 82.0 / 91.6 / 84.1; 1 000 config-2 reads 8.4 / 9.5 / 9.5; one 2 kb
    // read 2.4 / 4.0 / 2.5; 100 reads of 42 kb 120.7 / 127.4 / 123.3; 600 reads of 42 kb 437.9 / 404.5 / 434.1.  Two passes search what the reference
    // searches (+ 0.4 %) instead of 1.43 x (2 kb reads) or 5.9 x (42 kb reads) that, with 24 % / 80 % fewer k-mer tables - and pay every kernel's tail
    // twice: they win where the work saved is large and the batch deep enough to hide the tails, i.e. many long reads (1).  With EVERY wide range in
    // the first pass (2) the second pass has nothing long in it: pipelined over two contexts it costs nothing (final build, 20 steps, twice: 44.32 /
    // 44.35 ms a step in one pass, 44.42 / 44.32 with every wide range first, 45.30 / 45.22 in mode 1) while a lone launch takes 53.7 instead of
    // 51.8 ms.  So: batches too small for the four-per-wavefront passes (< 8 M bases, where a launch's own latency is what counts) keep one pass;
    // deep batches of long reads take mode 1; every other big batch mode 2 - the chain then searches what the reference searches.
    const char *e = getenv("MTR_TWO_PASS");
    if (e) return atoi(e);
    // Round 5: long reads take mode 1 from 2 M bases on (was 16 M).  What one pass searches beyond the reference grows with the read: x 5.9 the ranges and x 2.85
    // the DP calls on 42 kb reads (config 3's 100 reads = 4.2 M bases, the batch the bench line and the command line run); a lone launch pays the
    // second pass's tails (120.7 -> 127.4 ms), launches pipelined over two contexts - what both of them do - get the saved work.
    static const long long long_min = getenv("MTR_TWO_PASS_LONG_MIN_BASES") ? atoll(getenv("MTR_TWO_PASS_LONG_MIN_BASES")) : ((long long)2 << 20);
    if (bases >= long_min && bases / std::max(n_reads, 1) >= 8000) return 1;
    return bases >= 8000000 ? 2 : 0;
}
static unsigned st_sum(const unsigned long long *cls) { unsigned t = 0; for (int c = 0; c < ST_NCLS; c++) t += (unsigned)cls[c * 32]; return t; }
static bool use_staged(const mtr_ctx *)
{   // [measured, round 3] the chain is the faster arrangement for every batch: a single 2 kb read 3.0 against 19 ms, 2 000 reads 19
    // against 38 ms, 10 000 reads 61 against 75 ms for a launch that has the GPU to itself, and 49.5 against 52 ms a step when two
    // contexts keep launches in flight (since the alignments and revisions of big batches run four per wavefront).  MTR_STAGED=0
    // selects the per-read kernel (tests, diagnosis).
    const char *e = getenv("MTR_STAGED");
    return e ? atoi(e) != 0 : true;
}

static mtr_status launch_reads(mtr_ctx *ctx);
static mtr_status launch_staged(mtr_ctx *ctx)
{
    // a buffer of the chain that cannot be allocated (a GPU shared with other ranks or contexts): the per-read kernel takes the batch
    bool alloc_failed = false;
#define ST_ALLOC(call) do { if (!alloc_failed && (call) != hipSuccess) { (void)hipGetLastError(); alloc_failed = true; } } while (0)
    const int n = ctx->n_reads;
    int64_t sumL = 0; for (int i = 0; i < n; i++) sumL += ctx->lens[(size_t)i];
    StagedArgs s{};
    s.n_reads = n;
    size_t free_b = 0, tot_b = 0;
    if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess) free_b = (size_t)16 << 30;
    // parked candidates: 5 bytes per unit base and slot; [measured] 2 kb reads use ~52 bytes per read base, reads of long units (unit
    // 200 x 200 copies: ~1 KB per candidate, ~20 per range) several times that.  A batch that outgrows the arena is run once more with
    // four times the room per base (mtr_wait), then by the per-read kernel.
    const int64_t per_base = ctx->st_arena_per_base;
    s.arena_cap = std::min<int64_t>(sumL * per_base + (16 << 20), (int64_t)((double)free_b * 0.2) + (ctx->d_st_arena ? sumL * per_base : 0));
    s.kc_cap = (int32_t)std::min<int64_t>(ctx->total_rcap, sumL / 8 + 4096);
    s.dp_cap = (int32_t)std::min<int64_t>(0x7fffff00, sumL / 8 + 4096);
    s.sorted_cap = s.dp_cap + 4 * ST_QCLASSES;
    s.cand_cap = (int32_t)std::min<int64_t>(0x7fffff00, (int64_t)n * 8 + sumL / 256 + 1024);
    // lists in 64 sub-lists with a counter each from 4 M bases (k3_staged.hip.inc: one counter completes 88 M appends a second)
    s.nsub = sumL >= (4 << 20) ? 64 : 1;
    long test_rev_cap = -1, test_cont_cap = -1;
    if (const char *e = getenv("MTR_TEST_STAGED_CAPS")) {
        // tests only: "arena=<bytes>,kc=<n>,dp=<n>,cand=<n>,cont=<n>,rev=<n>" shrinks capacities so that every overflow path of the
        // chain is taken (the batch must then come out of the per-read kernel with the same records)
        auto cap = [&](const char *key) -> long { const char *q = strstr(e, key); return q ? atol(q + strlen(key)) : -1; };
        if (cap("arena=") >= 0) s.arena_cap = cap("arena=");
        if (cap("kc=") >= 0) s.kc_cap = (int32_t)cap("kc=");
        if (cap("dp=") >= 0) s.dp_cap = (int32_t)cap("dp=");
        if (cap("cand=") >= 0) s.cand_cap = (int32_t)cap("cand=");
        test_cont_cap = cap("cont="); test_rev_cap = cap("rev=");
        if (cap("nsub=") == 1 || cap("nsub=") == 64) s.nsub = (int32_t)cap("nsub=");
        s.sorted_cap = s.dp_cap + 4 * ST_QCLASSES;
    }
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_arena, (size_t)s.arena_cap));
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_kc, (size_t)s.kc_cap * 8 * 2));      /* (the block list, and behind it the list of the blocks with a revision due) */ ST_ALLOC(ensure_dev(ctx, ctx->d_st_dp, (size_t)s.dp_cap * sizeof(StDpItem)));
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_bincnt, (size_t)ST_NBINS * 4)); ST_ALLOC(ensure_dev(ctx, ctx->d_st_binstart, ((size_t)ST_NBINS + 1) * 4));
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_dpbin, (size_t)s.dp_cap * 4)); ST_ALLOC(ensure_dev(ctx, ctx->d_st_dprank, (size_t)s.dp_cap * 4));
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_sorted, (size_t)s.sorted_cap * 4)); ST_ALLOC(ensure_dev(ctx, ctx->d_st_classwave, 16 * 4));
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_cand, (size_t)s.cand_cap * sizeof(DevRecord))); ST_ALLOC(ensure_dev(ctx, ctx->d_st_flag, (size_t)std::max<int64_t>(ctx->total_rcap, 1) * 4));
    // device scalars, each on a 256-byte line of its own (same-line atomics complete one after the other), then the work queues
    const size_t st_scalar_bytes = 64 * 256 + (size_t)ST_N_QUEUES * WQ_WORDS * 4;
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_scalars, st_scalar_bytes));
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_wv, (size_t)ST_NCLS * (size_t)s.dp_cap * 4)); ST_ALLOC(ensure_dev(ctx, ctx->d_st_res, (size_t)s.dp_cap * 16 * 4));
    s.quad_min = staged_quad_min(sumL); s.wv_items = ctx->d_st_wv; s.dp_res = ctx->d_st_res;
    s.item_cap = (int32_t)std::min<int64_t>(0x7fffff00, ctx->total_rcap);
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_items, (size_t)std::max(s.item_cap, 1) * sizeof(int4)));
    s.item_tab = ctx->d_st_items;
    s.cont_cap = (int32_t)std::min<int64_t>(0x7fffff00, 2 * (int64_t)s.kc_cap);
    if (test_cont_cap >= 0) s.cont_cap = (int32_t)test_cont_cap;
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_cont, (size_t)ST_NCLS * (size_t)s.cont_cap * sizeof(int4)));
    s.cont = ctx->d_st_cont;
    s.k_first = 3;
    s.rev_cap = test_rev_cap >= 0 ? (int32_t)test_rev_cap : s.kc_cap;
    ST_ALLOC(ensure_dev(ctx, ctx->d_st_rev, (size_t)ST_NCLS * (size_t)s.rev_cap * 8));
    s.rev_items = (long long *)ctx->d_st_rev;
    unsigned long long *sc = ctx->d_st_scalars;
    s.item_off = ctx->d_item_off; s.n_items = (int32_t *)(sc + 0 * 32);
    s.arena = ctx->d_st_arena; ctx->st_last_arena_cap = s.arena_cap; ctx->st_last_nsub = s.nsub;
    s.kc_items = (long long *)ctx->d_st_kc; s.rb_items = s.kc_items + s.kc_cap;
    s.dp = ctx->d_st_dp;
    s.bin_count = ctx->d_st_bincnt; s.dp_bin = ctx->d_st_dpbin; s.dp_rank = ctx->d_st_dprank;
    s.bin_start = ctx->d_st_binstart; s.sorted = ctx->d_st_sorted; s.quad_cls = ctx->d_st_classwave;
    s.cand = ctx->d_st_cand; s.n_cand = (unsigned *)(sc + 5 * 32); s.cand_flag = ctx->d_st_flag;
    s.n_rev = (unsigned *)(sc + 32 * 32);   // ST_NCLS counters, 256 bytes apart
    s.work = (unsigned *)(sc + 64 * 32);
    // wide windows first (k3_staged.hip.inc): the chain in two passes.  MTR_TWO_PASS=0 / 1 overrides
    s.two_pass = staged_two_pass(sumL, n); s.pass = 0; s.defer_w = 160;
    // the alignment passes in a scattered order (k3_staged.hip.inc: st_scatter_stride): alone the kernel is 2 % slower (12.1 against 11.9 ms), two contexts'
    // launches overlapped are 0.6 ms a step faster (36.6 against 37.2 ms, two runs each) - what runs next to the other context's kernels is a steady mixture of
    // shapes instead of one shape at a time.  MTR_PASS_SHUFFLE=0: list order (development)
    { static const int shuffle = getenv("MTR_PASS_SHUFFLE") ? atoi(getenv("MTR_PASS_SHUFFLE")) : 1; s.pass_shuffle = shuffle; }
    if (s.two_pass) {
        ST_ALLOC(ensure_dev(ctx, ctx->d_st_ipass, (size_t)std::max(s.item_cap, 1))); ST_ALLOC(ensure_dev(ctx, ctx->d_st_plist0, (size_t)std::max(s.item_cap, 1) * 4));
        ST_ALLOC(ensure_dev(ctx, ctx->d_st_plist1, (size_t)std::max(s.item_cap, 1) * 4)); ST_ALLOC(ensure_dev(ctx, ctx->d_st_re2, (size_t)std::max<int64_t>(ctx->total_rcap, 1) * 4));
    }
    s.item_pass = ctx->d_st_ipass; s.pass_list[0] = ctx->d_st_plist0; s.pass_list[1] = ctx->d_st_plist1; s.re2 = ctx->d_st_re2;
    s.n_pass[0] = (unsigned *)(sc + 6 * 32); s.n_pass[1] = (unsigned *)(sc + 7 * 32);
    ctx->last_two_pass = s.two_pass != 0;
    ctx->last_quads = s.quad_min > 0;
    if (const char *e = getenv("MTR_TEST_STAGED_FLAGS")) s.test_flags = atoi(e);
    K1Layout y1 = k1_layout(ctx->Lmax); K2Layout y2 = k2_layout(ctx->Lmax);
    // few reads: the range finder as one wavefront per (read, pass) over per-READ scratch (launch_k1_parts), as in the
    // range-parallel mode; the unit kernels then take their per-wavefront scratch behind it
    const long parts_max = 256;
    bool parts = n <= parts_max;
    if (parts) { size_t tot = 0; parts = pick_waves(ctx, n, n, y1.total, &tot) == n; }
    const size_t per_wave = parts ? y2.total : std::max(y1.total, y2.total);
    size_t total = 0;
    const int waves = pick_waves(ctx, std::max(n * 8, ctx->n_cu * 16), waves_per_cu(), per_wave, &total);
    const int waves1 = std::min(waves, n);
    size_t total_dp = 0;                                    // the per-DP kernel runs at twice the occupancy (no LDS table, 64 VGPRs)
    const int waves_dp = pick_waves(ctx, std::max(n * 8, ctx->n_cu * 32), 32, per_wave, &total_dp);
    if (alloc_failed) {
        DBG("staged chain: a buffer could not be allocated (%.1f GB free): the per-read kernel takes the batch", (double)free_b / 1e9);
        dfree(ctx->d_st_arena); for (auto &c : ctx->caps) if (c.first == (void *)&ctx->d_st_arena) c.second = 0;
        ctx->last_staged = false;
        return launch_reads(ctx);
    }
    DBG("launch_staged: chain buffers ready (arena %.0f MB)", (double)s.arena_cap / 1e6);
    mtr_status st = ensure_scratch(ctx, std::max(std::max(total, total_dp), parts ? (size_t)n * y1.total : (size_t)0)); if (st != MTR_OK) return st;
    DBG("launch_staged: scratch ready (%.0f MB)", (double)ctx->scratch_bytes / 1e6);
    K1Args a1{}; k1_args(ctx, a1, per_wave);
    K2Args a{}; k2_args(ctx, a, per_wave);
    HIPCHK(hipMemsetAsync(ctx->d_trace_n, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_st_scalars, 0, st_scalar_bytes, ctx->stream));
    if (s.quad_min > 0) HIPCHK(hipMemsetAsync(ctx->d_st_bincnt, 0, (size_t)ST_NBINS * 4, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    if (parts) { mtr_status ps = launch_k1_parts(ctx); if (ps != MTR_OK) return ps; }
    else {
        HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(mtr_k1_ranges, dim3((unsigned)waves1), dim3(64), 0, ctx->stream, a1);
        HIPCHK(hipGetLastError());
    }
    // Few reads: not many more wavefronts than there can be work items.  Every wavefront of a persistent grid pulls from its
    // queue at least once, and pulls on one counter complete one after the other (11 ns each): 4 096 - 8 192 wavefronts starting
    // on a single read's dozens of items were 46 - 93 us per kernel, ~0.4 ms of a 3.2 ms call.  The bounds are generous
    // estimates from the bases of the batch (2 kb reads have ~0.11 ranges, 0.02 alignments, 0.002 (range, k) searches and
    // revisions per base; 42 kb reads fewer); a batch that has more items than wavefronts only takes longer.
    auto capped = [&](int g, int64_t per_bases) { return (int)std::min<int64_t>(g, std::max<int64_t>(64, sumL / per_bases + 64)); };
    // wavefronts per CU of the chain's service kernels (per block / per revision: a few dependent memory round trips per item and next to no instructions).  A
    // persistent grid holds its wavefront slots for the kernel's whole duration: sized by the chip (32 per CU) these kernels held a sixth of the step's slot time
    // for 3 % of its instructions.  MTR_SERVICE_WPC / MTR_SELECT_WPC (development) override.
    // [measured, round 6, one box, three rounds each] gather / rev_share / finish on 32 / 16 / 8 / 4 / 2 / 1 wavefronts per CU: the step 34.95 / 34.89 / 34.75 /
    // 34.6 / 34.7 / 34.8 ms, a lone launch 44.1 / 43.7 / 43.7 / 43.5 / 44.2 / 45.4 ms - their items are three contended atomics and a store fence each, and a
    // quarter of the wavefronts contend less; so four per CU for a batch of the headline's size, growing with the batch (a wavefront per 20 000 bases) up to the
    // 32 per CU that the 100 000-read batches had.  (The selection and the polish kernel on fewer wavefronts: nothing, or slower.)
    static const int service_wpc_env = getenv("MTR_SERVICE_WPC") ? atoi(getenv("MTR_SERVICE_WPC")) : 0;
    const int service_waves = service_wpc_env > 0 ? ctx->n_cu * service_wpc_env
                                                  : (int)std::min<int64_t>((int64_t)ctx->n_cu * 32, std::max<int64_t>((int64_t)ctx->n_cu * 4, sumL / 20000));
    static const int select_wpc = getenv("MTR_SELECT_WPC") ? atoi(getenv("MTR_SELECT_WPC")) : 16;
    static const int polish_wpc = getenv("MTR_POLISH_WPC") ? atoi(getenv("MTR_POLISH_WPC")) : 16;
    hipLaunchKernelGGL(mtr_k_items, dim3(1), dim3(1024), 0, ctx->stream, (const int32_t *)ctx->d_rcount, s);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(mtr_k_item_table, dim3((unsigned)std::min(n, 4096)), dim3(256), 0, ctx->stream, a, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev_ph[2], ctx->stream));       // ranges
    // one pass of the chain, from the walks to the candidate records (two passes: the work lists of the chain start from zero for the second)
    auto run_pass = [&](int pass) -> mtr_status {
        hipEvent_t *evp = pass == 0 ? ctx->ev_ph : ctx->ev_ph2;
        s.pass = pass;
        {   // the walks align nothing: a scratch layout without the cell region, and as many wavefronts as the occupancy allows
            K2Args aw = a;
            aw.cells_cap = 256;
            const size_t pw = k2_layout(ctx->Lmax, aw.cells_cap).total;
            size_t tw = 0;
            int ww = pick_waves(ctx, std::max(n * 8, ctx->n_cu * 16), waves_per_cu(), pw, &tw);
            if (tw > ctx->scratch_bytes) ww = std::min(ww, waves);
            aw.scratch_per_wave = pw;
            hipLaunchKernelGGL(mtr_k_walks, dim3((unsigned)capped(ww, 8)), dim3(64), 0, ctx->stream, aw, s);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(mtr_k_walks_k, dim3((unsigned)capped(ww, 32)), dim3(64), 0, ctx->stream, aw, s);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(mtr_k_gather, dim3((unsigned)capped(service_waves, 64)), dim3(64), 0, ctx->stream, a, s);      // (19 VGPRs, no LDS: eight wavefronts per SIMD hide its loads)
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(evp[3], ctx->stream));       // unit search (tables, seeds, walks) + the alignment items
        if (s.quad_min > 0) {
            hipLaunchKernelGGL(mtr_k_qbins, dim3(1), dim3(1024), 0, ctx->stream, s, 0);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(mtr_k_qscatter, dim3((unsigned)capped(512, 16384)), dim3(256), 0, ctx->stream, s, ctx->d_status);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(ctx->ev_dom[1][pass][0], ctx->stream));
            hipLaunchKernelGGL(mtr_k_dp2_quads, dim3((unsigned)capped(waves, 64)), dim3(64), 0, ctx->stream, a, s);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(ctx->ev_dom[1][pass][1], ctx->stream));
        }
        if (s.quad_min <= 0) {                                  // (a big batch: mtr_k_dp2_quads has run them)
            hipLaunchKernelGGL(mtr_k_dp2_waves, dim3((unsigned)capped(waves_dp, 16)), dim3(64), 0, ctx->stream, a, s);
            HIPCHK(hipGetLastError());
        }
        // (the bins' counts: mtr_k_qbins clears them behind its last read - the revisions are sorted next)
        HIPCHK(hipEventRecord(evp[4], ctx->stream));       // two-parameter alignments
        hipLaunchKernelGGL(mtr_k_select, dim3((unsigned)capped(std::min(waves, ctx->n_cu * select_wpc), 64)), dim3(64), 0, ctx->stream, a, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(evp[5], ctx->stream));       // selection
        if (s.quad_min > 0) {
            // four revisions per wavefront.  First every revision is polished (a work item each), then the revisions of a range whose polished
            // states are equal are joined and the ones that run are binned (the alignments' sort buffers are free again)
            {
                K2Args ap = a;
                ap.cells_cap = 256;                             // (polish aligns nothing: the scratch layout of the walks)
                const size_t pw = k2_layout(ctx->Lmax, ap.cells_cap).total;
                size_t tw = 0;
                int wp = pick_waves(ctx, std::max(n * 8, ctx->n_cu * 16), waves_per_cu(), pw, &tw);
                if (tw > ctx->scratch_bytes) wp = std::min(wp, waves);
                ap.scratch_per_wave = pw;
                hipLaunchKernelGGL(mtr_k_polish, dim3((unsigned)capped(std::min(wp, ctx->n_cu * polish_wpc), 256)), dim3(64), 0, ctx->stream, ap, s);
                HIPCHK(hipGetLastError());
            }
            hipLaunchKernelGGL(mtr_k_rev_share, dim3((unsigned)capped(service_waves, 64)), dim3(64), 0, ctx->stream, a, s);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(mtr_k_qbins, dim3(1), dim3(1024), 0, ctx->stream, s, 1);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(mtr_k_rscatter, dim3((unsigned)capped(512, 16384)), dim3(256), 0, ctx->stream, s, ctx->d_status);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(ctx->ev_dom[0][pass][0], ctx->stream));
            // eight slots per wavefront: MTR_REV_WAVES_PER_CU (development) bounds the grid - fewer wavefronts = more revisions in a row per slot
            static const int rev_wpc = getenv("MTR_REV_WAVES_PER_CU") ? atoi(getenv("MTR_REV_WAVES_PER_CU")) : 0;
            const int waves_rev = rev_wpc > 0 ? std::min(waves, ctx->n_cu * rev_wpc) : waves;
            hipLaunchKernelGGL(mtr_k_revise_quads, dim3((unsigned)capped(waves_rev, 256)), dim3(64), 0, ctx->stream, a, s);
            HIPCHK(hipEventRecord(ctx->ev_dom[0][pass][1], ctx->stream));
        } else hipLaunchKernelGGL(mtr_k_revise, dim3((unsigned)capped(waves, 64)), dim3(64), 0, ctx->stream, a, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(evp[6], ctx->stream));       // revisions
        hipLaunchKernelGGL(mtr_k_finish, dim3((unsigned)capped(service_waves, 64)), dim3(64), 0, ctx->stream, a, s);
        HIPCHK(hipGetLastError());
        return MTR_OK;
    };
    { mtr_status ps = run_pass(0); if (ps != MTR_OK) return ps; }
    if (s.two_pass) {
        hipLaunchKernelGGL(mtr_k_pass_mark, dim3((unsigned)std::min(n, 65535)), dim3(64), 0, ctx->stream, a, s);
        HIPCHK(hipGetLastError());
        // the lists of the chain start again: everything behind the scalars that live across the passes (items, candidates, the two pass lists)
        HIPCHK(hipMemsetAsync((uint8_t *)ctx->d_st_scalars + 8 * 256, 0, st_scalar_bytes - 8 * 256, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev_ph2[2], ctx->stream));
        { mtr_status ps = run_pass(1); if (ps != MTR_OK) return ps; }
    }
    ctx->st_cand_cap = s.cand_cap;
    SplitArgs sp{};
    sp.item_off = ctx->d_item_off; sp.cand = ctx->d_st_cand; sp.cand_flag = ctx->d_st_flag;
    hipLaunchKernelGGL(mtr_k_replay, dim3((unsigned)std::min(n, 65535)), dim3(64), 0, ctx->stream, a, sp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    DBG("launch_staged: every kernel of the chain enqueued");
    return MTR_OK;
}

// A read gets max_rec = 16 + Lmax/100 record slots; the kernels count the records of a read that found more (about one
// read in 10^5 of the synthetic sets) without storing them.  Such reads are run once more, alone, with exactly the room
// they need (the ranges are recomputed: the unit phase prunes them in place); the compaction then takes their
// records from the second buffer.  The reference has no such limit (std::set).
static mtr_status resolve_overflow(mtr_ctx *ctx)
{
    const int n = ctx->n_reads;
    std::vector<int32_t> cnt((size_t)n);
    HIPCHK(copy_sync(ctx, cnt.data(), ctx->d_reccount, (size_t)n * 4, hipMemcpyDeviceToHost));
    // A read the staged chain sent back (mtr_k_replay met a range the second pass had not searched) carries a mark instead of
    // a count: it most likely fits its max_rec slots, so it is given max_rec + 1 at first rather than the mark's worth of memory
    const int sent_back_mark = 4 * ctx->max_rec + 4096;
    ctx->ovf_reads.clear(); ctx->ovf_cap = 0;
    for (int i = 0; i < n; i++) if (cnt[(size_t)i] > ctx->max_rec) {
        ctx->ovf_reads.push_back(i);
        ctx->ovf_cap = std::max(ctx->ovf_cap, cnt[(size_t)i] == sent_back_mark ? ctx->max_rec + 1 : (int)cnt[(size_t)i]);
    }
    if (ctx->ovf_reads.empty()) return MTR_OK;
    const size_t m = ctx->ovf_reads.size();
    std::vector<int64_t> base((size_t)n, 0);
    HIPCHK(ensure_dev(ctx, ctx->d_rec_base, (size_t)n * 8)); HIPCHK(ensure_dev(ctx, ctx->d_ovf_order, m * 4));
    HIPCHK(hipMemcpyAsync(ctx->d_ovf_order, ctx->ovf_reads.data(), m * 4, hipMemcpyHostToDevice, ctx->stream));
    for (int attempt = 0; ; attempt++) {                // the run again reports true counts: once more only if one of them is larger
        DBG("resolve_overflow: %zu reads found more than %d records or were sent back: running them again with room for %d each", m, ctx->max_rec, ctx->ovf_cap);
        for (size_t k = 0; k < m; k++) base[(size_t)ctx->ovf_reads[k]] = (int64_t)k * ctx->ovf_cap;
        HIPCHK(ensure_dev(ctx, ctx->d_ovf_records, m * (size_t)ctx->ovf_cap * sizeof(DevRecord)));
        HIPCHK(hipMemcpyAsync(ctx->d_rec_base, base.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        ctx->sub_active = true;
        mtr_status s = launch_reads(ctx);               // one wavefront per read; events ev[2..3] are re-recorded, the
        ctx->sub_active = false;                        // kernel time reported stays the first run's (read before)
        if (s != MTR_OK) return s;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        s = check_status(ctx);
        if (s != MTR_OK) return s;
        HIPCHK(copy_sync(ctx, cnt.data(), ctx->d_reccount, (size_t)n * 4, hipMemcpyDeviceToHost));
        int need = 0;
        for (size_t k = 0; k < m; k++) need = std::max(need, (int)cnt[(size_t)ctx->ovf_reads[k]]);
        if (need <= ctx->ovf_cap) return MTR_OK;
        if (attempt >= 2) return MTR_ERR_OVERFLOW;       // counts of the same reads cannot keep growing
        ctx->ovf_cap = need;
    }
}

extern "C" mtr_status mtr_run_resident_async(mtr_ctx *ctx)
{
    if (!ctx) return MTR_ERR_BAD_ARG;
    if (ctx->n_reads <= 0) { ctx->err = "no batch uploaded"; return MTR_ERR_BAD_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    { mtr_status w = mtr_wait(ctx); if (w != MTR_OK && ctx->pending) return w; }
    ctx->run_status = MTR_OK; ctx->ran = false; ctx->first_failed = -1; ctx->ovf_reads.clear();
    HIPCHK(hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_fail_read, 0x7f, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N, ctx->stream));
    ctx->last_staged = use_staged(ctx);
    DBG("launch of %d reads: %s", ctx->n_reads, ctx->last_staged ? "staged chain" : "per-read kernel");
    mtr_status s = ctx->last_staged ? launch_staged(ctx) : launch_reads(ctx); if (s != MTR_OK) return s;
    ctx->pending = true;
    return MTR_OK;
}

extern "C" int32_t mtr_test_last_mode(const mtr_ctx *ctx) { return !ctx ? -1 : ctx->last_staged ? 2 : 0; }

extern "C" mtr_status mtr_wait(mtr_ctx *ctx)
{
    if (!ctx) return MTR_ERR_BAD_ARG;
    if (!ctx->pending) return ctx->run_status;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->pending = false; ctx->ran = false; ctx->run_status = MTR_ERR_HIP;   // until everything below succeeded
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    ctx->kt[0].ms = 0; ctx->kt[0].launches = 0;                // K1 runs inside the per-read kernel
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->kt[1].ms = ms; ctx->kt[1].launches = 1;
    for (int p = 2; p < MTR_N_PHASE_TIMES; p++) { ctx->kt[p].ms = 0; ctx->kt[p].launches = 0; }
    if (ctx->last_staged) {
        hipEvent_t prev = ctx->ev[2];
        for (int p = 2; p < MTR_N_PHASE_TIMES; p++) {
            // (two passes: the first pass ends with its revisions, then mtr_k_finish, the mark pass and the second pass's phases 3..6, whose
            //  durations are added to the first's; the last phase = finish + mark + finish + replay)
            hipEvent_t cur = p == MTR_N_PHASE_TIMES - 1 ? ctx->ev[3] : ctx->ev_ph[p];
            if (ctx->last_two_pass && p == MTR_N_PHASE_TIMES - 1) {
                float m1 = 0, m2 = 0;
                HIPCHK(hipEventElapsedTime(&m1, ctx->ev_ph[6], ctx->ev_ph2[2])); HIPCHK(hipEventElapsedTime(&m2, ctx->ev_ph2[6], ctx->ev[3]));
                ctx->kt[p].ms = m1 + m2; ctx->kt[p].launches = 2;
                break;
            }
            HIPCHK(hipEventElapsedTime(&ms, prev, cur)); ctx->kt[p].ms = ms; ctx->kt[p].launches = 1;
            if (ctx->last_two_pass && p >= 3) {
                float m2 = 0;
                HIPCHK(hipEventElapsedTime(&m2, ctx->ev_ph2[p - 1], ctx->ev_ph2[p])); ctx->kt[p].ms += m2; ctx->kt[p].launches = 2;
            }
            prev = cur;
        }
    }
    for (int d = 0; d < 2; d++) {                            // ids 8, 9: mtr_k_revise_quads and mtr_k_dp2_quads by themselves (both passes)
        ctx->kt[MTR_N_PHASE_TIMES + d].ms = 0; ctx->kt[MTR_N_PHASE_TIMES + d].launches = 0;
        if (!ctx->last_staged || !ctx->last_quads) continue;
        for (int pass = 0; pass < (ctx->last_two_pass ? 2 : 1); pass++) {
            float m = 0;
            HIPCHK(hipEventElapsedTime(&m, ctx->ev_dom[d][pass][0], ctx->ev_dom[d][pass][1]));
            ctx->kt[MTR_N_PHASE_TIMES + d].ms += m; ctx->kt[MTR_N_PHASE_TIMES + d].launches++;
        }
    }
    HIPCHK(copy_sync(ctx, ctx->counters, ctx->d_counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToHost));
    mtr_status st = check_status(ctx);
    if (ctx->last_staged && dbg()) {
        unsigned long long sc[40 * 32]; int32_t cw[10];
        std::vector<unsigned> wk((size_t)ST_N_QUEUES * WQ_WORDS);   // the sub-list counters live in work-queue slots (k3_staged.hip.inc: ST_SL_*)
        auto sub_sum = [&](int slot, bool wide) { unsigned long long t = 0; for (int q = 0; q < WQ_MAX; q++) { const unsigned *p = wk.data() + (size_t)slot * WQ_WORDS + (size_t)q * WQ_STRIDE; t += wide ? *(const unsigned long long *)p : (unsigned long long)*p; } return t; };
        if (copy_sync(ctx, sc, ctx->d_st_scalars, sizeof sc, hipMemcpyDeviceToHost) == hipSuccess && copy_sync(ctx, cw, ctx->d_st_classwave, sizeof cw, hipMemcpyDeviceToHost) == hipSuccess
            && copy_sync(ctx, wk.data(), (const uint8_t *)ctx->d_st_scalars + 64 * 256, wk.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long n_cont = 0;
            for (int slot : { 6, 7, 9, 11, 12, 13, 14, 15 }) n_cont += sub_sum(slot, false);
            unsigned long long n_wv = 0;
            for (int cls = 0; cls < ST_NCLS; cls++) n_wv += sub_sum(ST_SL_WV + cls, false);
            DBG("staged: items %d, ranges with a block %llu (+ %llu per-k work items), DP items %llu (%u work-list entries of one wavefront each), candidate arena %.1f MB, records parked %u; alignments four per wavefront: %d slots (8..1 columns per lane: %d %d %d %d %d %d %d %d)",
                (int)(int32_t)sc[0], sub_sum(3, false), n_cont, sub_sum(1, false), (unsigned)n_wv, (double)sub_sum(2, true) / 1e6, (unsigned)sc[5 * 32], cw[8], cw[1] - cw[0], cw[2] - cw[1], cw[3] - cw[2], cw[4] - cw[3], cw[5] - cw[4], cw[6] - cw[5], cw[7] - cw[6], cw[8] - cw[7]);
        }
    }
    if (st == MTR_ERR_OVERFLOW && ctx->last_staged) {
        // the batch outgrew a buffer of the staged mode: the per-read kernel takes it (same results)
        int32_t dst = 0;
        HIPCHK(copy_sync(ctx, &dst, ctx->d_status, 4, hipMemcpyDeviceToHost));
        if (dst == DEV_ERR_STAGED_OVERFLOW) {
            // the arena is full when one of its sub-arenas is (k3_staged.hip.inc: st_reserve_bytes; 64-bit cursors in work-queue slot 2)
            unsigned long long used = 0, worst = 0;
            {
                std::vector<unsigned> wk((size_t)WQ_WORDS);
                (void)copy_sync(ctx, wk.data(), (const uint8_t *)ctx->d_st_scalars + 64 * 256 + (size_t)2 * WQ_WORDS * 4, wk.size() * 4, hipMemcpyDeviceToHost);
                for (int q = 0; q < ctx->st_last_nsub; q++) { const unsigned long long u = *(const unsigned long long *)(wk.data() + (size_t)q * WQ_STRIDE); used += u; worst = std::max(worst, u); }
            }
            const bool arena_full = ctx->d_st_arena && worst > (unsigned long long)((ctx->st_last_arena_cap / ctx->st_last_nsub) & ~15ll) && ctx->st_arena_per_base < 1024 && !getenv("MTR_TEST_STAGED_CAPS");
            if (arena_full) ctx->st_arena_per_base *= 4;
            DBG("staged chain: a buffer overflowed (candidate arena %.1f of %.1f MB): %s", (double)used / 1e6, (double)ctx->st_last_arena_cap / 1e6,
                arena_full ? "once more with four times the arena per base" : "running the batch with the per-read kernel");
            ctx->last_staged = arena_full;
            HIPCHK(hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->d_fail_read, 0x7f, 4, ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N, ctx->stream));
            { mtr_status ls = arena_full ? launch_staged(ctx) : launch_reads(ctx); if (ls != MTR_OK) { ctx->run_status = ls; return ls; } }
            if (arena_full) {
                HIPCHK(hipStreamSynchronize(ctx->stream));
                int32_t d2 = 0;
                HIPCHK(copy_sync(ctx, &d2, ctx->d_status, 4, hipMemcpyDeviceToHost));
                if (d2 == DEV_ERR_STAGED_OVERFLOW) {
                    DBG("staged chain: overflow again, running the batch with the per-read kernel");
                    ctx->last_staged = false;
                    HIPCHK(hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
                    HIPCHK(hipMemsetAsync(ctx->d_fail_read, 0x7f, 4, ctx->stream));
                    HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N, ctx->stream));
                    { mtr_status ls = launch_reads(ctx); if (ls != MTR_OK) { ctx->run_status = ls; return ls; } }
                }
            }
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->kt[1].ms += ms; ctx->kt[1].launches = 2;
            for (int p = 2; p < MTR_N_KERNEL_TIMES; p++) { ctx->kt[p].ms = 0; ctx->kt[p].launches = 0; }
            HIPCHK(copy_sync(ctx, ctx->counters, ctx->d_counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToHost));
            st = check_status(ctx);
        }
    }
    if (st == MTR_ERR_DP_TOO_LARGE && ctx->last_staged) {
        // The chain aligns the candidates of EVERY range, also of ranges the reference's sequential loop would have removed after an
        // accepted repeat: a matrix beyond WrapDPsize in such a range is not the reference's failure.  The per-read kernel runs
        // the reference's own loop: its verdict (and its first failing read) counts.
        DBG("staged chain: a DP exceeded WrapDPsize; the per-read kernel decides whether the reference's loop reaches it");
        ctx->last_staged = false;
        HIPCHK(hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_fail_read, 0x7f, 4, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N, ctx->stream));
        { mtr_status ls = launch_reads(ctx); if (ls != MTR_OK) { ctx->run_status = ls; return ls; } }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->kt[1].ms += ms; ctx->kt[1].launches = 2;
        for (int p = 2; p < MTR_N_KERNEL_TIMES; p++) { ctx->kt[p].ms = 0; ctx->kt[p].launches = 0; }
        HIPCHK(copy_sync(ctx, ctx->counters, ctx->d_counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToHost));
        st = check_status(ctx);
    }
    if (st == MTR_ERR_DP_TOO_LARGE) {
        int32_t fr = -1;
        if (copy_sync(ctx, &fr, ctx->d_fail_read, 4, hipMemcpyDeviceToHost) == hipSuccess && fr >= 0 && fr < ctx->n_reads) ctx->first_failed = fr;
    }
    if (st == MTR_OK || st == MTR_ERR_DP_TOO_LARGE) {
        // reads that found more records than their slots are run again (also after a DP failure: the reads before the
        // failing one are still reported, as the reference has printed them when it exits)
        const mtr_status so = resolve_overflow(ctx);
        if (so != MTR_OK && so != MTR_ERR_DP_TOO_LARGE) st = so;
    }
    ctx->run_status = st;
    ctx->ran = (st == MTR_OK);
    return st;
}

extern "C" mtr_status mtr_run_resident(mtr_ctx *ctx)
{
    mtr_status s = mtr_run_resident_async(ctx); if (s != MTR_OK) return s;
    return mtr_wait(ctx);
}

// device array of per-read record sources for the compaction, or nullptr when every read's records are in its own slots
static mtr_status record_sources(mtr_ctx *ctx, const DevRecord *const **out)
{
    *out = nullptr;
    if (ctx->ovf_reads.empty()) return MTR_OK;
    const int n = ctx->n_reads;
    std::vector<const DevRecord *> src((size_t)n);
    for (int i = 0; i < n; i++) src[(size_t)i] = ctx->d_records + (size_t)i * (size_t)ctx->max_rec;
    for (size_t k = 0; k < ctx->ovf_reads.size(); k++) src[(size_t)ctx->ovf_reads[k]] = ctx->d_ovf_records + k * (size_t)ctx->ovf_cap;
    HIPCHK(ensure_dev(ctx, ctx->d_src, (size_t)n * sizeof(void *)));
    HIPCHK(copy_sync(ctx, ctx->d_src, src.data(), (size_t)n * sizeof(void *), hipMemcpyHostToDevice));
    *out = ctx->d_src;
    return MTR_OK;
}

__global__ void mtr_k_compact(const DevRecord *in, const DevRecord *const *src_of, const int32_t *cnt, const int64_t *off, int max_rec, int n_reads, DevRecord *out)
{
    // one block per read; records are copied as 16-byte words.  src_of (optional) = where each read's records are
    // (reads that were run again with more slots have theirs in the overflow buffer; every other read has at most max_rec)
    int rd = blockIdx.x;
    if (rd >= n_reads) return;
    int c = cnt[rd];
    if (!src_of && c > max_rec) c = max_rec;
    const uint4 *src = (const uint4 *)(src_of ? src_of[rd] : in + (size_t)rd * (size_t)max_rec);
    uint4 *dst = (uint4 *)(out + off[rd]);
    size_t words = (size_t)c * sizeof(DevRecord) / 16;
    for (size_t t = threadIdx.x; t < words; t += blockDim.x) dst[t] = src[t];
}

// ---- wire form (include/mtr_hip.h): 14 int32 | rep_period unit bytes padded to 4 | rep_period int32 scores ---------
__device__ __forceinline__ int wire_period(const DevRecord *r) { int p = r->f[3]; return p < 0 ? 0 : (p > MTRC_MAX_PERIOD ? MTRC_MAX_PERIOD : p); }
__global__ void mtr_k_wire_sizes(const DevRecord *in, const DevRecord *const *src_of, const int32_t *cnt, int max_rec, int n_reads, int64_t *bytes)
{
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= n_reads) return;
    int c = cnt[rd];
    if (!src_of && c > max_rec) c = max_rec;
    const DevRecord *src = src_of ? src_of[rd] : in + (size_t)rd * (size_t)max_rec;
    int64_t b = 0;
    for (int t = 0; t < c; t++) { const int p = wire_period(src + t); b += 56 + ((p + 3) & ~3) + 4 * p; }
    bytes[rd] = b;
}
__global__ void mtr_k_wire_pack(const DevRecord *in, const DevRecord *const *src_of, const int32_t *cnt, const int64_t *off, int max_rec, int n_reads, uint8_t *out)
{
    // one wavefront per read; every piece of a wire record is a whole number of dwords at a dword-aligned offset
    const int rd = blockIdx.x;
    if (rd >= n_reads) return;
    int c = cnt[rd];
    if (!src_of && c > max_rec) c = max_rec;
    const DevRecord *src = src_of ? src_of[rd] : in + (size_t)rd * (size_t)max_rec;
    uint32_t *dst = (uint32_t *)(out + off[rd]);
    for (int t = 0; t < c; t++) {
        const DevRecord *r = src + t;
        const int p = wire_period(r), uw = (p + 3) >> 2;
        const uint32_t *h = (const uint32_t *)r->f, *u = (const uint32_t *)r->unit, *sc = (const uint32_t *)r->unit_score;
        for (int q = threadIdx.x; q < 14 + uw + p; q += blockDim.x) {
            uint32_t v;
            if (q < 14) v = h[q];
            else if (q < 14 + uw) {
                v = u[q - 14];
                const int keep = p - 4 * (q - 14);                       // bytes of this word that belong to the unit
                if (keep < 4) v &= (1u << (8 * keep)) - 1u;
            } else v = sc[q - 14 - uw];
            dst[q] = v;
        }
        dst += 14 + uw + p;
    }
}

// status of the resident batch for the calls that read its results
static mtr_status results_ready(mtr_ctx *ctx, bool allow_dp_failure)
{
    { mtr_status w = mtr_wait(ctx); if (w != MTR_OK && !(allow_dp_failure && w == MTR_ERR_DP_TOO_LARGE)) return w; }
    if (ctx->n_reads <= 0 || (!ctx->ran && !(allow_dp_failure && ctx->run_status == MTR_ERR_DP_TOO_LARGE))) {
        if (ctx->run_status != MTR_OK) return ctx->run_status;
        ctx->err = "nothing has been run"; return MTR_ERR_BAD_ARG;
    }
    return MTR_OK;
}

extern "C" mtr_status mtr_get_first_failed_read(const mtr_ctx *ctx, int32_t *out)
{
    if (!ctx || !out) return MTR_ERR_BAD_ARG;
    *out = ctx->first_failed;
    return MTR_OK;
}

extern "C" mtr_status mtr_fetch_results(mtr_ctx *ctx, mtr_record **out_records, int32_t **out_counts, int64_t *out_total)
{
    if (!ctx || !out_records || !out_counts) return MTR_ERR_BAD_ARG;
    { mtr_status r = results_ready(ctx, false); if (r != MTR_OK) return r; }
    HIPCHK(hipSetDevice(ctx->device));
    const int n = ctx->n_reads;
    int32_t *counts = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!counts) return MTR_ERR_OOM;
    mtr_record *recs = nullptr;
    auto fail = [&](mtr_status st) { free(counts); free(recs); return st; };
#define FETCH_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return fail(MTR_ERR_HIP); } } while (0)
    FETCH_CHK(copy_sync(ctx, counts, ctx->d_reccount, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) off[(size_t)i + 1] = off[(size_t)i] + counts[i];
    const int64_t total = off[(size_t)n];
    recs = (mtr_record *)malloc(sizeof(mtr_record) * (size_t)std::max<int64_t>(total, 1));
    if (!recs) return fail(MTR_ERR_OOM);
    if (total > 0) {
        int64_t *d_off = ctx->d_recoff;
        FETCH_CHK(ensure_dev(ctx, ctx->d_out, (size_t)total * sizeof(DevRecord)));
        DevRecord *d_out = ctx->d_out;
        FETCH_CHK(hipMemcpyAsync(d_off, off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        const DevRecord *const *srcs = nullptr;
        { mtr_status st = record_sources(ctx, &srcs); if (st != MTR_OK) return fail(st); }
        hipLaunchKernelGGL(mtr_k_compact, dim3((unsigned)n), dim3(64), 0, ctx->stream, ctx->d_records, srcs, ctx->d_reccount, d_off, ctx->max_rec, n, d_out);
        FETCH_CHK(hipGetLastError());
        FETCH_CHK(hipMemcpyAsync(recs, d_out, (size_t)total * sizeof(DevRecord), hipMemcpyDeviceToHost, ctx->stream));
        FETCH_CHK(hipStreamSynchronize(ctx->stream));
    }
#undef FETCH_CHK
    *out_records = recs; *out_counts = counts; if (out_total) *out_total = total;
    return MTR_OK;
}

// sizes + offsets of the wire form of reads [0, n): counts and per-read byte sizes come to pinned host memory
static mtr_status wire_layout(mtr_ctx *ctx, int n, const DevRecord *const **srcs_out, int64_t *total_records, int64_t *total_bytes)
{
    HIPCHK(ensure_pinned(ctx->h_counts, ctx->h_counts_cap, (size_t)n * 4));
    HIPCHK(ensure_pinned(ctx->h_sizes, ctx->h_sizes_cap, ((size_t)n + 1) * 8));
    HIPCHK(ensure_dev(ctx, ctx->d_wire_bytes, (size_t)n * 8)); HIPCHK(ensure_dev(ctx, ctx->d_wire_off, ((size_t)n + 1) * 8));
    const DevRecord *const *srcs = nullptr;
    { mtr_status st = record_sources(ctx, &srcs); if (st != MTR_OK) return st; }
    *srcs_out = srcs;
    hipLaunchKernelGGL(mtr_k_wire_sizes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_records, srcs, ctx->d_reccount, ctx->max_rec, n, ctx->d_wire_bytes);
    HIPCHK(hipGetLastError());
    int32_t *counts = (int32_t *)ctx->h_counts; int64_t *sizes = (int64_t *)ctx->h_sizes;
    HIPCHK(hipMemcpyAsync(counts, ctx->d_reccount, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(sizes, ctx->d_wire_bytes, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int64_t recs = 0, off = 0;
    for (int i = 0; i < n; i++) { recs += counts[i]; const int64_t b = sizes[i]; sizes[i] = off; off += b; }   // in place: sizes -> offsets
    sizes[n] = off;
    *total_records = recs; *total_bytes = off;
    HIPCHK(hipMemcpyAsync(ctx->d_wire_off, sizes, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    return MTR_OK;
}

extern "C" mtr_status mtr_fetch_results_packed(mtr_ctx *ctx, int32_t n_reads_limit, const uint8_t **out_blob, int64_t *out_bytes,
                                               const int32_t **out_counts, int64_t *out_total_records)
{
    if (!ctx || !out_blob || !out_bytes || !out_counts) return MTR_ERR_BAD_ARG;
    { mtr_status r = results_ready(ctx, n_reads_limit >= 0); if (r != MTR_OK) return r; }
    HIPCHK(hipSetDevice(ctx->device));
    int n = ctx->n_reads;
    if (n_reads_limit >= 0 && n_reads_limit < n) n = n_reads_limit;
    if (ctx->run_status == MTR_ERR_DP_TOO_LARGE && (ctx->first_failed < 0 || n > ctx->first_failed)) { ctx->err = "only the reads before the first failed one can be fetched"; return MTR_ERR_DP_TOO_LARGE; }
    *out_blob = nullptr; *out_bytes = 0; *out_counts = nullptr; if (out_total_records) *out_total_records = 0;
    if (n == 0) return MTR_OK;
    const DevRecord *const *srcs = nullptr; int64_t recs = 0, bytes = 0;
    { mtr_status st = wire_layout(ctx, n, &srcs, &recs, &bytes); if (st != MTR_OK) return st; }
    HIPCHK(ensure_pinned(ctx->h_blob, ctx->h_blob_cap, (size_t)std::max<int64_t>(bytes, 4)));
    if (bytes > 0) {
        HIPCHK(ensure_dev(ctx, ctx->d_wire, (size_t)bytes));
        hipLaunchKernelGGL(mtr_k_wire_pack, dim3((unsigned)n), dim3(64), 0, ctx->stream, ctx->d_records, srcs, ctx->d_reccount, ctx->d_wire_off, ctx->max_rec, n, ctx->d_wire);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->h_blob, ctx->d_wire, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out_blob = (const uint8_t *)ctx->h_blob; *out_bytes = bytes; *out_counts = (const int32_t *)ctx->h_counts;
    if (out_total_records) *out_total_records = recs;
    return MTR_OK;
}

extern "C" mtr_status mtr_export_packed_device(mtr_ctx *ctx, void *d_dst, int64_t capacity_bytes, int32_t *counts_host,
                                               int64_t *out_total_records, int64_t *out_bytes)
{
    if (!ctx || !counts_host || !out_total_records || !out_bytes) return MTR_ERR_BAD_ARG;
    { mtr_status r = results_ready(ctx, false); if (r != MTR_OK) return r; }
    HIPCHK(hipSetDevice(ctx->device));
    const int n = ctx->n_reads;
    const DevRecord *const *srcs = nullptr; int64_t recs = 0, bytes = 0;
    { mtr_status st = wire_layout(ctx, n, &srcs, &recs, &bytes); if (st != MTR_OK) return st; }
    memcpy(counts_host, ctx->h_counts, (size_t)n * 4);
    *out_total_records = recs; *out_bytes = bytes;
    if (bytes == 0) { HIPCHK(hipStreamSynchronize(ctx->stream)); return MTR_OK; }
    if (!d_dst || bytes > capacity_bytes) { (void)hipStreamSynchronize(ctx->stream); ctx->err = "destination holds " + std::to_string(capacity_bytes) + " bytes, " + std::to_string(bytes) + " needed"; return MTR_ERR_OVERFLOW; }
    hipLaunchKernelGGL(mtr_k_wire_pack, dim3((unsigned)n), dim3(64), 0, ctx->stream, ctx->d_records, srcs, ctx->d_reccount, ctx->d_wire_off, ctx->max_rec, n, (uint8_t *)d_dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MTR_OK;
}

// host-only conversions between the wire form and mtr_record
extern "C" mtr_status mtr_unpack_records(const uint8_t *blob, int64_t bytes, int64_t n_records, mtr_record *out)
{
    if ((!blob && bytes > 0) || (!out && n_records > 0) || bytes < 0 || n_records < 0) return MTR_ERR_BAD_ARG;
    int64_t o = 0;
    for (int64_t i = 0; i < n_records; i++) {
        if (o + MTR_WIRE_HEADER_BYTES > bytes) return MTR_ERR_BAD_ARG;
        mtr_record *r = out + i;
        memcpy(r, blob + o, MTR_WIRE_HEADER_BYTES);
        const int p = r->rep_period;
        if (p < 0 || p > MTR_MAX_PERIOD) return MTR_ERR_BAD_ARG;
        const int64_t need = mtr_wire_record_bytes(p);
        if (o + need > bytes) return MTR_ERR_BAD_ARG;
        memcpy(r->unit, blob + o + MTR_WIRE_HEADER_BYTES, (size_t)p);
        r->unit[p] = 0;
        memcpy(r->unit_score, blob + o + MTR_WIRE_HEADER_BYTES + ((p + 3) & ~3), (size_t)p * 4);
        o += need;
    }
    return o == bytes ? MTR_OK : MTR_ERR_BAD_ARG;
}

extern "C" int64_t mtr_pack_records(const mtr_record *records, int64_t n_records, uint8_t *out, int64_t capacity)
{
    if ((!records && n_records > 0) || n_records < 0 || (!out && capacity > 0)) return -1;
    int64_t o = 0;
    for (int64_t i = 0; i < n_records; i++) {
        const mtr_record *r = records + i;
        const int p = r->rep_period < 0 ? 0 : (r->rep_period > MTR_MAX_PERIOD ? MTR_MAX_PERIOD : r->rep_period);
        const int64_t need = mtr_wire_record_bytes(p);
        if (o + need > capacity) return -1;
        memcpy(out + o, r, MTR_WIRE_HEADER_BYTES);
        const int up = (p + 3) & ~3;
        memcpy(out + o + MTR_WIRE_HEADER_BYTES, r->unit, (size_t)p);
        memset(out + o + MTR_WIRE_HEADER_BYTES + p, 0, (size_t)(up - p));
        memcpy(out + o + MTR_WIRE_HEADER_BYTES + up, r->unit_score, (size_t)p * 4);
        o += need;
    }
    return o;
}

extern "C" mtr_status mtr_process_batch(mtr_ctx *ctx, const uint8_t *bases, const int64_t *offsets, const int32_t *lens,
                                        int32_t n_reads, mtr_record **out_records, int32_t **out_counts, int64_t *out_total)
{
    mtr_status s = mtr_upload_batch(ctx, bases, offsets, lens, n_reads); if (s != MTR_OK) return s;
    s = mtr_run_resident(ctx); if (s != MTR_OK) return s;
    return mtr_fetch_results(ctx, out_records, out_counts, out_total);
}

extern "C" void mtr_free_results(mtr_record *records, int32_t *counts) { free(records); free(counts); }

extern "C" mtr_status mtr_get_bases_after_read(const mtr_ctx *ctx, int32_t read_idx, uint8_t out[2])
{
    if (!ctx || !out || read_idx < 0 || read_idx >= ctx->n_reads || ctx->after.size() < (size_t)ctx->n_reads * 2) return MTR_ERR_BAD_ARG;
    out[0] = ctx->after[(size_t)read_idx * 2]; out[1] = ctx->after[(size_t)read_idx * 2 + 1];
    return MTR_OK;
}

extern "C" mtr_status mtr_get_kernel_times(const mtr_ctx *ctx, mtr_kernel_time *out, int32_t n)
{
    if (!ctx || !out) return MTR_ERR_BAD_ARG;
    for (int i = 0; i < n && i < MTR_N_KERNEL_TIMES; i++) out[i] = ctx->kt[i];
    return MTR_OK;
}

extern "C" mtr_status mtr_get_counters(const mtr_ctx *ctx, int64_t *out, int32_t n)
{
    if (!ctx || !out) return MTR_ERR_BAD_ARG;
    for (int i = 0; i < n && i < CNT_N; i++) out[i] = (int64_t)ctx->counters[i];
    return MTR_OK;
}

extern "C" mtr_status mtr_test_ranges(mtr_ctx *ctx, int32_t **out_counts, int32_t **out_start, int32_t **out_end,
                                      int32_t **out_w, uint64_t **out_di_bits, int64_t *out_total)
{
    if (!ctx || !out_counts || !out_start || !out_end || !out_w || !out_di_bits) return MTR_ERR_BAD_ARG;
    if (ctx->n_reads <= 0) { ctx->err = "no batch uploaded"; return MTR_ERR_BAD_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N, ctx->stream));
    mtr_status s = launch_k1(ctx); if (s != MTR_OK) return s;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); ctx->kt[0].ms = ms; ctx->kt[0].launches = 1;
    s = check_status(ctx); if (s != MTR_OK) return s;
    const int n = ctx->n_reads;
    std::vector<int32_t> cnt((size_t)n), st((size_t)ctx->total_rcap), en((size_t)ctx->total_rcap), ww((size_t)ctx->total_rcap);
    std::vector<uint64_t> di((size_t)ctx->total_rcap);
    HIPCHK(copy_sync(ctx, cnt.data(), ctx->d_rcount, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(copy_sync(ctx, st.data(), ctx->d_rstart, st.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(copy_sync(ctx, en.data(), ctx->d_rend, en.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(copy_sync(ctx, ww.data(), ctx->d_rw, ww.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(copy_sync(ctx, di.data(), ctx->d_rdi, di.size() * 8, hipMemcpyDeviceToHost));
    int64_t total = 0; for (int i = 0; i < n; i++) total += cnt[(size_t)i];
    int32_t *oc = (int32_t *)malloc((size_t)n * 4), *os = (int32_t *)malloc((size_t)std::max<int64_t>(total, 1) * 4);
    int32_t *oe = (int32_t *)malloc((size_t)std::max<int64_t>(total, 1) * 4), *ow = (int32_t *)malloc((size_t)std::max<int64_t>(total, 1) * 4);
    uint64_t *od = (uint64_t *)malloc((size_t)std::max<int64_t>(total, 1) * 8);
    if (!oc || !os || !oe || !ow || !od) return MTR_ERR_OOM;
    int64_t p = 0;
    for (int i = 0; i < n; i++) {
        oc[i] = cnt[(size_t)i];
        for (int t = 0; t < cnt[(size_t)i]; t++, p++) {
            size_t q = (size_t)ctx->roff[(size_t)i] + (size_t)t;
            os[p] = st[q]; oe[p] = en[q]; ow[p] = ww[q]; od[p] = di[q];
        }
    }
    *out_counts = oc; *out_start = os; *out_end = oe; *out_w = ow; *out_di_bits = od; if (out_total) *out_total = total;
    return MTR_OK;
}

extern "C" mtr_status mtr_test_wrap_dp(mtr_ctx *ctx, int32_t n_tasks, const int32_t *read_idx, const int32_t *query_start,
                                       const int32_t *query_end, const uint8_t *units, const int32_t *unit_off,
                                       const int32_t *gain, const int32_t *mismatch, const int32_t *indel, int32_t *out8)
{
    if (!ctx || n_tasks <= 0 || !read_idx || !query_start || !query_end || !units || !unit_off || !gain || !mismatch || !indel || !out8) return MTR_ERR_BAD_ARG;
    if (ctx->n_reads <= 0) { ctx->err = "no batch uploaded"; return MTR_ERR_BAD_ARG; }
    HIPCHK(hipSetDevice(ctx->device));
    size_t cells = 1;
    for (int t = 0; t < n_tasks; t++) {
        int rd = read_idx[t];
        int U = unit_off[t + 1] - unit_off[t];
        if (rd < 0 || rd >= ctx->n_reads || query_start[t] < 0 || query_end[t] < query_start[t] || query_end[t] >= ctx->lens[(size_t)rd] || U <= 0 || U >= MTRC_MAX_PERIOD) { ctx->err = "bad DP task " + std::to_string(t); return MTR_ERR_BAD_ARG; }
        cells = std::max(cells, (size_t)(query_end[t] - query_start[t] + 1) * (size_t)(U + 1));   // rows may be padded to an even length
    }
    size_t per_wave = mtrc_align(cells + 256, 256), total = 0;   // + slack: the traceback's dword loads read a few bytes past a row
    int waves = pick_waves(ctx, n_tasks, 8, per_wave, &total);
    DBG("test_wrap_dp: %d tasks, cells %zu, waves %d, scratch %zu", n_tasks, cells, waves, total);
    mtr_status s = ensure_scratch(ctx, total); if (s != MTR_OK) return s;
    const size_t nt = (size_t)n_tasks;
    HIPCHK(ensure_dev(ctx, ctx->d_t_i32, (nt * 7 + 1) * 4)); HIPCHK(ensure_dev(ctx, ctx->d_t_units, (size_t)unit_off[n_tasks] + 16)); HIPCHK(ensure_dev(ctx, ctx->d_t_out, nt * 8 * 4));
    int32_t *d_i32 = ctx->d_t_i32; uint8_t *d_units = ctx->d_t_units; int32_t *d_out = ctx->d_t_out;
    int32_t *d_rd = d_i32, *d_qs = d_i32 + nt, *d_qe = d_i32 + 2 * nt, *d_g = d_i32 + 3 * nt, *d_m = d_i32 + 4 * nt, *d_d = d_i32 + 5 * nt, *d_uo = d_i32 + 6 * nt;
    HIPCHK(copy_sync(ctx, d_rd, read_idx, nt * 4, hipMemcpyHostToDevice)); HIPCHK(copy_sync(ctx, d_qs, query_start, nt * 4, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(ctx, d_qe, query_end, nt * 4, hipMemcpyHostToDevice)); HIPCHK(copy_sync(ctx, d_g, gain, nt * 4, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(ctx, d_m, mismatch, nt * 4, hipMemcpyHostToDevice)); HIPCHK(copy_sync(ctx, d_d, indel, nt * 4, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(ctx, d_uo, unit_off, (nt + 1) * 4, hipMemcpyHostToDevice)); HIPCHK(copy_sync(ctx, d_units, units, (size_t)unit_off[n_tasks], hipMemcpyHostToDevice));
    DpTestArgs a{};
    a.b = view(ctx); a.n_tasks = n_tasks; a.read_idx = d_rd; a.qs = d_qs; a.qe = d_qe; a.units = d_units; a.unit_off = d_uo;
    a.gain = d_g; a.mism = d_m; a.indel = d_d; a.out8 = d_out; a.scratch = ctx->d_scratch; a.scratch_per_wave = per_wave; a.cells_cap = cells;
    a.status = ctx->d_status; a.work_counter = ctx->d_work; a.counters = ctx->d_counters; a.dp16_max_rows = dp16_max_rows();
    HIPCHK(hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long) * CNT_N, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    DBG("test_wrap_dp: launching");
    hipLaunchKernelGGL(mtr_k_dp_test, dim3((unsigned)waves), dim3(64), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    DBG("test_wrap_dp: kernel done");
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->kt[1].ms = ms; ctx->kt[1].launches = 1;
    HIPCHK(copy_sync(ctx, ctx->counters, ctx->d_counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToHost));
    HIPCHK(copy_sync(ctx, out8, d_out, nt * 8 * 4, hipMemcpyDeviceToHost));
    return check_status(ctx);
}

extern "C" mtr_status mtr_alignments(mtr_ctx *ctx, int32_t n, const int32_t *read_idx, const mtr_record *records,
                                     uint8_t **out_ops, int64_t **out_off, int32_t **out_end)
{
    if (!ctx || n < 0 || (n > 0 && (!read_idx || !records)) || !out_ops || !out_off || !out_end) return MTR_ERR_BAD_ARG;
    if (ctx->n_reads <= 0) { ctx->err = "no batch uploaded"; return MTR_ERR_BAD_ARG; }
    { mtr_status w = mtr_wait(ctx); if (w != MTR_OK && w != MTR_ERR_DP_TOO_LARGE) return w; }   // after a DP failure: the reads before it are still printed
    HIPCHK(hipSetDevice(ctx->device));
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    if (!off) return MTR_ERR_OOM;
    off[0] = 0;
    if (n == 0) { *out_ops = (uint8_t *)malloc(1); *out_off = off; *out_end = (int32_t *)malloc(8); return MTR_OK; }
    const size_t nt = (size_t)n;
    std::vector<int32_t> h((nt * 6) + nt + 1);
    int32_t *h_rd = h.data(), *h_rs = h_rd + nt, *h_re = h_rs + nt, *h_g = h_re + nt, *h_m = h_g + nt, *h_d = h_m + nt, *h_uo = h_d + nt;
    std::vector<uint8_t> units; std::vector<int64_t> cap_off(nt + 1, 0);
    size_t cells = 1;
    for (int t = 0; t < n; t++) {
        const mtr_record &r = records[t];
        const int rd = read_idx[t], U = r.rep_period, rows = r.rep_end - r.rep_start + 1;
        if (rd < 0 || rd >= ctx->n_reads || U <= 0 || U >= MTRC_MAX_PERIOD || rows <= 0 || r.rep_start < 0 || r.rep_end > ctx->lens[(size_t)rd]) {
            free(off); ctx->err = "bad alignment task " + std::to_string(t); return MTR_ERR_BAD_ARG;
        }
        h_rd[t] = rd; h_rs[t] = r.rep_start; h_re[t] = r.rep_end; h_g[t] = r.match_gain; h_m[t] = r.mismatch_penalty; h_d[t] = r.indel_penalty;
        h_uo[t] = (int32_t)units.size();
        for (int j = 0; j < U; j++) { const char ch = r.unit[j]; units.push_back(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3); }
        // columns = rows + deletions; every deletion costs indel_penalty out of a score of at most match_gain per row
        const size_t max_del = (size_t)std::max(r.match_gain, 1) * (size_t)rows / (size_t)std::max(r.indel_penalty, 1);
        cap_off[(size_t)t + 1] = cap_off[(size_t)t] + (int64_t)(((size_t)rows + max_del + (size_t)U + 64 + 3) & ~(size_t)3);
        cells = std::max(cells, (size_t)rows * (size_t)(U + 1));
    }
    h_uo[n] = (int32_t)units.size();
    const size_t per_wave = mtrc_align(cells + 256, 256);
    size_t total = 0;
    const int waves = pick_waves(ctx, n, 8, per_wave, &total);
    mtr_status s = ensure_scratch(ctx, total); if (s != MTR_OK) { free(off); return s; }
    int32_t *ends = nullptr;
    const size_t ops_bytes = (size_t)cap_off[nt];
    auto fail = [&](mtr_status st) { free(off); free(ends); return st; };
#define ALN_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return fail(MTR_ERR_HIP); } } while (0)
    ALN_CHK(ensure_dev(ctx, ctx->d_al_i32, h.size() * 4)); ALN_CHK(ensure_dev(ctx, ctx->d_al_len, nt * 4)); ALN_CHK(ensure_dev(ctx, ctx->d_al_units, units.size() + 16));
    ALN_CHK(ensure_dev(ctx, ctx->d_al_ops, ops_bytes + 16)); ALN_CHK(ensure_dev(ctx, ctx->d_al_off, (nt + 1) * 8)); ALN_CHK(ensure_dev(ctx, ctx->d_al_ends, nt * 8));
    int32_t *d_i32 = ctx->d_al_i32, *d_len = ctx->d_al_len, *d_ends = ctx->d_al_ends; uint8_t *d_units = ctx->d_al_units, *d_ops = ctx->d_al_ops; int64_t *d_off = ctx->d_al_off;
    ALN_CHK(hipMemcpyAsync(d_i32, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    ALN_CHK(hipMemcpyAsync(d_units, units.data(), units.size(), hipMemcpyHostToDevice, ctx->stream));
    ALN_CHK(hipMemcpyAsync(d_off, cap_off.data(), (nt + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    AlignArgs a{};
    a.b = view(ctx); a.n_tasks = n;
    a.read_idx = d_i32; a.rep_start = d_i32 + nt; a.rep_end = d_i32 + 2 * nt; a.gain = d_i32 + 3 * nt; a.mism = d_i32 + 4 * nt; a.indel = d_i32 + 5 * nt;
    a.unit_off = d_i32 + 6 * nt; a.units = d_units;
    a.ops = d_ops; a.ops_off = d_off; a.ops_len = d_len; a.ends = d_ends;
    a.scratch = ctx->d_scratch; a.scratch_per_wave = per_wave; a.cells_cap = cells;
    a.status = ctx->d_status; a.work_counter = ctx->d_work; a.counters = ctx->d_counters; a.dp16_max_rows = dp16_max_rows();
    ALN_CHK(hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
    ALN_CHK(hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(mtr_k_align, dim3((unsigned)waves), dim3(64), 0, ctx->stream, a);
    ALN_CHK(hipGetLastError());
    std::vector<int32_t> len(nt);
    std::vector<uint8_t> raw(ops_bytes);
    ends = (int32_t *)malloc(nt * 8);
    if (!ends) return fail(MTR_ERR_OOM);
    ALN_CHK(hipMemcpyAsync(ends, d_ends, nt * 8, hipMemcpyDeviceToHost, ctx->stream));
    ALN_CHK(hipMemcpyAsync(len.data(), d_len, nt * 4, hipMemcpyDeviceToHost, ctx->stream));
    ALN_CHK(hipMemcpyAsync(raw.data(), d_ops, ops_bytes, hipMemcpyDeviceToHost, ctx->stream));
    ALN_CHK(hipStreamSynchronize(ctx->stream));
#undef ALN_CHK
    { mtr_status st = check_status(ctx); if (st != MTR_OK && !(st == MTR_ERR_DP_TOO_LARGE && ctx->run_status == MTR_ERR_DP_TOO_LARGE)) return fail(st); }
    for (int t = 0; t < n; t++) off[(size_t)t + 1] = off[(size_t)t] + len[(size_t)t];
    uint8_t *ops = (uint8_t *)malloc((size_t)std::max<int64_t>(off[nt], 1));
    if (!ops) return fail(MTR_ERR_OOM);
    for (int t = 0; t < n; t++) memcpy(ops + off[(size_t)t], raw.data() + cap_off[(size_t)t], (size_t)len[(size_t)t]);
    for (int t = 0; t < n; t++) ends[2 * t] = records[t].rep_start - 1 + ends[2 * t];       // window row -> read position
    *out_ops = ops; *out_off = off; *out_end = ends;
    return MTR_OK;
}

extern "C" mtr_status mtr_set_trace(mtr_ctx *ctx, int32_t max_events)
{
    if (!ctx || max_events < 0) return MTR_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    dfree(ctx->d_trace); ctx->trace_cap = 0;
    if (max_events > 0) { HIPCHK(hipMalloc(&ctx->d_trace, (size_t)max_events * 16 * 4)); ctx->trace_cap = max_events; }
    return MTR_OK;
}

extern "C" mtr_status mtr_get_trace(mtr_ctx *ctx, int32_t **out_events, int64_t *out_n)
{
    if (!ctx || !out_events || !out_n) return MTR_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    unsigned n = 0;
    HIPCHK(copy_sync(ctx, &n, ctx->d_trace_n, 4, hipMemcpyDeviceToHost));
    if ((int64_t)n > ctx->trace_cap) n = (unsigned)ctx->trace_cap;
    int32_t *ev = (int32_t *)malloc((size_t)std::max(1u, n) * 16 * 4);
    if (!ev) return MTR_ERR_OOM;
    if (n > 0) HIPCHK(copy_sync(ctx, ev, ctx->d_trace, (size_t)n * 16 * 4, hipMemcpyDeviceToHost));
    *out_events = ev; *out_n = n;
    return MTR_OK;
}

// ---- several GPUs in one process: the RCCL gather of the wire-form tables (include/mtr_hip.h, ABI 5) --------------------------------
#include "gather.hip.inc"
