// toyfhe_hip.hip -- context management and the C ABI of libtoyfhe_hip.so (see include/toyfhe_hip.h).
// gfx950 only.  No torch types, no oracle code, no CPU fallback: every compute entry point launches
// HIP kernels and fails with TFHE_E_HIP if the device is unavailable.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/toyfhe_hip.h"
#include "bfv_tables.h"
#include "host_math.h"
#include "kernels.h"
#include "ckks_kernels.h"
#include "sample_kernels.h"
#include "ntt_tables.h"
#include "dev_alloc.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(TFHE_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

struct prof_pair {
    hipEvent_t a, b;
    int64_t limb_polys;
};

}  // namespace

struct tfhe_ctx {
    int64_t N = 0;
    int logN = 0, L = 0;
    std::vector<u64> q, psi;
    std::vector<ntt_limb_t> limbs_host;
    ntt_limb_t* limbs_dev = nullptr;
    std::vector<void*> tabs;   // device twiddle tables (W, Winv and their fp64 twins per limb)
    int num_cus = 256;
    hipStream_t stream = nullptr;        // the stream launches go to: main_stream, or side_stream inside a forked region (lanes_t)
    hipStream_t main_stream = nullptr;   // the context's stream as its users know it (the slot the allocator records release events on)
    bool own_stream = false;
    // second lane (r06): rings that mix the two arithmetic policies (60-bit q0 / special prime beside 40-bit primes) run the
    // launches of one policy on side_stream beside those of the other, forked from and joined back into main_stream by events
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int lane_depth = 0;                  // > 0: inside a forked region (nested lanes_t objects only switch streams)
    bool lanes_broken = false;           // creating the second stream failed once: stay serial
    int variant = 0;
    // workspace (grown on demand, reused)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    bool ws_pooled = false;   // the block came from the recycling allocator (large per-call workspaces, ws_borrow_t)
    // profiling
    bool prof = false;
    std::vector<prof_pair> prof_pairs;
    // CKKS encode / decode tables (built on first use): FFT roots, twist, scatter / gather positions
    void *ckks_roots = nullptr, *ckks_tw = nullptr, *ckks_pos = nullptr, *ckks_gpos = nullptr;
    // exact-reconstruction tables of the window key switch, per level (built on first use)
    std::map<int, conv_tab_t*> ksw_tabs;
    std::vector<void*> ksw_allocs;
    std::mutex ksw_mu;
    // N zero words (the "+ c" stream of the fused contraction for the component that has no addend), made on first use
    // (under zero_mu; a failed attempt is retried by the next call)
    u64* zero_row = nullptr;
    std::mutex zero_mu;
};

namespace {

// pooled: the block is taken from (and, by ws_borrow_t, returned to) the recycling allocator instead of being owned by the
// context for its lifetime -- for the large per-call workspaces (tfhe_matmul_diag: tens of GiB).  A parked block is reused by
// the next call of the same shape without a synchronisation, counts against the allocator's cache bound, and is given back to
// the driver by tfhe_alloc_trim and by every out-of-memory retry of the library (devalloc::malloc_retry).
int ensure_ws(tfhe_ctx* c, size_t bytes, void** out, bool pooled = false) {
    if (bytes > c->ws_bytes) {
        if (c->ws) {
            // the slot is cleared BEFORE anything can return (r05, ADVICE r04: devalloc::release erases the block from its live
            // table even when it reports an error -- a context left pointing at it would hand a stale pointer to the next free)
            void* old = c->ws;
            const bool old_pooled = c->ws_pooled;
            c->ws = nullptr;
            c->ws_bytes = 0;
            c->ws_pooled = false;
            if (old_pooled) {
                HIP_TRY(devalloc::release(old));                 // parked behind events on every context stream: no wait
            } else {
                HIP_TRY(hipStreamSynchronize(c->stream));
                HIP_TRY(hipFree(old));
            }
        }
        size_t got = bytes;
        hipError_t e = pooled ? devalloc::alloc_ws(bytes, c->stream, &c->ws, &got) : devalloc::malloc_retry(&c->ws, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); c->ws = nullptr; return fail(TFHE_E_NOMEM, "workspace hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        c->ws_bytes = got;
        c->ws_pooled = pooled;
    }
    *out = c->ws;
    return TFHE_OK;
}
// A pooled workspace lives for ONE call: the context's own (long-lived, small) workspace is set aside while the call runs --
// the transforms inside it take their scratch from c->ws -- and comes back untouched when the call returns; the pooled block
// goes back to the allocator.  (First form, r04: the pooled block REPLACED the context's workspace; every large call then cost
// the next small operation a hipMalloc and itself a stream synchronisation + hipFree of that small block: the encrypted-MNIST
// pass went from 78 to 145 ms.)
struct ws_borrow_t {
    tfhe_ctx* c;
    void* ws = nullptr;
    size_t bytes = 0;
    bool pooled = false, active = false;
    explicit ws_borrow_t(tfhe_ctx* ctx) : c(ctx) {}
    void begin() {
        ws = c->ws; bytes = c->ws_bytes; pooled = c->ws_pooled; active = true;
        c->ws = nullptr; c->ws_bytes = 0; c->ws_pooled = false;
    }
    ~ws_borrow_t() {
        if (!active) return;
        if (c->ws) { if (c->ws_pooled) (void)devalloc::release(c->ws); else { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->ws); } }
        c->ws = ws; c->ws_bytes = bytes; c->ws_pooled = pooled;
    }
};
// workspaces above this size are per-call (pooled); TFHE_WS_KEEP_GIB overrides
size_t ws_keep_bytes() {
    static const size_t keep = [] { const char* e = getenv("TFHE_WS_KEEP_GIB"); const long g = e ? atol(e) : 0; return (size_t)(g > 0 ? g : 4) << 30; }();
    return keep;
}

// ---- two lanes ------------------------------------------------------------------------------------------------------------
// A ring that mixes fp64-size moduli with larger ones runs every transform / key-switch step as two sets of launches, one per
// arithmetic policy, over DISJOINT limb rows.  Serialised on one stream each set drains before the other starts, and on small
// batches (one ciphertext = a few dozen workgroup items) each leaves most of the 256 CUs idle.  lanes_t forks the context's
// stream: lane 0 stays on main_stream, lane 1 goes to side_stream, which starts after everything main_stream held at the fork;
// the outermost lanes_t joins the side lane back in its destructor (main waits for an event recorded on side), so the caller
// -- and the allocator, which records release events on main_stream only -- see ONE stream as before.  Nested lanes_t objects
// (a step inside a forked region) only switch lanes: data that stays within one policy's limbs flows in stream order on its
// own lane across steps without a join.  Whatever both lanes need (workspace growth: ensure_ws may synchronise or free) must be
// done BEFORE the fork.  TFHE_LANES=0 keeps everything on one stream (comparisons).
struct lanes_t {
    tfhe_ctx* c;
    bool owner = false, on = false;
    hipStream_t entry = nullptr;
    explicit lanes_t(tfhe_ctx* ctx, bool want = true) : c(ctx) {
        static const bool enabled = !(getenv("TFHE_LANES") && getenv("TFHE_LANES")[0] == '0');
        entry = c->stream;
        if (!want || !enabled || c->lanes_broken) return;
        if (c->lane_depth > 0) { on = true; c->lane_depth++; return; }
        if (!c->side_stream) {
            bool ok = hipStreamCreate(&c->side_stream) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); c->lanes_broken = true; return; }
        }
        if (hipEventRecord(c->ev_fork, c->main_stream) != hipSuccess || hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            c->lanes_broken = true;
            return;
        }
        owner = on = true;
        c->lane_depth = 1;
    }
    // launches that follow go to lane `l` (0 = main, 1 = side); without a fork both are the entry stream
    void use(int l) { if (on) c->stream = l ? c->side_stream : c->main_stream; }
    int join() {   // idempotent; the destructor calls it
        if (!on) return TFHE_OK;
        on = false;
        c->lane_depth--;
        c->stream = entry;
        if (!owner) return TFHE_OK;
        c->stream = c->main_stream;
        hipError_t e = hipEventRecord(c->ev_join, c->side_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->main_stream, c->ev_join, 0);
        if (e != hipSuccess) {   // never leave the side lane unordered: wait for it on the host
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->side_stream);
            c->lanes_broken = true;
        }
        return TFHE_OK;
    }
    ~lanes_t() { (void)join(); }
    lanes_t(const lanes_t&) = delete;
    lanes_t& operator=(const lanes_t&) = delete;
};

int make_sel(const tfhe_ctx* c, int limbs, const int32_t* idx, limb_sel_t* sel) {
    if (limbs < 1 || limbs > TFHE_MAX_LIMBS) return fail(TFHE_E_BADARG, "limbs=%d out of range [1,%d]", limbs, TFHE_MAX_LIMBS);
    sel->n = limbs;
    for (int j = 0; j < limbs; j++) {
        const int v = idx ? idx[j] : j;
        if (v < 0 || v >= c->L) return fail(TFHE_E_LEVEL_MISMATCH, "limb_idx[%d]=%d outside the ring's %d moduli", j, v, c->L);
        sel->idx[j] = v;
    }
    return TFHE_OK;
}


template <typename K>
int set_lds(K kern, size_t bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return TFHE_OK;
}

void prof_begin(tfhe_ctx* c, int64_t limb_polys) {
    if (!c->prof) return;
    prof_pair p;
    hipEventCreate(&p.a);
    hipEventCreate(&p.b);
    p.limb_polys = limb_polys;
    hipEventRecord(p.a, c->stream);
    c->prof_pairs.push_back(p);
}
void prof_end(tfhe_ctx* c) {
    if (!c->prof) return;
    hipEventRecord(c->prof_pairs.back().b, c->stream);
}

bool sel_fp(const tfhe_ctx* c, const limb_sel_t& sel, int x) {
    (void)x;  // sub-blocks of N > 2^14 transforms use the same fp64 block kernels (top stages stay u64)
    if (c->variant == 2) return false;
    for (int j = 0; j < sel.n; j++)
        if (!c->limbs_host[sel.idx[j]].Wd) return false;
    return true;
}

ntt_io_t io_plain() { ntt_io_t io; memset(&io, 0, sizeof io); return io; }
// a ring of mixed modulus sizes is split into one pass per arithmetic policy only when there is enough work to pay for the
// second set of launches (a single ciphertext at N = 2^16 is 0.9 M words and stays on the u64 kernels)
static long long mixed_min_words() {   // TFHE_MIXED_MIN_LOG2 overrides (measurements)
    static const long long v = [] { const char* e = getenv("TFHE_MIXED_MIN_LOG2"); const int l = e ? atoi(e) : -1; return 1ll << ((l >= 0 && l < 40) ? l : 20); }();
    return v;
}
#define TFHE_MIXED_MIN_WORDS mixed_min_words()

template <class A, int LOGB, int IOMODE = 0>
int launch_block_fwd(tfhe_ctx* c, const u64* src, u64* dst, int64_t rows, const limb_sel_t& sel, int x, const ntt_io_t& io) {
    if constexpr (IOMODE == 0) {
        if (io.mode == 1) return launch_block_fwd<A, LOGB, 1>(c, src, dst, rows, sel, x, io);
    }
    constexpr int LOGT = logt_for(LOGB);
    const size_t lds = (size_t)lds_words<LOGB, LOGT>() * 8;
    if constexpr (IOMODE == 1) {
        // digit lift, whole-transform blocks: read each source row once and produce all nw lifted transforms from it
        if (x == 0 && c->variant != 3 && io.nw >= 2 && rows % io.nw == 0) {
            auto lkern = k_ntt_fwd_lift<A, LOGB, LOGT>;
            static bool lattr_set = false;
            if (!lattr_set) { int rc = set_lds(lkern, lds); if (rc) return rc; lattr_set = true; }
            const unsigned litems = (unsigned)(rows / io.nw);
            const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(160 * 1024) / lds, (size_t)2048 >> LOGT}));
            const unsigned lgrid = std::min(litems, (unsigned)c->num_cus * per_cu);
            prof_begin(c, rows);
            hipLaunchKernelGGL(lkern, dim3(lgrid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, litems, io);
            prof_end(c);
            HIP_TRY(hipGetLastError());
            return TFHE_OK;
        }
    }
    if constexpr (std::is_same<A, ArithFp>::value && LOGB == 14 && IOMODE == 0) {
        if (x == 0 && c->variant == 0 && io.limb_mask == 0) {  // next row prefetched into registers under the middle pass: +6 % stand-alone, neutral inside the BFV pipeline
            auto pkern = k_ntt_fwd_pf<A, LOGB, LOGT>;
            static bool pattr_set = false;
            if (!pattr_set) { int rc = set_lds(pkern, lds); if (rc) return rc; pattr_set = true; }
            const unsigned pgrid = std::min((unsigned)rows, (unsigned)c->num_cus);
            prof_begin(c, rows);
            hipLaunchKernelGGL(pkern, dim3(pgrid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, (u32)rows);
            prof_end(c);
            HIP_TRY(hipGetLastError());
            return TFHE_OK;
        }
    }
    auto kern = k_ntt_fwd_block<A, LOGB, LOGT, IOMODE>;
    static bool attr_set = false;
    if (!attr_set) { int rc = set_lds(kern, lds); if (rc) return rc; attr_set = true; }
    // persistent workgroups: as many as are co-resident (LDS-limited), each loops over items
    const unsigned items = (unsigned)(rows << x);
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(160 * 1024) / lds, (size_t)2048 >> LOGT}));
    const unsigned grid = std::min(items, (unsigned)c->num_cus * per_cu);
    prof_begin(c, rows);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, x, items, io);
    prof_end(c);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
template <class A, int LOGB, int IOMODE = 0>
int launch_block_inv(tfhe_ctx* c, const u64* src, u64* dst, int64_t rows, const limb_sel_t& sel, int x, const ntt_io_t& io) {
    if constexpr (IOMODE == 0) {
        if (io.mode == 2) return launch_block_inv<A, LOGB, 2>(c, src, dst, rows, sel, x, io);
    }
    constexpr int LOGT = logt_for(LOGB);
    const size_t lds = (size_t)lds_words<LOGB, LOGT>() * 8;
    if constexpr (std::is_same<A, ArithFp>::value && LOGB == 14) {
        // whole 2^14 rows in fp64: the staged kernel (next row copied HBM -> LDS under the last pass)
        if (x == 0 && c->variant != 3 && io.limb_mask == 0) {
            auto skern = k_ntt_inv_staged<A, LOGB, LOGT, IOMODE>;
            static bool sattr_set = false;
            if (!sattr_set) { int rc = set_lds(skern, lds); if (rc) return rc; sattr_set = true; }
            const unsigned sgrid = std::min((unsigned)rows, (unsigned)c->num_cus);
            prof_begin(c, rows);
            hipLaunchKernelGGL(skern, dim3(sgrid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, (u32)rows, io);
            prof_end(c);
            HIP_TRY(hipGetLastError());
            return TFHE_OK;
        }
    }
    auto kern = k_ntt_inv_block<A, LOGB, LOGT, IOMODE>;
    static bool attr_set = false;
    if (!attr_set) { int rc = set_lds(kern, lds); if (rc) return rc; attr_set = true; }
    // persistent workgroups: as many as are co-resident (LDS-limited), each loops over items
    const unsigned items = (unsigned)(rows << x);
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(160 * 1024) / lds, (size_t)2048 >> LOGT}));
    const unsigned grid = std::min(items, (unsigned)c->num_cus * per_cu);
    prof_begin(c, rows);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, x, items, io);
    prof_end(c);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

// sub-blocks of N >= 2^16 rows, two per workgroup (k_ntt_*_subpair); plain I/O, 16-byte aligned rows
template <class A>
int launch_subpair(tfhe_ctx* c, bool inverse, const u64* src, u64* dst, int64_t rows, const limb_sel_t& sel, int x, u32 mask) {
    constexpr int LOGB = 14, LOGT = logt_for(LOGB);
    const size_t lds = (size_t)lds_words<LOGB, LOGT>() * 8;
    auto fk = k_ntt_fwd_subpair<A, LOGB, LOGT>;
    auto ik = k_ntt_inv_subpair<A, LOGB, LOGT>;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = set_lds(fk, lds);
        if (!rc) rc = set_lds(ik, lds);
        if (rc) return rc;
        attr_set = true;
    }
    const unsigned items = (unsigned)(rows << (x - 1));
    const unsigned grid = std::min(items, (unsigned)c->num_cus);
    prof_begin(c, rows);
    if (inverse) hipLaunchKernelGGL(ik, dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, x, items, mask);
    else hipLaunchKernelGGL(fk, dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, x, items, mask);
    prof_end(c);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

int launch_generic(tfhe_ctx* c, bool inverse, const u64* src, u64* dst, int64_t rows, const limb_sel_t& sel, const ntt_io_t& io) {
    const size_t lds = (size_t)c->N * 8;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = set_lds(k_ntt_fwd_generic, 128 * 1024);
        if (rc) return rc;
        rc = set_lds(k_ntt_inv_generic, 128 * 1024);
        if (rc) return rc;
        attr_set = true;
    }
    const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, c->N / 2));
    prof_begin(c, rows);
    if (inverse)
        hipLaunchKernelGGL(k_ntt_inv_generic, dim3((unsigned)rows), dim3(threads), lds, c->stream, src, dst, c->limbs_dev, sel, c->logN, io);
    else
        hipLaunchKernelGGL(k_ntt_fwd_generic, dim3((unsigned)rows), dim3(threads), lds, c->stream, src, dst, c->limbs_dev, sel, c->logN, io);
    prof_end(c);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

// N > 2^14 with one arithmetic policy (`fp`) for the block stage: x top stages on global memory + 2^14 blocks (needs an
// out-of-place intermediate), or the one-kernel variants of the fp64 policy.  io.limb_mask restricts every kernel of the
// call to those limbs (rings that mix modulus sizes: one call per policy).
int run_ntt_large(tfhe_ctx* c, bool inverse, const u64* src, u64* dst, int64_t rows, const limb_sel_t& sel, const ntt_io_t& io,
                  const ntt_io_t* iop, bool fp, bool wide_lift = false) {
    const int n = c->logN, x = n - 14;
    const bool pair15 = x == 1 && c->variant == 0 && fp;
    if (iop && x == 1 && !(pair15 && !inverse && io.mode == 1))
        return fail(TFHE_E_UNSUPPORTED, "fused NTT I/O transforms need N <= 2^14 (digit lift: N <= 2^16, fp64 policy)");
    if (pair15) {
        // N = 2^15, fp64 policy: top stage and both 2^14 sub-blocks in one kernel (one read + one write of the row)
        constexpr int LOGT = logt_for(14);
        const size_t lds = (size_t)lds_words<14, LOGT>() * 8;
        static bool pattr_set = false;
        if (!pattr_set) {
            int rc2 = set_lds(k_ntt_fwd_pair<ArithFp, 14, LOGT>, lds);
            if (!rc2) rc2 = set_lds(k_ntt_fwd_pair<ArithFp, 14, LOGT, true>, lds);
            if (!rc2) rc2 = set_lds(k_ntt_fwd_pair<ArithFpWide, 14, LOGT, true>, lds);
            if (!rc2) rc2 = set_lds(k_ntt_inv_pair<ArithFp, 14, LOGT>, lds);
            if (rc2) return rc2;
            pattr_set = true;
        }
        const unsigned grid = std::min((unsigned)rows, (unsigned)c->num_cus);
        prof_begin(c, rows);
        if (inverse) hipLaunchKernelGGL((k_ntt_inv_pair<ArithFp, 14, LOGT>), dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, (u32)rows, io.limb_mask);
        else if (io.mode == 1 && wide_lift) hipLaunchKernelGGL((k_ntt_fwd_pair<ArithFpWide, 14, LOGT, true>), dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, (u32)rows, io);
        else if (io.mode == 1) hipLaunchKernelGGL((k_ntt_fwd_pair<ArithFp, 14, LOGT, true>), dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, (u32)rows, io);
        else hipLaunchKernelGGL((k_ntt_fwd_pair<ArithFp, 14, LOGT>), dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, (u32)rows, io);
        prof_end(c);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    if (x > 3) return fail(TFHE_E_UNSUPPORTED, "N = 2^%d not supported (max 2^17)", n);
    // fp64 policy: two sub-blocks per workgroup (16-byte pieces on the natural-order side)
    const bool pairable = x >= 2 && c->variant == 0 && fp && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0;
    if (iop && !(pairable && x == 2 && !inverse && io.mode == 1 && src != dst))
        return fail(TFHE_E_UNSUPPORTED, "fused NTT I/O transforms need N <= 2^14 (digit lift: N <= 2^16, fp64 policy)");
    if (!inverse && pairable && x == 2 && src != dst) {
        // N = 2^16 forward in one kernel: top stages folded into the paired sub-block kernel (out of place only: the
        // row's other workgroup reads the same source words)
        constexpr int LOGT = logt_for(14);
        const size_t lds = (size_t)lds_words<14, LOGT>() * 8;
        auto qk = io.mode == 1 ? (wide_lift ? k_ntt_fwd_quad<ArithFpWide, 14, LOGT, true> : k_ntt_fwd_quad<ArithFp, 14, LOGT, true>)
                               : k_ntt_fwd_quad<ArithFp, 14, LOGT, false>;
        static bool qattr_set = false;
        if (!qattr_set) {
            int rc2 = set_lds(k_ntt_fwd_quad<ArithFp, 14, LOGT, true>, lds);
            if (!rc2) rc2 = set_lds(k_ntt_fwd_quad<ArithFpWide, 14, LOGT, true>, lds);
            if (!rc2) rc2 = set_lds(k_ntt_fwd_quad<ArithFp, 14, LOGT, false>, lds);
            if (rc2) return rc2;
            qattr_set = true;
        }
        const unsigned items = (unsigned)(rows << 1);
        const unsigned grid = std::min(items, (unsigned)c->num_cus);
        prof_begin(c, rows);
        hipLaunchKernelGGL(qk, dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, items, io);
        prof_end(c);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    // MEASURED (r03, same box, 512 polys x 7 limbs of 50 bits): 1.75 TB/s against 1.98 for the two-kernel path below (1.64 with
    // device-scope atomic loads of the parked words instead of an L1-invalidating acquire fence; 1.14 with a release fence, which
    // writes the L2 back) -- one 512-thread workgroup per CU runs its load, four sub-block transforms, parking, read-back, top
    // stages and stores strictly one after the other, where the two-kernel path has two workgroups per row in flight and
    // streaming kernels around them.  Off: the template is not instantiated unless -DTFHE_INV_QUAD=1.
#ifndef TFHE_INV_QUAD
#define TFHE_INV_QUAD 0
#endif
#if TFHE_INV_QUAD
    if (inverse && pairable && x == 2 && !iop) {
        // N = 2^16 inverse in one kernel (k_ntt_inv_quad): one row per workgroup pass, sub-block results parked in a
        // per-workgroup scratch row, top stages on the thread's own columns
        constexpr int LOGT = logt_for(14);
        const size_t lds = (size_t)lds_words<14, LOGT>() * 8;
        auto qk = k_ntt_inv_quad<ArithFp, 14, LOGT>;
        static bool iqattr_set = false;
        if (!iqattr_set) { int rc2 = set_lds(qk, lds); if (rc2) return rc2; iqattr_set = true; }
        const unsigned grid = (unsigned)std::min<int64_t>(rows, (int64_t)c->num_cus);
        void* scr = nullptr;
        int rc2 = ensure_ws(c, (size_t)grid * c->N * 8, &scr);
        if (rc2) return rc2;
        prof_begin(c, rows);
        hipLaunchKernelGGL(qk, dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, (u64*)scr, c->limbs_dev, sel, (u32)rows, io.limb_mask);
        prof_end(c);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
#endif
    // N = 2^16 inverse, out of place, in ONE pass (k_ntt_inv_quad2: the two top stages first, r04): 2 N 8 bytes per row instead of
    // the 4 N 8 of the sub-block kernel + k_ntt_inv_top<2> below; TFHE_INV_I2=0 keeps the two-kernel path (comparisons)
    static const bool inv_i2 = !(getenv("TFHE_INV_I2") && getenv("TFHE_INV_I2")[0] == '0');
    if (inverse && pairable && x == 2 && !iop && src != dst && inv_i2) {
        constexpr int LOGT = logt_for(14);
        const size_t lds = (size_t)lds_words<14, LOGT>() * 8;
        auto qk = k_ntt_inv_quad2<ArithFp, 14, LOGT>;
        static bool i2attr_set = false;
        if (!i2attr_set) { int rc2 = set_lds(qk, lds); if (rc2) return rc2; i2attr_set = true; }
        const unsigned items = (unsigned)(rows << 1);
        const unsigned grid = std::min(items, (unsigned)c->num_cus);
        prof_begin(c, rows);
        hipLaunchKernelGGL(qk, dim3(grid), dim3(1 << LOGT), lds, c->stream, src, dst, c->limbs_dev, sel, items, io.limb_mask);
        prof_end(c);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    void* tmp = nullptr;
    int rc = ensure_ws(c, (size_t)rows * c->N * 8, &tmp);
    if (rc) return rc;
    u64* t = (u64*)tmp;
    const dim3 tg((unsigned)((((c->N >> x) + 255) / 256) * rows));
    if (!inverse) {
        prof_begin(c, 0);
        switch (x) {
            case 1: hipLaunchKernelGGL(k_ntt_fwd_top<1>, tg, dim3(256), 0, c->stream, src, t, c->limbs_dev, sel, n, io.limb_mask); break;
            case 2: hipLaunchKernelGGL(k_ntt_fwd_top<2>, tg, dim3(256), 0, c->stream, src, t, c->limbs_dev, sel, n, io.limb_mask); break;
            default: hipLaunchKernelGGL(k_ntt_fwd_top<3>, tg, dim3(256), 0, c->stream, src, t, c->limbs_dev, sel, n, io.limb_mask); break;
        }
        prof_end(c);
        HIP_TRY(hipGetLastError());
        if (pairable) return launch_subpair<ArithFp>(c, false, t, dst, rows, sel, x, io.limb_mask);
        return fp ? launch_block_fwd<ArithFp, 14>(c, t, dst, rows, sel, x, io) : launch_block_fwd<ArithInt, 14>(c, t, dst, rows, sel, x, io);
    }
    if (pairable) rc = launch_subpair<ArithFp>(c, true, src, t, rows, sel, x, io.limb_mask);
    else rc = fp ? launch_block_inv<ArithFp, 14>(c, src, t, rows, sel, x, io) : launch_block_inv<ArithInt, 14>(c, src, t, rows, sel, x, io);
    if (rc) return rc;
    prof_begin(c, 0);
    switch (x) {
        case 1: hipLaunchKernelGGL(k_ntt_inv_top<1>, tg, dim3(256), 0, c->stream, t, dst, c->limbs_dev, sel, n, io.limb_mask); break;
        case 2: hipLaunchKernelGGL(k_ntt_inv_top<2>, tg, dim3(256), 0, c->stream, t, dst, c->limbs_dev, sel, n, io.limb_mask); break;
        default: hipLaunchKernelGGL(k_ntt_inv_top<3>, tg, dim3(256), 0, c->stream, t, dst, c->limbs_dev, sel, n, io.limb_mask); break;
    }
    prof_end(c);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

// forward / inverse transform of `rows` limb-polynomials; src == dst allowed
// Per-limb routing masks are 32 bits wide (ntt_io_t::limb_mask, the key-switch kernels' limb_mask); TFHE_MAX_LIMBS is 40
// and a special prime makes 41 working limbs, so rings with more than 32 selected limbs cannot be split by policy: they
// run whole on the policy that takes every limb (all fp64-size / all narrow: the fast kernels unmasked; otherwise the
// u64 / generic kernels unmasked).
static inline u32 mask_all(int n) { return n >= 32 ? ~0u : ((1u << n) - 1u); }
template <class F>
static inline u32 mask_of(int n, F pred) {  // bit j set where pred(j); for n > 32: all-ones if pred holds everywhere, else 0
    if (n > 32) {
        for (int j = 0; j < n; j++)
            if (!pred(j)) return 0u;
        return ~0u;
    }
    u32 m = 0;
    for (int j = 0; j < n; j++)
        if (pred(j)) m |= 1u << j;
    return m;
}

// `io` (optional) selects the fused global-I/O transforms of ntt_io_t; only for N <= 2^14 (single-block transforms)
int run_ntt(tfhe_ctx* c, bool inverse, const u64* src, u64* dst, int64_t rows, const limb_sel_t& sel, const ntt_io_t* iop = nullptr) {
    if (rows == 0) return TFHE_OK;
    const ntt_io_t io = iop ? *iop : io_plain();
    if (rows < 0 || (rows << std::max(0, c->logN - 14)) > 0x7fffffffll) return fail(TFHE_E_BADARG, "bad polynomial count");
    const int n = c->logN;
    const bool use_block = (c->variant != 1 && n >= 10) || n > 14;
    if (!use_block) return launch_generic(c, inverse, src, dst, rows, sel, io);
    if (n <= 14) {
        const bool fp = sel_fp(c, sel, 0);
        // a ring that mixes fp64-size moduli with larger ones (60-bit q0 / special prime next to 40-bit primes): one launch
        // per policy, each taking its limbs (ntt_io_t::limb_mask)
        const u32 fpmask = mask_of(sel.n, [&](int j) { return c->limbs_host[sel.idx[j]].Wd != nullptr; });
        const u32 all = mask_all(sel.n);
        if (!fp && c->variant == 0 && sel.n <= 32 && fpmask != 0 && fpmask != all && (io.mode != 1 || n >= 12) && (rows << n) >= TFHE_MIXED_MIN_WORDS) {
            // (digit-lift mode: the mask is on the TARGET limb j of item (b, i, j); the fp64 lift takes source limbs of either
            // size through ArithFpWide, instantiated for N >= 2^12)
            ntt_io_t a = io, b = io;
            a.limb_mask = fpmask;
            b.limb_mask = all & ~fpmask;
            // the two policies side by side (lanes_t): the u64 launches -- the long pole -- first, on the main lane
            lanes_t lanes(c);
            if (io.mode == 1) {
                int rc1 = TFHE_E_UNSUPPORTED;
                lanes.use(0);
                switch (n) {
                    case 12: rc1 = launch_block_fwd<ArithInt, 12, 1>(c, src, dst, rows, sel, 0, b); break;
                    case 13: rc1 = launch_block_fwd<ArithInt, 13, 1>(c, src, dst, rows, sel, 0, b); break;
                    case 14: rc1 = launch_block_fwd<ArithInt, 14, 1>(c, src, dst, rows, sel, 0, b); break;
                }
                if (rc1) return rc1;
                lanes.use(1);
                switch (n) {
                    case 12: return launch_block_fwd<ArithFpWide, 12, 1>(c, src, dst, rows, sel, 0, a);
                    case 13: return launch_block_fwd<ArithFpWide, 13, 1>(c, src, dst, rows, sel, 0, a);
                    default: return launch_block_fwd<ArithFpWide, 14, 1>(c, src, dst, rows, sel, 0, a);
                }
            }
            switch (n) {
#define CASE_(LB)                                                                                                  \
    case LB: {                                                                                                     \
        lanes.use(0);                                                                                              \
        int rc1 = inverse ? launch_block_inv<ArithInt, LB>(c, src, dst, rows, sel, 0, b) : launch_block_fwd<ArithInt, LB>(c, src, dst, rows, sel, 0, b); \
        if (rc1) return rc1;                                                                                       \
        lanes.use(1);                                                                                              \
        return inverse ? launch_block_inv<ArithFp, LB>(c, src, dst, rows, sel, 0, a) : launch_block_fwd<ArithFp, LB>(c, src, dst, rows, sel, 0, a); \
    }
                CASE_(10) CASE_(11) CASE_(12) CASE_(13) CASE_(14)
#undef CASE_
            }
        }
        switch (n) {
#define CASE_(LB)                                                                                                  \
    case LB:                                                                                                       \
        if (fp) return inverse ? launch_block_inv<ArithFp, LB>(c, src, dst, rows, sel, 0, io) : launch_block_fwd<ArithFp, LB>(c, src, dst, rows, sel, 0, io); \
        return inverse ? launch_block_inv<ArithInt, LB>(c, src, dst, rows, sel, 0, io) : launch_block_fwd<ArithInt, LB>(c, src, dst, rows, sel, 0, io);
            CASE_(10) CASE_(11) CASE_(12) CASE_(13) CASE_(14)
#undef CASE_
        }
    }
    // N > 2^14
    if (!iop && c->variant == 0) {
        const u32 fpmask = mask_of(sel.n, [&](int j) { return c->limbs_host[sel.idx[j]].Wd != nullptr; });
        const u32 all = mask_all(sel.n);
        if (sel.n <= 32 && fpmask != 0 && fpmask != all && (rows << n) >= TFHE_MIXED_MIN_WORDS) {  // mixed modulus sizes: one pass per policy, each over its limbs
            ntt_io_t a = io, b = io;
            a.limb_mask = fpmask;
            b.limb_mask = all & ~fpmask;
            // the transform scratch of both policies (disjoint rows of it) is sized before the fork: growing it synchronises
            void* tmp = nullptr;
            int rc1 = ensure_ws(c, (size_t)rows * c->N * 8, &tmp);
            if (rc1) return rc1;
            lanes_t lanes(c);
            // (which policy is enqueued first makes no difference: TFHE_LANE_FP_FIRST A/B, profiles/r06b_lane_order_ab.txt)
            lanes.use(0);   // the u64 launches (top stages + block kernel: the long pole) first, on the main lane
            rc1 = run_ntt_large(c, inverse, src, dst, rows, sel, b, nullptr, false);
            if (rc1) return rc1;
            lanes.use(1);
            return run_ntt_large(c, inverse, src, dst, rows, sel, a, nullptr, true);
        }
    }
    return run_ntt_large(c, inverse, src, dst, rows, sel, io, iop, sel_fp(c, sel, n - 14));
}


bool bfv_core_fusable(const tfhe_ctx* c, const limb_sel_t& sel) { return c->variant == 0 && c->logN >= 12 && c->logN <= 14 && sel_fp(c, sel, 0); }
// forward transforms + tensor + inverse transforms of one BFV multiplication chunk in one kernel (fp64 policy, N = 2^12 .. 2^14:
// the reference's own BFV tests run at 2^11 - 2^12, test/bfv_crt.jl:8, and its MNIST parameters at 2^13, infer.jl:97);
// *done = false when the configuration is not covered.  scratch: one row per workgroup.
#ifndef TFHE_GRID_MULT_CORE  // workgroups per CU slot in the grids of the two fused kernels (> 1: the dispatcher balances the items)
#define TFHE_GRID_MULT_CORE 1u
#endif
#ifndef TFHE_GRID_MULT_KS
#define TFHE_GRID_MULT_KS 1u
#endif
template <int LOGB>
static int launch_bfv_core_fused_n(tfhe_ctx* c, const u64* Ea, const u64* Eb, u64* T, u64* scratch, int64_t nct, const limb_sel_t& sel,
                                   const core_alt_t& alt, bool out_double) {
    constexpr int LOGT = logt_for(LOGB);
    const size_t lds = fused_lds_bytes<LOGB, LOGT, TFHE_TWL_CORE>();
    auto kern = out_double ? k_bfv_core_fused<ArithFp, LOGB, LOGT, true> : k_bfv_core_fused<ArithFp, LOGB, LOGT, false>;
    static bool attr_set = false;
    if (!attr_set) {
        int rc = set_lds(k_bfv_core_fused<ArithFp, LOGB, LOGT, true>, lds);
        if (!rc) rc = set_lds(k_bfv_core_fused<ArithFp, LOGB, LOGT, false>, lds);
        if (rc) return rc;
        attr_set = true;
    }
    const unsigned items = (unsigned)(nct * sel.n);
    // one 512-thread workgroup fills a CU at 2^14; the 256-thread workgroups of the smaller rings leave room for a second one
    const unsigned grid = std::min(items, (LOGB == 14 ? 1u : 2u) * (unsigned)c->num_cus * TFHE_GRID_MULT_CORE);
    prof_begin(c, (int64_t)items * 7);  // limb transforms inside this launch: 4 forward + 3 inverse per item
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1 << LOGT), lds, c->stream, Ea, Eb, T, scratch, c->limbs_dev, sel, items, alt);
    prof_end(c);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
int launch_bfv_core_fused(tfhe_ctx* c, const u64* Ea, const u64* Eb, u64* T, u64* scratch, int64_t nct, const limb_sel_t& sel, bool* done,
                          const core_alt_t* altp = nullptr, bool out_double = false) {
    *done = false;
    if (!bfv_core_fusable(c, sel) || nct * sel.n > 0x7fffffffll) return TFHE_OK;
    core_alt_t alt;
    if (altp) alt = *altp; else { memset(&alt, 0, sizeof alt); alt.a = alt.b = nullptr; }
    int rc;
    switch (c->logN) {
        case 12: rc = launch_bfv_core_fused_n<12>(c, Ea, Eb, T, scratch, nct, sel, alt, out_double); break;
        case 13: rc = launch_bfv_core_fused_n<13>(c, Ea, Eb, T, scratch, nct, sel, alt, out_double); break;
        default: rc = launch_bfv_core_fused_n<14>(c, Ea, Eb, T, scratch, nct, sel, alt, out_double); break;
    }
    if (rc) return rc;
    *done = true;
    return TFHE_OK;
}

// grid of the row-wise kernels (one limb row per blockIdx.x): few rows -- a single ciphertext at N = 2^16 is 14 -- are
// split over blockIdx.y so that the launch still covers the chip (about 2048 workgroups, at least 1024 coefficients each)
static inline dim3 row_grid(unsigned rows, size_t n) {
    const unsigned want = rows ? (2048u + rows - 1) / rows : 1u;
    const unsigned cap = (unsigned)std::max<size_t>(1, n / 1024);
    return dim3(rows, std::max(1u, std::min(want, cap)));
}

template <int OP>
int run_pointwise(tfhe_ctx* c, const u64* a, const u64* b, const u64* acc, u64* dst, int64_t count, int limbs,
                  const int32_t* idx, const scal_arg_t* sc) {
    if (!c) return fail(TFHE_E_BADARG, "null context");
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    if (count == 0) return TFHE_OK;
    if (count < 0 || count * limbs > 0x7fffffffll) return fail(TFHE_E_BADARG, "bad polynomial count");
    scal_arg_t s0;
    if (!sc) { memset(&s0, 0, sizeof s0); sc = &s0; }
    hipLaunchKernelGGL(k_pointwise<OP>, row_grid((unsigned)(count * limbs), (size_t)c->N), dim3(256), 0, c->stream, a, b, acc, dst,
                       c->limbs_dev, sel, *sc, (u32)c->N);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* tfhe_last_error(void) { return g_err.c_str(); }

int tfhe_device_count(int* n) {
    if (!n) return fail(TFHE_E_BADARG, "null out pointer");
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; return fail(TFHE_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return TFHE_OK;
}
int tfhe_set_device(int dev) { HIP_TRY(hipSetDevice(dev)); return TFHE_OK; }

int tfhe_ctx_create(int64_t N, int L, const uint64_t* q, const uint64_t* psi, tfhe_ctx** out) {
    using namespace hostmath;
    if (!out) return fail(TFHE_E_BADARG, "null out pointer");
    *out = nullptr;
    if (N < 2 || (N & (N - 1)) || N > (1 << 17)) return fail(TFHE_E_BADARG, "N=%lld must be a power of two in [2, 2^17]", (long long)N);
    if (L < 1 || L > TFHE_MAX_LIMBS || !q) return fail(TFHE_E_BADARG, "L=%d out of range [1,%d]", L, TFHE_MAX_LIMBS);
    int logN = 0;
    while ((1ll << logN) < N) logN++;
    for (int l = 0; l < L; l++) {
        if (q[l] < 3 || q[l] >= (1ull << 62) || !is_prime(q[l])) return fail(TFHE_E_BADARG, "q[%d]=%llu is not a prime below 2^62", l, (unsigned long long)q[l]);
        if ((q[l] - 1) % (2 * (u64)N)) return fail(TFHE_E_BADARG, "q[%d]=%llu: 2N does not divide q-1 (no 2N-th root of unity; the reference's naive ψ=0 path is host-only)", l, (unsigned long long)q[l]);
        for (int m = 0; m < l; m++)
            if (q[m] == q[l]) return fail(TFHE_E_BADARG, "q[%d] repeats q[%d]", l, m);
    }
    tfhe_ctx* c = new tfhe_ctx();
    c->N = N; c->logN = logN; c->L = L;
    c->q.assign(q, q + L);
    c->psi.resize(L);
    c->limbs_host.resize(L);
    ntt_host_tabs_t HT;
    auto up = [&](const void* host, size_t bytes, const void** dev) -> bool {
        if (!host || !bytes) { *dev = nullptr; return true; }
        void* d = nullptr;
        if (devalloc::malloc_retry(&d, bytes) != hipSuccess) return false;
        c->tabs.push_back(d);
        if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
        *dev = d;
        return true;
    };
    for (int l = 0; l < L; l++) {
        const u64 ql = q[l];
        u64 p = (psi && psi[l]) ? psi[l] : minimal_primitive_root(ql, 2 * (u64)N);
        ntt_limb_t& LL = c->limbs_host[l];
        if (build_ntt_tables_all(N, ql, p, HT, &LL) != 0) {  // pow2_cyc_rings.jl:31,61 (+ primitivity: psi^N == -1)
            tfhe_ctx_destroy(c);
            return fail(TFHE_E_BADARG, "psi[%d]=%llu is not a primitive 2N-th root of unity mod q[%d]", l, (unsigned long long)p, l);
        }
        c->psi[l] = p;
        const size_t tb = (size_t)N * sizeof(twd_t), fb = (size_t)N * sizeof(ftwd_t);
        bool ok = up(HT.W.data(), tb, (const void**)&LL.W) && up(HT.Wi.data(), tb, (const void**)&LL.Winv) &&
                  up(LL.Wb ? HT.Wb.data() : nullptr, tb, (const void**)&LL.Wb) && up(LL.Winvb ? HT.Wib.data() : nullptr, tb, (const void**)&LL.Winvb) &&
                  up(LL.Wd ? HT.Wd.data() : nullptr, fb, (const void**)&LL.Wd) && up(LL.Winvd ? HT.Wid.data() : nullptr, fb, (const void**)&LL.Winvd) &&
                  up(LL.Wdb ? HT.Wdb.data() : nullptr, fb, (const void**)&LL.Wdb) && up(LL.Winvdb ? HT.Widb.data() : nullptr, fb, (const void**)&LL.Winvdb) &&
                  up(LL.i2_t0 ? HT.I2t0.data() : nullptr, HT.I2t0.size() * sizeof(ftwd_t), (const void**)&LL.i2_t0) &&
                  up(LL.i2_winvb ? HT.I2wib.data() : nullptr, HT.I2wib.size() * sizeof(ftwd_t), (const void**)&LL.i2_winvb);
        if (!ok) {
            tfhe_ctx_destroy(c);
            return fail(TFHE_E_HIP, "allocating the twiddle tables failed (no usable HIP device?)");
        }
    }
    if (devalloc::malloc_retry(&c->limbs_dev, L * sizeof(ntt_limb_t)) != hipSuccess) { tfhe_ctx_destroy(c); return fail(TFHE_E_HIP, "hipMalloc failed"); }
    hipMemcpy(c->limbs_dev, c->limbs_host.data(), L * sizeof(ntt_limb_t), hipMemcpyHostToDevice);
    if (hipStreamCreate(&c->stream) != hipSuccess) { tfhe_ctx_destroy(c); return fail(TFHE_E_HIP, "hipStreamCreate failed"); }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            c->num_cus = cus;
    }
    c->own_stream = true;
    c->main_stream = c->stream;
    devalloc::register_stream(&c->main_stream);
    *out = c;
    return TFHE_OK;
}

int tfhe_ctx_destroy(tfhe_ctx* c) {
    if (!c) return TFHE_OK;
    devalloc::unregister_stream(&c->main_stream);
    if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    c->stream = c->main_stream;
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto* t : c->tabs) hipFree(t);
    for (auto* t : c->ksw_allocs) hipFree(t);
    if (c->limbs_dev) hipFree(c->limbs_dev);
    if (c->ws) { if (c->ws_pooled) (void)devalloc::release(c->ws); else hipFree(c->ws); }
    if (c->zero_row) hipFree(c->zero_row);
    for (auto& p : c->prof_pairs) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return TFHE_OK;
}

int tfhe_ctx_psi(const tfhe_ctx* c, uint64_t* out) {
    if (!c || !out) return fail(TFHE_E_BADARG, "null argument");
    memcpy(out, c->psi.data(), c->L * 8);
    return TFHE_OK;
}
int tfhe_ctx_set_stream(tfhe_ctx* c, void* s) {
    if (!c) return fail(TFHE_E_BADARG, "null context");
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    hipStream_t old = c->own_stream ? c->stream : nullptr, fresh = (hipStream_t)s;
    if (!s) HIP_TRY(hipStreamCreate(&fresh));
    devalloc::set_stream(&c->main_stream, fresh);                  // the allocator reads the slot under its mutex (tfhe_free)
    c->stream = fresh;
    c->own_stream = (s == nullptr);
    if (old) hipStreamDestroy(old);
    return TFHE_OK;
}
int tfhe_ctx_sync(tfhe_ctx* c) {
    if (!c) return fail(TFHE_E_BADARG, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TFHE_OK;
}
// Work submitted to `c` after this call starts after everything submitted to `producer` before it (one recorded event, no
// host wait): the way two contexts with their own streams hand device buffers to each other.
int tfhe_ctx_wait_for(tfhe_ctx* c, tfhe_ctx* producer) {
    if (!c || !producer) return fail(TFHE_E_BADARG, "null context");
    if (c == producer || c->stream == producer->stream) return TFHE_OK;
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, producer->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev, 0);
    hipEventDestroy(ev);  // deferred by the runtime until the event has completed
    if (e != hipSuccess) return fail(TFHE_E_HIP, "tfhe_ctx_wait_for: %s", hipGetErrorString(e));
    return TFHE_OK;
}
int tfhe_ctx_set_ntt_variant(tfhe_ctx* c, int v) {
    if (!c || v < 0 || v > 3) return fail(TFHE_E_BADARG, "variant must be 0, 1, 2 or 3");
    c->variant = v;
    return TFHE_OK;
}

int tfhe_malloc(size_t bytes, void** p) {
    if (!p) return fail(TFHE_E_BADARG, "null out pointer");
    hipError_t e = devalloc::alloc(bytes, p);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? TFHE_E_NOMEM : TFHE_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return TFHE_OK;
}
int tfhe_free(void* p) { if (p) HIP_TRY(devalloc::release(p)); return TFHE_OK; }
int tfhe_alloc_stats(uint64_t* live_bytes, uint64_t* cached_bytes, uint64_t* hip_mallocs, uint64_t* reuses) {
    devalloc::state_t& s = devalloc::S();
    std::lock_guard<std::mutex> g(s.mu);
    if (live_bytes) *live_bytes = s.live_bytes;
    if (cached_bytes) *cached_bytes = s.cached_bytes;
    if (hip_mallocs) *hip_mallocs = (uint64_t)s.n_hip_malloc;
    if (reuses) *reuses = (uint64_t)s.n_reuse;
    return TFHE_OK;
}
int tfhe_alloc_trim(void) {
    devalloc::state_t& s = devalloc::S();
    std::lock_guard<std::mutex> g(s.mu);
    devalloc::trim_locked(s);
    return TFHE_OK;
}
int tfhe_memcpy_h2d(void* d, const void* s, size_t n) { HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); return TFHE_OK; }
int tfhe_memcpy_d2h(void* d, const void* s, size_t n) { HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); return TFHE_OK; }
int tfhe_memcpy_d2d(tfhe_ctx* c, void* d, const void* s, size_t n) {
    if (!c) return fail(TFHE_E_BADARG, "null context");
    HIP_TRY(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, c->stream));
    return TFHE_OK;
}
// component p of a packed batch [count][polys][words] <-> a batch of single polynomials [count][words]: one strided copy
int tfhe_pack_poly(tfhe_ctx* c, uint64_t* packed, const uint64_t* src, int polys, int p, size_t words, int64_t count) {
    if (!c || !packed || !src) return fail(TFHE_E_BADARG, "null argument");
    if (polys < 1 || p < 0 || p >= polys || count < 0) return fail(TFHE_E_BADARG, "bad component index");
    if (count == 0 || words == 0) return TFHE_OK;
    HIP_TRY(hipMemcpy2DAsync(packed + (size_t)p * words, (size_t)polys * words * 8, src, words * 8, words * 8, (size_t)count,
                             hipMemcpyDeviceToDevice, c->stream));
    return TFHE_OK;
}
int tfhe_unpack_poly(tfhe_ctx* c, uint64_t* dst, const uint64_t* packed, int polys, int p, size_t words, int64_t count) {
    if (!c || !packed || !dst) return fail(TFHE_E_BADARG, "null argument");
    if (polys < 1 || p < 0 || p >= polys || count < 0) return fail(TFHE_E_BADARG, "bad component index");
    if (count == 0 || words == 0) return TFHE_OK;
    HIP_TRY(hipMemcpy2DAsync(dst, words * 8, packed + (size_t)p * words, (size_t)polys * words * 8, words * 8, (size_t)count,
                             hipMemcpyDeviceToDevice, c->stream));
    return TFHE_OK;
}
// dst[k][:] = src[:] for k < count: one ring element (a plaintext, a key) broadcast over a batch
__global__ __launch_bounds__(256) void k_broadcast(u64* __restrict__ dst, const u64* __restrict__ src, size_t words) {
    const size_t w = (size_t)blockIdx.x * 512 + threadIdx.x * 2;
    if (w + 1 < words) {
        const ulonglong2 v = *(const ulonglong2*)(src + w);
        *(ulonglong2*)(dst + (size_t)blockIdx.y * words + w) = v;
    } else if (w < words) {
        dst[(size_t)blockIdx.y * words + w] = src[w];
    }
}
int tfhe_broadcast_poly(tfhe_ctx* c, uint64_t* dst, const uint64_t* src, size_t words, int64_t count) {
    if (!c || !dst || !src) return fail(TFHE_E_BADARG, "null argument");
    if (count < 0 || count > 65535 || (words & 1)) return fail(TFHE_E_BADARG, "count must be in [0, 65535], words even");
    if (count == 0 || words == 0) return TFHE_OK;
    hipLaunchKernelGGL(k_broadcast, dim3((unsigned)((words + 511) / 512), (unsigned)count), dim3(256), 0, c->stream, dst, src, words);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
int tfhe_memset(tfhe_ctx* c, void* d, int byte, size_t n) {
    if (!c) return fail(TFHE_E_BADARG, "null context");
    HIP_TRY(hipMemsetAsync(d, byte, n, c->stream));
    return TFHE_OK;
}

int tfhe_nntt(tfhe_ctx* c, const uint64_t* src, uint64_t* dst, int64_t count, int limbs, const int32_t* idx) {
    if (!c || !src || !dst) return fail(TFHE_E_BADARG, "null argument");
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    return run_ntt(c, false, src, dst, count * limbs, sel);
}
int tfhe_inntt(tfhe_ctx* c, const uint64_t* src, uint64_t* dst, int64_t count, int limbs, const int32_t* idx) {
    if (!c || !src || !dst) return fail(TFHE_E_BADARG, "null argument");
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    return run_ntt(c, true, src, dst, count * limbs, sel);
}

int tfhe_add(tfhe_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* d, int64_t n, int l, const int32_t* i) { return run_pointwise<OP_ADD>(c, a, b, nullptr, d, n, l, i, nullptr); }
int tfhe_sub(tfhe_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* d, int64_t n, int l, const int32_t* i) { return run_pointwise<OP_SUB>(c, a, b, nullptr, d, n, l, i, nullptr); }
int tfhe_neg(tfhe_ctx* c, const uint64_t* a, uint64_t* d, int64_t n, int l, const int32_t* i) { return run_pointwise<OP_NEG>(c, a, nullptr, nullptr, d, n, l, i, nullptr); }
int tfhe_mul(tfhe_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* d, int64_t n, int l, const int32_t* i) { return run_pointwise<OP_MUL>(c, a, b, nullptr, d, n, l, i, nullptr); }
int tfhe_mad(tfhe_ctx* c, const uint64_t* acc, const uint64_t* a, const uint64_t* b, uint64_t* d, int64_t n, int l, const int32_t* i) { return run_pointwise<OP_MAD>(c, a, b, acc, d, n, l, i, nullptr); }
int tfhe_scalar_mul(tfhe_ctx* c, const uint64_t* scal, const uint64_t* a, uint64_t* d, int64_t n, int l, const int32_t* idx) {
    if (!c || !scal) return fail(TFHE_E_BADARG, "null argument");
    limb_sel_t sel;
    int rc = make_sel(c, l, idx, &sel);
    if (rc) return rc;
    scal_arg_t sc;
    memset(&sc, 0, sizeof sc);
    for (int j = 0; j < l; j++) { const u64 q = c->q[sel.idx[j]]; sc.s[j] = hostmath::make_tw(scal[j] % q, q); }
    return run_pointwise<OP_SCAL>(c, a, nullptr, nullptr, d, n, l, idx, &sc);
}

int tfhe_dot(tfhe_ctx* c, const uint64_t* acc, const uint64_t* const* a, const uint64_t* const* b, int n_terms, uint64_t* dst,
             int64_t count, int limbs, const int32_t* idx) {
    if (!c || !a || !b || !dst) return fail(TFHE_E_BADARG, "null argument");
    if (n_terms < 1) return fail(TFHE_E_BADARG, "tfhe_dot needs at least one term");
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    if (count < 0) return fail(TFHE_E_BADARG, "negative count");
    if (count == 0) return TFHE_OK;
    for (int k = 0; k < n_terms; k++)
        if (!a[k] || !b[k]) return fail(TFHE_E_BADARG, "null operand %d", k);
    const u64* running = acc;
    for (int k0 = 0; k0 < n_terms; k0 += TFHE_DOT_MAX) {   // more than 64 terms: further launches accumulate onto dst
        dot_arg_t D;
        D.n = std::min(TFHE_DOT_MAX, n_terms - k0);
        for (int k = 0; k < D.n; k++) { D.a[k] = a[k0 + k]; D.b[k] = b[k0 + k]; }
        hipLaunchKernelGGL(k_dot, row_grid((unsigned)(count * limbs), (size_t)c->N), dim3(256), 0, c->stream, D, running, dst, c->limbs_dev, sel, (u32)c->N);
        HIP_TRY(hipGetLastError());
        running = dst;
    }
    return TFHE_OK;
}

int tfhe_lincomb(tfhe_ctx* c, const uint64_t* scalars, const uint64_t* const* a, int n_terms, uint64_t* dst, int64_t count, int limbs,
                 const int32_t* idx) {
    if (!c || !scalars || !a || !dst) return fail(TFHE_E_BADARG, "null argument");
    if (n_terms < 1) return fail(TFHE_E_BADARG, "tfhe_lincomb needs at least one term");
    if (n_terms > TFHE_DOT_MAX) return fail(TFHE_E_UNSUPPORTED, "tfhe_lincomb takes at most %d terms per call", TFHE_DOT_MAX);
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    if (count < 0) return fail(TFHE_E_BADARG, "negative count");
    if (count == 0) return TFHE_OK;
    for (int k = 0; k < n_terms; k++)
        if (!a[k]) return fail(TFHE_E_BADARG, "null operand %d", k);
    std::vector<u64> sc((size_t)n_terms * limbs);
    for (int k = 0; k < n_terms; k++)
        for (int j = 0; j < limbs; j++) {
            const u64 q = c->q[sel.idx[j]], v = scalars[(size_t)k * limbs + j];
            if (v >= q) return fail(TFHE_E_BADARG, "scalar %d, limb %d is not a residue", k, j);
            sc[(size_t)k * limbs + j] = v;
        }
    void* dsc = nullptr;
    hipError_t e = devalloc::alloc(sc.size() * 8, &dsc);
    if (e != hipSuccess) return fail(TFHE_E_NOMEM, "hipMalloc(%zu): %s", sc.size() * 8, hipGetErrorString(e));
    e = hipMemcpyAsync(dsc, sc.data(), sc.size() * 8, hipMemcpyHostToDevice, c->stream);   // pageable source: staged before the call returns
    if (e != hipSuccess) {
        (void)hipGetLastError();
        devalloc::release(dsc);
        return fail(TFHE_E_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
    }
    dot_arg_t D;
    D.n = n_terms;
    for (int k = 0; k < n_terms; k++) { D.a[k] = a[k]; D.b[k] = nullptr; }
    hipLaunchKernelGGL(k_lincomb, row_grid((unsigned)(count * limbs), (size_t)c->N), dim3(256), 0, c->stream, D, (const u64*)dsc, dst, c->limbs_dev, sel, (u32)c->N);
    hipError_t le = hipGetLastError();
    devalloc::release(dsc);   // parked until the launch above has run
    if (le != hipSuccess) return fail(TFHE_E_HIP, "k_lincomb: %s", hipGetErrorString(le));
    return TFHE_OK;
}

int tfhe_lincomb_many(tfhe_ctx* c, const uint64_t* scalars, const uint64_t* const* a, int n_terms, uint64_t* const* dst, int n_out, int64_t count,
                      int limbs, const int32_t* idx) {
    if (!c || !scalars || !a || !dst) return fail(TFHE_E_BADARG, "null argument");
    if (n_terms < 1 || n_out < 1) return fail(TFHE_E_BADARG, "tfhe_lincomb_many needs at least one term and one output");
    if (n_terms > TFHE_DOT_MAX) return fail(TFHE_E_UNSUPPORTED, "tfhe_lincomb_many takes at most %d terms per call", TFHE_DOT_MAX);
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    if (count < 0) return fail(TFHE_E_BADARG, "negative count");
    for (int k = 0; k < n_terms; k++)
        if (!a[k]) return fail(TFHE_E_BADARG, "null operand %d", k);
    for (int o = 0; o < n_out; o++)
        if (!dst[o]) return fail(TFHE_E_BADARG, "null output %d", o);
    const size_t per_out = (size_t)n_terms * limbs;
    for (int o = 0; o < n_out; o++)
        for (int k = 0; k < n_terms; k++)
            for (int j = 0; j < limbs; j++)
                if (scalars[(size_t)o * per_out + (size_t)k * limbs + j] >= c->q[sel.idx[j]])
                    return fail(TFHE_E_BADARG, "scalar of output %d, term %d, limb %d is not a residue", o, k, j);
    if (count == 0) return TFHE_OK;
    void* dsc = nullptr;
    hipError_t e = devalloc::alloc((size_t)n_out * per_out * 8, &dsc);
    if (e != hipSuccess) return fail(TFHE_E_NOMEM, "hipMalloc(%zu): %s", (size_t)n_out * per_out * 8, hipGetErrorString(e));
    hipError_t le = hipMemcpyAsync(dsc, scalars, (size_t)n_out * per_out * 8, hipMemcpyHostToDevice, c->stream);   // pageable source: staged before the call returns
    dot_arg_t D;
    D.n = n_terms;
    for (int k = 0; k < n_terms; k++) { D.a[k] = a[k]; D.b[k] = nullptr; }
    const dim3 grid = row_grid((unsigned)(count * limbs), (size_t)c->N);
    for (int o0 = 0; o0 < n_out && le == hipSuccess; o0 += 4) {   // four outputs per pass over the operands
        const int no = std::min(4, n_out - o0);
        lincomb_out_t O;
        for (int o = 0; o < 4; o++) O.dst[o] = o < no ? dst[o0 + o] : nullptr;
        const u64* sc = (const u64*)dsc + (size_t)o0 * per_out;
        switch (no) {
            case 1: hipLaunchKernelGGL(k_lincomb_many<1>, grid, dim3(256), 0, c->stream, D, sc, O, c->limbs_dev, sel, (u32)c->N); break;
            case 2: hipLaunchKernelGGL(k_lincomb_many<2>, grid, dim3(256), 0, c->stream, D, sc, O, c->limbs_dev, sel, (u32)c->N); break;
            case 3: hipLaunchKernelGGL(k_lincomb_many<3>, grid, dim3(256), 0, c->stream, D, sc, O, c->limbs_dev, sel, (u32)c->N); break;
            default: hipLaunchKernelGGL(k_lincomb_many<4>, grid, dim3(256), 0, c->stream, D, sc, O, c->limbs_dev, sel, (u32)c->N); break;
        }
        le = hipGetLastError();
    }
    devalloc::release(dsc);   // parked until the launches above have run
    if (le != hipSuccess) return fail(TFHE_E_HIP, "k_lincomb_many: %s", hipGetErrorString(le));
    return TFHE_OK;
}

int tfhe_tensor(tfhe_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, int64_t batch, int limbs, const int32_t* idx) {
    if (!c || !a || !b || !out) return fail(TFHE_E_BADARG, "null argument");
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    if (batch == 0) return TFHE_OK;
    hipLaunchKernelGGL(k_tensor, row_grid((unsigned)(batch * limbs), (size_t)c->N), dim3(256), 0, c->stream, a, b, out, c->limbs_dev, sel, (u32)c->N);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

static int do_rescale(tfhe_ctx* c, const u64* src, u64* dst, int64_t count, const limb_sel_t& sel) {
    if (sel.n < 2) return fail(TFHE_E_LEVEL_MISMATCH, "modswitch needs at least 2 limbs");
    rescale_arg_t ra;
    memset(&ra, 0, sizeof ra);
    const u64 ql = c->q[sel.idx[sel.n - 1]];
    for (int j = 0; j < sel.n - 1; j++) {
        const u64 qj = c->q[sel.idx[j]];
        ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(ql % qj, qj), qj);
    }
    if (count == 0) return TFHE_OK;
    static const bool row_major = getenv("TFHE_RESCALE_ROWS") && getenv("TFHE_RESCALE_ROWS")[0] == '1';
    if (!row_major && c->N >= 512 && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {   // coefficient-major: the last limb read once per polynomial
        const unsigned want = (unsigned)((2048 + count - 1) / count), cap = (unsigned)(c->N / 512);
        hipLaunchKernelGGL(k_rescale_cm, dim3((unsigned)count, std::max(1u, std::min(want, cap))), dim3(256), 0, c->stream, src, dst, c->limbs_dev, sel, ra, (u32)c->N);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    hipLaunchKernelGGL(k_rescale, row_grid((unsigned)(count * (sel.n - 1)), (size_t)c->N), dim3(256), 0, c->stream, src, dst, c->limbs_dev, sel, ra, (u32)c->N);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
int tfhe_rescale(tfhe_ctx* c, const uint64_t* src, uint64_t* dst, int64_t count, int limbs, const int32_t* idx) {
    if (!c || !src || !dst) return fail(TFHE_E_BADARG, "null argument");
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    return do_rescale(c, src, dst, count, sel);
}

int tfhe_select_limbs(tfhe_ctx* c, const uint64_t* src, uint64_t* dst, int64_t count, int src_limbs, const int32_t* which, int nw) {
    if (!c || !src || !dst || !which) return fail(TFHE_E_BADARG, "null argument");
    if (nw < 1 || nw > TFHE_MAX_LIMBS) return fail(TFHE_E_BADARG, "bad limb count");
    limb_sel_t w;
    w.n = nw;
    for (int j = 0; j < nw; j++) {
        if (which[j] < 0 || which[j] >= src_limbs) return fail(TFHE_E_LEVEL_MISMATCH, "which[%d]=%d outside source limbs", j, which[j]);
        w.idx[j] = which[j];
    }
    if (count == 0) return TFHE_OK;
    hipLaunchKernelGGL(k_select, row_grid((unsigned)(count * nw), (size_t)c->N), dim3(256), 0, c->stream, src, dst, w, src_limbs, (u32)c->N);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

static int do_galois(tfhe_ctx* c, const u64* src, u64* dst, u64 g, int64_t rows, const limb_sel_t& sel) {
    if ((g & 1) == 0) return fail(TFHE_E_BADARG, "galois element %llu must be odd", (unsigned long long)g);
    if (src == dst) return fail(TFHE_E_BADARG, "tfhe_galois cannot run in place");
    const u64 m = 2 * (u64)c->N;
    g %= m;
    u64 ginv = 1;  // inverse modulo 2N by Newton iteration (g odd)
    for (int i = 0; i < 6; i++) ginv = (ginv * (2 - g * ginv)) & (m - 1);
    if (rows == 0) return TFHE_OK;
    if (c->variant == 0 && c->logN >= 11 && c->logN <= 14 && rows >= (int64_t)c->num_cus / 2 && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
        // rows that fit the LDS, enough of them to fill the chip: scatter through the LDS, coalesced on both sides
        constexpr int T = 512;
        const size_t lds = (size_t)c->N * 8;
        static bool gattr = false;
        if (!gattr) { int rc = set_lds(k_galois_lds<T>, 128 * 1024); if (rc) return rc; gattr = true; }
        const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / lds));
        const unsigned grid = (unsigned)std::min<int64_t>(rows, (int64_t)c->num_cus * per_cu);
        hipLaunchKernelGGL(k_galois_lds<T>, dim3(grid), dim3(T), lds, c->stream, src, dst, c->limbs_dev, sel, g, (u32)c->N, (u32)rows);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
#ifndef TFHE_GALOIS_XCD
#define TFHE_GALOIS_XCD 1
#endif
    if (TFHE_GALOIS_XCD && c->variant == 0 && c->logN >= 15 && rows % sel.n == 0 && rows / sel.n >= 16) {
        // whole polynomials, enough of them: the XCD-cooperative scatter (k_galois_xcd)
        hipLaunchKernelGGL(k_galois_xcd, dim3(8 * TFHE_ROT_TAIL_SLOTS), dim3(256), 0, c->stream, src, dst, c->limbs_dev, sel, g, (u32)c->N, (u32)(rows / sel.n));
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    hipLaunchKernelGGL(k_galois, row_grid((unsigned)rows, (size_t)c->N), dim3(256), 0, c->stream, src, dst, c->limbs_dev, sel, ginv, (u32)c->N);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
int tfhe_galois(tfhe_ctx* c, const uint64_t* src, uint64_t* dst, uint64_t g, int64_t count, int limbs, const int32_t* idx) {
    if (!c || !src || !dst) return fail(TFHE_E_BADARG, "null argument");
    limb_sel_t sel;
    int rc = make_sel(c, limbs, idx, &sel);
    if (rc) return rc;
    return do_galois(c, src, dst, g, count * limbs, sel);
}

// ---- keyswitch ------------------------------------------------------------------------------------
// One chunk of key switching (rlwe_she.jl:315-347).  By linearity of the NTT the expanded inputs never need
// transforming:  out_s = c_s + INTT(S_s)  (plain)   or   c_s + modswitch-part of INTT(S_s)  (ModulusRaised),
// with S_s = Σ_i evk_{i,s} ⊙ NTT(digit_i)  (see k_ks_inner, k_ks_rescale_add).
//   dig: [batch][level][nw][N]  NTT'd digits      S: [batch][2][nw][N]
// the fused key switches (k_ks_fused at N = 2^14, k_ks_fused_sub at 2^15) read the key as doubles, prepared once per call
// (k_evk_to_f64 in keyswitch_impl)
static bool ks_fused14(const tfhe_ctx* c, int Lk, int level, int special) {
    limb_sel_t w;
    w.n = special ? level + 1 : level;
    for (int j = 0; j < level; j++) w.idx[j] = j;
    if (special) w.idx[level] = Lk - 1;
    if (c->variant != 0) return false;
    if (c->logN == 14) return sel_fp(c, w, 0);
    if (c->logN == 13) {  // the 256 x 32 geometry: same fused kernel (it takes 256 VGPRs + 222 AGPRs: one workgroup per CU)
        static const bool on = !(getenv("TFHE_FUSED13") && getenv("TFHE_FUSED13")[0] == '0');
        return on && sel_fp(c, w, 0);
    }
    if (c->logN == 16) {  // k_ks_fused_sub at X = 2
        static const bool on16 = !(getenv("TFHE_FUSED16") && getenv("TFHE_FUSED16")[0] == '0');
        return on16 && level >= 2 && sel_fp(c, w, 2);
    }
    return c->logN == 15 && level >= 2 && sel_fp(c, w, 1);  // k_ks_fused_sub
}
// pre-lifted c[end] rows (centred doubles from the BFV contraction) are understood by k_ks_fused only
static bool ks_prelift_ok(const tfhe_ctx* c, int level) {
    if (c->logN != 14 || !ks_fused14(c, level, level, 0)) return false;
    u64 lo = ~0ull, hi = 0;  // the bit-cast lift keeps |digit| <= q_i / 2 unreduced: needs q_i <= 2 q_j for every pair
    for (int j = 0; j < level; j++) { lo = std::min(lo, c->q[j]); hi = std::max(hi, c->q[j]); }
    return hi <= 2 * lo;
}
// Part A of the unfused key switch: the NTT-domain RNS digits of c[end], dig [batch][level][nw][N]
// (centred lift of limb i into every working limb, rlwe_she.jl:326-329, then forward transforms).
static int ks_digits_fwd(tfhe_ctx* c, const ks_arg_t& A, const u64* ct, u64* dig, int64_t batch) {
    const int level = A.level, nw = A.nw, polys = A.polys;
    const u32 n = (u32)c->N;
    int rc;
    const bool lift_fused = c->logN <= 14 || ((c->logN == 15 || c->logN == 16) && c->variant == 0 && sel_fp(c, A.w, c->logN - 14) &&
                                              (((uintptr_t)ct | (uintptr_t)dig) & 15u) == 0);
    // rings that mix fp64-size moduli with larger ones (infer.jl:97-112: 60-bit q0 and special prime next to 40-bit primes)
    // at N = 2^15 / 2^16: the fp64-size working limbs take the lift-fused one-kernel transforms (the fp64 lift reads source
    // limbs of either size), the others go through the digit buffer and the u64 kernels
    const u32 fpmask = mask_of(nw, [&](int j) { return c->limbs_host[A.w.idx[j]].Wd != nullptr; });
    const u32 allmask = mask_all(nw);
    const bool lift_mixed = !lift_fused && nw <= 32 && (c->logN == 15 || c->logN == 16) && c->variant == 0 && fpmask != 0 && fpmask != allmask &&
                            (((uintptr_t)ct | (uintptr_t)dig) & 15u) == 0;
    if (lift_fused) {
        // digits: centred lift of limb i of c[end] into every working limb, fused into the forward NTT's loads
        ntt_io_t io = io_plain();
        io.mode = 1; io.level = (u32)level; io.nw = (u32)nw; io.polys = (u32)polys;
        rc = run_ntt(c, false, ct, dig, batch * level * nw, A.w, &io);
        if (rc) return rc;
    } else if (lift_mixed) {
        ntt_io_t io = io_plain();
        io.mode = 1; io.level = (u32)level; io.nw = (u32)nw; io.polys = (u32)polys; io.limb_mask = fpmask;
        const int64_t rows = batch * level * nw;
        const int x = c->logN - 14;
        void* tmp = nullptr;
        rc = ensure_ws(c, (size_t)rows * c->N * 8, &tmp);   // (before the fork: growing the workspace synchronises)
        if (rc) return rc;
        lanes_t lanes(c);   // the fp64-size working limbs on the side lane, beside the u64 ones
        lanes.use(1);
        rc = run_ntt_large(c, false, ct, dig, rows, A.w, io, &io, true, true);
        if (rc) return rc;
        lanes.use(0);
        // the larger working limbs: lift fused into the top-stage kernel (into the transform scratch), then the u64 block kernels
        ntt_io_t iw = io;
        iw.limb_mask = allmask & ~fpmask;
        const dim3 tg((unsigned)((((c->N >> x) + 255) / 256) * rows));
        if (x == 1) hipLaunchKernelGGL(k_ntt_fwd_top_lift<1>, tg, dim3(256), 0, c->stream, ct, (u64*)tmp, c->limbs_dev, A.w, c->logN, iw);
        else hipLaunchKernelGGL(k_ntt_fwd_top_lift<2>, tg, dim3(256), 0, c->stream, ct, (u64*)tmp, c->limbs_dev, A.w, c->logN, iw);
        HIP_TRY(hipGetLastError());
        ntt_io_t ib = io_plain();
        ib.limb_mask = allmask & ~fpmask;
        rc = launch_block_fwd<ArithInt, 14>(c, (const u64*)tmp, dig, rows, A.w, x, ib);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_ks_digits, row_grid((unsigned)(batch * level * nw), (size_t)c->N), dim3(256), 0, c->stream, ct, dig, c->limbs_dev, A, n, 0u);
        HIP_TRY(hipGetLastError());
        rc = run_ntt(c, false, dig, dig, batch * level * nw, A.w);
        if (rc) return rc;
    }
    return TFHE_OK;
}

// tfhe_matmul_diag, evaluation-domain form: U[grp][j] = NTT_j( [P[grp]] mod q_j ) (x P^-1 unless *scaled comes back false) for
// the `groups` special-limb rows P (coefficient domain, canonical) and the limbs j of `sl`.  At N = 2^16 the lift rides on the
// forward transforms' loads (ntt_io_t::lift_unsigned: k_ntt_fwd_quad for the fp64-size limbs, k_ntt_fwd_top_lift + the u64 block
// kernels for the others) and the factor P^-1 is left to k_md_acc; otherwise k_md_lift writes the scaled lifts to `LF` and plain
// transforms follow.
// Whether the unsigned lift can ride on the forward transforms' loads (ntt_io_t::lift_unsigned) for these target limbs and this
// special prime: N = 2^15 / 2^16 through the pair / quad kernels (+ top-stage lift for the u64 limbs); N <= 2^14 through
// run_ntt's digit-lift mode -- except when every target limb is fp64-size and the special prime is not (the uniform fp64 policy
// reads its source as one double).
static bool md_lift_is_fused(const tfhe_ctx* c, const ks_arg_t& A, const limb_sel_t& sl) {
    static const bool unfused = getenv("TFHE_MD_LIFT_UNFUSED") && getenv("TFHE_MD_LIFT_UNFUSED")[0] == '1';
    const int level = A.level;
    if (unfused || level > 32 || c->logN > 16) return false;
    if (c->logN >= 15) return c->variant == 0;
    const bool special_fp = c->limbs_host[A.w.idx[level]].Wd != nullptr;
    return special_fp || !sel_fp(c, sl, 0);
}
static int md_lift_fwd(tfhe_ctx* c, const ks_arg_t& A, const limb_sel_t& sl, const rescale_arg_t& ra, const u64* P, u64* LF, u64* U,
                       int64_t groups, bool* scaled) {
    const int level = A.level;
    const u32 n = (u32)c->N;
    const int64_t rows = groups * level;
    if (LF == nullptr) {   // the caller sized the workspace for the fused form (md_lift_is_fused)
        ntt_io_t io = io_plain();
        io.mode = 1; io.level = 1; io.nw = (u32)level; io.polys = 1; io.lift_unsigned = 1;
        *scaled = false;
        if (c->logN <= 14) return run_ntt(c, false, P, U, rows, sl, &io);
        const int x = c->logN - 14;
        const u32 tmask = mask_of(level, [&](int j) { return c->limbs_host[sl.idx[j]].Wd != nullptr; });   // fp64-size target limbs
        const u32 tall = mask_all(level);
        const bool special_fp = c->limbs_host[A.w.idx[level]].Wd != nullptr;
        if ((rows << x) > 0x7fffffffll || (((uintptr_t)P | (uintptr_t)U) & 15u) != 0) return fail(TFHE_E_BADARG, "md_lift_fwd: row count / alignment");
        int rc;
        void* tmp = nullptr;
        if (tmask != tall) {   // (before the fork: growing the workspace synchronises)
            rc = ensure_ws(c, (size_t)rows * c->N * 8, &tmp);
            if (rc) return rc;
        }
        lanes_t lanes(c, tmask != 0 && tmask != tall);   // both kinds of limbs: side by side
        if (tmask) {   // the fp64-size limbs: one kernel per row (pair) (ArithFpWide reads a source above 2^52 in two halves)
            ntt_io_t a = io;
            a.limb_mask = tmask == tall ? 0u : tmask;
            lanes.use(1);
            rc = run_ntt_large(c, false, P, U, rows, sl, a, &a, true, !special_fp);
            if (rc) return rc;
            lanes.use(0);
        }
        if (tmask != tall) {   // the larger limbs: lift fused into the top stages (into the transform scratch), then the u64 block kernels
            ntt_io_t w = io;
            w.limb_mask = tmask ? (tall & ~tmask) : 0u;
            const dim3 tg((unsigned)((((c->N >> x) + 255) / 256) * rows));
            if (x == 1) hipLaunchKernelGGL(k_ntt_fwd_top_lift<1>, tg, dim3(256), 0, c->stream, P, (u64*)tmp, c->limbs_dev, sl, c->logN, w);
            else hipLaunchKernelGGL(k_ntt_fwd_top_lift<2>, tg, dim3(256), 0, c->stream, P, (u64*)tmp, c->limbs_dev, sl, c->logN, w);
            HIP_TRY(hipGetLastError());
            ntt_io_t ib = io_plain();
            ib.limb_mask = w.limb_mask;
            rc = launch_block_fwd<ArithInt, 14>(c, (const u64*)tmp, U, rows, sl, x, ib);
            if (rc) return rc;
        }
        return TFHE_OK;
    }
    hipLaunchKernelGGL(k_md_lift, row_grid((unsigned)rows, (size_t)c->N), dim3(256), 0, c->stream, P, LF, c->limbs_dev, sl, ra, n);
    HIP_TRY(hipGetLastError());
    *scaled = true;
    return run_ntt(c, false, LF, U, rows, sl);
}

static int do_galois(tfhe_ctx* c, const u64* src, u64* dst, u64 g, int64_t rows, const limb_sel_t& sel);
// S_s = sum_i evk_{i,s} (.) digit_i over the working limbs (NTT domain), S: [batch][2][nw][N]
// keys != nullptr: `nkeys` keys against the same digits in one launch, key r writing S + r * batch*2*nw*N (tfhe_matmul_diag)
static int ks_inner_launch(tfhe_ctx* c, const ks_arg_t& A, int Lk, const u64* evk, const u64* dig, u64* S, int64_t batch,
                           const uint64_t* const* keys = nullptr, int nkeys = 0, const u64* epi_x = nullptr) {
    const int nw = A.nw;
    ks_keys_t K;
    memset(&K, 0, sizeof K);
    if (keys) {
        K.n = nkeys;
        K.s_stride = (size_t)batch * 2 * nw * (size_t)c->N;
        for (int r = 0; r < nkeys; r++) K.key[r] = keys[r];
        evk = keys[0];
        K.epi_x = epi_x;
    }
    const unsigned gy = keys ? (unsigned)nkeys : 1u;
    const u32 n = (u32)c->N;
    const unsigned gx = (n + 255) / 256;
    // enough workgroups to fill the chip: split the batch into slices (the key is re-read once per slice)
    // (several keys in one launch multiply the grid by gy)
    const unsigned bsplit = (unsigned)std::max<int64_t>(1, std::min<int64_t>(batch, (4096 + nw * gx * gy - 1) / (nw * gx * gy)));
    // working limbs below 2^52: the two-coefficient, carry-free kernel; the others: the generic one (more than 32 working
    // limbs: all or nothing, see mask_of)
    // several keys: groups of eight x-blocks (one per XCD), each followed by its keys (ks_multi_key_block)
    auto multi = [&](unsigned nx) { return keys ? dim3(((nx + 7u) / 8u) * 8u * gy) : dim3(nx); };
    u32 nmask = mask_of(nw, [&](int j) { return (c->limbs_host[A.w.idx[j]].q >> 52) == 0; });
    const u32 amask = mask_all(nw);
    if (n % 2 != 0) nmask = 0;
    lanes_t lanes(c, nmask != 0 && nmask != amask);   // both kernels: side by side (disjoint working limbs)
    if (nmask) {
        lanes.use(1);
        const unsigned gx2 = (n / 2 + 255) / 256;
        const unsigned bs2 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(batch, (4096 + nw * gx2 * gy - 1) / (nw * gx2 * gy)));
        auto kn = epi_x ? k_ks_inner_n2<8, true> : k_ks_inner_n2<8, false>;
        hipLaunchKernelGGL(kn, multi((unsigned)nw * gx2 * bs2), dim3(256), 0, c->stream, evk, dig, S, c->limbs_dev, A, Lk, n, (u32)batch, bs2,
                           nmask == amask ? 0u : nmask, K);
    }
    if (nmask != amask) {
        lanes.use(0);
        auto kg = epi_x ? k_ks_inner<8, true> : k_ks_inner<8, false>;
        hipLaunchKernelGGL(kg, multi((unsigned)nw * gx * bsplit), dim3(256), 0, c->stream, evk, dig, S, c->limbs_dev, A, Lk, n, (u32)batch, bsplit,
                           nmask ? (amask & ~nmask) : 0u, K);
    }
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
// Part B: S_s = sum_i evk_{i,s} (.) digit_i, inverse transforms, and the tail out_s = ct_s + ... (with the special prime: the
// ModulusRaised contraction).  `ct` only supplies the addends.  `tbuf` ([batch][2][nw][N]) receives the sub-block inverse at
// N = 2^16; the plain key switch passes the digit buffer (free by then).
// Hoisted rotations (g != 0): `evk` is the key of x -> x^g prepared by tfhe_galois_key_prepare (its NTT-domain rows permuted by
// g^-1), so the sums are those of the rotated digits up to that permutation: S' = sigma_g^-1(S).  The automorphism is applied
// to INTT(S') in the coefficient domain (a signed permutation) BEFORE the tail -- the ModulusRaised floor does not commute
// with sign changes -- into `tbuf`, which must not alias the digits (they are reused by the next rotation).
// whether ks_finish takes the N = 2^16 path (paired sub-block inverse + k_ks_top_tail<2>)
static bool ks_tail16(const tfhe_ctx* c, const ks_arg_t& A) {
    const u32 fpm = mask_of(A.nw, [&](int j) { return c->limbs_host[A.w.idx[j]].Wd != nullptr; });
    return c->logN == 16 && c->variant == 0 && fpm != 0 && A.level >= 2;
}
// g_tail != 0: a rotation finished in the tail (k_ks_top_tail_rot<2>; N = 2^16 path only): `evk` is the prepared key, `ct` the
// UNrotated input
// `outer`: the caller's forked region (ks_chunk) -- the steps up to the inverse sub-block transforms stay on their lanes, the
// join comes before the tail, which reads every limb
static int ks_finish(tfhe_ctx* c, const ks_arg_t& A, int Lk, const u64* evk, const u64* ct, u64* out, int64_t batch, u64* S,
                     const u64* dig, u64* tbuf, u64 g, u64 g_tail = 0, lanes_t* outer = nullptr) {
    const int level = A.level, nw = A.nw, polys = A.polys, special = A.special;
    const u32 n = (u32)c->N;
    const u32 add_s = polys == 3 ? 2u : 1u;  // c2 starts from zero for a 2-element input (rlwe_she.jl:324)
    int rc;
    const unsigned gx = (n + 255) / 256;
    rc = ks_inner_launch(c, A, Lk, evk, dig, S, batch);
    if (rc) return rc;
    const u32 amask = mask_all(nw);
    (void)gx;
    if (g != 0) {  // hoisted rotation: INTT, automorphism, tail
        if (outer) outer->join();
        rc = run_ntt(c, true, S, S, batch * 2 * nw, A.w);
        if (rc) return rc;
        rc = do_galois(c, S, tbuf, g, batch * 2 * nw, A.w);
        if (rc) return rc;
        if (special) {
            rescale_arg_t ra;
            memset(&ra, 0, sizeof ra);
            const u64 P = c->q[Lk - 1];
            for (int j = 0; j < level; j++) ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
            hipLaunchKernelGGL(k_ks_rescale_add, row_grid((unsigned)(batch * 2 * level), (size_t)c->N), dim3(256), 0, c->stream, tbuf, ct, out, c->limbs_dev, A, ra, n, add_s);
        } else {
            HIP_TRY(hipMemcpyAsync(out, tbuf, (size_t)batch * 2 * level * c->N * 8, hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(k_ks_add_ct, row_grid((unsigned)(batch * 2 * level), (size_t)c->N), dim3(256), 0, c->stream, ct, out, c->limbs_dev, A, n, add_s);
        }
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    const u32 fpm = mask_of(nw, [&](int j) { return c->limbs_host[A.w.idx[j]].Wd != nullptr; });
    const bool tail16 = ks_tail16(c, A) && (((uintptr_t)S | (uintptr_t)tbuf) & 15u) == 0;
    if (g_tail && !tail16) return fail(TFHE_E_UNSUPPORTED, "internal: rotation in the tail needs the N = 2^16 sub-block path");
    if (tail16) {
        // N = 2^16: the paired sub-block inverse into the (now free) digit buffer, then the two inverse top stages together with
        // the tail (k_ks_top_tail<2>) instead of k_ntt_inv_top<2> + k_ks_rescale_add / k_ks_add_ct.  Rings of mixed modulus sizes:
        // the larger limbs' sub-blocks come from the u64 block kernel (same sub-block layout, lazy [0, 2q) outputs).
        {
            lanes_t lanes(c, fpm != amask);
            lanes.use(1);
            rc = launch_subpair<ArithFp>(c, true, S, tbuf, batch * 2 * nw, A.w, 2, fpm == amask ? 0u : fpm);
            if (rc) return rc;
            if (fpm != amask) {
                lanes.use(0);
                ntt_io_t iw = io_plain();
                iw.limb_mask = amask & ~fpm;
                rc = launch_block_inv<ArithInt, 14>(c, S, tbuf, batch * 2 * nw, A.w, 2, iw);
                if (rc) return rc;
            }
        }
        if (outer) outer->join();
        rescale_arg_t ra;
        memset(&ra, 0, sizeof ra);
        if (special) {
            const u64 P = c->q[Lk - 1];
            for (int j = 0; j < level; j++) ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
        }
        if (g_tail) hipLaunchKernelGGL(k_ks_top_tail_rot<2>, dim3(8 * TFHE_ROT_TAIL_SLOTS), dim3(256), 0, c->stream, tbuf, ct, out, c->limbs_dev, A, ra, n, add_s, g_tail, (u32)(batch * 2));
        else hipLaunchKernelGGL(k_ks_top_tail<2>, row_grid((unsigned)(batch * 2 * level), (size_t)c->N / 4), dim3(256), 0, c->stream, tbuf, ct, out, c->limbs_dev, A, ra, n, add_s);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    if (outer) outer->join();
    if (special) {
        rc = run_ntt(c, true, S, S, batch * 2 * nw, A.w);
        if (rc) return rc;
        rescale_arg_t ra;
        memset(&ra, 0, sizeof ra);
        const u64 P = c->q[Lk - 1];
        for (int j = 0; j < level; j++) ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
        hipLaunchKernelGGL(k_ks_rescale_add, row_grid((unsigned)(batch * 2 * level), (size_t)c->N), dim3(256), 0, c->stream, S, ct, out, c->limbs_dev, A, ra, n, add_s);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    if (c->logN <= 14) {  // out = c + INTT(S), the addition fused into the inverse NTT's stores
        ntt_io_t io = io_plain();
        io.mode = 2; io.gsz = (u32)(2 * level); io.src_gstride = io.gsz; io.dst_gstride = io.gsz;
        io.add_rows = add_s * (u32)level; io.add_gstride = (u32)(polys * level); io.addend = ct;
        return run_ntt(c, true, S, out, batch * 2 * nw, A.w, &io);
    }
    rc = run_ntt(c, true, S, out, batch * 2 * nw, A.w);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ks_add_ct, row_grid((unsigned)(batch * 2 * level), (size_t)c->N), dim3(256), 0, c->stream, ct, out, c->limbs_dev, A, n, add_s);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

// g_tail != 0: a rotation finished in the tail (k_ks_top_tail_rot) -- `ct` is the UNrotated input and `evd` was prepared for g
static int ks_chunk(tfhe_ctx* c, int Lk, int level, int special, const u64* evk, const u64* ct, int polys, u64* out, int64_t batch,
                    u64* S, u64* dig, const u64* evd, bool prelifted, u64 g_tail = 0) {
    const int nw = special ? level + 1 : level;
    ks_arg_t A;
    memset(&A, 0, sizeof A);
    A.level = level; A.nw = nw; A.special = special; A.polys = polys;
    A.w.n = nw;
    for (int j = 0; j < level; j++) A.w.idx[j] = j;
    if (special) A.w.idx[level] = Lk - 1;  // downswitch_keyelement, modulusraising.jl:43-49
    const u32 n = (u32)c->N;
    const u32 add_s = polys == 3 ? 2u : 1u;  // c2 starts from zero for a 2-element input (rlwe_she.jl:324)
    int rc;
    if (evd && (c->logN == 14 || c->logN == 13)) {  // ks_fused14
        // everything in one kernel: digit lift, forward transforms, key inner product and the two inverse transforms;
        // with the special prime the transformed sums go to S and the contraction kernel finishes (modulusraising.jl:42)
        const unsigned items = (unsigned)(batch * nw);
        // Special prime: two launches -- the special limb of every ciphertext first (its coefficient rows t_P into S), then the
        // ciphertext limbs with the ModulusRaised contraction and "+ c" in their final store (k_ks_fused SPMODE 1 / 2, ArithFpMD).
        // No T, no tail kernel.  TFHE_KS_TAIL=1 keeps the one-launch form with k_ks_rescale_add (comparisons).
        static const bool ks_tail = getenv("TFHE_KS_TAIL") && getenv("TFHE_KS_TAIL")[0] == '1';
        if (special && !prelifted && !ks_tail) {
            {
                std::lock_guard<std::mutex> zg(c->zero_mu);         // a failed attempt (transient out-of-memory) is retried by the next call
                if (!c->zero_row) {
                    void* z = nullptr;
                    if (devalloc::malloc_retry(&z, (size_t)c->N * 8) == hipSuccess) {
                        if (hipMemset(z, 0, (size_t)c->N * 8) == hipSuccess) c->zero_row = (u64*)z;
                        else { (void)hipGetLastError(); (void)hipFree(z); }
                    } else (void)hipGetLastError();
                }
                if (!c->zero_row) return fail(TFHE_E_NOMEM, "allocating the zero row failed");
            }
            const u64 P = c->q[Lk - 1];
            for (int j = 0; j < level; j++) A.pinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
            const unsigned it1 = (unsigned)batch, it2 = (unsigned)(batch * level);
            auto launch2 = [&](auto k1, auto k2, int LOGT, size_t lds, unsigned per_cu, bool& attr_set) -> int {
                if (!attr_set) {
                    int r2 = set_lds(k1, lds);
                    if (!r2) r2 = set_lds(k2, lds);
                    if (r2) return r2;
                    attr_set = true;
                }
                prof_begin(c, (int64_t)items * (level + 2));
                hipLaunchKernelGGL(k1, dim3(std::min(it1, per_cu * (unsigned)c->num_cus)), dim3(1u << LOGT), lds, c->stream, evd, ct, S, c->limbs_dev, A, Lk, it1,
                                   (const u64*)nullptr, (const u64*)nullptr);
                hipLaunchKernelGGL(k2, dim3(std::min(it2, per_cu * (unsigned)c->num_cus)), dim3(1u << LOGT), lds, c->stream, evd, ct, out, c->limbs_dev, A, Lk, it2,
                                   (const u64*)S, (const u64*)c->zero_row);
                prof_end(c);
                return TFHE_OK;
            };
            if (c->logN == 14 && g_tail) {   // rotation finished in the second launch's stores (SPMODE 3; `ct` is the unrotated input)
                constexpr int LOGT = logt_for(14);
                static bool a14r = false;
                A.rot_g = (u32)(g_tail & (2 * (u64)c->N - 1));
                rc = launch2(k_ks_fused<ArithFp, 14, LOGT, false, 1>, k_ks_fused<ArithFp, 14, LOGT, false, 3>, LOGT, fused_lds_bytes<14, LOGT, TFHE_TWL_KS>(), 1u, a14r);
            } else if (g_tail) {
                return fail(TFHE_E_UNSUPPORTED, "internal: rotation in the store is an N = 2^14 path");
            } else if (c->logN == 14) {
                constexpr int LOGT = logt_for(14);
                static bool a14 = false;
                rc = launch2(k_ks_fused<ArithFp, 14, LOGT, false, 1>, k_ks_fused<ArithFp, 14, LOGT, false, 2>, LOGT, fused_lds_bytes<14, LOGT, TFHE_TWL_KS>(), 1u, a14);
            } else {
                constexpr int LOGT = logt_for(13);
                static bool a13 = false;
                rc = launch2(k_ks_fused<ArithFp, 13, LOGT, false, 1>, k_ks_fused<ArithFp, 13, LOGT, false, 2>, LOGT, fused_lds_bytes<13, LOGT, TFHE_TWL_KS>(), 2u, a13);
            }
            if (rc) return rc;
            HIP_TRY(hipGetLastError());
            return TFHE_OK;
        }
        if (g_tail) return fail(TFHE_E_UNSUPPORTED, "internal: rotation in the store needs the two-launch special-prime form");
        if (c->logN == 14) {
            constexpr int LOGT = logt_for(14);
            const size_t lds = fused_lds_bytes<14, LOGT, TFHE_TWL_KS>();
            auto fk = prelifted ? k_ks_fused<ArithFp, 14, LOGT, true> : k_ks_fused<ArithFp, 14, LOGT, false>;
            static bool fattr_set = false;
            if (!fattr_set) {
                rc = set_lds(k_ks_fused<ArithFp, 14, LOGT, true>, lds);
                if (!rc) rc = set_lds(k_ks_fused<ArithFp, 14, LOGT, false>, lds);
                if (rc) return rc;
                fattr_set = true;
            }
            const unsigned grid = std::min(items, (unsigned)c->num_cus * TFHE_GRID_MULT_KS);
            prof_begin(c, (int64_t)items * (level + 2));  // limb transforms inside this launch: `level` forward + 2 inverse per item
            hipLaunchKernelGGL(fk, dim3(grid), dim3(1 << LOGT), lds, c->stream, evd, ct, special ? S : out, c->limbs_dev, A, Lk, items, (const u64*)nullptr, (const u64*)nullptr);
            prof_end(c);
        } else {  // N = 2^13: 256 threads x 32 elements, 65 KiB of LDS; 478 registers per thread, so one workgroup per CU is resident
                  // (capped to two resident workgroups it measured 5 % slower)
            constexpr int LOGT = logt_for(13);
            const size_t lds = fused_lds_bytes<13, LOGT, TFHE_TWL_KS>();
            auto fk = k_ks_fused<ArithFp, 13, LOGT, false>;
            static bool fattr13_set = false;
            if (!fattr13_set) { rc = set_lds(fk, lds); if (rc) return rc; fattr13_set = true; }
            const unsigned grid = std::min(items, 2u * (unsigned)c->num_cus);
            prof_begin(c, (int64_t)items * (level + 2));
            hipLaunchKernelGGL(fk, dim3(grid), dim3(1 << LOGT), lds, c->stream, evd, ct, special ? S : out, c->limbs_dev, A, Lk, items, (const u64*)nullptr, (const u64*)nullptr);
            prof_end(c);
        }
        HIP_TRY(hipGetLastError());
        if (!special) return TFHE_OK;
        rescale_arg_t ra;
        memset(&ra, 0, sizeof ra);
        const u64 P = c->q[Lk - 1];
        for (int j = 0; j < level; j++) ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
        hipLaunchKernelGGL(k_ks_rescale_add, row_grid((unsigned)(batch * 2 * level), (size_t)c->N), dim3(256), 0, c->stream, S, ct, out, c->limbs_dev, A, ra, n, add_s);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    if (evd && (c->logN == 15 || c->logN == 16)) {  // ks_fused14: variant 0, fp64 policy, level >= 2
        // N = 2^15 / 2^16 on fp64-size moduli: per-sub-block fused key switch (k_ks_fused_sub) into T = dig ([batch][2][nw] rows,
        // level >= 2 makes room), then the inverse top stages over T together with the tail.  At 2^16 (X = 2) the four quarters
        // of a digit row stream through the LDS (dma_stream_load, no spills: 33.0 k against 32.1 k key switches/s for the
        // three-kernel path on 7 x 50 bit; with register loads the X = 2 load phase either spilled 237 registers or ran as 32
        // serial rounds: 26-28 k).  At 2^15 the register loads stay (the streamed form measured 36.2 k against 40.2 k, cfg#3).
        // TFHE_FUSED16=0 keeps the three-kernel path at 2^16.
        constexpr int LOGT = logt_for(14);
        const size_t lds = (size_t)lds_words<14, LOGT>() * 8;
        const int x = c->logN - 14;
        // moduli below 2^42 (the 40-bit chains of the reference's CKKS rings): the range plan of that size class -- no forward
        // sweeps, no reduced product operands, one sweep per inverse (ArithFpS): cfg#3 36.5 k -> 38.1 k key switches/s
        bool small = true;
        for (int j = 0; j < nw; j++) small = small && c->q[A.w.idx[j]] < TFHE_FPS_QMAX;
        static const bool fps_on = !(getenv("TFHE_FPS") && getenv("TFHE_FPS")[0] == '0');
        auto fk = x == 2 ? ((small && fps_on) ? k_ks_fused_sub<ArithFpS, 14, LOGT, 2> : k_ks_fused_sub<ArithFp, 14, LOGT, 2>)
                         : ((small && fps_on) ? k_ks_fused_sub<ArithFpS, 14, LOGT, 1> : k_ks_fused_sub<ArithFp, 14, LOGT, 1>);
        static bool sattr_set = false;
        if (!sattr_set) {
            rc = set_lds(k_ks_fused_sub<ArithFp, 14, LOGT, 1>, lds);
            if (!rc) rc = set_lds(k_ks_fused_sub<ArithFpS, 14, LOGT, 1>, lds);
            if (!rc) rc = set_lds(k_ks_fused_sub<ArithFp, 14, LOGT, 2>, lds);
            if (!rc) rc = set_lds(k_ks_fused_sub<ArithFpS, 14, LOGT, 2>, lds);
            if (rc) return rc;
            sattr_set = true;
        }
        const unsigned items = (unsigned)((batch * nw) << x);
        const unsigned grid = std::min(items, (unsigned)c->num_cus);
        prof_begin(c, (int64_t)batch * nw * (level + 2));
        hipLaunchKernelGGL(fk, dim3(grid), dim3(1 << LOGT), lds, c->stream, evd, ct, dig, c->limbs_dev, A, Lk, items);
        prof_end(c);
        HIP_TRY(hipGetLastError());
        // inverse top stage + "+ c" / special-prime contraction in one pass over the sub-block results
        rescale_arg_t ra;
        memset(&ra, 0, sizeof ra);
        if (special) {
            const u64 P = c->q[Lk - 1];
            for (int j = 0; j < level; j++) ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
        }
        if (g_tail) {   // rotation: the automorphism rides on the tail's stores (XCD-cooperative scatter)
            const dim3 rg(8 * TFHE_ROT_TAIL_SLOTS);
            if (x == 1) hipLaunchKernelGGL(k_ks_top_tail_rot<1>, rg, dim3(256), 0, c->stream, dig, ct, out, c->limbs_dev, A, ra, n, add_s, g_tail, (u32)(batch * 2));
            else hipLaunchKernelGGL(k_ks_top_tail_rot<2>, rg, dim3(256), 0, c->stream, dig, ct, out, c->limbs_dev, A, ra, n, add_s, g_tail, (u32)(batch * 2));
        } else if (x == 1) hipLaunchKernelGGL(k_ks_top_tail<1>, row_grid((unsigned)(batch * 2 * level), (size_t)c->N / 2), dim3(256), 0, c->stream, dig, ct, out, c->limbs_dev, A, ra, n, add_s);
        else hipLaunchKernelGGL(k_ks_top_tail<2>, row_grid((unsigned)(batch * 2 * level), (size_t)c->N / 4), dim3(256), 0, c->stream, dig, ct, out, c->limbs_dev, A, ra, n, add_s);
        HIP_TRY(hipGetLastError());
        return TFHE_OK;
    }
    // Rings of mixed modulus sizes at N = 2^16 (the three-kernel path): when the digit transforms, the key sums and the inverse
    // sub-block transforms all split the working limbs the same way (fp64-size = "narrow"), the two policies run as two lanes
    // from the lift to the inverse with ONE join before the tail; otherwise every step forks and joins by itself.
    const u32 fpm = mask_of(nw, [&](int j) { return c->limbs_host[A.w.idx[j]].Wd != nullptr; });
    const u32 nrw = mask_of(nw, [&](int j) { return (c->limbs_host[A.w.idx[j]].q >> 52) == 0; });
    const bool two_lanes = nw <= 32 && c->variant == 0 && c->logN == 16 && fpm != 0 && fpm != mask_all(nw) && fpm == nrw && ks_tail16(c, A) &&
                           (((uintptr_t)ct | (uintptr_t)dig | (uintptr_t)S) & 15u) == 0;
    if (two_lanes) {
        void* tmp = nullptr;
        rc = ensure_ws(c, (size_t)batch * level * nw * c->N * 8, &tmp);   // (no-op under keyswitch_impl; before the fork)
        if (rc) return rc;
    }
    lanes_t outer(c, two_lanes);
    rc = ks_digits_fwd(c, A, ct, dig, batch);
    if (rc) return rc;
    return ks_finish(c, A, Lk, evk, ct, out, batch, S, dig, dig, 0, g_tail, &outer);
}

static int ks_check(tfhe_ctx* c, int Lk, int level, int special, const void* evk, int n_digits, const void* ct, int polys, const void* out, int64_t batch) {
    if (!c || !evk || !ct || !out) return fail(TFHE_E_BADARG, "null argument");
    if (polys != 2 && polys != 3) return fail(TFHE_E_BADARG, "keyswitch needs a 2- or 3-element ciphertext (rlwe_she.jl:318), got %d", polys);
    if (Lk < 1 || Lk > c->L) return fail(TFHE_E_LEVEL_MISMATCH, "key_limbs=%d outside [1,%d]", Lk, c->L);
    const int maxlevel = special ? Lk - 1 : Lk;
    if (level < 1 || level > maxlevel) return fail(TFHE_E_LEVEL_MISMATCH, "level=%d outside [1,%d]", level, maxlevel);
    if (n_digits < level) return fail(TFHE_E_PARAMS_MISMATCH, "evaluation key has %d components, level %d needs %d", n_digits, level, level);
    if (batch < 0) return fail(TFHE_E_BADARG, "negative batch");
    return TFHE_OK;
}

static u64 inv_mod_2n(u64 g, u64 twoN);
// key_prepared (rotations only, tfhe_rotate_prepared): `evk` is the output of tfhe_galois_key_prepare for `galois` -- the paths that
// finish the rotation in their tail consume it as it is (no per-call k_ntt_perm / permuting conversion); the others take the
// hoisted form of tfhe_rotate_many, which is defined on prepared keys (same bits)
static int keyswitch_impl(tfhe_ctx* c, int Lk, int level, int special, const u64* evk, const u64* ct, int polys, u64* out, int64_t batch,
                          u64 galois, bool rotate, bool prelifted = false, bool key_prepared = false) {
    if (prelifted && (rotate || special || !ks_prelift_ok(c, level))) return fail(TFHE_E_UNSUPPORTED, "internal: pre-lifted rows need the fused key switch");
    const int nw = special ? level + 1 : level;
    const size_t N = (size_t)c->N;
    const bool f14 = ks_fused14(c, Lk, level, special);
    // Rotations through the sub-block fused key switch (N = 2^15 / 2^16): no rotated copy of the input -- the key is prepared while
    // it is converted (k_evk_to_f64), the key sums are those of the unrotated digits, the automorphism rides on the tail's stores
    // (k_ks_top_tail_rot).  Same bits (the hoisting identity of tfhe_rotate_many).  TFHE_ROT_TAIL=0 keeps the separate pass.
    static const bool rot_tail_on = !(getenv("TFHE_ROT_TAIL") && getenv("TFHE_ROT_TAIL")[0] == '0');
    // The tail scatters into `out` while other workgroups still read the unrotated `ct`: the two RANGES must be disjoint (r05, ADVICE
    // r04: pointer inequality let a partially overlapping out / ct race) -- otherwise the rotated-copy path below runs.
    const bool io_disjoint = (const char*)(out + (size_t)batch * 2 * level * N) <= (const char*)ct ||
                             (const char*)(ct + (size_t)batch * polys * level * N) <= (const char*)out;
    bool rot_in_tail = rotate && f14 && c->logN >= 15 && rot_tail_on && io_disjoint && batch >= 8;   // (one (ciphertext, component) per XCD at a time)
    // N = 2^14 with the special prime (the two-launch fused key switch): the rotation rides on the in-kernel contraction's stores
    // (k_ks_fused SPMODE 3) -- each workgroup scatters the row it owns
    static const bool ks_tail_env = getenv("TFHE_KS_TAIL") && getenv("TFHE_KS_TAIL")[0] == '1';
    if (rotate && f14 && c->logN == 14 && special && !prelifted && !ks_tail_env && rot_tail_on && io_disjoint) rot_in_tail = true;
    // the same for the N = 2^16 three-kernel path (rings with moduli beyond the fp64 size: the reference's CKKS ring): the key is
    // prepared into the workspace (k_ntt_perm), the rotation rides on k_ks_top_tail_rot<2>
    bool rot_key_prep = false;
    if (rotate && !f14 && rot_tail_on && io_disjoint && batch >= 8) {
        ks_arg_t KA;
        memset(&KA, 0, sizeof KA);
        KA.level = level; KA.nw = nw;
        KA.w.n = nw;
        for (int j = 0; j < level; j++) KA.w.idx[j] = j;
        if (special) KA.w.idx[level] = Lk - 1;
        if (ks_tail16(c, KA)) rot_in_tail = rot_key_prep = true;
    }
    if (key_prepared && !rot_in_tail) {
        const uint64_t* one[1] = {evk};
        return tfhe_rotate_many(c, Lk, level, special, one, level, 1, &galois, 1, ct, out, batch);
    }
    if (key_prepared) rot_key_prep = false;   // the three-kernel path reads the prepared key where it lies
    // chunk the batch so that the digit tensor stays at a few GiB.  The fused key switches (ks_fused14) never write the digit
    // rows: at N = 2^15 the "digit" buffer only carries the sub-block sums T (2 nw rows per ciphertext), so a whole batch is one
    // launch (cfg#3, 512 ciphertexts: 248 + 248 + 16 before -- the 16 ran as two nearly empty item rounds of k_ks_fused_sub)
    const size_t dig_rows = f14 ? (size_t)2 * nw : (size_t)level * nw;
    const size_t per_ct = ((size_t)2 * nw + dig_rows + (rotate ? (size_t)polys * level : 0)) * N * 8;
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>({batch, (int64_t)512, (int64_t)((8192ull << 20) / per_ct)}));
    void* ws = nullptr;
    // NTT of N > 2^14 uses the context workspace as well: keep ours separate by over-allocating (the fused paths run no
    // stand-alone transform)
    const size_t ntt_tmp = (c->logN > 14 && !f14) ? (size_t)chunk * std::max(2, level) * nw * N * 8 : 0;
    const size_t evd_bytes = f14 ? (size_t)level * 2 * nw * N * 8                  // the key rows of this call as doubles
                                 : (rot_key_prep ? (size_t)level * 2 * Lk * N * 8 : 0);   // ... or the prepared key of a rotation
    int rc = ensure_ws(c, ntt_tmp + chunk * per_ct + evd_bytes, &ws);
    if (rc) return rc;
    u64* base = (u64*)((char*)ws + ntt_tmp);
    u64* acc = base;
    u64* dig = acc + (size_t)chunk * 2 * nw * N;
    u64* rot = dig + (size_t)chunk * dig_rows * N;
    u64* evd = nullptr;
    if (f14) {
        evd = (u64*)((char*)ws + ntt_tmp + chunk * per_ct);
        ks_arg_t KA;
        memset(&KA, 0, sizeof KA);
        KA.level = level; KA.nw = nw; KA.special = special; KA.polys = polys;
        KA.w.n = nw;
        for (int j = 0; j < level; j++) KA.w.idx[j] = j;
        if (special) KA.w.idx[level] = Lk - 1;
        hipLaunchKernelGGL(k_evk_to_f64, row_grid((unsigned)(level * 2 * nw), N), dim3(256), 0, c->stream, evk, evd, c->limbs_dev, KA, Lk, (u32)N, c->logN <= 14 ? 1 : 0, c->logN == 16 ? 2 : 0,
                           (rot_in_tail && !key_prepared) ? inv_mod_2n(galois, 2 * (u64)c->N) : (u64)0, 1);
        HIP_TRY(hipGetLastError());
    }
    if (rot_key_prep) {
        u64* keyp = (u64*)((char*)ws + ntt_tmp + chunk * per_ct);
        hipLaunchKernelGGL(k_ntt_perm, row_grid((unsigned)(level * 2 * Lk), N), dim3(256), 0, c->stream, evk, keyp, inv_mod_2n(galois, 2 * (u64)c->N), (u32)N);
        HIP_TRY(hipGetLastError());
        evk = keyp;
    }
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = std::min(chunk, batch - b0);
        const u64* cin = ct + (size_t)b0 * polys * level * N;
        if (rotate && !rot_in_tail) {
            limb_sel_t s;
            s.n = level;
            for (int j = 0; j < level; j++) s.idx[j] = j;
            rc = do_galois(c, cin, rot, galois, nb * polys * level, s);
            if (rc) return rc;
            cin = rot;
        }
        rc = ks_chunk(c, Lk, level, special, evk, cin, polys, out + (size_t)b0 * 2 * level * N, nb, acc, dig, evd, prelifted, rot_in_tail ? galois : (u64)0);
        if (rc) return rc;
    }
    return TFHE_OK;
}

int tfhe_keyswitch(tfhe_ctx* c, int Lk, int level, int special, const uint64_t* evk, int n_digits, const uint64_t* ct, int polys, uint64_t* out, int64_t batch) {
    int rc = ks_check(c, Lk, level, special, evk, n_digits, ct, polys, out, batch);
    if (rc) return rc;
    return keyswitch_impl(c, Lk, level, special, evk, ct, polys, out, batch, 0, false);
}
// Hoisted rotations: rotate(gk_r, c) for r < n_rot from ONE digit decomposition.  The centred RNS digits commute with the
// automorphism (a signed permutation of coefficients; the centred representative of -x is the negative of that of x, q odd),
// and in the NTT domain the automorphism is the permutation galois_ntt_pos.  With the key rows permuted by g^-1 once per key
// (tfhe_galois_key_prepare) the inner product runs coalesced on the transformed digits of the UNrotated ciphertext and yields
// sigma_g^-1 of the sums; the automorphism is applied after the inverse transform, over 2 nw rows instead of level nw:
// level * nw forward transforms once instead of per rotation.  Bit-identical to n_rot calls of tfhe_rotate.
static u64 inv_mod_2n(u64 g, u64 twoN) {  // g odd, twoN a power of two: Newton iteration
    u64 x = g;
    for (int i = 0; i < 6; i++) x *= 2 - g * x;
    return x & (twoN - 1);
}
// evk_out = the Galois key of x -> x^g with every NTT-domain row permuted by g^-1 (tfhe_rotate_many with prepared = 1)
int tfhe_galois_key_prepare(tfhe_ctx* c, int Lk, int n_digits, uint64_t g, const uint64_t* evk, uint64_t* evk_out) {
    if (!c || !evk || !evk_out || evk == evk_out) return fail(TFHE_E_BADARG, "null or aliased argument");
    if (Lk < 1 || Lk > c->L || n_digits < 1) return fail(TFHE_E_LEVEL_MISMATCH, "key shape outside the ring");
    if ((g & 1) == 0 || g >= 2 * (u64)c->N) return fail(TFHE_E_BADARG, "galois element must be odd and below 2N");
    const u64 ginv = inv_mod_2n(g, 2 * (u64)c->N);
    hipLaunchKernelGGL(k_ntt_perm, row_grid((unsigned)(n_digits * 2 * Lk), (size_t)c->N), dim3(256), 0, c->stream, evk, evk_out, ginv, (u32)c->N);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

int tfhe_rotate_many(tfhe_ctx* c, int Lk, int level, int special, const uint64_t* const* evks, int n_digits, int prepared,
                     const uint64_t* galois, int n_rot, const uint64_t* ct, uint64_t* out, int64_t batch) {
    if (!evks || !galois || n_rot < 0) return fail(TFHE_E_BADARG, "null argument");
    if (n_rot == 0) return TFHE_OK;
    for (int r = 0; r < n_rot; r++) {
        int rc = ks_check(c, Lk, level, special, evks[r], n_digits, ct, 2, out, batch);
        if (rc) return rc;
        if ((galois[r] & 1) == 0 || galois[r] >= 2 * (u64)c->N) return fail(TFHE_E_BADARG, "galois element must be odd and below 2N");
    }
    const int nw = special ? level + 1 : level, polys = 2;
    const size_t N = (size_t)c->N;
    if (!prepared && c->logN <= 14 && ks_fused14(c, Lk, level, special)) {
        // N = 2^14 on fp64-size moduli: the fused key switch (digits never leave the registers) beats the hoisted three-kernel
        // path (measured 126 k against 110 k rotations/s at 6 limbs + special prime) -- same bits either way
        for (int r = 0; r < n_rot; r++) {
            int rc = keyswitch_impl(c, Lk, level, special, evks[r], ct, 2, out + (size_t)r * batch * 2 * level * N, batch, galois[r], true);
            if (rc) return rc;
        }
        return TFHE_OK;
    }
    ks_arg_t A;
    memset(&A, 0, sizeof A);
    A.level = level; A.nw = nw; A.special = special; A.polys = polys;
    A.w.n = nw;
    for (int j = 0; j < level; j++) A.w.idx[j] = j;
    if (special) A.w.idx[level] = Lk - 1;
    // workspace per ciphertext: S (2 nw rows) + digits (level nw) + T (2 nw) + rotated input (2 level); + one prepared key
    const size_t per_ct = ((size_t)4 * nw + (size_t)level * nw + (size_t)polys * level) * N * 8;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>({batch, (int64_t)512, (int64_t)((8192ull << 20) / per_ct)}));
    const size_t ntt_tmp = c->logN > 14 ? (size_t)chunk * std::max(2, level) * nw * N * 8 : 0;
    const size_t key_bytes = prepared ? 0 : (size_t)n_digits * 2 * Lk * N * 8;
    void* ws = nullptr;
    int rc = ensure_ws(c, ntt_tmp + chunk * per_ct + key_bytes, &ws);
    if (rc) return rc;
    u64* S = (u64*)((char*)ws + ntt_tmp);
    u64* dig = S + (size_t)chunk * 2 * nw * N;
    u64* T = dig + (size_t)chunk * level * nw * N;
    u64* rot = T + (size_t)chunk * 2 * nw * N;
    u64* keytmp = rot + (size_t)chunk * polys * level * N;
    limb_sel_t sl;
    sl.n = level;
    for (int j = 0; j < level; j++) sl.idx[j] = j;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = std::min(chunk, batch - b0);
        const u64* cin = ct + (size_t)b0 * polys * level * N;
        rc = ks_digits_fwd(c, A, cin, dig, nb);
        if (rc) return rc;
        for (int r = 0; r < n_rot; r++) {
            const u64* key = evks[r];
            if (!prepared) {
                rc = tfhe_galois_key_prepare(c, Lk, n_digits, galois[r], evks[r], keytmp);
                if (rc) return rc;
                key = keytmp;
            }
            rc = do_galois(c, cin, rot, galois[r], nb * polys * level, sl);   // the addend sigma_g(c_1) (and sigma_g(c_2), unused)
            if (rc) return rc;
            rc = ks_finish(c, A, Lk, key, rot, out + ((size_t)r * batch + b0) * 2 * level * N, nb, S, dig, T, galois[r]);
            if (rc) return rc;
        }
    }
    return TFHE_OK;
}

int tfhe_rotate(tfhe_ctx* c, int Lk, int level, int special, const uint64_t* evk, int n_digits, uint64_t g, const uint64_t* ct, uint64_t* out, int64_t batch) {
    int rc = ks_check(c, Lk, level, special, evk, n_digits, ct, 2, out, batch);
    if (rc) return rc;
    if ((g & 1) == 0) return fail(TFHE_E_BADARG, "galois element must be odd");
    return keyswitch_impl(c, Lk, level, special, evk, ct, 2, out, batch, g, true);
}
// the same with the key as tfhe_galois_key_prepare left it (r06): a caller that rotates by ONE Galois element again and again
// (infer.jl:140-149: 63 chained rotations per matrix product, one key) prepares it once instead of once per call
int tfhe_rotate_prepared(tfhe_ctx* c, int Lk, int level, int special, const uint64_t* evk_prepared, int n_digits, uint64_t g, const uint64_t* ct, uint64_t* out,
                         int64_t batch) {
    int rc = ks_check(c, Lk, level, special, evk_prepared, n_digits, ct, 2, out, batch);
    if (rc) return rc;
    if ((g & 1) == 0 || g >= 2 * (u64)c->N) return fail(TFHE_E_BADARG, "galois element must be odd and below 2N");
    return keyswitch_impl(c, Lk, level, special, evk_prepared, ct, 2, out, batch, g, true, false, true);
}

// ---- diagonal matrix-vector product (infer.jl:140-149, test/ckks_matmul.jl:33-41) in one call ---------------------------
// out = diag_0 (.) c + sum_r diag_{r+1} (.) rotate(gk_r, c): the hoisted rotations of tfhe_rotate_many (one digit decomposition
// of c) with every step -- key sums, inverse transforms, automorphism + tail, forward transforms, accumulation -- run over ALL
// rotations at once instead of rotation by rotation (R x more rows per launch: about 15 launches per product instead of
// about 8 R + 2 R + 2), and the rotated ciphertexts never return to the caller.  Same arithmetic, term by term, as
// rotate_many -> nntt -> dot: bit-identical results.
int tfhe_matmul_diag(tfhe_ctx* c, int Lk, int level, int special, const uint64_t* const* evks, int n_digits, const uint64_t* galois,
                     int n_rot, const uint64_t* diags, const uint64_t* ct, uint64_t* out, int64_t batch) {
    if (!evks || !galois || !diags || n_rot < 0) return fail(TFHE_E_BADARG, "null argument");
    if (n_rot > TFHE_DOT_MAX) return fail(TFHE_E_UNSUPPORTED, "tfhe_matmul_diag takes at most %d rotations per call", TFHE_DOT_MAX);
    for (int r = 0; r < n_rot; r++) {
        int rc = ks_check(c, Lk, level, special, evks[r], n_digits, ct, 2, out, batch);
        if (rc) return rc;
        if ((galois[r] & 1) == 0 || galois[r] >= 2 * (u64)c->N) return fail(TFHE_E_BADARG, "galois element must be odd and below 2N");
    }
    if (n_rot == 0) {
        int rc = ks_check(c, Lk, level, special, diags, n_digits, ct, 2, out, batch);
        if (rc) return rc;
    }
    if (batch == 0) return TFHE_OK;
    const int nw = special ? level + 1 : level, polys = 2, R = n_rot;
    const size_t N = (size_t)c->N;
    const u32 n = (u32)c->N;
    ks_arg_t A;
    memset(&A, 0, sizeof A);
    A.level = level; A.nw = nw; A.special = special; A.polys = polys;
    A.w.n = nw;
    for (int j = 0; j < level; j++) A.w.idx[j] = j;
    if (special) A.w.idx[level] = Lk - 1;
    limb_sel_t sl;
    sl.n = level;
    for (int j = 0; j < level; j++) sl.idx[j] = j;
    rescale_arg_t ra;
    memset(&ra, 0, sizeof ra);
    if (special) {
        const u64 P = c->q[Lk - 1];
        for (int j = 0; j < level; j++) A.pinv[j] = ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
    }
    rot_tail_arg_t G;
    memset(&G, 0, sizeof G);
    for (int r = 0; r < R; r++) G.g[r] = galois[r];
    // With a special prime the rotations are finished in the evaluation domain (k_md_*: only the special limb of every key sum is
    // inverse-transformed); TFHE_MD_COEFF=1 keeps the coefficient-domain tail (k_ks_rot_tail) for comparisons.
    static const bool md_coeff = getenv("TFHE_MD_COEFF") && getenv("TFHE_MD_COEFF")[0] == '1';
    const bool eval_form = special && R > 0 && !md_coeff;
    // workspace per ciphertext: digits (level nw rows) + S / T (R 2 nw) + rotated ciphertexts (R 2 level) + the ciphertext's own transform (2 level)
    // (+ evaluation-domain form: the lifted special limbs before their transforms (R 2 level) and the special limbs themselves (R 2))
    // (the untransformed lifts only where the lift is not fused into the transforms' loads)
    const bool need_lf = eval_form && !(md_lift_is_fused(c, A, sl) && (size_t)R * 2 * level * 4 * 512 <= 0x7fffffffull);
    const size_t per_ct = ((size_t)level * nw + (size_t)R * 2 * nw + (size_t)R * 2 * level + (size_t)2 * level +
                           (eval_form ? (need_lf ? (size_t)R * 2 * level : 0) + (size_t)R * 2 : 0)) * N * 8;
    // The keys of all R rotations are read once per chunk (2.8 GB at N = 2^16, 6 limbs + special prime, 63 rotations): a chunk as
    // large as the memory allows (32 GiB of workspace unless TFHE_MD_WS_GIB says otherwise; halved while the allocation fails)
    static const size_t ws_cap_env = [] { const char* e = getenv("TFHE_MD_WS_GIB"); const long g = e ? atol(e) : 0; return (size_t)(g > 0 ? g : 32) << 30; }();
    // ... and never more than half of what the device has free right now (counting what this context and the allocator's cache
    // already hold, both of which the allocation below can reuse): other contexts, BFV plans and processes keep their room
    size_t ws_cap = ws_cap_env;
    {
        // (hipMemGetInfo is a driver round trip -- milliseconds in a process with many allocations -- and this call sits in the
        // launch path of a host-bound circuit: the reading is kept for two seconds)
        // (r05, ADVICE r04: the reading and the allocator's cached bytes are those of THIS context's device -- a process driving
        // several GPUs computed the cap from whichever device had asked last and from the cache summed over all of them)
        struct mi_t { size_t free_bytes = 0; std::chrono::steady_clock::time_point at{}; };
        static std::mutex mi_mu;
        static std::map<int, mi_t> mi_dev;
        size_t fr = 0, tot = 0;
        bool have = false;
        const int cur_dev = devalloc::current_device();          // what hipMemGetInfo reads and where the workspace will be allocated
        {
            std::lock_guard<std::mutex> g(mi_mu);
            mi_t& mi = mi_dev[cur_dev];
            const auto now = std::chrono::steady_clock::now();
            if (mi.free_bytes && now - mi.at < std::chrono::seconds(2)) { fr = mi.free_bytes; have = true; }
            else if (hipMemGetInfo(&fr, &tot) == hipSuccess) { mi.free_bytes = fr; mi.at = now; have = true; }
            else (void)hipGetLastError();
        }
        if (have) {
            size_t cached = 0;
            {
                devalloc::state_t& as = devalloc::S();
                std::lock_guard<std::mutex> g(as.mu);
                auto it = as.devs.find(cur_dev);
                if (it != as.devs.end()) cached = it->second.cached_bytes;
            }
            ws_cap = std::min(ws_cap, std::max<size_t>((fr + c->ws_bytes + cached) / 2, (size_t)1 << 30));
        }
    }
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>({batch, (int64_t)512, (int64_t)(ws_cap / per_ct)}));
    size_t ntt_tmp = 0;
    void* ws = nullptr;
    int rc;
    ws_borrow_t borrow(c);    // a workspace above ws_keep_bytes() is this call's: back to the allocator on every return path
    for (bool first_try = true;; first_try = false) {
        const size_t ntt_rows = (size_t)chunk * std::max<size_t>({(size_t)level * nw, (size_t)R * 2 * nw, (size_t)2 * level});
        ntt_tmp = c->logN > 14 ? ntt_rows * N * 8 : 0;
        const size_t need = ntt_tmp + chunk * per_ct;
        if (first_try && need > ws_keep_bytes() && need > c->ws_bytes) borrow.begin();
        rc = ensure_ws(c, need, &ws, borrow.active);
        if (rc != TFHE_E_NOMEM || chunk == 1) break;
        chunk = (chunk + 1) / 2;
    }
    if (rc) return rc;
    u64* dig = (u64*)((char*)ws + ntt_tmp);
    u64* S = dig + (size_t)chunk * level * nw * N;
    u64* ROT = S + (size_t)chunk * R * 2 * nw * N;
    u64* X = ROT + (size_t)chunk * R * 2 * level * N;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = std::min(chunk, batch - b0);
        const u64* cin = ct + (size_t)b0 * polys * level * N;
        if (eval_form) {
            u64* const U = ROT;                                  // [R][nb][2][level][N]
            u64* const LF = need_lf ? X + (size_t)chunk * 2 * level * N : nullptr;   // lifted special limbs, untransformed (same shape)
            u64* const PB = X + (size_t)chunk * 2 * level * N + (need_lf ? (size_t)chunk * R * 2 * level * N : 0);   // [R][nb][2][N]
            const u32 groups = (u32)((int64_t)R * nb * 2);
            rc = run_ntt(c, false, cin, X, nb * 2 * level, sl);
            if (rc) return rc;
            rc = ks_digits_fwd(c, A, cin, dig, nb);
            if (rc) return rc;
            rc = ks_inner_launch(c, A, Lk, nullptr, dig, S, nb, evks, R, X);   // limbs j < level leave as V = S' P^-1 + X0 [s = 0]
            if (rc) return rc;
            hipLaunchKernelGGL(k_md_special_perm, dim3(8 * TFHE_ROT_TAIL_SLOTS), dim3(256), 0, c->stream, S, PB, G, n, (u32)nw, (u32)level, (u32)nb, groups);
            HIP_TRY(hipGetLastError());
            limb_sel_t sp;
            sp.n = 1;
            sp.idx[0] = Lk - 1;
            rc = run_ntt(c, true, PB, PB, (int64_t)groups, sp);
            if (rc) return rc;
            bool scaled = true;
            rc = md_lift_fwd(c, A, sl, ra, PB, LF, U, (int64_t)groups, &scaled);
            if (rc) return rc;
            auto acc = scaled ? k_md_acc<false> : k_md_acc<true>;
            hipLaunchKernelGGL(acc, dim3(8 * TFHE_MD_SLOTS), dim3(256), 0, c->stream, X, S, U, diags, out + (size_t)b0 * 2 * level * N, c->limbs_dev, sl,
                               G, ra, n, (u32)nw, (u32)R, (u32)nb);
            HIP_TRY(hipGetLastError());
            continue;
        }
        if (R) {
            rc = ks_digits_fwd(c, A, cin, dig, nb);
            if (rc) return rc;
            rc = ks_inner_launch(c, A, Lk, nullptr, dig, S, nb, evks, R);   // key sums of the unrotated digits against every prepared key
            if (rc) return rc;
            rc = run_ntt(c, true, S, S, (int64_t)R * nb * 2 * nw, A.w);
            if (rc) return rc;
            hipLaunchKernelGGL(k_ks_rot_tail, dim3(8 * TFHE_ROT_TAIL_SLOTS), dim3(256), 0, c->stream, S, cin, ROT, c->limbs_dev, A, ra, G, n, (u32)nb,
                               (u32)((int64_t)R * nb * 2));
            HIP_TRY(hipGetLastError());
            rc = run_ntt(c, false, ROT, ROT, (int64_t)R * nb * 2 * level, sl);
            if (rc) return rc;
        }
        rc = run_ntt(c, false, cin, X, nb * 2 * level, sl);
        if (rc) return rc;
        hipLaunchKernelGGL(k_matmul_acc, row_grid((unsigned)(nb * 2 * level), N), dim3(256), 0, c->stream, X, ROT, diags, out + (size_t)b0 * 2 * level * N,
                           c->limbs_dev, sl, n, (u32)R, (u32)(nb * 2 * level));
        HIP_TRY(hipGetLastError());
    }
    return TFHE_OK;
}

// ---- digit-window key switch (relin_window != 0, rlwe_she.jl:330-338) -----------------------------------------------
static int ksw_table(tfhe_ctx* c, int level, const conv_tab_t** out) {
    *out = nullptr;
    if (level == 1) return TFHE_OK;
    std::lock_guard<std::mutex> g(c->ksw_mu);
    auto it = c->ksw_tabs.find(level);
    if (it != c->ksw_tabs.end()) { *out = it->second; return TFHE_OK; }
    conv_host_t H;
    build_conv_host(std::vector<u64>(c->q.begin(), c->q.begin() + level), std::vector<u64>(), &H);
    conv_tab_t T = H.tab;
    auto up = [&](const std::vector<u64>& v, const u64** d) -> bool {
        *d = nullptr;
        if (v.empty()) return true;
        void* p = nullptr;
        if (devalloc::malloc_retry(&p, v.size() * 8) != hipSuccess) return false;
        c->ksw_allocs.push_back(p);
        if (hipMemcpy(p, v.data(), v.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return false;
        *d = (const u64*)p;
        return true;
    };
    void* dt = nullptr;
    if (!up(H.C, &T.C) || !up(H.M, &T.M) || !up(H.Aw, &T.Aw) || devalloc::malloc_retry(&dt, sizeof T) != hipSuccess)
        return fail(TFHE_E_NOMEM, "allocating the window-digit tables failed");
    c->ksw_allocs.push_back(dt);
    HIP_TRY(hipMemcpy(dt, &T, sizeof T, hipMemcpyHostToDevice));
    c->ksw_tabs[level] = (conv_tab_t*)dt;
    *out = (conv_tab_t*)dt;
    return TFHE_OK;
}

int tfhe_keyswitch_window(tfhe_ctx* c, int key_limbs, int level, int special, int window_bits, const uint64_t* evk, int n_windows,
                          const uint64_t* ct, int polys, uint64_t* out, int64_t batch) {
    if (!c || !evk || !ct || !out) return fail(TFHE_E_BADARG, "null argument");
    if (polys != 2 && polys != 3) return fail(TFHE_E_BADARG, "keyswitch needs a 2- or 3-element ciphertext (rlwe_she.jl:318), got %d", polys);
    const int Lk = key_limbs;
    if (Lk < 1 || Lk > c->L) return fail(TFHE_E_BADARG, "key_limbs=%d outside [1,%d]", Lk, c->L);
    if (level < 1 || level > (special ? Lk - 1 : Lk)) return fail(TFHE_E_LEVEL_MISMATCH, "level=%d outside [1,%d]", level, special ? Lk - 1 : Lk);
    if (batch < 0) return fail(TFHE_E_BADARG, "negative batch");
    hostmath::bigint Q = hostmath::big_from(1);
    u64 qmin = special ? c->q[Lk - 1] : ~0ull;
    for (int j = 0; j < level; j++) { Q = hostmath::big_mul_u64(Q, c->q[j]); qmin = std::min(qmin, c->q[j]); }
    if (window_bits < 1 || window_bits > 32 || (qmin >> window_bits) == 0)
        return fail(TFHE_E_BADARG, "window of %d bits: need 1 <= w <= 32 and 2^w below every modulus", window_bits);
    int qbits = 0;
    for (int w = (int)Q.size() - 1; w >= 0 && !qbits; w--)
        if (Q[w]) qbits = w * 64 + hostmath::bitlen(Q[w]);
    // ndigits(modulus of the CIPHERTEXT ring, base = 2^w), rlwe_she.jl:333; a key made for a larger ring (a higher level, or the
    // ModulusRaised key ring Q P, modulusraising.jl:28-32) has more components: the loop rlwe_she.jl:340 uses the first `need`
    const int need = (qbits + window_bits - 1) / window_bits;
    if (n_windows < need)
        return fail(TFHE_E_PARAMS_MISMATCH, "evaluation key has %d components, a %d-bit modulus in %d-bit windows has %d digits", n_windows, qbits, window_bits, need);
    if (batch == 0) return TFHE_OK;
    const conv_tab_t* T = nullptr;
    int rc = ksw_table(c, level, &T);
    if (rc) return rc;
    const size_t N = (size_t)c->N;
    const u32 n = (u32)c->N;
    const int nw = special ? level + 1 : level;  // working limbs: [0 .. level-1] (+ the special prime, downswitch_keyelement modulusraising.jl:43-49)
    const size_t per_ct = ((size_t)2 * nw + (size_t)need * nw) * N * 8;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>({batch, (int64_t)512, (int64_t)((8192ull << 20) / per_ct)}));
    const size_t ntt_tmp = c->logN > 14 ? (size_t)chunk * need * nw * N * 8 : 0;
    void* ws = nullptr;
    rc = ensure_ws(c, ntt_tmp + chunk * per_ct, &ws);
    if (rc) return rc;
    u64* S = (u64*)((char*)ws + ntt_tmp);
    u64* dig = S + (size_t)chunk * 2 * nw * N;
    ks_arg_t A;   // inner product: `level` field = number of digits, nw = working limbs
    memset(&A, 0, sizeof A);
    A.level = need; A.nw = nw; A.special = special ? 1 : 0; A.polys = polys;
    A.w.n = nw;
    for (int j = 0; j < level; j++) A.w.idx[j] = j;
    if (special) A.w.idx[level] = Lk - 1;
    ks_arg_t Al = A;  // row bookkeeping of the tail kernels: `level` = ciphertext limbs
    Al.level = level;
    const u32 add_s = polys == 3 ? 2u : 1u;
    const unsigned gx = (n + 255) / 256;
    rescale_arg_t ra;
    memset(&ra, 0, sizeof ra);
    if (special) {
        const u64 P = c->q[Lk - 1];
        for (int j = 0; j < level; j++) ra.qlinv[j] = hostmath::make_tw(hostmath::invmod_prime(P % c->q[j], c->q[j]), c->q[j]);
    }
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = std::min(chunk, batch - b0);
        const u64* cin = ct + (size_t)b0 * polys * level * N;
        u64* cout = out + (size_t)b0 * 2 * level * N;
        hipLaunchKernelGGL(k_ks_window_digits, dim3((unsigned)(nb * gx)), dim3(256), 0, c->stream, cin, dig, T, level, window_bits, need, polys, n, gx, nw);
        HIP_TRY(hipGetLastError());
        rc = run_ntt(c, false, dig, dig, nb * need * nw, A.w);
        if (rc) return rc;
        const unsigned bsplit = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nb, (4096 + nw * gx - 1) / (nw * gx)));
        ks_keys_t K1;
        memset(&K1, 0, sizeof K1);
        hipLaunchKernelGGL(k_ks_inner<8>, dim3((unsigned)nw * gx * bsplit), dim3(256), 0, c->stream, evk, dig, S, c->limbs_dev, A, Lk, n, (u32)nb, bsplit, 0u, K1);
        HIP_TRY(hipGetLastError());
        if (special) {
            // ModulusRaised: c1 = P c + S over [q_0 .. q_{level-1}, P], contracted by modswitch (modulusraising.jl:35-42):
            // out_j = c_j + (T_j - [T_P]) P^-1 with T = INTT(S)  (the tail of the RNS-digit path, k_ks_rescale_add)
            rc = run_ntt(c, true, S, S, nb * 2 * nw, A.w);
            if (rc) return rc;
            hipLaunchKernelGGL(k_ks_rescale_add, row_grid((unsigned)(nb * 2 * level), (size_t)c->N), dim3(256), 0, c->stream, S, cin, cout, c->limbs_dev, Al, ra, n, add_s);
            HIP_TRY(hipGetLastError());
        } else if (c->logN <= 14) {
            ntt_io_t io = io_plain();
            io.mode = 2; io.gsz = (u32)(2 * level); io.src_gstride = io.gsz; io.dst_gstride = io.gsz;
            io.add_rows = add_s * (u32)level; io.add_gstride = (u32)(polys * level); io.addend = cin;
            rc = run_ntt(c, true, S, cout, nb * 2 * level, A.w, &io);
            if (rc) return rc;
        } else {
            rc = run_ntt(c, true, S, cout, nb * 2 * level, A.w);
            if (rc) return rc;
            hipLaunchKernelGGL(k_ks_add_ct, row_grid((unsigned)(nb * 2 * level), (size_t)c->N), dim3(256), 0, c->stream, cin, cout, c->limbs_dev, Al, n, add_s);
            HIP_TRY(hipGetLastError());
        }
    }
    return TFHE_OK;
}

// ---- CKKS encode / decode (float; ckksencoding.jl:56-97, ckks.jl:35-59) -------------------------------------------
static int ckks_tables(tfhe_ctx* c) {
    std::lock_guard<std::mutex> g(c->ksw_mu);
    if (c->ckks_roots) return TFHE_OK;
    const size_t n = (size_t)c->N, n2 = n / 2;
    const int logn = c->logN;
    std::vector<double> roots(n2 * 2 + 2), tw(n * 2);
    for (size_t t = 0; t < n2; t++) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)t / (long double)n;
        roots[2 * t] = (double)cosl(a);
        roots[2 * t + 1] = (double)sinl(a);
    }
    for (size_t k = 0; k < n; k++) {
        // ckksencoding.jl:84: the angle 2k/(2N)*pi is evaluated in Float64 first, then exponentiated
        const double ang = ((double)(2 * k) / (double)(2 * n)) * 3.14159265358979323846;
        tw[2 * k] = cos(ang);
        tw[2 * k + 1] = sin(ang);
    }
    auto brev = [&](u32 x) { u32 r = 0; for (int i = 0; i < logn; i++) r |= ((x >> i) & 1u) << (logn - 1 - i); return r; };
    std::vector<u32> pos(2 * n2), gpos(n2);
    const u64 M = 2 * n;
    u64 e = 1;
    for (size_t i = 0; i < n2; i++) {
        e = (e * 3) % M;                       // 3^(i+1) mod 2N   (ZmstarPermutation, ckksencoding.jl:49-54)
        pos[2 * i] = brev((u32)(e >> 1));
        pos[2 * i + 1] = brev((u32)(((M - e) % M) >> 1));
        gpos[i] = brev((u32)(e >> 1));
    }
    auto up = [&](const void* h, size_t bytes, void** d) -> bool {
        if (devalloc::malloc_retry(d, bytes ? bytes : 8) != hipSuccess) return false;
        c->ksw_allocs.push_back(*d);
        return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    void *r = nullptr, *t = nullptr, *p = nullptr, *gp = nullptr;
    if (!up(roots.data(), (n2 ? n2 : 1) * 16, &r) || !up(tw.data(), n * 16, &t) || !up(pos.data(), pos.size() * 4, &p) ||
        !up(gpos.data(), gpos.size() * 4, &gp))
        return fail(TFHE_E_NOMEM, "allocating the CKKS tables failed");
    c->ckks_tw = t; c->ckks_pos = p; c->ckks_gpos = gp; c->ckks_roots = r;
    return TFHE_OK;
}
static int ckks_check(tfhe_ctx* c, int level, uint64_t smant, const void* a, const void* b, int64_t batch) {
    if (!c || !a || !b) return fail(TFHE_E_BADARG, "null argument");
    if (level < 1 || level > c->L) return fail(TFHE_E_LEVEL_MISMATCH, "level=%d outside [1,%d]", level, c->L);
    if (smant == 0) return fail(TFHE_E_BADARG, "scale must be positive");
    if (c->N < 4) return fail(TFHE_E_UNSUPPORTED, "CKKS encoding needs N >= 4");
    if (batch < 0) return fail(TFHE_E_BADARG, "negative batch");
    return TFHE_OK;
}
static int ckks_fft(tfhe_ctx* c, cplx_t* buf, int64_t nb, bool dif, bool conj_roots) {
    const u32 n = (u32)c->N;
    const dim3 grid((n / 2 + 255) / 256, (unsigned)nb);
    if (dif) for (u32 h = n / 2; h >= 1; h >>= 1) hipLaunchKernelGGL(k_fft_pass, grid, dim3(256), 0, c->stream, buf, (const cplx_t*)c->ckks_roots, n, h, 1, conj_roots ? 1 : 0);
    else for (u32 h = 1; h < n; h <<= 1) hipLaunchKernelGGL(k_fft_pass, grid, dim3(256), 0, c->stream, buf, (const cplx_t*)c->ckks_roots, n, h, 0, conj_roots ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

int tfhe_ckks_encode(tfhe_ctx* c, int level, uint64_t scale_mant, int scale_exp2, const double* slots, uint64_t* out, int64_t batch) {
    int rc = ckks_check(c, level, scale_mant, slots, out, batch);
    if (rc || batch == 0) return rc;
    rc = ckks_tables(c);
    if (rc) return rc;
    const u32 n = (u32)c->N;
    limb_sel_t sel;
    sel.n = level;
    for (int j = 0; j < level; j++) sel.idx[j] = j;
    const int64_t chunk = std::min<int64_t>(batch, 4096);
    void* ws = nullptr;
    rc = ensure_ws(c, (size_t)chunk * n * 16, &ws);
    if (rc) return rc;
    cplx_t* buf = (cplx_t*)ws;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = std::min(chunk, batch - b0);
        hipLaunchKernelGGL(k_ckks_scatter, dim3((n / 2 + 255) / 256, (unsigned)nb), dim3(256), 0, c->stream,
                           (const cplx_t*)slots + (size_t)b0 * (n / 2), buf, (const u32*)c->ckks_pos, n);
        rc = ckks_fft(c, buf, nb, false, true);  // inverse transform: conjugate roots, 1/N applied below
        if (rc) return rc;
        hipLaunchKernelGGL(k_ckks_encode_finish, dim3((n + 255) / 256, (unsigned)nb), dim3(256), 0, c->stream, buf, (const cplx_t*)c->ckks_tw,
                           out + (size_t)b0 * level * n, c->limbs_dev, sel, scale_mant, scale_exp2, n);
        HIP_TRY(hipGetLastError());
    }
    return TFHE_OK;
}

int tfhe_ckks_decode(tfhe_ctx* c, int level, uint64_t scale_mant, int scale_exp2, const uint64_t* in, double* slots, int64_t batch) {
    int rc = ckks_check(c, level, scale_mant, in, slots, batch);
    if (rc || batch == 0) return rc;
    rc = ckks_tables(c);
    if (rc) return rc;
    const conv_tab_t* T = nullptr;
    rc = ksw_table(c, level, &T);
    if (rc) return rc;
    const u32 n = (u32)c->N;
    const int64_t chunk = std::min<int64_t>(batch, 4096);
    void* ws = nullptr;
    rc = ensure_ws(c, (size_t)chunk * n * 16, &ws);
    if (rc) return rc;
    cplx_t* buf = (cplx_t*)ws;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = std::min(chunk, batch - b0);
        hipLaunchKernelGGL(k_ckks_decode_start, dim3((n + 255) / 256, (unsigned)nb), dim3(256), 0, c->stream, in + (size_t)b0 * level * n, buf,
                           (const cplx_t*)c->ckks_tw, T, level, c->q[0], scale_mant, scale_exp2, n);
        HIP_TRY(hipGetLastError());
        rc = ckks_fft(c, buf, nb, true, false);
        if (rc) return rc;
        hipLaunchKernelGGL(k_ckks_gather, dim3((n / 2 + 255) / 256, (unsigned)nb), dim3(256), 0, c->stream, buf,
                           (cplx_t*)slots + (size_t)b0 * (n / 2), (const u32*)c->ckks_gpos, n);
        HIP_TRY(hipGetLastError());
    }
    return TFHE_OK;
}

// ---- device-side samplers (poly.jl:7-23, crt.jl:277-279; stream definition in sample_kernels.h) --------------------------
int tfhe_sample_uniform(tfhe_ctx* c, int level, uint64_t seed, uint32_t stream, uint64_t first_poly, uint64_t* out, int64_t count) {
    if (!c || !out) return fail(TFHE_E_BADARG, "null argument");
    if (level < 1 || level > c->L) return fail(TFHE_E_LEVEL_MISMATCH, "level=%d outside [1,%d]", level, c->L);
    if (count < 0) return fail(TFHE_E_BADARG, "negative count");
    if (first_poly + (u64)count > (1ull << 32)) return fail(TFHE_E_BADARG, "polynomial counter exceeds 2^32");
    const u32 n = (u32)c->N;
    for (int64_t p0 = 0; p0 < count; p0 += 32768) {
        const unsigned np = (unsigned)std::min<int64_t>(32768, count - p0);
        hipLaunchKernelGGL(k_sample_uniform, dim3((n + 255) / 256, (unsigned)level, np), dim3(256), 0, c->stream, out + (size_t)p0 * level * n,
                           c->limbs_dev, level, seed, stream, n, first_poly + (u64)p0);
    }
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}
int tfhe_sample_gaussian(tfhe_ctx* c, int level, double sigma, uint64_t multiplier, uint64_t seed, uint32_t stream, uint64_t first_poly,
                         uint64_t* out, int64_t count) {
    if (!c || !out) return fail(TFHE_E_BADARG, "null argument");
    if (level < 1 || level > c->L) return fail(TFHE_E_LEVEL_MISMATCH, "level=%d outside [1,%d]", level, c->L);
    if (!(sigma >= 0) || sigma > 1e15) return fail(TFHE_E_BADARG, "sigma out of range");
    if (count < 0) return fail(TFHE_E_BADARG, "negative count");
    if (first_poly + (u64)count > (1ull << 32)) return fail(TFHE_E_BADARG, "polynomial counter exceeds 2^32");
    const u32 n = (u32)c->N;
    for (int64_t p0 = 0; p0 < count; p0 += 32768) {
        const unsigned np = (unsigned)std::min<int64_t>(32768, count - p0);
        hipLaunchKernelGGL(k_sample_gaussian, dim3((n + 255) / 256, np), dim3(256), 0, c->stream, out + (size_t)p0 * level * n, c->limbs_dev,
                           level, sigma, multiplier, seed, stream, n, first_poly + (u64)p0);
    }
    HIP_TRY(hipGetLastError());
    return TFHE_OK;
}

// ---- profiling / events ---------------------------------------------------------------------------
int tfhe_prof_enable(tfhe_ctx* c, int on) {
    if (!c) return fail(TFHE_E_BADARG, "null context");
    c->prof = on != 0;
    return TFHE_OK;
}
int tfhe_prof_read(tfhe_ctx* c, int64_t* launches, int64_t* limb_polys, double* total_ms) {
    if (!c) return fail(TFHE_E_BADARG, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    int64_t nl = 0, np = 0;
    double ms = 0;
    for (auto& p : c->prof_pairs) {
        float t = 0;
        hipEventElapsedTime(&t, p.a, p.b);
        if (p.limb_polys > 0) { nl++; np += p.limb_polys; ms += t; }
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    c->prof_pairs.clear();
    if (launches) *launches = nl;
    if (limb_polys) *limb_polys = np;
    if (total_ms) *total_ms = ms;
    return TFHE_OK;
}
int tfhe_event_create(void** ev) {
    if (!ev) return fail(TFHE_E_BADARG, "null out pointer");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = (void*)e;
    return TFHE_OK;
}
int tfhe_event_destroy(void* ev) { if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return TFHE_OK; }
int tfhe_event_record(tfhe_ctx* c, void* ev) {
    if (!c || !ev) return fail(TFHE_E_BADARG, "null argument");
    HIP_TRY(hipEventRecord((hipEvent_t)ev, c->stream));
    return TFHE_OK;
}
int tfhe_event_elapsed_ms(void* a, void* b, float* ms) {
    if (!a || !b || !ms) return fail(TFHE_E_BADARG, "null argument");
    HIP_TRY(hipEventSynchronize((hipEvent_t)b));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return TFHE_OK;
}

}  // extern "C"

#include "bfv_api.inc"
#include "comm_api.inc"

#ifdef TFHE_KS_TRACE
extern "C" int tfhe_debug_kstrace(unsigned long long* out, unsigned* n, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(tfhe_kst), sizeof(unsigned long long) * 4096);
    hipMemcpyFromSymbol(n, HIP_SYMBOL(tfhe_kst_n), sizeof(unsigned));
    if (reset) { unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(tfhe_kst_n), &z, sizeof z); }
    return 0;
}
#endif
