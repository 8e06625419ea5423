// kernels.h -- __global__ kernels for gfx950 (MI355X).  Wave64; one workgroup per RNS limb-polynomial
// (or per block of it) for the NTTs, one workgroup per limb row for the streaming kernels, one
// thread per coefficient for the cross-limb kernels.  Integer work: no MFMA by design.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "bfv_core.h"
#include "bfv_fast.h"
#include "ntt_core.h"

typedef u64 u64x2_t __attribute__((ext_vector_type(2)));

// Minimum waves per SIMD the row kernels are compiled for (2 = the two waves per SIMD a 512-thread workgroup occupies).
// With 3 (a 168-VGPR cap) a streaming kernel of another chunk can be co-resident; measured (tools/two_lane_probe.py) that
// buys +5 % from overlap but costs more in the transforms, so the default stays 2.
#ifndef TFHE_NTT_WAVES
#define TFHE_NTT_WAVES 2
#endif

// The thread index as an opaque per-row value: stops loop-invariant code motion from parking 30-60 VGPRs of precomputed
// per-element addresses (global and LDS) across the whole row loop; they are re-derived with one add each instead.
__device__ __forceinline__ u32 fresh_tid() {
    u32 t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

struct limb_sel_t {  // which context modulus each buffer limb uses (crtselect, src/crt.jl:185-211)
    int n;
    int idx[TFHE_MAX_LIMBS];
};

// ------------------------------------------------------------------------------------------------
// NTT: register-blocked LDS kernel.  Block b = ((poly*limbs + j) << x) + sb handles sub-block sb of
// limb j of polynomial `poly`; x = log2(N) - LOGB (0 when the whole limb fits one LDS block).
// ------------------------------------------------------------------------------------------------

// Early LDS stores (ntt_core.h, fwd_compute / inv_compute `lds_early`), per kernel family: bit set = the pass writes its results to
// LDS from inside its last butterfly stage.  Measured per site (r05, profiles/LOG.md; registers from the ISA):
//   bit 0  k_ntt_fwd_quad middle pass     182 -> 234 VGPRs, no scratch: N = 2^16 forward 2.75 -> 3.00 TB/s (mixed ring 2.00 -> 2.12)   ON
//   bit 5  k_ntt_inv_staged middle pass   140 -> 166 VGPRs: inverse at 2^14 + 0.7 %                                                     ON
//   bit 7  k_ntt_inv_pair middle pass     scratch 164 -> 96 bytes                                                                       ON
//   bit 8  k_ntt_inv_subpair middle pass  218 -> 222 VGPRs: cfg#5 key switch + 0.7 %, reference-shaped MNIST + 2 %                     ON
//   bit 11 k_ntt_fwd_pf middle pass       234 -> 238 VGPRs: no change (3.93 / 3.95 TB/s)                                                off
//   bit 3  fwd_schedule (block kernels)   u64 forward 117 -> 188 VGPRs: 60-bit transforms 2.06 -> 1.91 TB/s                              off
//   bits 1, 2, 6, 9, 10 (quad first pass, k_ntt_fwd_pair, inv_schedule, k_ntt_inv_quad2, k_ntt_fwd_pf first pass): 80 - 830 bytes of scratch  off
// The fused kernels sit at the 256-VGPR cap: the same change spills 13-20 accumulator registers per digit there (headline 61.2 k ->
// 56.4 k ciphertext-mul/s, cfg#3 40.1 k -> 30.4 k key switches/s) -- they keep the store phase behind the pass.
#ifndef TFHE_ES_SITES
#define TFHE_ES_SITES 0x1a1
#endif
#define TFHE_ES(bit) (((TFHE_ES_SITES) >> (bit)) & 1)
// ---- simple schedule: one workgroup per item, passes separated by barriers ----
template <class A, int LOGB, int LOGT, int S0>
__device__ __forceinline__ void fwd_schedule(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 tid,
                                             u32 pre, int x, u32 sbrev, const lift_t* lift) {
    constexpr int K = pass_k_fwd(LOGB, LOGT, S0);
    constexpr bool LAST = (S0 + K == LOGB);
    ntt_fwd_pass<A, LOGB, LOGT, S0, K, S0 == 0, LAST, TFHE_ES(3) != 0>(lds, gsrc, gdst, C, tid, pre, x, sbrev, lift);
    if constexpr (!LAST) {
        __syncthreads();
        fwd_schedule<A, LOGB, LOGT, S0 + K>(lds, gsrc, gdst, C, tid, pre, x, sbrev, lift);
    }
}
// ---- inverse schedule with the twiddles of pass p+1 requested before the LDS exchange that ends pass p (8-byte fp64
// twiddles: a whole pass's set fits the register budget of a 512-thread workgroup).  +9 % for the inverse transform; the
// same idea measured slower for the forward one, which keeps the simple schedule. ----
// AO: the policy whose out_inv_* forms the words of the final store (ArithFpD: reduced doubles instead of canonical words)
template <class A, int LOGB, int LOGT, int SEND, bool SCALE = true, class AO = A>
__device__ __forceinline__ void inv_schedule_ptw(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 tid,
                                                 u32 pre, const u64* addend, const typename A::tw* tw_cur, u64* keep = nullptr) {
    constexpr int K = pass_k_inv(LOGB, LOGT, SEND);
    constexpr int S0 = SEND - K;
    constexpr bool FROM_GLOBAL = (SEND == LOGB), TO_GLOBAL = (S0 == 0);
    typedef pgeom<LOGB, LOGT, S0, K> G;
    u64 raw[G::E];
    typename A::elem v[G::E];
    inv_load_data<LOGB, LOGT, S0, K, FROM_GLOBAL>(raw, lds, gsrc, tid, 0, 0u);
    // the last pass (S0 == 0) has workgroup-uniform twiddles: loaded inside (scalar); the first one loads its own
    // alongside the operands; the middle ones come prefetched
    inv_compute<A, LOGB, LOGT, S0, K, FROM_GLOBAL, SCALE, (TO_GLOBAL || FROM_GLOBAL) ? 0 : K>(v, raw, tw_cur, C, tid, pre);
    if constexpr (!TO_GLOBAL) {
        constexpr int K2 = pass_k_inv(LOGB, LOGT, S0);
        typedef pgeom<LOGB, LOGT, S0 - K2, K2> G2;
        typename A::tw tw_next[G2::SETS * G2::NTW];
        if constexpr (S0 - K2 != 0) inv_load_tw<A, LOGB, LOGT, S0 - K2, K2, false>(tw_next, C, tid, pre);
        inv_store<A, LOGB, LOGT, S0, K, FROM_GLOBAL, SCALE>(v, lds, gdst, C, tid, nullptr);
        __syncthreads();
        inv_schedule_ptw<A, LOGB, LOGT, S0, SCALE, AO>(lds, gsrc, gdst, C, tid, pre, addend, tw_next, keep);
    } else {
        inv_store<AO, LOGB, LOGT, S0, K, FROM_GLOBAL, SCALE>(v, lds, gdst, C, tid, addend, keep);
    }
}

template <class A, int LOGB, int LOGT, int SEND, bool SCALE>
__device__ __forceinline__ void inv_schedule(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 tid,
                                             u32 pre, int x, u32 sbrev, const u64* addend) {
    constexpr int K = pass_k_inv(LOGB, LOGT, SEND);
    constexpr int S0 = SEND - K;
    ntt_inv_pass<A, LOGB, LOGT, S0, K, SEND == LOGB, S0 == 0, SCALE, TFHE_ES(6) != 0>(lds, gsrc, gdst, C, tid, pre, x, sbrev, addend);
    if constexpr (S0 != 0) {
        __syncthreads();
        inv_schedule<A, LOGB, LOGT, S0, SCALE>(lds, gsrc, gdst, C, tid, pre, x, sbrev, addend);
    }
}

// Workgroups loop over items i = blockIdx.x, blockIdx.x + gridDim.x, ... (item = (row << x) + sb, row = poly*limbs + j
// in the plain mode; see ntt_io_t for the grouped / digit-lift / addend row maps).
// Item walk of the block kernels for N > 2^LOGB (x > 0).  The 2^x sub-blocks of a row interleave in the row's natural
// order at 8-byte granularity (forward output / inverse input at stride 2^x), so every 128-byte line is shared by 2^x
// sub-blocks.  Workgroups are dispatched to the 8 XCDs round-robin (workgroup b -> XCD b % 8); the walk gives the
// sub-blocks of one row to workgroups of ONE XCD in the same iteration, so the partial lines meet in that XCD's L2
// (stores merge before write-back, loads hit after the first miss) instead of crossing the fabric 2^x times.
// Returns the item ((row << x) | sub-block) for position `it` of workgroup `b`, or ~0u past the end; identity when the
// grid does not tile.
__device__ __forceinline__ u32 xcd_walk_item(u32 it, u32 b, u32 grid, int x, u32 nitems) {
    const u32 nsb = 1u << x;
    if (x == 0 || (grid % (8u * nsb)) != 0u) {
        const u32 item = it * grid + b;
        return item < nitems ? item : ~0u;
    }
    const u32 xcd = b & 7u, slot = b >> 3, groups = (grid >> 3) >> x;
    const u32 row = it * (8u * groups) + xcd * groups + (slot >> x);
    const u32 item = (row << x) | (slot & (nsb - 1u));
    return item < nitems ? item : ~0u;
}
// Walk of the fused kernels' (ciphertext b, limb j) items, numbered item = b * nb + j.  Workgroup w runs on XCD w & 7, and
// every XCD has its own 4 MiB L2: with the plain walk (item = it * grid + w) each XCD meets all nb limbs in turn, and the
// per-limb constants it streams -- four 128 KiB twiddle tables per limb, in the key switch also 2 * level key rows per limb --
// are 8.5 MB (17 limbs) / 19 MB (keys) per XCD: they miss L2 and come from the Infinity Cache every time (the PMC read
// traffic of k_bfv_core_fused was 2.2 x its algorithmic bytes).  Here XCD x takes the x-th eighth of the items in LIMB-MAJOR
// order (at most ceil(nb / 8) + 1 limbs), its workgroups striding through that share.  Returns ~0u past the end.
// Used by k_bfv_core_fused (+ 1.1 % on the headline).
#ifndef TFHE_XCD_LIMB
#define TFHE_XCD_LIMB 1
#endif
template <bool ON = true>
__device__ __forceinline__ u32 xcd_limb_niter(u32 grid, u32 nitems) {
    if (!ON || !TFHE_XCD_LIMB || (grid & 7u)) return (nitems + grid - 1) / grid;
    const u32 per = (nitems + 7u) >> 3, nslots = grid >> 3;
    return (per + nslots - 1) / nslots;
}
template <bool ON = true>
__device__ __forceinline__ u32 xcd_limb_walk(u32 it, u32 wg, u32 grid, u32 nb, u32 nitems) {
    if (!ON || !TFHE_XCD_LIMB || (grid & 7u)) {
        const u32 item = it * grid + wg;
        return item < nitems ? item : ~0u;
    }
    const u32 per = (nitems + 7u) >> 3, nslots = grid >> 3, xcd = wg & 7u, slot = wg >> 3;
    const u32 k = it * nslots + slot, idx = xcd * per + k;  // position in this XCD's share / in limb-major order
    if (k >= per || idx >= nitems) return ~0u;
    const u32 B = nitems / nb, j = idx / B, b = idx - j * B;
    return b * nb + j;
}
// Masked launches (ntt_io_t::limb_mask: rings of mixed modulus sizes, one launch per arithmetic policy): the items of the
// masked-in limbs numbered DENSELY.  Walking all items and skipping the others resonates with the walk's period -- with 4 or 6
// limbs per group some workgroups met none of their policy's items and others two to four times their share (the u64 transforms
// of the encrypted-MNIST products ran 4 x longer than their work).  mask bits lie below `limbs`; mask == 0: identity.
// An odd number of limbs is coprime to the walks' power-of-two periods and meets every limb evenly as it is: those launches keep
// the skipping walk (dense_mask returns 0; the reference's 7-limb ring measured 1 % slower with the dense numbering).
__device__ __forceinline__ u32 dense_mask(u32 limbs, u32 mask) { return (limbs & 1u) ? 0u : mask; }
__device__ __forceinline__ u32 dense_count(u32 nitems, int x, u32 limbs, u32 mask) {
    return mask ? ((((nitems >> x) / limbs) * (u32)__popc(mask)) << x) : nitems;
}
__device__ __forceinline__ u32 dense_item(u32 d, int x, u32 limbs, u32 mask) {
    if (!mask || d == ~0u) return d;
    const u32 nact = (u32)__popc(mask), sb = d & ((1u << x) - 1u), pld = d >> x, g = pld / nact, a = pld % nact;
    u32 m = mask;
    for (u32 i = 0; i < a; i++) m &= m - 1u;   // drop the a lowest set bits
    return ((g * limbs + (u32)__ffs((int)m) - 1u) << x) | sb;
}
template <class A, int LOGB, int LOGT, int IOMODE>
__global__ __launch_bounds__(1 << LOGT, TFHE_NTT_WAVES) void k_ntt_fwd_block(const u64* __restrict__ src, u64* __restrict__ dst,
                                                              const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int x,
                                                              u32 nitems, ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const size_t ntot = (size_t)1 << (LOGB + x);
    const u32 dmask = dense_mask((u32)sel.n, io.limb_mask), dn = dense_count(nitems, x, (u32)sel.n, dmask), niter = (dn + gridDim.x - 1) / gridDim.x;
    bool first = true;
    for (u32 it = 0; it < niter; it++) {
        const u32 item = dense_item(xcd_walk_item(it, blockIdx.x, gridDim.x, x, dn), x, (u32)sel.n, dmask);
        if (item == ~0u) continue;
        const u32 sb = item & ((1u << x) - 1), pl = item >> x;
        u32 srow = pl, drow = pl, j = pl % (u32)sel.n;
        lift_t lf;
        const lift_t* lift = nullptr;
        if constexpr (IOMODE == 1) {  // digit i of ciphertext b lifted into working limb j
            const u32 per_ct = io.level * io.nw, b = pl / per_ct, rem = pl % per_ct, i = rem / io.nw;
            j = rem % io.nw;
            srow = (b * io.polys + io.polys - 1) * io.level + i;
            const ntt_limb_t& Li = LT[sel.idx[i]];
            const ntt_limb_t& Lj = LT[sel.idx[j]];
            lf.qi = Li.q; lf.half = Li.q >> 1; lf.qj = Lj.q; lf.bj = Lj.br;
            if (io.lift_unsigned) lf.qi = lf.half = ~0ull;
            lift_wide_consts<A>(lf);
            lift = &lf;
        }
        if (io.limb_mask && !((io.limb_mask >> j) & 1u)) continue;  // the other policy's launch takes this limb
        const typename A::ctx C = A::make(LT[sel.idx[j]]);
        if (!first) __syncthreads();  // the previous item's last pass has read LDS
        first = false;
        fwd_schedule<A, LOGB, LOGT, 0>(lds, src + srow * ntot + ((size_t)sb << LOGB), dst + drow * ntot, C, fresh_tid(),
                                       (1u << x) + sb, x, brev_bits(sb, x), lift);
    }
}
template <class T>
__device__ __forceinline__ void pin_vgpr(T& x) { asm volatile("" : "+v"(x)); }
// make `idx` (an address component of later loads) depend on `v`: loads through read-only __restrict__ pointers are
// invariant to the compiler and float above TFHE_SCHED_FENCE, so a piece-wise load phase needs a data dependence to keep
// piece h+1's requests behind piece h's arithmetic (otherwise every piece is requested at once and spilled)
__device__ __forceinline__ void order_after(u32& idx, u64 v) { asm volatile("" : "+v"(idx) : "v"(v)); }
#define TFHE_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// Register-prefetch variant of the forward block kernel (whole rows, plain I/O): the operands of item i+1 are loaded
// into registers underneath the MIDDLE pass of item i, one load per few butterflies during its first half
// (row_prefetcher as the progress hook), so that the first pass of item i+1 starts without waiting on HBM.
// vmcnt is one in-order counter shared by loads and stores, hence the ordering rules:
//   * the middle pass's own twiddles are requested a pass ahead and have landed before the prefetch starts;
//   * the last pass's twiddle loads queue behind the prefetch (it has landed by then);
//   * everything is awaited BEFORE the stores of item i are issued -- a later wait would include their drain.
template <int LOGB, int LOGT>
struct row_prefetcher {
    static constexpr int E = 1 << (LOGB - LOGT);
    u64* raw;
    const u64* g;  // next row + tid
    bool on;
    __device__ __forceinline__ void operator()(int before, int after, int total) const {
        const int lo2 = 2 * E * before / total, hi2 = 2 * E * after / total;  // all requests in the first half of the pass
        const int lo = lo2 < E ? lo2 : E, hi = hi2 < E ? hi2 : E;
        if (hi > lo && on) {
#pragma unroll
            for (int i = lo; i < hi; i++) raw[i] = g[(size_t)i << LOGT];
        }
    }
};
template <class A, int LOGB, int LOGT>
__global__ __launch_bounds__(1 << LOGT, TFHE_NTT_WAVES) void k_ntt_fwd_pf(const u64* __restrict__ src, u64* __restrict__ dst,
                                                           const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nitems) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    static_assert(K3 >= 1 && pass_k_fwd(LOGB, LOGT, K1 + K2) == K3, "three-pass schedule expected");
    typedef pgeom<LOGB, LOGT, K1, K2> G2;
    typedef pgeom<LOGB, LOGT, K1 + K2, K3> G3;
    static_assert(G2::SETS == 1, "middle pass: one register set");
    constexpr int E = G3::E;
    u32 item = blockIdx.x;
    if (item >= nitems) return;
    u64 raw[E];
    fwd_load_data<LOGB, LOGT, 0, K1, true, false>(raw, lds, src + ((size_t)item << LOGB), threadIdx.x);
    for (bool first = true;; first = false) {
        const u32 tid = fresh_tid();
        const typename A::ctx C = A::make(LT[sel.idx[item % (u32)sel.n]]);
        u64* gdst = dst + ((size_t)item << LOGB);
        const u32 next = item + gridDim.x;
        if (!first) __syncthreads();  // the previous item's last pass has read LDS
        typename A::tw tw2[G2::SETS * G2::NTW];
        {
            typename A::elem v[E];
            if constexpr (TFHE_ES(10) != 0) fwd_load_tw<A, LOGB, LOGT, K1, K2, false>(tw2, C, tid, 1u);  // arrive underneath the exchange
            fwd_compute<A, LOGB, LOGT, 0, K1, true, false, 0>(v, raw, nullptr, C, tid, 1u, nullptr, no_hook(), TFHE_ES(10) ? lds : nullptr);
            if constexpr (TFHE_ES(10) == 0) {
                fwd_load_tw<A, LOGB, LOGT, K1, K2, false>(tw2, C, tid, 1u);
                fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
            }
        }
        __syncthreads();
        {
            u64 r2[E];
            typename A::elem v[E];
            fwd_load_data<LOGB, LOGT, K1, K2, false, false>(r2, lds, nullptr, tid);
#pragma unroll
            for (int i = 0; i < G2::SETS * G2::NTW; i++) pin_vgpr(tw2[i].w);  // landed before the prefetch starts
            const row_prefetcher<LOGB, LOGT> pf{raw, src + ((size_t)(next < nitems ? next : item) << LOGB) + tid, next < nitems};
            fwd_compute<A, LOGB, LOGT, K1, K2, false, false, K2, -1, row_prefetcher<LOGB, LOGT>>(v, r2, tw2, C, tid, 1u, nullptr, pf, TFHE_ES(11) ? lds : nullptr);
            if (!TFHE_ES(11)) fwd_store<A, LOGB, LOGT, K1, K2, false>(v, lds, nullptr, C, tid, 0, 0u);
        }
        __syncthreads();
        {
            u64 r3[E];
            typename A::elem v[E];
            fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(r3, lds, nullptr, tid);
            fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, r3, nullptr, C, tid, 1u);
            TFHE_SCHED_FENCE();
            TFHE_WAIT_VM0();  // the prefetched row is in registers; nothing older than the stores below is pending
#pragma unroll
            for (int i = 0; i < E; i++) pin_vgpr(raw[i]);
            fwd_store<A, LOGB, LOGT, K1 + K2, K3, true>(v, lds, gdst, C, tid, 0, 0u);
        }
        if (next >= nitems) break;
        item = next;
    }
}


// Digit-lift forward transforms of the key switch (ntt_io_t mode 1) with the source row read ONCE: item = (ciphertext b,
// digit i); the residues of limb i of c[end] stay in registers while the workgroup lifts them into each of the nw working
// limbs j in turn and transforms (rows (b*level + i)*nw + j of dst).  Whole-transform blocks only (x == 0).
template <class A, int LOGB, int LOGT>
__global__ __launch_bounds__(1 << LOGT, TFHE_NTT_WAVES) void k_ntt_fwd_lift(const u64* __restrict__ src, u64* __restrict__ dst,
                                                             const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nitems,
                                                             ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0);
    typedef pgeom<LOGB, LOGT, 0, K1> G1;
    const u32 tid = threadIdx.x;
    bool first = true;
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
        const u32 b = item / io.level, i = item % io.level;
        const u64* grow = src + ((size_t)((b * io.polys + io.polys - 1) * io.level + i) << LOGB);
        u64 raw[G1::E];
        fwd_load_data<LOGB, LOGT, 0, K1, true, false>(raw, lds, grow, tid);
        lift_t lf;
        lf.qi = LT[sel.idx[i]].q;
        lf.half = lf.qi >> 1;
        if (io.lift_unsigned) lf.qi = lf.half = ~0ull;   // ntt_io_t::lift_unsigned
        for (u32 j = 0; j < io.nw; j++) {
            if (io.limb_mask && !((io.limb_mask >> j) & 1u)) continue;  // the other policy's launch lifts into this limb
            const u32 tid = fresh_tid();
            const ntt_limb_t& Lj = LT[sel.idx[j]];
            lf.qj = Lj.q;
            lf.bj = Lj.br;
            lift_wide_consts<A>(lf);
            const typename A::ctx C = A::make(Lj);
            u64* gdst = dst + ((size_t)(item * io.nw + j) << LOGB);
            if (!first) __syncthreads();  // the previous transform's last pass has read LDS
            first = false;
            {
                typename A::elem v[G1::E];
                fwd_compute<A, LOGB, LOGT, 0, K1, true, false, 0>(v, raw, nullptr, C, tid, 1u, &lf);
                fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
            }
            __syncthreads();
            fwd_schedule<A, LOGB, LOGT, K1>(lds, nullptr, gdst, C, tid, 1u, 0, 0u, nullptr);
        }
    }
}

template <class A, int LOGB, int LOGT, int IOMODE>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_inv_block(const u64* __restrict__ src, u64* __restrict__ dst,
                                                              const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int x,
                                                              u32 nitems, ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const size_t ntot = (size_t)1 << (LOGB + x);
    const u32 dmask = dense_mask((u32)sel.n, io.limb_mask), dn = dense_count(nitems, x, (u32)sel.n, dmask), niter = (dn + gridDim.x - 1) / gridDim.x;
    bool first = true;
    for (u32 it = 0; it < niter; it++) {
        const u32 item = dense_item(xcd_walk_item(it, blockIdx.x, gridDim.x, x, dn), x, (u32)sel.n, dmask);
        if (item == ~0u) continue;
        const u32 sb = item & ((1u << x) - 1), pl = item >> x;
        u32 srow = pl, drow = pl, j = pl % (u32)sel.n;
        const u64* addend = nullptr;
        if constexpr (IOMODE == 2) {
            const u32 g = pl / io.gsz, w = pl % io.gsz;
            srow = g * io.src_gstride + w;
            drow = g * io.dst_gstride + w;
            j = w % (u32)sel.n;
            if (w < io.add_rows) addend = io.addend + (size_t)(g * io.add_gstride + w) * ntot;
        }
        if (io.limb_mask && !((io.limb_mask >> j) & 1u)) continue;  // the other policy's launch takes this limb
        const typename A::ctx C = A::make(LT[sel.idx[j]]);
        if (!first) __syncthreads();
        first = false;
        if (A::prefetch_tw && x == 0) {  // fp64 policy, whole transform: 8-byte twiddles prefetched a pass ahead
            if constexpr (A::prefetch_tw)
                inv_schedule_ptw<A, LOGB, LOGT, LOGB>(lds, src + srow * ntot, dst + drow * ntot, C, fresh_tid(), 1u, addend, nullptr);
        } else {
            if (x == 0)
                inv_schedule<A, LOGB, LOGT, LOGB, true>(lds, src + srow * ntot, dst + drow * ntot, C, threadIdx.x, 1u, 0, 0u, addend);
            else
                inv_schedule<A, LOGB, LOGT, LOGB, false>(lds, src + srow * ntot, dst + drow * ntot + ((size_t)sb << LOGB), C,
                                                         threadIdx.x, (1u << x) + sb, x, brev_bits(sb, x), nullptr);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Staged variant (whole-transform blocks, x == 0): the workgroup keeps walking items, and while it runs
// the LAST pass of item i out of registers the LDS is idle -- so the 128 KiB row of item i+1 is copied
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write) underneath that pass.  The first
// pass of item i+1 then reads its operands from LDS (lane-linear image of the row: conflict-free for the
// stride-1-across-lanes first-pass patterns) instead of waiting on HBM.  Order inside an item:
//     barrier | pass 1 (operands from the staged row) | barrier | pass-1 results -> LDS | ... | pass-3
//     operands -> registers | barrier | LDS-DMA of the next row | pass-3 butterflies | wait for the DMA |
//     stores of item i  (they drain underneath passes 1-2 of item i+1)
// vmcnt is one in-order counter for loads, so every ordinary vector load whose result is needed while the
// DMA is in flight (last-pass twiddles, addends) is issued and waited for BEFORE the DMA is started.
// ------------------------------------------------------------------------------------------------
#ifdef TFHE_TRACE  // design aid (tools/ntt_ablate.hip): shader-clock stamps of one workgroup's phases
__device__ unsigned long long tfhe_trace[64 * 16];
#define TFHE_STAMP(k)                                                                                   \
    do {                                                                                                \
        if (blockIdx.x == (TFHE_TRACE) && threadIdx.x == 0 && itc < 64) {                               \
            tfhe_trace[itc * 16 + (k)] = __builtin_amdgcn_s_memtime();                                  \
            if ((k) == 0) tfhe_trace[itc * 16 + 15] = __builtin_amdgcn_s_memrealtime();                 \
        }                                                                                               \
    } while (0)
#else
#define TFHE_STAMP(k) ((void)0)
#endif

__device__ __forceinline__ void glds16(const void* gsrc, u32 lds_byte) {  // one 1-KiB wave-wide piece
#ifdef TFHE_ABL_NOMEM
    return;
#endif
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte)
                 : "memory");
}
// Streaming load phase of the kernels that form a sub-block's first-pass operands from NSRC strided parts of a source row
// (the 2^15 / 2^16 transforms: halves / quarters at distance 2^LOGB words).  While those operands are being formed the LDS
// holds no row image, so the parts stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs) and are read back from
// there, a few KiB per wave in flight while the arithmetic of the current step runs -- instead of a few rounds of (register
// loads, wait, arithmetic) with every round's latency exposed.  WAVE-PRIVATE: a wave copies exactly the words its own lanes
// will read (register position e of lane l is row word wave*64 + l + (e << LOGT) of each part: 512 contiguous bytes per wave,
// so one 1-KiB DMA instruction carries two such segments, lanes 0-31 one and lanes 32-63 the other), into its own ring of
// RING steps -- no barrier inside the phase, only vmcnt waits (the counter is in order: waiting for all but the youngest
// k instructions).  consume(e, q) receives the NSRC words of register position e.  The caller puts a barrier between the
// previous LDS readers and this call, and one before the next LDS writer.
template <int LOGB, int LOGT, int NSRC, int EC, class F>
__device__ __forceinline__ void dma_stream_load(u64* lds, const u64* s, u32 tid, F&& consume) {
    constexpr int E = 1 << (LOGB - LOGT), NSTEP = E / EC, RING = 4, AHEAD = RING - 1;
    constexpr int IPS = EC * NSRC / 2;                                    // DMA instructions per step (two segments each)
    constexpr int WAVES = 1 << (LOGT - 6);
    static_assert((EC * NSRC) % 2 == 0 && NSTEP >= AHEAD, "step geometry");
    static_assert(WAVES * RING * IPS * 128 <= (int)lds_words<LOGB, LOGT>(), "the rings fit the row image");
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63u, hl = lane >> 5, l32 = lane & 31u;
    const u32 ring_w = wave * (RING * IPS * 128u);                        // this wave's ring, in words
    const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) u64*)lds + (ring_w << 3);
    const u64* sw = s + wave * 64u + l32 * 2u;
    auto issue = [&](int c) {
#pragma unroll
        for (int i = 0; i < IPS; i++) {
            const u32 sg = 2u * (u32)i + hl, r = sg / NSRC, k = sg % NSRC;   // lanes 0-31: segment 2i, lanes 32-63: segment 2i+1
            const u64* g = sw + ((size_t)k << LOGB) + ((size_t)(c * EC + (int)r) << LOGT);
            glds16(g, lds0 + ((u32)((c % RING) * IPS + i) << 10));
        }
    };
#pragma unroll
    for (int c = 0; c < AHEAD; c++) issue(c);
#pragma unroll
    for (int c = 0; c < NSTEP; c++) {
        const int later = (NSTEP - 1 - c) < (AHEAD - 1) ? (NSTEP - 1 - c) : (AHEAD - 1);   // steps issued after c and still in flight
        if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPS) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u64 q[EC][NSRC];
        const u64* buf = lds + ring_w + (size_t)(c % RING) * IPS * 128 + lane;
#pragma unroll
        for (int r = 0; r < EC; r++)
#pragma unroll
            for (int k = 0; k < NSRC; k++) {
                constexpr int dummy = 0; (void)dummy;
                const int sg = r * NSRC + k;
                q[r][k] = buf[(sg >> 1) * 128 + (sg & 1) * 64];
            }
        if (c + AHEAD < NSTEP) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the slot of step c-1 ... c is read before it is refilled
            issue(c + AHEAD);
        }
#pragma unroll
        for (int r = 0; r < EC; r++) consume(c * EC + r, q[r]);
    }
}
// LDS-DMA of one 2^LOGB-word row into the lane-linear LDS image, PER_WAVE 1-KiB pieces per wave; as a progress hook
// it issues piece k once the butterfly count passes k/PER_WAVE of the pass
template <int LOGB, int LOGT>
struct row_stager {
    static constexpr int PER_WAVE = ((8 << LOGB) / 1024) >> (LOGT - 6);
    const char* g;
    u32 l0;
    bool on;
    __device__ __forceinline__ row_stager(u64* lds, const u64* grow, u32 tid, bool enable) {
        const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63u;
        l0 = (u32)(size_t)(__attribute__((address_space(3))) u64*)lds + wave * (PER_WAVE * 1024u);
        g = (const char*)grow + (size_t)wave * (PER_WAVE * 1024) + lane * 16;
        on = enable;
    }
    __device__ __forceinline__ void piece(int i) const { glds16(g + i * 1024, l0 + (u32)i * 1024u); }
    __device__ __forceinline__ void all() const {
#pragma unroll
        for (int i = 0; i < PER_WAVE; i++) piece(i);
    }
    __device__ __forceinline__ void operator()(int before, int after, int total) const {
        const int lo = PER_WAVE * before / total, hi = PER_WAVE * after / total;
        if (hi > lo && on) {
#pragma unroll
            for (int i = lo; i < hi; i++) piece(i);
        }
    }
};

struct item_rows_t {
    u32 srow, drow, j;
    bool has_add;
    u32 arow;
};
template <int IOMODE>
__device__ __forceinline__ item_rows_t item_rows(u32 pl, const limb_sel_t& sel, const ntt_io_t& io, lift_t& lf,
                                                 const ntt_limb_t* LT) {
    item_rows_t r{pl, pl, pl % (u32)sel.n, false, 0u};
    if constexpr (IOMODE == 1) {
        const u32 per_ct = io.level * io.nw, b = pl / per_ct, rem = pl % per_ct, i = rem / io.nw;
        r.j = rem % io.nw;
        r.srow = (b * io.polys + io.polys - 1) * io.level + i;
        const ntt_limb_t& Li = LT[sel.idx[i]];
        const ntt_limb_t& Lj = LT[sel.idx[r.j]];
        lf.qi = Li.q; lf.half = Li.q >> 1; lf.qj = Lj.q; lf.bj = Lj.br;
        if (io.lift_unsigned) lf.qi = lf.half = ~0ull;
    } else if constexpr (IOMODE == 2) {
        const u32 g = pl / io.gsz, w = pl % io.gsz;
        r.srow = g * io.src_gstride + w;
        r.drow = g * io.dst_gstride + w;
        r.j = w % (u32)sel.n;
        r.has_add = w < io.add_rows;
        r.arow = g * io.add_gstride + w;
    }
    return r;
}

template <class A, int LOGB, int LOGT, int IOMODE>
__global__ __launch_bounds__(1 << LOGT, IOMODE == 2 ? 2 : TFHE_NTT_WAVES) void k_ntt_inv_staged(const u64* __restrict__ src, u64* __restrict__ dst,
                                                               const ntt_limb_t* __restrict__ LT, limb_sel_t sel,
                                                               u32 nitems, ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_inv(LOGB, LOGT, LOGB), K2 = pass_k_inv(LOGB, LOGT, LOGB - K1), K3 = LOGB - K1 - K2;
    static_assert(K3 >= 1 && pass_k_inv(LOGB, LOGT, K3) == K3, "three-pass schedule expected");
    typedef pgeom<LOGB, LOGT, 0, K3> G3;
    constexpr int E = G3::E;
    const u32 tid = threadIdx.x;
    u32 item = blockIdx.x;
    if (item >= nitems) return;
    lift_t lf;
    item_rows_t R = item_rows<IOMODE>(item, sel, io, lf, LT);
    row_stager<LOGB, LOGT>(lds, src + ((size_t)R.srow << LOGB), tid, true).all();
    TFHE_WAIT_VM0();
    u32 itc = 0;
    (void)itc;
    for (;;) {
        const u32 tid = IOMODE == 2 ? threadIdx.x : fresh_tid();  // (the addend variant measured faster with hoisted addresses)
        const typename A::ctx C = A::make(LT[sel.idx[R.j]]);
        u64* gdst = dst + ((size_t)R.drow << LOGB);
        __syncthreads();
        TFHE_STAMP(0);
        {
            u64 raw[E];
            typename A::elem v[E];
            inv_load_data<LOGB, LOGT, LOGB - K1, K1, true>(raw, lds, lds, tid, 0, 0u);
            inv_compute<A, LOGB, LOGT, LOGB - K1, K1, true, true, 0>(v, raw, nullptr, C, tid, 1u);
            __syncthreads();
            TFHE_STAMP(1);
            inv_store<A, LOGB, LOGT, LOGB - K1, K1, true, true>(v, lds, nullptr, C, tid);
        }
        __syncthreads();
        TFHE_STAMP(2);
        ntt_inv_pass<A, LOGB, LOGT, K3, K2, false, false, true, TFHE_ES(5) != 0>(lds, nullptr, nullptr, C, tid, 1u, 0, 0u);
        __syncthreads();
        TFHE_STAMP(3);
        const u32 next = item + gridDim.x;
        {
            u64 raw[E], add[IOMODE == 2 ? E : 1];
            typename A::elem v[E];
            inv_load_data<LOGB, LOGT, 0, K3, false>(raw, lds, nullptr, tid, 0, 0u);
            const bool has_add = IOMODE == 2 && R.has_add;
            if constexpr (IOMODE == 2) {
                if (has_add) {
                    const u64* ap = io.addend + ((size_t)R.arow << LOGB);
#pragma unroll
                    for (int r = 0; r < E; r++) add[r] = ap[tid + ((u32)r << LOGT)];
#pragma unroll
                    for (int r = 0; r < E; r++) pin_vgpr(add[r]);
                }
            }
#pragma unroll
            for (int i = 0; i < E; i++) pin_vgpr(raw[i]);
            __syncthreads();  // LDS is free
            TFHE_STAMP(4);
            item_rows_t Rn = R;
            if (next < nitems) Rn = item_rows<IOMODE>(next, sel, io, lf, LT);
            const row_stager<LOGB, LOGT> stager(lds, src + ((size_t)Rn.srow << LOGB), tid, next < nitems);
            inv_compute<A, LOGB, LOGT, 0, K3, false, true, 0, -1, row_stager<LOGB, LOGT>>(v, raw, nullptr, C, tid, 1u, stager);
            u64 o[E];
#pragma unroll
            for (int i = 0; i < E; i++) o[i] = A::out_inv_scaled(v[i], C);
            if constexpr (IOMODE == 2) {
                if (has_add) {
#pragma unroll
                    for (int i = 0; i < E; i++) o[i] = addmod(o[i], add[i], C.q);
                }
            }
            TFHE_SCHED_FENCE();
            TFHE_STAMP(6);
            TFHE_WAIT_VM0();
            TFHE_STAMP(7);
            static_assert(G3::SETS == 1 && G3::LO == LOGT, "last inverse pass: one register set, stride 2^LOGT");
#pragma unroll
            for (int r = 0; r < E; r++) gdst[tid + ((u32)r << LOGT)] = o[r];
            TFHE_STAMP(8);
            R = Rn;
        }
        if (next >= nitems) break;
        item = next;
        itc++;
    }
}

// ------------------------------------------------------------------------------------------------
// N = 2^(LOGB+1) in ONE kernel (fp64 policy): the workgroup takes a whole row, does the top stage in registers and runs
// the two 2^LOGB sub-blocks one after the other, holding the waiting half in registers (64 VGPRs) -- one read and one
// write of the row instead of the extra pass of k_ntt_*_top plus strided block I/O.
//   forward: a = lo + W[1] hi -> block 0,  b = lo - W[1] hi -> block 1; block outputs are interleaved in natural order
//            (position 2 nat + sb).
//   inverse: sub-block sb reads words 2 nat + sb; out_lo = (r0 + r1) N^-1, out_hi = (r0 - r1) W[1]^-1 N^-1.
// ------------------------------------------------------------------------------------------------
// LIFT: the rows are key-switch digits (ntt_io_t mode 1): item (b, i, j) reads limb i of c[end] of ciphertext b and lifts
// it, centred, into working limb j while loading (rlwe_she.jl:326-329) -- the digit rows are never stored untransformed.
template <class A, int LOGB, int LOGT, bool LIFT = false>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_fwd_pair(const u64* __restrict__ src, u64* __restrict__ dst,
                                                             const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nitems,
                                                             ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    static_assert(K3 >= 1 && pass_k_fwd(LOGB, LOGT, K1 + K2) == K3, "three-pass schedule expected");
    typedef pgeom<LOGB, LOGT, K1 + K2, K3> G3;
    constexpr int E = G3::E;
    const u32 tid = threadIdx.x;
    bool first = true;
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
        u32 srow = item, j = item % (u32)sel.n;
        lift_t lf;
        if constexpr (LIFT) {
            const u32 per_ct = io.level * io.nw, b = item / per_ct, rem = item % per_ct, i = rem / io.nw;
            j = rem % io.nw;
            srow = (b * io.polys + io.polys - 1) * io.level + i;
            const ntt_limb_t& Li = LT[sel.idx[i]];
            const ntt_limb_t& Lj = LT[sel.idx[j]];
            lf.qi = Li.q; lf.half = Li.q >> 1; lf.qj = Lj.q; lf.bj = Lj.br;
            if (io.lift_unsigned) lf.qi = lf.half = ~0ull;
            lift_wide_consts<A>(lf);
        }
        if (io.limb_mask && !((io.limb_mask >> j) & 1u)) continue;  // the other policy's launch takes this limb
        const typename A::ctx C = A::make(LT[sel.idx[j]]);
        const u64* s = src + ((size_t)srow << (LOGB + 1));
        u64* d = dst + ((size_t)item << (LOGB + 1));
        u64 wa[E], wb[E];  // the two sub-blocks' first-pass operands (element bits)
        {
            // (dma_stream_load, which pays in k_ntt_fwd_quad with its four rounds of loads, measured 10 % SLOWER here -- 2.2
            // against 2.45 TB/s: two rounds of register loads keep nearly the whole row in flight already)
            const typename A::tw w1 = A::ld_fwd(C, 1u);
#pragma unroll
            for (int h = 0; h < 2; h++) {  // two halves: bounds the raw operands in flight next to the 128 result registers
#pragma unroll
                for (int r = h * (E / 2); r < (h + 1) * (E / 2); r++) {
                    const u32 k = tid + ((u32)r << LOGT);
                    // lifted digits enter loosely (|v| <= p, ArithFp::from_global_lift): lo + t <= 1.88 p before the reduction
                    const double lo = LIFT ? A::from_global_lift(s[k], C, lf, true) : fp_from_u64(s[k]);
                    const double hv = LIFT ? A::from_global_lift(s[k + (1u << LOGB)], C, lf, true) : fp_from_u64(s[k + (1u << LOGB)]);
                    const double t = fp_mulmod_c(hv, w1, C.p, C.pinv);
                    wa[r] = A::to_lds(fp_reduce(lo + t, C.p, C.pinv));
                    wb[r] = A::to_lds(fp_reduce(lo - t, C.p, C.pinv));
                }
                TFHE_SCHED_FENCE();
            }
        }
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
            const u32 pre = 2u + (u32)sb;
            if (!first) __syncthreads();  // the previous transform's last pass has read LDS
            first = false;
            {
                typename A::elem v[E];
                fwd_compute<A, LOGB, LOGT, 0, K1, false, false, 0>(v, sb ? wb : wa, nullptr, C, tid, pre);
                fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
            }
            __syncthreads();
            ntt_fwd_pass<A, LOGB, LOGT, K1, K2, false, false, TFHE_ES(2) != 0>(lds, nullptr, nullptr, C, tid, pre, 0, 0u);
            __syncthreads();
            {
                u64 r3[E];
                typename A::elem v[E];
                fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(r3, lds, nullptr, tid);
                fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, r3, nullptr, C, tid, pre);
                // natural-order position 2 nat + sb: the two sub-blocks fill alternate words of the same lines (L2 merges them)
                fwd_store<A, LOGB, LOGT, K1 + K2, K3, true>(v, lds, d, C, tid, 1, (u32)sb);
            }
        }
    }
}
template <class A, int LOGB, int LOGT>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_inv_pair(const u64* __restrict__ src, u64* __restrict__ dst,
                                                             const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nitems,
                                                             u32 limb_mask) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_inv(LOGB, LOGT, LOGB), S1 = LOGB - K1;
    typedef pgeom<LOGB, LOGT, S1, K1> G1;
    constexpr int E = G1::E;
    constexpr int K2 = pass_k_inv(LOGB, LOGT, S1), KL = S1 - K2;  // middle and last pass widths
    static_assert(KL >= 1 && pass_k_inv(LOGB, LOGT, KL) == KL, "three-pass schedule expected");
    const u32 tid = threadIdx.x;
    bool first = true;
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
        if (limb_mask && !((limb_mask >> (item % (u32)sel.n)) & 1u)) continue;
        const typename A::ctx C = A::make(LT[sel.idx[item % (u32)sel.n]]);
        const u64* s = src + ((size_t)item << (LOGB + 1));
        u64* d = dst + ((size_t)item << (LOGB + 1));
        double res0[E];
#pragma unroll
        for (int sb = 0; sb < 2; sb++) {
            const u32 pre = 2u + (u32)sb;
            if (!first) __syncthreads();
            first = false;
            {
                u64 raw[E];  // sub-block sb = words 2 nat + sb of the row (the other sub-block's pass re-reads the lines from L2)
                typename A::elem v[E];
                inv_load_data<LOGB, LOGT, S1, K1, true>(raw, lds, s, tid, 1, (u32)sb);
                inv_compute<A, LOGB, LOGT, S1, K1, true, false, 0>(v, raw, nullptr, C, tid, pre);
                inv_store<A, LOGB, LOGT, S1, K1, true, false>(v, lds, nullptr, C, tid);
            }
            __syncthreads();
            ntt_inv_pass<A, LOGB, LOGT, KL, K2, false, false, false, TFHE_ES(7) != 0>(lds, nullptr, nullptr, C, tid, pre, 0, 0u);
            __syncthreads();
            {
                u64 r3[E];
                typename A::elem v[E];
                inv_load_data<LOGB, LOGT, 0, KL, false>(r3, lds, nullptr, tid, 0, 0u);
                inv_compute<A, LOGB, LOGT, 0, KL, false, false, 0>(v, r3, nullptr, C, tid, pre);
                if (sb == 0) {
#pragma unroll
                    for (int i = 0; i < E; i++) res0[i] = fp_reduce(v[i], C.p, C.pinv);
                } else {  // top stage of the 2^(LOGB+1) transform with N^-1 folded in, then natural-order stores
#pragma unroll
                    for (int r = 0; r < E; r++) {
                        const double y = fp_reduce(v[r], C.p, C.pinv);
                        const double a = res0[r] + y, dd = res0[r] - y;
                        const u32 j = tid + ((u32)r << LOGT);
                        d[j] = fp_canon(fp_mulmod_c(a, C.ninv, C.p, C.pinv), C.p, C.pinv);
                        d[j + (1u << LOGB)] = fp_canon(fp_mulmod_c(dd, C.w1n, C.p, C.pinv), C.p, C.pinv);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Sub-block kernels for N = 2^(LOGB+x), x >= 2 (after / before the k_ntt_*_top stages): one workgroup runs the TWO
// sub-blocks sb and sb + 2^(x-1) of a row one after the other.  Their words are neighbours on the natural-order side
// (positions (nat << x) + brv_x(sb) and + 1), so that side moves as 16-byte pieces -- half as many partial-line
// transactions as one sub-block per workgroup, and a line is shared by 2^(x-1) workgroups (of one XCD: xcd_walk_item).
//   forward: the first sub-block's canonical outputs wait in registers (64 VGPRs) for the second's;
//   inverse: both sub-blocks' inputs are read together, the second's wait in registers.
// ------------------------------------------------------------------------------------------------
template <class A, int LOGB, int LOGT>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_fwd_subpair(const u64* __restrict__ src, u64* __restrict__ dst,
                                                                const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int x,
                                                                u32 nitems, u32 limb_mask) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    static_assert(K3 >= 1 && pass_k_fwd(LOGB, LOGT, K1 + K2) == K3, "three-pass schedule expected");
    typedef pgeom<LOGB, LOGT, K1 + K2, K3> G3;
    constexpr int E = G3::E;
    const size_t ntot = (size_t)1 << (LOGB + x);
    const u32 niter = (nitems + gridDim.x - 1) / gridDim.x, hmask = (1u << (x - 1)) - 1u;
    bool first = true;
    for (u32 it = 0; it < niter; it++) {
        const u32 item = xcd_walk_item(it, blockIdx.x, gridDim.x, x - 1, nitems);
        if (item == ~0u) continue;
        const u32 ph = item & hmask, pl = item >> (x - 1);
        if (limb_mask && !((limb_mask >> (pl % (u32)sel.n)) & 1u)) continue;
        const typename A::ctx C = A::make(LT[sel.idx[pl % (u32)sel.n]]);
        u64* d = dst + pl * ntot + brev_bits(ph, x);  // even word offset: 16-byte aligned pieces
        u64 held[E];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 sb = ph + ((u32)half << (x - 1)), pre = (1u << x) + sb;
            const u64* g = src + pl * ntot + ((size_t)sb << LOGB);
            const u32 tid = fresh_tid();
            if (!first) __syncthreads();  // the previous transform's last pass has read LDS
            first = false;
            ntt_fwd_pass<A, LOGB, LOGT, 0, K1, true, false, TFHE_ES(4) != 0>(lds, g, nullptr, C, tid, pre, 0, 0u);
            __syncthreads();
            ntt_fwd_pass<A, LOGB, LOGT, K1, K2, false, false, TFHE_ES(4) != 0>(lds, nullptr, nullptr, C, tid, pre, 0, 0u);
            __syncthreads();
            {
                u64 r3[E];
                typename A::elem v[E];
                fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(r3, lds, nullptr, tid);
                fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, r3, nullptr, C, tid, pre);
#pragma unroll
                for (int u = 0; u < G3::SETS; u++) {
                    u32 c0, hi, base;
                    G3::template coords<true>(tid, u, c0, hi, base);
#pragma unroll
                    for (int r = 0; r < G3::R; r++) {
                        const int e = u * G3::R + r;
                        const u64 o = A::out_fwd(v[e], C);
                        if (half == 0) {
                            held[e] = o;
                        } else {
                            const u32 nat = (brev_bits((u32)r, K3) << (LOGB - K3)) + c0;
                            u64x2_t w;
                            w.x = held[e];
                            w.y = o;
                            *(u64x2_t*)(d + ((u64)nat << x)) = w;
                        }
                    }
                }
            }
        }
    }
}
// N = 2^(LOGB+2), forward, fp64 policy, ONE kernel: the workgroup reads all four quarters of the row, forms the first-pass
// operands of its two sub-blocks (sb = ph and ph + 2: two of the four outputs of the two top stages, 4 modular products
// per point instead of the 2 a full radix-4 would spend on them) and runs them as in k_ntt_fwd_subpair.  The row's other
// workgroup (same XCD, same iteration: xcd_walk_item) reads the same lines from L2 -- one HBM read and one write of the
// row instead of two of each with k_ntt_fwd_top in front.
// LIFT: key-switch digit rows (ntt_io_t mode 1), lifted while loading as in k_ntt_fwd_pair.
template <class A, int LOGB, int LOGT, bool LIFT = false>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_fwd_quad(const u64* __restrict__ src, u64* __restrict__ dst,
                                                             const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nitems,
                                                             ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int x = 2;
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    static_assert(K3 >= 1 && pass_k_fwd(LOGB, LOGT, K1 + K2) == K3, "three-pass schedule expected");
    typedef pgeom<LOGB, LOGT, K1 + K2, K3> G3;
    constexpr int E = G3::E;
    const size_t ntot = (size_t)1 << (LOGB + x);
    const u32 dmask = dense_mask((u32)sel.n, io.limb_mask), dn = dense_count(nitems, 1, (u32)sel.n, dmask), niter = (dn + gridDim.x - 1) / gridDim.x;
    bool first = true;
    for (u32 it = 0; it < niter; it++) {
        const u32 item = dense_item(xcd_walk_item(it, blockIdx.x, gridDim.x, x - 1, dn), 1, (u32)sel.n, dmask);
        if (item == ~0u) continue;
        const u32 ph = item & 1u, pl = item >> 1;
        u32 srow = pl, j = pl % (u32)sel.n;
        lift_t lf;
        if constexpr (LIFT) {
            const u32 per_ct = io.level * io.nw, b = pl / per_ct, rem = pl % per_ct, i = rem / io.nw;
            j = rem % io.nw;
            srow = (b * io.polys + io.polys - 1) * io.level + i;
            const ntt_limb_t& Li = LT[sel.idx[i]];
            const ntt_limb_t& Lj = LT[sel.idx[j]];
            lf.qi = Li.q; lf.half = Li.q >> 1; lf.qj = Lj.q; lf.bj = Lj.br;
            if (io.lift_unsigned) lf.qi = lf.half = ~0ull;  // no centring, and never `loose` (from_global_lift)
            lift_wide_consts<A>(lf);
        }
        if (io.limb_mask && !((io.limb_mask >> j) & 1u)) continue;  // the other policy's launch takes this limb
        const typename A::ctx C = A::make(LT[sel.idx[j]]);
        const u64* s = src + srow * ntot;
        u64* d = dst + pl * ntot + brev_bits(ph, x);
        u64 w[2][E];  // the two sub-blocks' first-pass operands (element bits)
        {
            // load phase: the four quarters stream through the LDS (dma_stream_load), top-stage arithmetic per chunk
            const u32 tid = fresh_tid();
            const typename A::tw w1 = A::ld_fwd(C, 1u), w2 = A::ld_fwd(C, 2u), w3 = A::ld_fwd(C, 3u);
            const double sgn = ph ? -1.0 : 1.0;
            if (!first) __syncthreads();  // the previous item's last pass has read its LDS image
            dma_stream_load<LOGB, LOGT, 4, 2>(lds, s, tid, [&](int e, const u64* q) {
                double xin[4];
#pragma unroll
                for (int k = 0; k < 4; k++) xin[k] = LIFT ? A::from_global_lift(q[k], C, lf, true) : fp_from_u64(q[k]);
                const double x0 = xin[0], x1 = xin[1];
                const double t2 = fp_mulmod_c(xin[2], w1, C.p, C.pinv);
                const double t3 = fp_mulmod_c(xin[3], w1, C.p, C.pinv);
                // |y| <= 1.88 p, products <= 1.21 p (fp64arith.h); the sub-blocks take reduced operands
                const double u0 = fp_mulmod_c(x1 + t3, w2, C.p, C.pinv), u1 = fp_mulmod_c(x1 - t3, w3, C.p, C.pinv);
                w[0][e] = A::to_lds(fp_reduce(fp_fma(sgn, u0, x0 + t2), C.p, C.pinv));
                w[1][e] = A::to_lds(fp_reduce(fp_fma(sgn, u1, x0 - t2), C.p, C.pinv));
            });
            first = false;
        }
        u64 held[E];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 sb = ph + ((u32)half << (x - 1)), pre = (1u << x) + sb;
            const u32 tid = fresh_tid();
            __syncthreads();  // every wave has read the last chunk / the previous transform's last pass has read LDS
            {
                typename A::elem v[E];
                fwd_compute<A, LOGB, LOGT, 0, K1, false, false, 0>(v, w[half], nullptr, C, tid, pre, nullptr, no_hook(), TFHE_ES(1) ? lds : nullptr);
                if (!TFHE_ES(1)) fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
            }
            __syncthreads();
            ntt_fwd_pass<A, LOGB, LOGT, K1, K2, false, false, TFHE_ES(0) != 0>(lds, nullptr, nullptr, C, tid, pre, 0, 0u);
            __syncthreads();
            static_assert(G3::SETS == 2, "last pass: two register sets");
            // one register set at a time: halves the transient next to the 64 held / waiting registers
            auto last_set = [&](auto uc) {
                constexpr int U = decltype(uc)::value;
                u64 r3[E];
                typename A::elem v[E];
                fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true, U>(r3, lds, nullptr, tid);
                fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0, U>(v, r3, nullptr, C, tid, pre);
                u32 c0, hi, base;
                G3::template coords<true>(tid, U, c0, hi, base);
#pragma unroll
                for (int r = 0; r < G3::R; r++) {
                    const int e = U * G3::R + r;
                    const u64 o = A::out_fwd(v[e], C);
                    if (half == 0) {
                        held[e] = o;
                    } else {
                        const u32 nat = (brev_bits((u32)r, K3) << (LOGB - K3)) + c0;
                        u64x2_t ww;
                        ww.x = held[e];
                        ww.y = o;
                        *(u64x2_t*)(d + ((u64)nat << x)) = ww;
                    }
                }
            };
            last_set(std::integral_constant<int, 0>());
            last_set(std::integral_constant<int, 1>());
        }
    }
}
template <class A, int LOGB, int LOGT>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_inv_subpair(const u64* __restrict__ src, u64* __restrict__ dst,
                                                                const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int x,
                                                                u32 nitems, u32 limb_mask) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_inv(LOGB, LOGT, LOGB), S1 = LOGB - K1;
    typedef pgeom<LOGB, LOGT, S1, K1> G1;
    constexpr int E = G1::E;
    constexpr int K2 = pass_k_inv(LOGB, LOGT, S1), KL = S1 - K2;  // middle and last pass widths
    static_assert(KL >= 1 && pass_k_inv(LOGB, LOGT, KL) == KL, "three-pass schedule expected");
    const size_t ntot = (size_t)1 << (LOGB + x);
    const u32 niter = (nitems + gridDim.x - 1) / gridDim.x, hmask = (1u << (x - 1)) - 1u;
    bool first = true;
    for (u32 it = 0; it < niter; it++) {
        const u32 item = xcd_walk_item(it, blockIdx.x, gridDim.x, x - 1, nitems);
        if (item == ~0u) continue;
        const u32 ph = item & hmask, pl = item >> (x - 1);
        if (limb_mask && !((limb_mask >> (pl % (u32)sel.n)) & 1u)) continue;
        const typename A::ctx C = A::make(LT[sel.idx[pl % (u32)sel.n]]);
        const u64* s = src + pl * ntot + brev_bits(ph, x);
        u64 raw[2][E];
        {
            const u32 tid = fresh_tid();
#pragma unroll
            for (int u = 0; u < G1::SETS; u++) {
                u32 c0, hi, base;
                G1::template coords<true>(tid, u, c0, hi, base);
#pragma unroll
                for (int r = 0; r < G1::R; r++) {
                    const u32 nat = (brev_bits((u32)r, K1) << (LOGB - K1)) + c0;
                    const u64x2_t w = *(const u64x2_t*)(s + ((u64)nat << x));
                    raw[0][u * G1::R + r] = w.x;
                    raw[1][u * G1::R + r] = w.y;
                }
            }
            TFHE_SCHED_FENCE();
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 sb = ph + ((u32)half << (x - 1)), pre = (1u << x) + sb;
            u64* g = dst + pl * ntot + ((size_t)sb << LOGB);
            const u32 tid = fresh_tid();
            if (!first) __syncthreads();
            first = false;
            {
                typename A::elem v[E];
                inv_compute<A, LOGB, LOGT, S1, K1, true, false, 0>(v, raw[half], nullptr, C, tid, pre);
                inv_store<A, LOGB, LOGT, S1, K1, true, false>(v, lds, nullptr, C, tid);
            }
            __syncthreads();
            ntt_inv_pass<A, LOGB, LOGT, KL, K2, false, false, false, TFHE_ES(8) != 0>(lds, nullptr, nullptr, C, tid, pre, 0, 0u);
            __syncthreads();
            ntt_inv_pass<A, LOGB, LOGT, 0, KL, false, true, false>(lds, nullptr, g, C, tid, pre, 0, 0u);
        }
    }
}

// N = 2^(LOGB+2) inverse transform in ONE kernel, one row per workgroup pass: the four 2^LOGB sub-blocks are inverse-transformed
// two at a time (sub-blocks ph and ph + 2 share their natural-order 16-byte pieces, as k_ntt_inv_subpair), their results --
// reduced doubles -- parked in a per-workgroup scratch row (4 x 2^LOGB words; a thread reads back exactly the words it
// wrote, so there is no exchange between threads and no synchronisation: the scratch only extends the register file, and at
// 512 KiB per workgroup it lives in the L2 / Infinity Cache), then every thread runs the two top stages on its columns
// (4 sub-block values per column: two Gentleman-Sande stages, N^-1 folded into the last) and stores the row.  HBM traffic is
// one read and one write of the row (2 N 8 bytes) instead of the 4 N 8 of k_ntt_inv_subpair + k_ntt_inv_top<2>; a row is
// read completely before any of it is written, so in place is fine.
template <class A, int LOGB, int LOGT>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_inv_quad(const u64* __restrict__ src, u64* __restrict__ dst, u64* __restrict__ scratch,
                                                             const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nrows, u32 limb_mask) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int X = 2;
    constexpr int K1 = pass_k_inv(LOGB, LOGT, LOGB), S1 = LOGB - K1;
    typedef pgeom<LOGB, LOGT, S1, K1> G1;
    constexpr int E = G1::E;
    constexpr int K2 = pass_k_inv(LOGB, LOGT, S1), KL = S1 - K2;  // middle and last pass widths
    static_assert(KL >= 1 && pass_k_inv(LOGB, LOGT, KL) == KL, "three-pass schedule expected");
    typedef pgeom<LOGB, LOGT, 0, KL> GL;
    static_assert(GL::SETS * GL::R == E, "last pass geometry");
    const size_t ntot = (size_t)1 << (LOGB + X);
    u64* const scr = scratch + ((size_t)blockIdx.x << (LOGB + X));
    bool first = true;
    for (u32 row = blockIdx.x; row < nrows; row += gridDim.x) {
        if (limb_mask && !((limb_mask >> (row % (u32)sel.n)) & 1u)) continue;
        const typename A::ctx C = A::make(LT[sel.idx[row % (u32)sel.n]]);
#pragma unroll 1
        for (u32 ph = 0; ph < 2; ph++) {
            const u64* s = src + row * ntot + brev_bits(ph, X);
            u64 raw[2][E];
            {
                const u32 tid = fresh_tid();
#pragma unroll
                for (int u = 0; u < G1::SETS; u++) {
                    u32 c0, hi, base;
                    G1::template coords<true>(tid, u, c0, hi, base);
#pragma unroll
                    for (int r = 0; r < G1::R; r++) {
                        const u32 nat = (brev_bits((u32)r, K1) << (LOGB - K1)) + c0;
                        const u64x2_t w = *(const u64x2_t*)(s + ((u64)nat << X));
                        raw[0][u * G1::R + r] = w.x;
                        raw[1][u * G1::R + r] = w.y;
                    }
                }
                TFHE_SCHED_FENCE();
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const u32 sb = ph + ((u32)half << (X - 1)), pre = (1u << X) + sb;
                u64* g = scr + ((size_t)sb << LOGB);
                const u32 tid = fresh_tid();
                if (!first) __syncthreads();
                first = false;
                {
                    typename A::elem v[E];
                    inv_compute<A, LOGB, LOGT, S1, K1, true, false, 0>(v, raw[half], nullptr, C, tid, pre);
                    inv_store<A, LOGB, LOGT, S1, K1, true, false>(v, lds, nullptr, C, tid);
                }
                __syncthreads();
                ntt_inv_pass<A, LOGB, LOGT, KL, K2, false, false, false>(lds, nullptr, nullptr, C, tid, pre, 0, 0u);
                __syncthreads();
                ntt_inv_pass<ArithFpD, LOGB, LOGT, 0, KL, false, true, false>(lds, nullptr, g, C, tid, pre, 0, 0u);   // parked as reduced doubles
            }
        }
        // top stages on this thread's own columns (it wrote exactly these words: same thread -> element map in every sub-block).
        // The vector L1 is write-through but may still hold the PREVIOUS row's scratch words (same addresses): one acquire
        // fence invalidates it, then the reads are plain coalesced loads.
        TFHE_WAIT_VM0();                                   // this wave's parking stores have been written through to the L2
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // L1 invalidate only (a release fence would write the L2 back: 1.1 TB/s)
        {
            const u32 tid = fresh_tid();
            const typename A::tw w2 = A::ld_inv(C, 2u), w3 = A::ld_inv(C, 3u);
            u64* o = dst + row * ntot;
            constexpr int CH = 8;  // columns per batch of loads
#pragma unroll
            for (int u = 0; u < GL::SETS; u++) {
                u32 c0, hi, base;
                GL::template coords<false>(tid, u, c0, hi, base);
#pragma unroll
                for (int r0 = 0; r0 < GL::R; r0 += CH) {
                    u64 zw[CH][4];
#pragma unroll
                    for (int r = 0; r < CH; r++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            zw[r][q] = scr[((size_t)q << LOGB) + base + ((u32)(r0 + r) << GL::LO)];
#pragma unroll
                    for (int r = 0; r < CH; r++) {
                        double z0 = A::from_lds(zw[r][0]), z1 = A::from_lds(zw[r][1]), z2 = A::from_lds(zw[r][2]), z3 = A::from_lds(zw[r][3]);
                        A::bf_inv(z0, z1, w2, C);          // stage 1: pairs (0,1) and (2,3), twiddles Winv[2], Winv[3]
                        A::bf_inv(z2, z3, w3, C);
                        A::bf_inv_scaled(z0, z2, C);       // stage 0 with N^-1: pairs (0,2) and (1,3)
                        A::bf_inv_scaled(z1, z3, C);
                        const u32 j = base + ((u32)(r0 + r) << GL::LO);
                        o[j] = A::out_inv_scaled(z0, C);
                        o[j + (1u << LOGB)] = A::out_inv_scaled(z1, C);
                        o[j + (2u << LOGB)] = A::out_inv_scaled(z2, C);
                        o[j + (3u << LOGB)] = A::out_inv_scaled(z3, C);
                    }
                }
            }
        }
    }
}

// top stages of N > 2^LOGB transforms: one column per thread, rows = count*limbs
template <int X>
__global__ __launch_bounds__(256) void k_ntt_fwd_top(const u64* __restrict__ src, u64* __restrict__ dst,
                                                      const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int logn, u32 limb_mask) {
    const u64 stride = (u64)1 << (logn - X);
    const u32 chunks = (u32)((stride + 255) / 256);
    const u32 row = blockIdx.x / chunks;
    if (limb_mask && !((limb_mask >> (row % (u32)sel.n)) & 1u)) return;  // ntt_io_t::limb_mask
    const ntt_limb_t L = LT[sel.idx[row % (u32)sel.n]];
    const u64 col = (u64)(blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (col < stride) ntt_fwd_top<X>(src + ((size_t)row << logn), dst + ((size_t)row << logn), L.W, L.q, col, stride);
}
// the same top stages with the key switch's digit lift fused into the loads (ntt_io_t mode 1: row (b, i, j) reads limb i of
// c[end] of ciphertext b and lifts it, centred, into working limb j): the u64 working limbs of rings that mix modulus sizes at
// N > 2^14 -- the digit rows are never stored untransformed
template <int X>
__global__ __launch_bounds__(256) void k_ntt_fwd_top_lift(const u64* __restrict__ ct, u64* __restrict__ dst,
                                                           const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int logn, ntt_io_t io) {
    constexpr int R = 1 << X;
    const u64 stride = (u64)1 << (logn - X);
    const u32 chunks = (u32)((stride + 255) / 256);
    const u32 row = blockIdx.x / chunks, j = row % io.nw, i = (row / io.nw) % io.level, b = row / (io.nw * io.level);
    if (io.limb_mask && !((io.limb_mask >> j) & 1u)) return;
    const ntt_limb_t L = LT[sel.idx[j]];
    lift_t lf;
    lf.qi = LT[sel.idx[i]].q; lf.half = lf.qi >> 1; lf.qj = L.q; lf.bj = L.br;
    if (io.lift_unsigned) lf.qi = lf.half = ~0ull;
    const u64 col = (u64)(blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (col >= stride) return;
    const u64* s = ct + ((size_t)((b * io.polys + io.polys - 1) * io.level + i) << logn);
    u64* d = dst + ((size_t)row << logn);
    u64 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = lift_digit(s[col + (u64)r * stride], lf);
#pragma unroll
    for (int dd = 0; dd < X; dd++) {
        const int half = 1 << (X - 1 - dd);
#pragma unroll
        for (int g = 0; g < (1 << dd); g++) {
            const tw_t w = ld_tw(L.W, (1u << dd) + (u32)g);
#pragma unroll
            for (int k = 0; k < half; k++) bfly_fwd(v[(g << (X - dd)) + k], v[(g << (X - dd)) + k + half], w, L.q);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) d[col + (u64)r * stride] = csub(csub(v[r], 2 * L.q), L.q);
}
template <int X>
__global__ __launch_bounds__(256) void k_ntt_inv_top(const u64* __restrict__ src, u64* __restrict__ dst,
                                                      const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int logn, u32 limb_mask) {
    const u64 stride = (u64)1 << (logn - X);
    const u32 chunks = (u32)((stride + 255) / 256);
    const u32 row = blockIdx.x / chunks;
    if (limb_mask && !((limb_mask >> (row % (u32)sel.n)) & 1u)) return;  // ntt_io_t::limb_mask
    const ntt_limb_t L = LT[sel.idx[row % (u32)sel.n]];
    const u64 col = (u64)(blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (col < stride) ntt_inv_top<X>(src + ((size_t)row << logn), dst + ((size_t)row << logn), L, col, stride);
}

// generic radix-2 kernel, any N <= 2^14, natural order in/out
__global__ void k_ntt_fwd_generic(const u64* __restrict__ src, u64* __restrict__ dst, const ntt_limb_t* __restrict__ LT,
                                  limb_sel_t sel, int logn, ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 pl = blockIdx.x, n = 1u << logn;
    u32 srow = pl, row = pl, j = pl % (u32)sel.n;
    lift_t lf;
    bool lift = false;
    if (io.mode == 1) {
        const u32 per_ct = io.level * io.nw, b = pl / per_ct, rem = pl % per_ct, i = rem / io.nw;
        j = rem % io.nw;
        srow = (b * io.polys + io.polys - 1) * io.level + i;
        lf.qi = LT[sel.idx[i]].q; lf.half = lf.qi >> 1; lf.qj = LT[sel.idx[j]].q; lf.bj = LT[sel.idx[j]].br;
        if (io.lift_unsigned) lf.qi = lf.half = ~0ull;
        lift = true;
    } else if (io.gsz) {
        const u32 g = pl / io.gsz, w = pl % io.gsz;
        srow = g * io.src_gstride + w;
        row = g * io.dst_gstride + w;
        j = w % (u32)sel.n;
    }
    const ntt_limb_t L = LT[sel.idx[j]];
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 v = src[((size_t)srow << logn) + i];
        lds[i] = lift ? lift_digit(v, lf) : v;
    }
    __syncthreads();
    for (int s = 0; s < logn; s++) {
        for (u32 b = threadIdx.x; b < n / 2; b += blockDim.x) ntt_generic_fwd_stage(lds, L.W, L.q, logn, s, b);
        __syncthreads();
    }
    for (u32 i = threadIdx.x; i < n; i += blockDim.x)
        dst[((size_t)row << logn) + i] = csub(csub(lds[brev_bits(i, logn)], 2 * L.q), L.q);
}
__global__ void k_ntt_inv_generic(const u64* __restrict__ src, u64* __restrict__ dst, const ntt_limb_t* __restrict__ LT,
                                  limb_sel_t sel, int logn, ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 pl = blockIdx.x, n = 1u << logn;
    u32 srow = pl, row = pl, j = pl % (u32)sel.n;
    const u64* addend = nullptr;
    if (io.gsz) {
        const u32 g = pl / io.gsz, w = pl % io.gsz;
        srow = g * io.src_gstride + w;
        row = g * io.dst_gstride + w;
        j = w % (u32)sel.n;
        if (io.mode == 2 && w < io.add_rows) addend = io.addend + ((size_t)(g * io.add_gstride + w) << logn);
    }
    const ntt_limb_t L = LT[sel.idx[j]];
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) lds[brev_bits(i, logn)] = src[((size_t)srow << logn) + i];
    __syncthreads();
    for (int s = logn - 1; s >= 0; s--) {
        for (u32 b = threadIdx.x; b < n / 2; b += blockDim.x) ntt_generic_inv_stage(lds, L, logn, s, b);
        __syncthreads();
    }
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        u64 o = csub(lds[i], L.q);
        if (addend) o = addmod(o, addend[i], L.q);
        dst[((size_t)row << logn) + i] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming limb-wise kernels: one workgroup per row (= one limb of one polynomial), so the modulus
// is workgroup-uniform (scalar registers).  N is a power of two >= 2.
// ------------------------------------------------------------------------------------------------
enum { OP_ADD = 0, OP_SUB = 1, OP_NEG = 2, OP_MUL = 3, OP_MAD = 4, OP_SCAL = 5 };

struct scal_arg_t {
    tw_t s[TFHE_MAX_LIMBS];
};

template <int OP>
__global__ __launch_bounds__(256) void k_pointwise(const u64* __restrict__ a, const u64* __restrict__ b,
                                                    const u64* __restrict__ c, u64* __restrict__ dst,
                                                    const ntt_limb_t* __restrict__ LT, limb_sel_t sel, scal_arg_t sc,
                                                    u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)sel.n;
    const ntt_limb_t L = LT[sel.idx[j]];
    const u64 q = L.q;
    const size_t base = (size_t)row * n;
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        const u64 x = a[base + i];
        u64 r;
        if (OP == OP_ADD) r = addmod(x, b[base + i], q);
        else if (OP == OP_SUB) r = submod(x, b[base + i], q);
        else if (OP == OP_NEG) r = negmod(x, q);
        else if (OP == OP_MUL) r = mulmod(x, b[base + i], L.br);
        else if (OP == OP_MAD) r = addmod(c[base + i], mulmod(x, b[base + i], L.br), q);
        else r = shoup_full(x, sc.s[j], q);
        dst[base + i] = r;
    }
}

// dst = (acc +) sum_k a_k .* b_k, limb-wise: the accumulation loop of the diagonal matrix-vector products (infer.jl:140-149:
// `result += rotated * diagonal_k`, each a ring multiplication and a ring addition in the reference) as ONE pass over the
// operands.  Exact: full 128-bit products are summed and reduced (Barrett) every `chunk` terms, chunk = 2^(62 - bits(q)) capped
// at the term count -- the same canonical residues as the one-by-one mulmod / addmod sequence.
#define TFHE_DOT_MAX 64
struct dot_arg_t {
    const u64* a[TFHE_DOT_MAX];
    const u64* b[TFHE_DOT_MAX];
    int n;
};
__global__ __launch_bounds__(256) void k_dot(dot_arg_t D, const u64* __restrict__ acc, u64* __restrict__ dst,
                                              const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)sel.n;
    const ntt_limb_t L = LT[sel.idx[j]];
    const size_t base = (size_t)row * n;
    int bits = 0;
    while ((L.q >> bits) != 0) bits++;
    const int chunk = bits >= 62 ? 1 : (62 - bits >= 6 ? 64 : (1 << (62 - bits)));   // products summed between two reductions
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        u64 r = acc ? acc[base + i] : 0;
        for (int k0 = 0; k0 < D.n; k0 += chunk) {
            acc128 s{r, 0};
            const int k1 = k0 + chunk < D.n ? k0 + chunk : D.n;
            for (int k = k0; k < k1; k++) acc_mac(s, D.a[k][base + i], D.b[k][base + i]);
            r = barrett_reduce128(s.lo, s.hi, L.br);
        }
        dst[base + i] = r;
    }
}

// dst[row] = sum_k scal[k][j] * a_k[row] (limb-wise, either domain): the scalar-weighted sum of ring elements behind a
// convolution with plaintext scalar weights (infer.jl:127-129: 49 ciphertexts times 49 scalars per channel -- per term one
// scalar_mul pow2_cyc_rings.jl:177-185 and one + :200-214).  scal: device array [K][limbs], residues of the scalars.
__global__ __launch_bounds__(256) void k_lincomb(dot_arg_t D, const u64* __restrict__ scal, u64* __restrict__ dst,
                                                  const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)sel.n;
    const ntt_limb_t L = LT[sel.idx[j]];
    const size_t base = (size_t)row * n;
    int bits = 0;
    while ((L.q >> bits) != 0) bits++;
    const int chunk = bits >= 62 ? 1 : (62 - bits >= 6 ? 64 : (1 << (62 - bits)));   // products summed between two reductions
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        u64 r = 0;
        for (int k0 = 0; k0 < D.n; k0 += chunk) {
            acc128 s{r, 0};
            const int k1 = k0 + chunk < D.n ? k0 + chunk : D.n;
            for (int k = k0; k < k1; k++) acc_mac(s, D.a[k][base + i], scal[(size_t)k * sel.n + j]);
            r = barrett_reduce128(s.lo, s.hi, L.br);
        }
        dst[base + i] = r;
    }
}

// NO such sums of the SAME operands in one pass (the output channels of a convolution layer, infer.jl:127-131: each channel
// weighs the same 49 encrypted inputs): every operand word is read once for all outputs.  scal: [NO][K][limbs]; dst: NO rows sets.
struct lincomb_out_t {
    u64* dst[4];
};
template <int NO>
__global__ __launch_bounds__(256) void k_lincomb_many(dot_arg_t D, const u64* __restrict__ scal, lincomb_out_t O,
                                                       const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)sel.n;
    const ntt_limb_t L = LT[sel.idx[j]];
    const size_t base = (size_t)row * n, ostride = (size_t)D.n * sel.n;
    int bits = 0;
    while ((L.q >> bits) != 0) bits++;
    const int chunk = bits >= 62 ? 1 : (62 - bits >= 6 ? 64 : (1 << (62 - bits)));   // products summed between two reductions
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        u64 r[NO];
#pragma unroll
        for (int o = 0; o < NO; o++) r[o] = 0;
        for (int k0 = 0; k0 < D.n; k0 += chunk) {
            acc128 s[NO];
#pragma unroll
            for (int o = 0; o < NO; o++) s[o] = acc128{r[o], 0};
            const int k1 = k0 + chunk < D.n ? k0 + chunk : D.n;
            for (int k = k0; k < k1; k++) {
                const u64 x = D.a[k][base + i];
#pragma unroll
                for (int o = 0; o < NO; o++) acc_mac(s[o], x, scal[(size_t)o * ostride + (size_t)k * sel.n + j]);
            }
#pragma unroll
            for (int o = 0; o < NO; o++) r[o] = barrett_reduce128(s[o].lo, s[o].hi, L.br);
        }
#pragma unroll
        for (int o = 0; o < NO; o++) O.dst[o][base + i] = r[o];
    }
}

// Diagonal matrix-vector product, accumulation step: out[b][s][j] = diag[0][j] (.) X[b][s][j] + sum_r diag[r+1][j] (.) ROT[r][b][s][j]
// (NTT domain; the loop `result += rotated_k * diagonal_k` of infer.jl:140-149 / test/ckks_matmul.jl:33-41 over all terms in
// one pass: the canonical residues of the term-by-term sum).  diag: [R+1][limbs][N], shared by the batch.
__global__ __launch_bounds__(256) void k_matmul_acc(const u64* __restrict__ X, const u64* __restrict__ ROT, const u64* __restrict__ diag,
                                                     u64* __restrict__ out, const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 n,
                                                     u32 nrot, u32 rows_per_rot) {
    const u32 row = blockIdx.x, j = row % (u32)sel.n;   // row = (b * 2 + s) * limbs + j
    const ntt_limb_t L = LT[sel.idx[j]];
    const size_t base = (size_t)row * n, rstride = (size_t)rows_per_rot * n, dstride = (size_t)sel.n * n;
    const u64* dj = diag + (size_t)j * n;
    int bits = 0;
    while ((L.q >> bits) != 0) bits++;
    const u32 chunk = bits >= 62 ? 1u : (62 - bits >= 6 ? 64u : (1u << (62 - bits)));
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        u64 r = 0;
        for (u32 k0 = 0; k0 <= nrot; k0 += chunk) {
            acc128 s{r, 0};
            const u32 k1 = k0 + chunk <= nrot ? k0 + chunk : nrot + 1;
            for (u32 k = k0; k < k1; k++) {
                const u64 x = k == 0 ? X[base + i] : ROT[(size_t)(k - 1) * rstride + base + i];
                acc_mac(s, x, dj[(size_t)k * dstride + i]);
            }
            r = barrett_reduce128(s.lo, s.hi, L.br);
        }
        out[base + i] = r;
    }
}

// tensor (rlwe_she.jl:255-258) in the NTT domain: a,b [batch][2][limbs][N] -> out [batch][3][limbs][N]
__global__ __launch_bounds__(256) void k_tensor(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out,
                                                 const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)sel.n, ct = row / (u32)sel.n;
    const ntt_limb_t L = LT[sel.idx[j]];
    const size_t ps = (size_t)sel.n * n;  // poly stride
    const u64 *a0 = a + (size_t)ct * 2 * ps + (size_t)j * n, *a1 = a0 + ps;
    const u64 *b0 = b + (size_t)ct * 2 * ps + (size_t)j * n, *b1 = b0 + ps;
    u64 *o0 = out + (size_t)ct * 3 * ps + (size_t)j * n, *o1 = o0 + ps, *o2 = o1 + ps;
    auto one = [&](u64 x0, u64 x1, u64 y0, u64 y1, u64& r0, u64& r1, u64& r2) {
        r0 = mulmod(x0, y0, L.br);
        acc128 acc{0, 0};
        acc_mac(acc, x0, y1);
        if (L.br.sh <= 58) {  // two products fit the Barrett window when q < 2^61
            acc_mac(acc, x1, y0);
            r1 = barrett_reduce128(acc.lo, acc.hi, L.br);
        } else {
            r1 = addmod(barrett_reduce128(acc.lo, acc.hi, L.br), mulmod(x1, y0, L.br), L.q);
        }
        r2 = mulmod(x1, y1, L.br);
    };
    if ((n & 1u) == 0) {  // two coefficients per thread: 16-byte loads and stores
        for (u32 i = (blockIdx.y * blockDim.x + threadIdx.x) * 2; i < n; i += gridDim.y * blockDim.x * 2) {
            const u64x2_t x0 = *(const u64x2_t*)(a0 + i), x1 = *(const u64x2_t*)(a1 + i);
            const u64x2_t y0 = *(const u64x2_t*)(b0 + i), y1 = *(const u64x2_t*)(b1 + i);
            u64x2_t r0, r1, r2;
#pragma unroll
            for (int v = 0; v < 2; v++) {
                u64 t0, t1, t2;
                one(x0[v], x1[v], y0[v], y1[v], t0, t1, t2);
                r0[v] = t0; r1[v] = t1; r2[v] = t2;
            }
            *(u64x2_t*)(o0 + i) = r0;
            *(u64x2_t*)(o1 + i) = r1;
            *(u64x2_t*)(o2 + i) = r2;
        }
        return;
    }
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) one(a0[i], a1[i], b0[i], b1[i], o0[i], o1[i], o2[i]);
}

// modswitch (crt.jl:215-228): rows = count*(limbs-1)
struct rescale_arg_t {
    tw_t qlinv[TFHE_MAX_LIMBS];  // q_last^-1 mod q_j
};
__global__ __launch_bounds__(256) void k_rescale(const u64* __restrict__ src, u64* __restrict__ dst,
                                                  const ntt_limb_t* __restrict__ LT, limb_sel_t sel, rescale_arg_t ra,
                                                  u32 n) {
    const u32 nl = (u32)sel.n, row = blockIdx.x, j = row % (nl - 1), p = row / (nl - 1);
    const ntt_limb_t L = LT[sel.idx[j]];
    const u64* cj = src + ((size_t)p * nl + j) * n;
    const u64* cl = src + ((size_t)p * nl + nl - 1) * n;
    u64* d = dst + (size_t)row * n;
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        const u64 last = barrett_reduce128(cl[i], 0, L.br);  // unsigned representative of c_last, mod q_j
        d[i] = shoup_full(submod(cj[i], last, L.q), ra.qlinv[j], L.q);
    }
}

// The same, coefficient-major (r04): a thread takes two coefficients of ONE polynomial through all its limbs, so the last limb's
// words are read once -- row by row (above) the blocks of the nl - 1 output rows of a polynomial run on different XCDs and each
// fetched the last limb for itself: 2.3-2.5 x the algorithmic bytes at an HBM-bound kernel (profiles/r04v_pmc_configs.txt).
__global__ __launch_bounds__(256) void k_rescale_cm(const u64* __restrict__ src, u64* __restrict__ dst,
                                                     const ntt_limb_t* __restrict__ LT, limb_sel_t sel, rescale_arg_t ra,
                                                     u32 n) {
    const u32 nl = (u32)sel.n, p = blockIdx.x;
    const u64* sp = src + (size_t)p * nl * n;
    u64* dp = dst + (size_t)p * (nl - 1) * n;
    for (u32 i = (blockIdx.y * blockDim.x + threadIdx.x) * 2u; i < n; i += gridDim.y * blockDim.x * 2u) {
        const u64x2_t l = *(const u64x2_t*)(sp + (size_t)(nl - 1) * n + i);
        // the limbs in blocks of JB: every word of a block is requested before the first is used (one limb at a time the loop ran at
        // the latency of its loads)
        constexpr u32 JB = 8;
        for (u32 j0 = 0; j0 + 1 < nl; j0 += JB) {
            u64x2_t cj[JB];
#pragma unroll
            for (u32 t = 0; t < JB; t++) {
                const u32 j = j0 + t + 1 < nl ? j0 + t : nl - 2;
                cj[t] = *(const u64x2_t*)(sp + (size_t)j * n + i);
            }
#pragma unroll
            for (u32 t = 0; t < JB; t++) {
                const u32 j = j0 + t;
                if (j + 1 < nl) {
                    const ntt_limb_t& L = LT[sel.idx[j]];
                    const u64 q = L.q;
                    u64x2_t o;
                    o.x = shoup_full(submod(cj[t].x, barrett_reduce128(l.x, 0, L.br), q), ra.qlinv[j], q);
                    o.y = shoup_full(submod(cj[t].y, barrett_reduce128(l.y, 0, L.br), q), ra.qlinv[j], q);
                    *(u64x2_t*)(dp + (size_t)j * n + i) = o;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_select(const u64* __restrict__ src, u64* __restrict__ dst, limb_sel_t which,
                                                 int src_limbs, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)which.n, p = row / (u32)which.n;
    const u64* s = src + ((size_t)p * src_limbs + which.idx[j]) * n;
    u64* d = dst + (size_t)row * n;
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) d[i] = s[i];
}

// apply_galois_element (pow2_cyc_rings.jl:321-329) in gather form: out[r] = ± in[i], g*i ≡ r (mod N).
// ginv = g^-1 mod 2N.  i0 = r*ginv mod 2N; i0 < N: +in[i0]; else -in[i0-N].
__global__ __launch_bounds__(256) void k_galois(const u64* __restrict__ src, u64* __restrict__ dst,
                                                 const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u64 ginv, u32 n) {
    const u32 row = blockIdx.x;
    const u64 q = LT[sel.idx[row % (u32)sel.n]].q;
    const size_t base = (size_t)row * n;
    const u64 mask2n = 2ull * n - 1;
    for (u32 r = blockIdx.y * blockDim.x + threadIdx.x; r < n; r += gridDim.y * blockDim.x) {
        const u64 i0 = ((u64)r * ginv) & mask2n;
        const u64 v = src[base + (i0 & (n - 1))];
        dst[base + r] = (i0 >= n) ? negmod(v, q) : v;
    }
}

// The same automorphism for rows beyond the LDS (N >= 2^15) in scatter form with an XCD-cooperative walk (see k_ks_rot_tail):
// the workgroups of one XCD share one polynomial (its `limbs` rows: the same index map, different moduli for the sign) at a
// time, so the scattered 8-byte stores land in windows that stay in that XCD's L2 until they are complete; reads are coalesced.
// rows = polys * limbs, row r uses modulus sel.idx[r % limbs].
__global__ __launch_bounds__(256) void k_galois_xcd(const u64* __restrict__ src, u64* __restrict__ dst,
                                                     const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u64 g, u32 n, u32 npolys) {
    __shared__ u64 lq[TFHE_MAX_LIMBS];
    const u32 limbs = (u32)sel.n;
    for (u32 j = threadIdx.x; j < limbs; j += blockDim.x) lq[j] = LT[sel.idx[j]].q;
    __syncthreads();
    const u32 xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const u64 mask2n = 2ull * n - 1;
    for (u32 p = xcd; p < npolys; p += 8u) {
        const u64* s0 = src + (size_t)p * limbs * n;
        u64* d0 = dst + (size_t)p * limbs * n;
        for (u32 i = slot * blockDim.x + threadIdx.x; i < n; i += nslot * blockDim.x) {
            const u64 t = ((u64)i * g) & mask2n;
            const bool neg = t >= n;
            const u32 m = (u32)t & (n - 1);
            for (u32 j = 0; j < limbs; j++) {
                const u64 v = s0[(size_t)j * n + i];
                d0[(size_t)j * n + m] = neg ? negmod(v, lq[j]) : v;
            }
        }
    }
}

// The same automorphism for rows that fit the LDS (N <= 2^14) in scatter form through the LDS: the row is read with coalesced
// 16-byte loads, every word is written to LDS position g*i mod N (odd stride: the 32 lanes of a half-wave hit 32 different
// bank pairs -- conflict-free), and the permuted row is read back linearly and stored with coalesced 16-byte stores.  The
// gather form above moves 8 scattered bytes per lane and is bound by the texture addresser (about 8 B/clk/CU: 2.9 TB/s);
// this one streams.  Persistent workgroups, one row at a time; the stores of a row drain under the loads of the next.
template <int T>
__global__ __launch_bounds__(T) void k_galois_lds(const u64* __restrict__ src, u64* __restrict__ dst,
                                                   const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u64 g, u32 n, u32 nrows) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u64 mask2n = 2ull * n - 1;
    const u32 tid = threadIdx.x, per = n / (2 * T);           // 16-byte pieces per thread (n >= 2 T)
    for (u32 row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u64 q = LT[sel.idx[row % (u32)sel.n]].q;
        const u64x2_t* s2 = (const u64x2_t*)(src + (size_t)row * n);
        u64x2_t* d2 = (u64x2_t*)(dst + (size_t)row * n);
        for (u32 k0 = 0; k0 < per; k0 += 8) {                 // eight pieces in flight per thread
            u64x2_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k0 + k < per) v[k] = s2[tid + (k0 + k) * T];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k0 + k < per) {
                    const u64 i = 2ull * (tid + (k0 + k) * T);
                    const u64 t0 = (g * i) & mask2n, t1 = (g * (i + 1)) & mask2n;
                    lds[t0 & (n - 1)] = t0 >= n ? negmod(v[k].x, q) : v[k].x;
                    lds[t1 & (n - 1)] = t1 >= n ? negmod(v[k].y, q) : v[k].y;
                }
        }
        __syncthreads();
        for (u32 k = 0; k < per; k++) d2[tid + k * T] = *(const u64x2_t*)(lds + 2 * (tid + k * T));
        __syncthreads();
    }
}

// (For N = 2^15 / 2^16 a residue-class variant -- the automorphism maps the class i = r (mod 2^X) onto r' = g r as an affine map of
// the 2^14 class indices, so each class can go through the same LDS scatter with stride-2^X global accesses -- was measured: no
// gain at 2^15 (30.2 k vs 30.3 k rotations/s on cfg#3) and slower at 2^16 (24.5 k vs 27.4 k): the gather form stays there.)

// ------------------------------------------------------------------------------------------------
// keyswitch pieces (rlwe_she.jl:315-347, modulusraising.jl:35-49)
// ------------------------------------------------------------------------------------------------
// NTT-domain action of the Galois automorphism x -> x^g (pow2_cyc_rings.jl:321-329) in the library's natural order
// a^[k] = a(psi^(2k+1)):  (sigma_g a)^[k] = a(psi^(g (2k+1))) = a^[k'] with 2k' + 1 = g (2k+1) mod 2N -- a pure permutation.
TFHE_HD u32 galois_ntt_pos(u32 k, u64 g, u32 n) { return (u32)(((g * (2ull * k + 1ull)) - 1ull) >> 1) & (n - 1u); }

// dst[row][m] = src[row][pi_g(m)]: the automorphism applied to NTT-domain rows (key preparation of the hoisted rotations)
__global__ __launch_bounds__(256) void k_ntt_perm(const u64* __restrict__ src, u64* __restrict__ dst, u64 g, u32 n) {
    const u64* s = src + (size_t)blockIdx.x * n;
    u64* d = dst + (size_t)blockIdx.x * n;
    for (u32 m = blockIdx.y * blockDim.x + threadIdx.x; m < n; m += gridDim.y * blockDim.x) d[m] = s[galois_ntt_pos(m, g, n)];
}

struct ks_arg_t {
    int level, nw, special, polys;
    u32 rot_g;                   // k_ks_fused SPMODE 3: the Galois element (mod 2N) of a rotation finished in the final store
    limb_sel_t w;                // working limbs: key limbs 0..level-1 (+ special prime)
    tw_t pinv[TFHE_MAX_LIMBS];   // P^-1 mod q_j (special) for j < level: the epilogue of the key sums in tfhe_matmul_diag (ks_keys_t::epi_x)
};

// S[b][s][j] = Σ_i evk[i][s'][w[j]] * D[b][i][j]  (NTT domain); s = 0 (c1) uses masked, s = 1 (c2) uses mask
// (rlwe_she.jl:340-344).  One thread owns coefficient k of limb j for the WHOLE chunk of ciphertexts, so the
// evaluation-key values are loaded once into registers and reused across the batch (the key is shared by
// all ciphertexts; re-reading it per ciphertext was the dominant traffic).  grid = nw * ceil(N/256).
// several keys against the same digits in one launch (the rotations of a diagonal product, tfhe_matmul_diag): the workgroup's
// position selects the key and the slice of S it writes (ks_multi_key_block); n == 0: the single key `evk`
struct ks_keys_t {
    int n;
    size_t s_stride;  // words of S per key
    const u64* epi_x; // EPI kernels (tfhe_matmul_diag, evaluation-domain form): X [batch][2][level][N]; limb j < level of the sums
                      // leaves as V = S P^-1 (+ X[b][0][j] for s = 0) instead of S (k_md_v folded into the store)
    const u64* key[TFHE_DOT_MAX];
};
// Layout of the key sums in the EPI kernels (tfhe_matmul_diag, evaluation-domain form), per ciphertext a block of 2 nw n words:
// every working limb j holds the two components INTERLEAVED (word j 2n + 2k + s) -- k_md_acc and k_md_special_perm gather both
// with one 16-byte load per rotated position (a gather costs a cache-line access per lane whatever its width: r04, MNIST pass
// 73.2 -> 69.0 ms with the ciphertext limbs interleaved).
__device__ __forceinline__ size_t epi_pair(u32 b, u32 j, u32 k, u32 nw, u32 n) { return ((size_t)b * nw + j) * 2 * n + 2 * (size_t)k; }
// Several keys in one launch: a 1-D grid in which the workgroups of ONE (limb, tile, batch slice) for all the keys sit next to
// each other on ONE XCD (workgroup b runs on XCD b & 7): they stream the same digit words at the same time, so the digit rows
// are fetched from HBM once per launch instead of once per key (the keys and the sums are per key anyway).
// Returns false for the padding workgroups of the last group of eight.
__device__ __forceinline__ bool ks_multi_key_block(const ks_keys_t& K, u32 nx, u32& bx, u32& key) {
    const u32 bid = blockIdx.x, q = bid >> 3;
    key = q % (u32)K.n;
    bx = (q / (u32)K.n) * 8u + (bid & 7u);
    return bx < nx;
}
template <int DCH, bool EPI = false>
__global__ __launch_bounds__(256) void k_ks_inner(const u64* __restrict__ evk, const u64* __restrict__ dig,
                                                   u64* __restrict__ S, const ntt_limb_t* __restrict__ LT, ks_arg_t A,
                                                   int Lk, u32 n, u32 batch, u32 bsplit, u32 limb_mask, ks_keys_t K) {
    // bx = (slice * nw + j) * gx + tile; slice = which part of the batch this workgroup owns
    const u32 gx = (n + 255) / 256;
    u32 bx = blockIdx.x;
    if (K.n) {
        u32 key;
        if (!ks_multi_key_block(K, gx * (u32)A.nw * bsplit, bx, key)) return;
        evk = K.key[key]; S += (size_t)key * K.s_stride;
    }
    const u32 tile = bx % gx, j = (bx / gx) % (u32)A.nw, slice = bx / (gx * (u32)A.nw);
    if (limb_mask && !((limb_mask >> j) & 1u)) return;  // rings of mixed modulus sizes: the narrow kernel takes the other limbs
    const u32 k = tile * 256 + threadIdx.x;
    const u32 per = (batch + bsplit - 1) / bsplit, b_lo = slice * per, b_hi = b_lo + per < batch ? b_lo + per : batch;
    if (k >= n) return;
    const ntt_limb_t L = LT[A.w.idx[j]];
    // Lazy sums of full products, one reduction per `lazy` terms.  Moduli up to 52 bits: the 128-bit sum goes straight into the
    // Barrett reduction (z < 2^(sh+64)).  Wider moduli (r04; the 60-bit q0 / special prime of the reference's CKKS rings used to
    // be reduced after EVERY term, 2 x level Barrett reductions per coefficient -- the kernel ran at 0.9 VALU issue): the high
    // word is folded first, z' = hi (2^64 mod q) + lo < 2^(sh+64) for up to 2^(126 - 2 bits) - 1 terms (15 at 61 bits, 3 at 62),
    // then ONE Barrett reduction -- the same canonical residue.
    const u32 qbits = L.br.sh + 2;
    const bool fold = qbits > 52;
    const int lazy = fold ? (qbits <= 61 ? DCH : 3) : (1 << 10);
    static_assert(DCH <= 15, "fold budget at 61 bits");
    const u64 c64 = fold ? barrett_reduce128(0, 1, L.br) : 0;   // 2^64 mod q
    auto reduce = [&](const acc128& s) -> u64 {
        if (!fold) return barrett_reduce128(s.lo, s.hi, L.br);
        u64 lo, hi;
        mul64_full(s.hi, c64, lo, hi);
        const u64 l2 = lo + s.lo;
        return barrett_reduce128(l2, hi + (l2 < lo), L.br);
    };
    for (int i0 = 0; i0 < A.level; i0 += DCH) {
        u64 mk[DCH], md[DCH];
#pragma unroll
        for (int ii = 0; ii < DCH; ii++) {
            const int i = i0 + ii < A.level ? i0 + ii : A.level - 1;
            mk[ii] = evk[(((size_t)i * 2 + 0) * Lk + A.w.idx[j]) * n + k];
            md[ii] = evk[(((size_t)i * 2 + 1) * Lk + A.w.idx[j]) * n + k];
        }
        // two ciphertexts per round: their digit words are requested together (one ciphertext at a time the loop was bound by the
        // round trip of its `level` digit loads: 30 % of the multiplier rate, 1.5 TB/s)
        for (u32 b0 = b_lo; b0 < b_hi; b0 += 2) {
            const u32 bb[2] = {b0, b0 + 1 < b_hi ? b0 + 1 : b0};
            u64 d[2][DCH], r1[2], r2[2];
            size_t p1[2], p2[2];   // where the two sums of ciphertext bb[h] live
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (EPI) {
                    p1[h] = epi_pair(bb[h], j, k, (u32)A.nw, n);
                    p2[h] = p1[h] + 1;
                } else {
                    p1[h] = (((size_t)bb[h] * 2 + 0) * A.nw + j) * n + k;
                    p2[h] = (((size_t)bb[h] * 2 + 1) * A.nw + j) * n + k;
                }
                r1[h] = i0 ? S[p1[h]] : 0;
                r2[h] = i0 ? S[p2[h]] : 0;
#pragma unroll
                for (int ii = 0; ii < DCH; ii++)
                    if (i0 + ii < A.level) d[h][ii] = dig[(((size_t)bb[h] * A.level + i0 + ii) * A.nw + j) * n + k];
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                acc128 s1{0, 0}, s2{0, 0};
                int pend = 0;
#pragma unroll
                for (int ii = 0; ii < DCH; ii++) {
                    if (i0 + ii < A.level) {
                        acc_mac(s1, md[ii], d[h][ii]);
                        acc_mac(s2, mk[ii], d[h][ii]);
                        if (++pend == lazy) {
                            r1[h] = addmod(r1[h], reduce(s1), L.q);
                            r2[h] = addmod(r2[h], reduce(s2), L.q);
                            s1 = acc128{0, 0}; s2 = acc128{0, 0}; pend = 0;
                        }
                    }
                }
                if (pend) {
                    r1[h] = addmod(r1[h], reduce(s1), L.q);
                    r2[h] = addmod(r2[h], reduce(s2), L.q);
                }
                if constexpr (EPI) {
                    if (i0 + DCH >= A.level && j < (u32)A.level) {
                        r1[h] = addmod(shoup_full(r1[h], A.pinv[j], L.q), K.epi_x[(((size_t)bb[h] * 2) * A.level + j) * n + k], L.q);
                        r2[h] = shoup_full(r2[h], A.pinv[j], L.q);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (h == 0 || bb[1] != bb[0]) {
                    if (EPI) {
                        u64x2_t w;
                        w.x = r1[h]; w.y = r2[h];
                        *(u64x2_t*)(S + p1[h]) = w;
                    } else {
                        S[p1[h]] = r1[h];
                        S[p2[h]] = r2[h];
                    }
                }
            }
        }
    }
}

// Same sums for working moduli below 2^52, two coefficients per thread (16-byte loads / stores) and carry-free
// 26-bit-split accumulation (modarith.h acc52): the kernel is bound by the digit stream, not by the multiplier.
template <int DCH, bool EPI = false>
__global__ __launch_bounds__(256) void k_ks_inner_n2(const u64* __restrict__ evk, const u64* __restrict__ dig,
                                                      u64* __restrict__ S, const ntt_limb_t* __restrict__ LT, ks_arg_t A,
                                                      int Lk, u32 n, u32 batch, u32 bsplit, u32 limb_mask, ks_keys_t K) {
    static_assert(DCH + 1 <= 16, "acc52 term budget");
    const u32 gx = (n / 2 + 255) / 256;
    u32 bx = blockIdx.x;
    if (K.n) {
        u32 key;
        if (!ks_multi_key_block(K, gx * (u32)A.nw * bsplit, bx, key)) return;
        evk = K.key[key]; S += (size_t)key * K.s_stride;
    }
    const u32 tile = bx % gx, j = (bx / gx) % (u32)A.nw, slice = bx / (gx * (u32)A.nw);
    if (limb_mask && !((limb_mask >> j) & 1u)) return;
    const u32 k = (tile * 256 + threadIdx.x) * 2;
    const u32 per = (batch + bsplit - 1) / bsplit, b_lo = slice * per, b_hi = b_lo + per < batch ? b_lo + per : batch;
    if (k >= n) return;
    const barrett_t br = LT[A.w.idx[j]].br;
    for (int i0 = 0; i0 < A.level; i0 += DCH) {
        u32 mk0[DCH][2], mk1[DCH][2], md0[DCH][2], md1[DCH][2];
#pragma unroll
        for (int ii = 0; ii < DCH; ii++) {
            const int i = i0 + ii < A.level ? i0 + ii : A.level - 1;
            const u64x2_t a = *(const u64x2_t*)(evk + (((size_t)i * 2 + 0) * Lk + A.w.idx[j]) * n + k);
            const u64x2_t d = *(const u64x2_t*)(evk + (((size_t)i * 2 + 1) * Lk + A.w.idx[j]) * n + k);
#pragma unroll
            for (int v = 0; v < 2; v++) {
                mk0[ii][v] = (u32)a[v] & 0x3ffffffu; mk1[ii][v] = (u32)(a[v] >> 26);
                md0[ii][v] = (u32)d[v] & 0x3ffffffu; md1[ii][v] = (u32)(d[v] >> 26);
            }
        }
        for (u32 b = b_lo; b < b_hi; b++) {
            // EPI: the two components interleaved (epi_pair): words {s1[k], s2[k]}, {s1[k+1], s2[k+1]}
            constexpr bool pairs = EPI;
            u64x2_t *s1p, *s2p;
            if (EPI) {
                s1p = (u64x2_t*)(S + epi_pair(b, j, k, (u32)A.nw, n));
                s2p = s1p + 1;
            } else {
                s1p = (u64x2_t*)(S + (((size_t)b * 2 + 0) * A.nw + j) * n + k);
                s2p = (u64x2_t*)(S + (((size_t)b * 2 + 1) * A.nw + j) * n + k);
            }
            u64x2_t p1 = {0, 0}, p2 = {0, 0};
            if (i0) {
                p1 = *s1p; p2 = *s2p;
                if (pairs) { const u64x2_t a = p1, c = p2; p1.x = a.x; p1.y = c.x; p2.x = a.y; p2.y = c.y; }
            }
            acc52 s1[2] = {{p1[0], 0, 0}, {p1[1], 0, 0}}, s2[2] = {{p2[0], 0, 0}, {p2[1], 0, 0}};
            u64x2_t dv[DCH];
#pragma unroll
            for (int ii = 0; ii < DCH; ii++)
                if (i0 + ii < A.level) dv[ii] = *(const u64x2_t*)(dig + (((size_t)b * A.level + i0 + ii) * A.nw + j) * n + k);
#pragma unroll
            for (int ii = 0; ii < DCH; ii++) {
                if (i0 + ii < A.level) {
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const u32 d0 = (u32)dv[ii][v] & 0x3ffffffu, d1 = (u32)(dv[ii][v] >> 26);
                        acc52_mac(s1[v], d0, d1, md0[ii][v], md1[ii][v]);
                        acc52_mac(s2[v], d0, d1, mk0[ii][v], mk1[ii][v]);
                    }
                }
            }
            u64x2_t r1, r2;
#pragma unroll
            for (int v = 0; v < 2; v++) {
                u64 lo, hi;
                acc52_fold(s1[v], lo, hi);
                r1[v] = barrett_reduce128(lo, hi, br);
                acc52_fold(s2[v], lo, hi);
                r2[v] = barrett_reduce128(lo, hi, br);
            }
            if constexpr (EPI) {
                if (i0 + DCH >= A.level && j < (u32)A.level) {
                    const u64x2_t x = *(const u64x2_t*)(K.epi_x + (((size_t)b * 2) * A.level + j) * n + k);
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        r1[v] = addmod(shoup_full(r1[v], A.pinv[j], br.q), x[v], br.q);
                        r2[v] = shoup_full(r2[v], A.pinv[j], br.q);
                    }
                }
            }
            if (pairs) {
                u64x2_t a, c;
                a.x = r1[0]; a.y = r2[0]; c.x = r1[1]; c.y = r2[1];
                *s1p = a;
                *s2p = c;
            } else {
                *s1p = r1;
                *s2p = r2;
            }
        }
    }
}

#ifdef TFHE_KS_TRACE  // design aid (tools/ks_trace.py): 100 MHz stamps of workgroup 0's phases in the fused key switch
__device__ unsigned long long tfhe_kst[4096];
__device__ unsigned tfhe_kst_n;
__device__ __forceinline__ void kst(unsigned tag) {
    __builtin_amdgcn_sched_barrier(0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned n = tfhe_kst_n;
        if (n < 4096) {
            tfhe_kst[n] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memrealtime() & ((1ull << 56) - 1));
            tfhe_kst_n = n + 1;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
#else
#define kst(tag) ((void)0)
#endif
// ---- shared pieces of the fused kernels: a forward transform that ends in registers (last pass's natural-order map)
// and an inverse transform that starts from registers in that same map ----
// middle-pass twiddles of the fused kernels from LDS tables (ArithFpL), per kernel: bit 0 -- forward transforms, bit 1 -- inverse
// transforms (those keep their register prefetch when the bit is clear); 0: vector loads throughout, no tables.
// MEASURED (r03, rocprof on one box, alternating bench runs): key switch 1 445 / 1 452 us with the forward / both tables against
// 1 455 us without, the core kernel 1 905 against 1 870 us (+ 2 %), step 59.9 k against 59.8 k -- nothing: with the twiddles
// pinned to one cache line (-DTFHE_ABL_NOTW=3) the two kernels gain 5 %, which is the whole prize, and the LDS reads cost
// about what the L2 round trips did.  Off by default; the code stays as the record of the experiment.
#ifndef TFHE_TWL_KS
#define TFHE_TWL_KS 0
#endif
#ifndef TFHE_TWL_CORE
#define TFHE_TWL_CORE 0
#endif
// words of the LDS twiddle tables of the fused kernels (one per direction): every stage below the boundary pass
template <int LOGB, int LOGT>
constexpr u32 fused_tw_entries() {
    return 1u << (pass_k_fwd(LOGB, LOGT, 0) + pass_k_fwd(LOGB, LOGT, pass_k_fwd(LOGB, LOGT, 0)));
}
template <int LOGB, int LOGT>
constexpr u32 fused_tw_words() {  // padded (tw_lds_pos)
    return fused_tw_entries<LOGB, LOGT>() + (fused_tw_entries<LOGB, LOGT>() >> 4);
}
// fill the LDS twiddle tables of limb `C` (ArithFpL); the first reader is behind the barrier that follows the first pass, and
// the previous item's last reader (its inverse middle pass) is behind the barrier that precedes its last pass
template <class A, int LOGB, int LOGT, int MASK>
__device__ __forceinline__ void fused_fill_tw(u64* lds, typename A::ctx& C) {
    constexpr u32 TE = fused_tw_entries<LOGB, LOGT>(), T = 1u << LOGT;
    static_assert(TE % T == 0, "table size vs workgroup size");
    double* wl = reinterpret_cast<double*>(lds + lds_words<LOGB, LOGT>());
    double* wil = wl + fused_tw_words<LOGB, LOGT>();
    const u32 tid = fresh_tid();
    typedef __attribute__((address_space(1))) const double* gptr_t;
    const gptr_t gw = (gptr_t)C.W, gwi = (gptr_t)C.Winv;
#pragma unroll
    for (u32 i = 0; i < TE; i += T) {
        if (MASK & 1) wl[tw_lds_pos(i + tid)] = gw[i + tid];
        if (MASK & 2) wil[tw_lds_pos(i + tid)] = gwi[i + tid];
    }
    C.Wl = wl;
    C.Winvl = wil;
}
template <class A, int LOGB, int LOGT, bool PRELIFT = false, bool TWL = false>
__device__ __forceinline__ void fused_fwd_to_regs(u64* lds, const u64* grow, const typename A::ctx& C, bool& first,
                                                  typename A::elem* v, const lift_t* lift = nullptr) {
    typedef typename std::conditional<TWL, ArithFpL, A>::type AM;  // policy of the middle pass
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    constexpr int E = 1 << (LOGB - LOGT);
    const u32 tid = fresh_tid();
    {
        u64 raw[E];
#ifdef TFHE_ABL_NOROWS  // design aid: operands from arithmetic (wrong results, no row traffic)
#pragma unroll
        for (int i = 0; i < E; i++) { raw[i] = (u64)(tid * 131u + (u32)i * 7919u) + (u64)(size_t)grow; pin_vgpr(raw[i]); }
#else
        fwd_load_data<LOGB, LOGT, 0, K1, true, false>(raw, lds, grow, tid);
#endif
        if (!first) __syncthreads();  // the previous transform's last pass has read LDS
        first = false;
        if constexpr (PRELIFT) {
            // The words are centred doubles already (|d| <= q_i / 2 <= p: the host admits this path only when no modulus is
            // more than twice another): element bits, as if they came from LDS.  Every operand
            // is pinned before the first butterfly: left to itself the compiler spreads the waits for these loads over the
            // butterflies, which measured 22 % slower on the whole kernel (1.61 ms pinned, 2.05 ms unpinned, 1.67 ms with
            // the in-kernel lift, whose conversion loop has the same effect as the pins).
#pragma unroll
            for (int i = 0; i < E; i++) pin_vgpr(raw[i]);
            TFHE_SCHED_FENCE();
            fwd_compute<A, LOGB, LOGT, 0, K1, false, false, 0>(v, raw, nullptr, C, tid, 1u);
        } else {
            fwd_compute<A, LOGB, LOGT, 0, K1, true, false, 0>(v, raw, nullptr, C, tid, 1u, lift);
        }
        fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
    }
    __syncthreads();
    kst(2);
    ntt_fwd_pass<AM, LOGB, LOGT, K1, K2, false, false>(lds, nullptr, nullptr, C, tid, 1u, 0, 0u);
    __syncthreads();
    kst(3);
    {
        u64 r3[E];
        fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(r3, lds, nullptr, tid);
        fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, r3, nullptr, C, tid, 1u);
    }
#ifdef TFHE_KS_TRACE
#pragma unroll
    for (int i = 0; i < E; i++) pin_vgpr(v[i]);
#endif
    kst(4);
}
// inverse transform of elements held in the forward-last-pass register map (v is reduced here and consumed); result
// (+ addend) to gdst
// SCALE = false: the caller has folded N^-1 into its operands (k_ks_fused: into the key rows), the last stage is a plain
// butterfly instead of two scaling products per pair
template <class A, int LOGB, int LOGT, bool SCALE = true, bool TWL = false, class AO = A>
__device__ __forceinline__ void fused_inv_from_regs(u64* lds, typename A::elem* v, u64* gdst, const typename A::ctx& C, const u64* addend,
                                                    u64* keep = nullptr, u32 pre = 1u) {   // pre: sub-block prefix (2^x + sb) of a larger transform
    constexpr int KI1 = pass_k_inv(LOGB, LOGT, LOGB);
    constexpr int E = 1 << (LOGB - LOGT);
    const u32 tid = fresh_tid();
    __syncthreads();  // the previous transform's last pass has read LDS
    constexpr int S1 = LOGB - KI1, K2 = pass_k_inv(LOGB, LOGT, S1);
    typedef pgeom<LOGB, LOGT, S1 - K2, K2> G2;
    if constexpr (TWL) {
        // middle pass with its twiddles from the LDS table (no register prefetch, no vector loads), last pass as before
        static_assert(S1 - K2 != 0, "three-pass inverse expected");
        {
#pragma unroll
            for (int e = 0; e < E; e++) v[e] = fp_reduce(v[e], C.p, C.pinv);
            inv_compute<A, LOGB, LOGT, S1, KI1, true, SCALE, 0, -1, no_hook, true>(v, nullptr, nullptr, C, tid, 1u);
            inv_store<A, LOGB, LOGT, S1, KI1, true, SCALE>(v, lds, nullptr, C, tid);
        }
        __syncthreads();
        ntt_inv_pass<ArithFpL, LOGB, LOGT, S1 - K2, K2, false, false, SCALE>(lds, nullptr, nullptr, C, tid, 1u, 0, 0u);
        __syncthreads();
        ntt_inv_pass<AO, LOGB, LOGT, 0, S1 - K2, false, true, SCALE>(lds, nullptr, gdst, C, tid, 1u, 0, 0u, addend);
    } else {
        typename A::tw tw_next[G2::SETS * G2::NTW];  // middle-pass twiddles, requested before the exchange
        {
#pragma unroll
            for (int e = 0; e < E; e++) v[e] = fp_reduce(v[e], C.p, C.pinv);
            inv_compute<A, LOGB, LOGT, S1, KI1, true, SCALE, 0, -1, no_hook, true>(v, nullptr, nullptr, C, tid, pre);
            if constexpr (S1 - K2 != 0) inv_load_tw<A, LOGB, LOGT, S1 - K2, K2, false>(tw_next, C, tid, pre);
            inv_store<A, LOGB, LOGT, S1, KI1, true, SCALE>(v, lds, nullptr, C, tid);
        }
        __syncthreads();
        inv_schedule_ptw<A, LOGB, LOGT, S1, SCALE, AO>(lds, nullptr, gdst, C, tid, pre, addend, tw_next, keep);
    }
}

// N = 2^(LOGB+2) inverse, fp64 policy, ONE kernel with the two TOP stages FIRST (r04): decimation in frequency on the natural-order
// input.  With k = k' + m M (M = 2^LOGB, m < 4) and i = 4 i' + c:
//     a[4 i' + c] = N^-1 psi^{-(4i'+c)} sum_k A[k] w^{-ik} = INTT'( B_c )[i'],   B_c[k'] = psi^{-c (2k'+1)} sum_m A[k' + m M] I^{-c m},
// I = psi^{N/2} (I^2 = -1), INTT' = the M-point negacyclic inverse over psi^4 -- whose twiddle tables are the first quarter of this
// limb's own (W[k] = psi^brv(k): brv_16(k) = 4 brv_14(k) for k < 2^14) -- scaled by this ring's N^-1.  So the four output classes c
// are independent M-point transforms of combinations of the four input quarters: the mirror image of k_ntt_fwd_quad.  A
// workgroup reads the four quarters (LDS-DMA stream, as there), forms the operands of classes 2 ph and 2 ph + 1 (the row's other
// workgroup -- same XCD, same iteration -- takes the other two and finds the lines in L2), runs the two transforms and stores
// 16-byte pieces at words 4 i' + 2 ph: one read and one write of the row (2 N 8 bytes) where k_ntt_inv_subpair + k_ntt_inv_top<2>
// move it twice, and every stored word is a FINAL coefficient (k_ntt_inv_subpair's are sub-block results that still need the top
// stages of all four sub-blocks).  Out of place only (the sibling reads the same source words).
//   class 2 ph    : S = (A0 + A2) + sgn (A1 + A3)                      twiddle 1 (ph = 0) / psi^{-2(2k'+1)}
//   class 2 ph + 1: S = (A0 - A2) + sgn I^-1 (A1 - A3)                 twiddle psi^{-(2k'+1)} / psi^{-3(2k'+1)},   sgn = +1 / -1
// The per-point twiddles of a thread's points k' = tid + (e << LOGT) follow from its first one by the uniform factor
// psi^{-c 2^(LOGT+1)} (ntt_limb_t::i2_t0, i2_g): no table loads inside the DMA stream (vmcnt is hand-counted there).
// Ranges (fp64arith.h): inputs canonical (< p): S <= 4 p, exact; products <= p (1/2 + 1.5 a 4) = 2 p; the transforms reduce first.
template <class A, int LOGB, int LOGT>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_inv_quad2(const u64* __restrict__ src, u64* __restrict__ dst,
                                                              const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nitems, u32 limb_mask) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int x = 2;
    constexpr int K1 = pass_k_inv(LOGB, LOGT, LOGB), S1 = LOGB - K1;
    typedef pgeom<LOGB, LOGT, S1, K1> G1;
    constexpr int E = G1::E;
    constexpr int K2 = pass_k_inv(LOGB, LOGT, S1), KL = S1 - K2;  // middle and last pass widths
    static_assert(KL >= 1 && pass_k_inv(LOGB, LOGT, KL) == KL, "three-pass schedule expected");
    typedef pgeom<LOGB, LOGT, 0, KL> GL;
    static_assert(GL::SETS == 1 && GL::LO == LOGT, "last pass: element r of a thread is word tid + (r << LOGT)");
    static_assert(G1::SETS << (LOGT + K1) == 1 << LOGB, "first pass: register position e <-> point tid + (e << LOGT)");
    const size_t ntot = (size_t)1 << (LOGB + x);
    const u32 dmask = dense_mask((u32)sel.n, limb_mask), dn = dense_count(nitems, 1, (u32)sel.n, dmask), niter = (dn + gridDim.x - 1) / gridDim.x;
    bool first = true;
    for (u32 it = 0; it < niter; it++) {
        const u32 item = dense_item(xcd_walk_item(it, blockIdx.x, gridDim.x, x - 1, dn), 1, (u32)sel.n, dmask);
        if (item == ~0u) continue;
        const u32 ph = item & 1u, pl = item >> 1, j = pl % (u32)sel.n;
        if (limb_mask && !((limb_mask >> j) & 1u)) continue;  // the other policy's launch takes this limb
        const ntt_limb_t& L = LT[sel.idx[j]];
        typename A::ctx C = A::make(L);
        C.Winvb = L.i2_winvb;  // boundary pass of the M-point transform over psi^4 (pre = 1: a whole transform of that ring)
        const u64* s = src + pl * ntot;
        u64* d = dst + pl * ntot + 2u * ph;
        typename A::elem w[2][E];  // first-pass operands of the two classes, in the inverse first pass's register map
        {
            const u32 tid = fresh_tid();
            typedef __attribute__((address_space(1))) const double* gptr_t;
            const gptr_t t0 = (gptr_t)L.i2_t0;
            // ph = 0: classes 0 (no twiddle: the constant 1 keeps the phase branch-free) and 1;  ph = 1: classes 2 and 3
            double ta = ph ? t0[(1u << LOGT) + tid] : 1.0;
            double tb = t0[(ph ? (2u << LOGT) : 0u) + tid];
            const ftw_t ga{ph ? L.i2_g[1] : 1.0}, gb{ph ? L.i2_g[2] : L.i2_g[0]}, iinv{L.i2_iinv};
            const double sgn = ph ? -1.0 : 1.0;
            pin_vgpr(ta);
            pin_vgpr(tb);  // landed before the DMA stream starts (its waits count DMA instructions only)
            if (!first) __syncthreads();  // the previous item's last pass has read its LDS image
            dma_stream_load<LOGB, LOGT, 4, 2>(lds, s, tid, [&](int e, const u64* q) {
                const int idx = (e % G1::SETS) * G1::R + (int)brev_bits((u32)(e / G1::SETS), K1);
                const double a0 = fp_from_u64(q[0]), a1 = fp_from_u64(q[1]), a2 = fp_from_u64(q[2]), a3 = fp_from_u64(q[3]);
                // the twiddle recurrence advances WITH the data: left free, the scheduler runs the (data-independent) chain to its end
                // ahead of the first DMA wait and parks 2 x 32 twiddles in registers (32 spilled in that form)
                asm volatile("" : "+v"(ta), "+v"(tb) : "v"(a0));
                const double slo = fp_fma(sgn, a1 + a3, a0 + a2);
                const double t = fp_mulmod_c(a1 - a3, iinv, C.p, C.pinv);
                const double shi = fp_fma(sgn, t, a0 - a2);
                w[0][idx] = fp_mulmod_c(slo, ftw_t{ta}, C.p, C.pinv);
                w[1][idx] = fp_mulmod_c(shi, ftw_t{tb}, C.p, C.pinv);
                ta = fp_mulmod_c(ta, ga, C.p, C.pinv);
                tb = fp_mulmod_c(tb, gb, C.p, C.pinv);
            });
            first = false;
            TFHE_SCHED_FENCE();  // the first pass's twiddle requests stay behind the load phase (register pressure)
        }
        u64 held[E];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 tid = fresh_tid();
            TFHE_SCHED_FENCE();
            // the plain three-pass schedule (as k_ntt_inv_subpair; the twiddle prefetch of inv_schedule_ptw costs 62 registers next
            // to the 64 of the waiting class and the 64 of the held results: 166 spilled in the first version of this kernel)
            __syncthreads();  // every wave has read the last DMA chunk / the previous transform's last pass has read LDS
            {
                typename A::elem* v = w[half];
#pragma unroll
                for (int e = 0; e < E; e++) v[e] = fp_reduce(v[e], C.p, C.pinv);
                inv_compute<A, LOGB, LOGT, S1, K1, true, true, 0, -1, no_hook, true>(v, nullptr, nullptr, C, tid, 1u);
                inv_store<A, LOGB, LOGT, S1, K1, true, true>(v, lds, nullptr, C, tid);
            }
            __syncthreads();
            ntt_inv_pass<A, LOGB, LOGT, KL, K2, false, false, true, TFHE_ES(9) != 0>(lds, nullptr, nullptr, C, tid, 1u, 0, 0u);
            __syncthreads();
            {
                u64 r3[E];
                typename A::elem v[E];
                inv_load_data<LOGB, LOGT, 0, KL, false>(r3, lds, nullptr, tid, 0, 0u);
                inv_compute<A, LOGB, LOGT, 0, KL, false, true, 0>(v, r3, nullptr, C, tid, 1u);
                if (half == 0) {
#pragma unroll
                    for (int r = 0; r < E; r++) held[r] = A::out_inv_scaled(v[r], C);
                } else {
#pragma unroll
                    for (int r = 0; r < E; r++) {
                        u64x2_t ww;
                        ww.x = held[r];
                        ww.y = A::out_inv_scaled(v[r], C);
                        *(u64x2_t*)(d + ((u64)(tid + ((u32)r << LOGT)) << x)) = ww;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole RNS-digit key switch for one (ciphertext b, working limb j) per workgroup pass, fp64 policy,
// whole-transform blocks:   out_s[b][j] = c_s[b][j] + INTT_j( Σ_i evk_{i,s}[j] ⊙ NTT_j(lift_{i→j}(c_end[b][i])) )
// The digit transforms never leave the CU: after the last forward pass each thread holds 2^(LOGB-LOGT) transform values
// in registers, multiplies them by the two key components (streamed from L2; the key is shared by the whole batch) and
// accumulates; the two accumulators then go straight into the inverse transform, whose first pass uses the same
// natural-order register map.  HBM traffic per ciphertext: `level` source rows + (polys-1)*level addend rows read and
// 2*level rows written, instead of writing and re-reading level*level digit rows and 2*level accumulator rows.
// ------------------------------------------------------------------------------------------------
// The key rows a call needs, as doubles: evd[(i*2 + comp)*nw + j] = double(evk[(i*2 + comp)*Lk + w.idx[j]]) (bit patterns).
// One small launch per key switch call (level*2*nw rows) takes the two u64 -> double conversions per key word out of the
// fused kernel's inner product, where every word is used once per ciphertext of the batch.
// fold_ninv: the words are multiplied by N^-1 mod q_j on the way (k_ks_fused runs its inverse transforms unscaled).
__global__ __launch_bounds__(256) void k_evk_to_f64(const u64* __restrict__ evk, u64* __restrict__ evd,
                                                     const ntt_limb_t* __restrict__ LT, ks_arg_t KA, int Lk, u32 n, int fold_ninv, int x,
                                                     u64 ginv, int pair) {
    const u32 row = blockIdx.x, j = row % (u32)KA.nw, ic = row / (u32)KA.nw;
    const ntt_limb_t& L = LT[KA.w.idx[j]];
    const u64* s = evk + ((size_t)ic * Lk + KA.w.idx[j]) * n;
    // pair (k_ks_fused, r05): the two components of digit i interleaved word by word -- evd[(i nw + j)][k][comp] -- so that the
    // product phase takes both key words of a position with ONE 16-byte load (half the load instructions in flight for the same bytes)
    u64* d = pair ? evd + (((size_t)(ic >> 1) * KA.nw + j) * n) * 2 + (ic & 1u) : evd + (size_t)row * n;
    for (u32 k = blockIdx.y * blockDim.x + threadIdx.x; k < n; k += gridDim.y * blockDim.x) {
        // ginv != 0: the key of x -> x^g PREPARED on the way (rows permuted by g^-1, as tfhe_galois_key_prepare) -- the rotation in the
        // tail (k_ks_top_tail_rot): the key sums of the UNrotated digits are then sigma_g^-1 of the rotated ciphertext's
        double v = fp_from_u64(s[ginv ? galois_ntt_pos(k, ginv, n) : k]);
        if (fold_ninv) {
            v = fp_mulmod_c(v, L.ninv_d, L.pd, L.pinvd);
            v = v < 0.0 ? v + L.pd : v;
        }
        u64 b;
        __builtin_memcpy(&b, &v, 8);
        // x > 0 (k_ks_fused_sub at N = 2^16, x = 2): a sub-block reads the words at positions (nat << x) + c -- stored class by
        // class (c major), so that its lanes read consecutive words instead of every 2^x-th one
        const size_t pos = x ? ((size_t)(k & ((1u << x) - 1u)) * (n >> x)) + (k >> x) : (size_t)k;
        d[pair ? pos * 2 : pos] = b;
    }
}
// PRELIFT: the rows of c[end] arrive as centred doubles (bfv_contract_narrow<.., LIFTED>): the lift is a bit cast
template <int LOGB, int LOGT, int MASK>
constexpr size_t fused_lds_bytes() {
    return ((size_t)lds_words<LOGB, LOGT>() + (MASK ? 2 * (size_t)fused_tw_words<LOGB, LOGT>() : 0)) * 8;
}
// SPMODE (special prime, N <= 2^LOGB): the ModulusRaised contraction inside the kernel instead of a tail kernel over T.
//   1: the items are the SPECIAL limb of every ciphertext only (nitems = batch); their two coefficient rows t_P go to
//      T [batch][2][N] (`out`)
//   2: the items are the `level` ciphertext limbs (nitems = batch * level); the final store of each is
//      out_s[b][j] = (INTT(S_j) - [t_P]) P^-1 + c_s[b][j]  (ArithFpMD) with t_P read from `tsp` (launch 1's rows) -- nothing
//      intermediate is written, k_ks_rescale_add's 32 row moves per ciphertext become 2 written + 2 (x level, L2 hits) read.
//   0: as before (no special prime: "+ c" in the store; special prime: sums to T for k_ks_rescale_add).
template <class A, int LOGB, int LOGT, bool PRELIFT = false, int SPMODE = 0>
__global__ __launch_bounds__(1 << LOGT) void k_ks_fused(const u64* __restrict__ evd, const u64* __restrict__ ct,
                                                         u64* __restrict__ out, const ntt_limb_t* __restrict__ LT,
                                                         ks_arg_t KA, int Lk, u32 nitems, const u64* __restrict__ tsp = nullptr,
                                                         const u64* __restrict__ zero_row = nullptr) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    static_assert(K3 >= 1 && pass_k_fwd(LOGB, LOGT, K1 + K2) == K3, "three-pass forward schedule expected");
    constexpr int KI1 = pass_k_inv(LOGB, LOGT, LOGB);
    static_assert(KI1 == K3, "forward last pass and inverse first pass must share the register map");
    typedef pgeom<LOGB, LOGT, 0, K1> G1;
    typedef pgeom<LOGB, LOGT, K1 + K2, K3> G3;
    constexpr int E = G3::E;
    // special != 0 (ModulusRaised): the working limbs are the `level` ciphertext limbs plus the special prime (nw = level + 1);
    // the transformed sums go to T [batch][2][nw][N] for k_ks_rescale_add instead of being added to c and written to out
    const u32 level = (u32)KA.level, polys = (u32)KA.polys, nw = (u32)KA.nw;
    const u32 add_s = polys == 3 ? 2u : 1u;
    bool first = true;
#ifndef TFHE_XCD_KS  // measured: nothing on the headline's key switch, - 6 % on a 7-limb one (one limb per XCD: its 32 CUs then
#define TFHE_XCD_KS 0  // read the same key rows in step, through the same L2 channels) -- the plain walk stays
#endif
    const u32 niter = xcd_limb_niter<TFHE_XCD_KS>(gridDim.x, nitems);
    for (u32 it = 0; it < niter; it++) {
        const u32 nper = SPMODE == 1 ? 1u : (SPMODE >= 2 ? level : nw);   // items per ciphertext
        const u32 item = xcd_limb_walk<TFHE_XCD_KS>(it, blockIdx.x, gridDim.x, nper, nitems);
        if (item == ~0u) break;
        const u32 b = item / nper, j = SPMODE == 1 ? level : item % nper;
        const ntt_limb_t& Lj = LT[KA.w.idx[j]];
        typename A::ctx C = A::make(Lj);
        if constexpr (TFHE_TWL_KS != 0) fused_fill_tw<A, LOGB, LOGT, TFHE_TWL_KS>(lds, C);
        lift_t lf;
        lf.qj = Lj.q;
        lf.bj = Lj.br;
        typename A::elem acc[2][E];
#pragma unroll
        for (int e = 0; e < E; e++) acc[0][e] = acc[1][e] = 0;
        for (u32 i = 0; i < level; i++) {
            const u32 tid = fresh_tid();
            lf.qi = LT[KA.w.idx[i]].q;
            lf.half = lf.qi >> 1;
            const u64* grow = ct + ((size_t)((b * polys + polys - 1) * level + i) << LOGB);
            typename A::elem v[E];
            kst(1);
            fused_fwd_to_regs<A, LOGB, LOGT, PRELIFT, (TFHE_TWL_KS & 1) != 0>(lds, grow, C, first, v, &lf);
            // multiply-accumulate with the key: component 1 (masked) feeds out_0, component 0 (mask) feeds out_1
            const u64x2_t* e_pair = (const u64x2_t*)(evd + ((((size_t)i * nw + j) << LOGB) << 1));   // key words as doubles, (mask, masked) per position (k_evk_to_f64, pair)
#pragma unroll
            for (int u = 0; u < G3::SETS; u++) {
                u32 c0, hi, base;
                G3::template coords<true>(tid, u, c0, hi, base);
#pragma unroll
                for (int r = 0; r < G3::R; r++) {
                    const u32 nat = (brev_bits((u32)r, K3) << (LOGB - K3)) + c0;
                    const int e = u * G3::R + r;
#ifdef TFHE_ABL_NOKEYS  // design aid: key words from arithmetic (wrong results, no key traffic)
                    const typename A::tw k1{(double)(nat | 1u) * 4097.0 + C.pinv}, k0{(double)(nat | 3u) * 257.0 + C.pinv};
#else
                    const u64x2_t kw = e_pair[nat];
                    const typename A::tw k1{A::from_lds(kw.y)}, k0{A::from_lds(kw.x)};
#endif
                    // range: y reduced to |y| <= p/2, so every term is <= (1/2 + 0.75 a) p = 0.69 p and eight of them stay
                    // below the 7.9 p exactness limit (fp64arith.h); the accumulators are swept every eighth digit
                    const double y = A::pre_product(v[e], C);
                    acc[0][e] += fp_mulmod_c(y, k1, C.p, C.pinv);
                    acc[1][e] += fp_mulmod_c(y, k0, C.p, C.pinv);
                }
            }
            if ((i & 7u) == 7u && i + 1 < level) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    acc[0][e] = fp_reduce(acc[0][e], C.p, C.pinv);
                    acc[1][e] = fp_reduce(acc[1][e], C.p, C.pinv);
                }
            }
        }
        // inverse transforms of the two accumulators; the first pass takes them from registers (same natural-order map)
#ifdef TFHE_KS_TRACE
#pragma unroll
        for (int e = 0; e < E; e++) { pin_vgpr(acc[0][e]); pin_vgpr(acc[1][e]); }
#endif
        kst(5);
#pragma unroll
        for (int sidx = 0; sidx < 2; sidx++) {
            if (sidx) kst(6);
            if constexpr (SPMODE >= 2) {   // 3: a rotation finished in the store (ArithFpMDR; KA.rot_g, `ct` the unrotated input)
                C.md_pinv = (double)KA.pinv[j].w;   // P^-1 mod q_j < 2^52: exact
                C.md_c = (u32)sidx < add_s ? ct + ((size_t)((b * polys + sidx) * level + j) << LOGB) : zero_row;
                C.md_g = KA.rot_g;
                u64* gdst = out + ((size_t)((b * 2 + sidx) * level + j) << LOGB);
                typedef typename std::conditional<SPMODE == 3, ArithFpMDR, ArithFpMD>::type AOUT;
                fused_inv_from_regs<A, LOGB, LOGT, false, (TFHE_TWL_KS & 2) != 0, AOUT>(lds, acc[sidx], gdst, C, tsp + ((size_t)(b * 2 + sidx) << LOGB));
            } else if constexpr (SPMODE == 1) {
                u64* gdst = out + ((size_t)(b * 2 + sidx) << LOGB);
                fused_inv_from_regs<A, LOGB, LOGT, false, (TFHE_TWL_KS & 2) != 0>(lds, acc[sidx], gdst, C, nullptr);
            } else {
                const u64* addend = (!KA.special && (u32)sidx < add_s) ? ct + ((size_t)((b * polys + sidx) * level + j) << LOGB) : nullptr;
                u64* gdst = out + ((size_t)((b * 2 + sidx) * nw + j) << LOGB);
                fused_inv_from_regs<A, LOGB, LOGT, false, (TFHE_TWL_KS & 2) != 0>(lds, acc[sidx], gdst, C, addend);  // N^-1 is in the key rows (k_evk_to_f64)
            }
        }
        kst(7);
    }
}

// ------------------------------------------------------------------------------------------------
// The same for N = 2^(LOGB+X), X = 1, 2: one (ciphertext b, working limb j, sub-block sb) per workgroup pass.  In the
// transform domain the key product is pointwise, so each of the 2^X sub-blocks of the transform can be carried on its
// own: the workgroup forms its sub-block's operands from the 2^X parts of the lifted source row (its one output of the
// top stages, as k_ntt_fwd_pair / k_ntt_fwd_quad), runs the sub-block's forward passes, multiplies by the key words at
// the natural-order positions (nat << X) + brv_X(sb) and accumulates in registers; the two accumulators then go through
// the sub-block's inverse passes and are written, canonical, to T [batch][2][nw][2^X][2^LOGB].  The inverse top stages
// (which need all sub-blocks) are k_ntt_inv_top<X> over T, followed by the usual tail (k_ks_add_ct / k_ks_rescale_add).
// The digit rows and their transforms never reach HBM; the sub-block items of a (b, j) run on one XCD (xcd_walk_item)
// and share the source rows in its L2.
// ------------------------------------------------------------------------------------------------
// load phase of k_ks_fused_sub: LDS-DMA streaming (dma_stream_load) at X = 2, piece-wise register loads at X = 1 (measured)
#ifndef TFHE_FSUB_DMA2
#define TFHE_FSUB_DMA2 1
#endif
#ifndef TFHE_FSUB_DMA1
#define TFHE_FSUB_DMA1 0
#endif
template <class A, int LOGB, int LOGT, int X>
__global__ __launch_bounds__(1 << LOGT) void k_ks_fused_sub(const u64* __restrict__ evd, const u64* __restrict__ ct,
                                                             u64* __restrict__ T, const ntt_limb_t* __restrict__ LT,
                                                             ks_arg_t KA, int Lk, u32 nitems) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    static_assert(K3 >= 1 && pass_k_fwd(LOGB, LOGT, K1 + K2) == K3, "three-pass forward schedule expected");
    constexpr int KI1 = pass_k_inv(LOGB, LOGT, LOGB), S1 = LOGB - KI1;
    static_assert(KI1 == K3, "forward last pass and inverse first pass must share the register map");
    typedef pgeom<LOGB, LOGT, K1 + K2, K3> G3;
    constexpr int E = G3::E;
    const u32 level = (u32)KA.level, polys = (u32)KA.polys, nw = (u32)KA.nw;
    const u32 niter = (nitems + gridDim.x - 1) / gridDim.x;
    bool first = true;
    for (u32 it = 0; it < niter; it++) {
        const u32 item = xcd_walk_item(it, blockIdx.x, gridDim.x, X, nitems);
        if (item == ~0u) continue;
        const u32 sb = item & ((1u << X) - 1u), pl = item >> X, b = pl / nw, j = pl % nw, pre = (1u << X) + sb;
        const ntt_limb_t& Lj = LT[KA.w.idx[j]];
        const typename A::ctx C = A::make(Lj);
        lift_t lf;
        lf.qj = Lj.q;
        lf.bj = Lj.br;
        // top stages: sub-block sb = (h, q) takes x0 +- W[1] x1 (X = 1), or (x0 +- W[1] x2) +- W[2+h] (x1 +- W[1] x3) (X = 2)
        const typename A::tw w1 = A::ld_fwd(C, 1u), w2 = A::ld_fwd(C, 2u + (sb >> 1));
        const double sgn_h = (sb >> (X - 1)) ? -1.0 : 1.0, sgn_q = (sb & 1u) ? -1.0 : 1.0;
        typename A::elem acc[2][E];
#pragma unroll
        for (int e = 0; e < E; e++) acc[0][e] = acc[1][e] = 0;
        for (u32 i = 0; i < level; i++) {
            const u32 tid = fresh_tid();
            lf.qi = LT[KA.w.idx[i]].q;
            lf.half = lf.qi >> 1;
            const u64* grow = ct + ((size_t)((b * polys + polys - 1) * level + i) << (LOGB + X));
            typename A::elem v[E];
            {
                u64 op[E];  // first-pass operands of this sub-block (element bits)
                // (the LDS-DMA streaming of dma_stream_load, which pays in k_ntt_fwd_quad, measured 7-9 % SLOWER here: cfg#3
                // 32.6 k against 35.1 k key switches/s -- the piece-wise register loads below stay)
                // (r04: the same phase software-pipelined by hand -- inline-asm requests, pieces of 4 points, 24 requests in flight under
                // the arithmetic of a piece, hand-counted vmcnt waits; bit-exact, scratch unchanged -- measured 7 % SLOWER: 34.1 k against
                // 36.6 k key switches/s at cfg#3, although a build without the row loads gains 13 %.  profiles/LOG.md.)
                constexpr int PC = 4 * X, PE = E / PC;  // pieces of the load phase (bounds the raw words in flight)
                if constexpr (X == 2 ? TFHE_FSUB_DMA2 : TFHE_FSUB_DMA1) {
                    // the 2^X parts: streamed through the LDS as in k_ntt_fwd_quad (the LDS holds no row image here)
                    if (!first) __syncthreads();  // the previous transform's last pass has read LDS
                    dma_stream_load<LOGB, LOGT, 1 << X, 2>(lds, grow, tid, [&](int e, const u64* q) {
                        double xin[1 << X];
#pragma unroll
                        for (int m = 0; m < (1 << X); m++) xin[m] = A::from_global_lift(q[m], C, lf, true);
                        double z;
                        if constexpr (X == 1) {
                            z = fp_fma(sgn_h, fp_mulmod_c(xin[1], w1, C.p, C.pinv), xin[0]);
                        } else {
                            const double ya = fp_fma(sgn_h, fp_mulmod_c(xin[2], w1, C.p, C.pinv), xin[0]);
                            const double yb = fp_fma(sgn_h, fp_mulmod_c(xin[3], w1, C.p, C.pinv), xin[1]);
                            z = fp_fma(sgn_q, fp_mulmod_c(yb, w2, C.p, C.pinv), ya);
                        }
                        op[e] = A::to_lds(A::top_reduce(z, C));
                    });
                } else
#pragma unroll
                for (int h = 0; h < PC; h++) {
                    u64 q[1 << X][PE];
                    u32 tp = tid;
                    if (h > 0) order_after(tp, op[h * PE - 1]);
#pragma unroll
                    for (int r = 0; r < PE; r++) {
                        const u32 k = tp + ((u32)(h * PE + r) << LOGT);
#pragma unroll
                        for (int m = 0; m < (1 << X); m++) {
#ifdef TFHE_ABL_NOROWS  // design aid: operands from arithmetic (wrong results, no row traffic)
                            q[m][r] = (u64)(k * 131u + (u32)m * 7919u) + (u64)(size_t)grow; pin_vgpr(q[m][r]);
#elif defined(TFHE_ABL_TOPONCE)  // design aid (r06, wrong results): a sub-block workgroup loads and lifts ONE of the row's 2^X parts and
                            // pays no top-stage product -- the upper bound of "the top stage paid once per row" 
                            if (m == 0) q[m][r] = grow[k + ((sb & ((1u << X) - 1u)) << LOGB)];
#else
                            q[m][r] = grow[k + ((u32)m << LOGB)];
#endif
                        }
                    }
                    TFHE_SCHED_FENCE();
#pragma unroll
                    for (int r = 0; r < PE; r++) {
                        // loosely lifted digits (|v| <= p): sums <= 1.88 p, products <= 1.21 p, <= 3.09 p before the reduction
#ifdef TFHE_ABL_TOPONCE
                        const double z = A::from_global_lift(q[0][r], C, lf, true);
#else
                        double xin[1 << X];
#pragma unroll
                        for (int m = 0; m < (1 << X); m++) xin[m] = A::from_global_lift(q[m][r], C, lf, true);
                        double z;
                        if constexpr (X == 1) {
                            z = fp_fma(sgn_h, fp_mulmod_c(xin[1], w1, C.p, C.pinv), xin[0]);
                        } else {
                            const double ya = fp_fma(sgn_h, fp_mulmod_c(xin[2], w1, C.p, C.pinv), xin[0]);
                            const double yb = fp_fma(sgn_h, fp_mulmod_c(xin[3], w1, C.p, C.pinv), xin[1]);
                            z = fp_fma(sgn_q, fp_mulmod_c(yb, w2, C.p, C.pinv), ya);
                        }
#endif
                        op[h * PE + r] = A::to_lds(A::top_reduce(z, C));
                    }
                    TFHE_SCHED_FENCE();
                }
                constexpr bool DMA = X == 2 ? TFHE_FSUB_DMA2 : TFHE_FSUB_DMA1;
                if (!DMA && !first) __syncthreads();  // the previous transform's last pass has read LDS
                first = false;
                fwd_compute<A, LOGB, LOGT, 0, K1, false, false, 0>(v, op, nullptr, C, tid, pre);
                if (DMA) __syncthreads();  // every wave has read its ring
                fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
            }
            __syncthreads();
            ntt_fwd_pass<A, LOGB, LOGT, K1, K2, false, false>(lds, nullptr, nullptr, C, tid, pre, 0, 0u);
            __syncthreads();
            {
                u64 r3[E];
                fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(r3, lds, nullptr, tid);
                fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, r3, nullptr, C, tid, pre);
            }
            // multiply-accumulate with the key words at natural-order positions 2 nat + sb
            // (doubles from k_evk_to_f64.  X = 2: stored class by class, the sub-block's words are consecutive -- 33.1 k -> 34.5 k key
            // switches/s on 7 x 50 bit; at X = 1 the same layout cost 24 more spilled registers and 2.8 % at cfg#3: interleaved there)
            constexpr bool CM = X == 2;
            const size_t eoff = CM ? ((size_t)brev_bits(sb, X) << LOGB) : (size_t)brev_bits(sb, X);
            const u64x2_t* e_pair = (const u64x2_t*)(evd + ((((size_t)i * nw + j) << (LOGB + X)) << 1)) + eoff;   // (mask, masked) per position (k_evk_to_f64, pair)
#pragma unroll
            for (int u = 0; u < G3::SETS; u++) {
                u32 c0, hi, base;
                G3::template coords<true>(tid, u, c0, hi, base);
#pragma unroll
                for (int r = 0; r < G3::R; r++) {
                    const u32 nat = (brev_bits((u32)r, K3) << (LOGB - K3)) + c0;
                    const int e = u * G3::R + r;
                    const u64x2_t kw = e_pair[CM ? nat : nat << X];
                    const typename A::tw k1{A::from_lds(kw.y)}, k0{A::from_lds(kw.x)};
                    const double y = A::pre_product(v[e], C);  // range: as k_ks_fused
                    acc[0][e] += fp_mulmod_c(y, k1, C.p, C.pinv);
                    acc[1][e] += fp_mulmod_c(y, k0, C.p, C.pinv);
                }
            }
            if ((i & 7u) == 7u && i + 1 < level) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    acc[0][e] = fp_reduce(acc[0][e], C.p, C.pinv);
                    acc[1][e] = fp_reduce(acc[1][e], C.p, C.pinv);
                }
            }
        }
        // the sub-block's inverse passes on the two accumulators (first pass from registers: same natural-order map)
#pragma unroll
        for (int sidx = 0; sidx < 2; sidx++) {
            u64* gdst = T + ((size_t)((b * 2 + sidx) * nw + j) << (LOGB + X)) + ((size_t)sb << LOGB);
            // (r04: the middle pass's twiddles requested before the exchange, as in k_ks_fused)
            fused_inv_from_regs<A, LOGB, LOGT, false, false, A>(lds, acc[sidx], gdst, C, nullptr, nullptr, pre);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The inside of a BFV multiplication for one (ciphertext b, limb j of ℛbig) per workgroup pass, fp64 policy, whole-
// transform blocks:   T_k[b][j] = INTT_j( tensor_k( NTT_j(a0), NTT_j(a1), NTT_j(b0), NTT_j(b1) ) ),  k = 0, 1, 2
// (enc_mul, rlwe_she.jl:255-258, between mul_expand and mul_contract).  The four forward transforms, the tensor product
// and the three inverse transforms never leave the CU: NTT(a0), NTT(a1) are held in registers (2 x 32 doubles per
// thread) while b0 and b1 are transformed; products are formed in place and go straight into the inverse transform
// (its first pass shares the forward last pass's natural-order register map).  One intermediate row (a1 b0) makes a
// round trip through a per-workgroup scratch row (L2 / Infinity Cache).  HBM traffic per limb: 4 rows read + 3 written
// (+ 2 scratch) instead of 21 row moves for forward kernel + tensor kernel + inverse kernel.
// ------------------------------------------------------------------------------------------------
struct core_alt_t {  // limbs of ℛbig that are limbs of ℛ: read from the input ciphertexts ([nct][2][ns][N]) instead of E
    const u64 *a, *b;
    int ns;
    signed char idx[TFHE_MAX_LIMBS];  // limb j of ℛbig -> limb of ℛ, or -1
};
// OUTD: the three result rows leave as reduced doubles (ArithFpD) for the narrow contraction (k_bfv_contract_fast<.., TD>)
template <class A, int LOGB, int LOGT, bool OUTD = false>
__global__ __launch_bounds__(1 << LOGT) void k_bfv_core_fused(const u64* __restrict__ Ea, const u64* __restrict__ Eb,
                                                               u64* __restrict__ T, u64* __restrict__ scratch,
                                                               const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 nitems,
                                                               core_alt_t alt) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K3 = LOGB - pass_k_fwd(LOGB, LOGT, 0) - pass_k_fwd(LOGB, LOGT, pass_k_fwd(LOGB, LOGT, 0));
    static_assert(pass_k_inv(LOGB, LOGT, LOGB) == K3, "forward last pass and inverse first pass must share the register map");
    typedef pgeom<LOGB, LOGT, LOGB - K3, K3> G3;
    constexpr int E = G3::E;
    typedef typename std::conditional<OUTD, ArithFpD, A>::type AO;
    const u32 nb = (u32)sel.n;
    u64* const srow = scratch + ((size_t)blockIdx.x << LOGB);
    bool first = true;
    const u32 niter = xcd_limb_niter(gridDim.x, nitems);
    for (u32 it = 0; it < niter; it++) {
        const u32 item = xcd_limb_walk(it, blockIdx.x, gridDim.x, nb, nitems);
        if (item == ~0u) break;
        const u32 b = item / nb, j = item % nb;
        typename A::ctx C = A::make(LT[sel.idx[j]]);
        if constexpr (TFHE_TWL_CORE != 0) fused_fill_tw<A, LOGB, LOGT, TFHE_TWL_CORE>(lds, C);
        const size_t r0 = ((size_t)(b * 2 + 0) * nb + j) << LOGB, r1 = ((size_t)(b * 2 + 1) * nb + j) << LOGB;
        u64* const t0 = T + (((size_t)(b * 3 + 0) * nb + j) << LOGB);
        u64* const t1 = T + (((size_t)(b * 3 + 1) * nb + j) << LOGB);
        u64* const t2 = T + (((size_t)(b * 3 + 2) * nb + j) << LOGB);
        typename A::elem A0[E], A1[E], v[E];
        // range (fp64arith.h): the held transforms are reduced to |A| <= p/2; the running one stays lazy (|v| <= 7.68 p at the end
        // of the forward plan), so |A v| / p <= 3.84 p and a product is exact with |r| <= (1/2 + 1.5 a 3.84) p = 1.94 p;
        // a0 b1 + a1 b0 <= 3.9 p < 7.9 p; fused_inv_from_regs reduces before the inverse butterflies
        const u64 *pa0 = Ea + r0, *pa1 = Ea + r1, *pb0 = Eb + r0, *pb1 = Eb + r1;
        if (alt.a && alt.idx[j] >= 0) {
            const size_t s0 = ((size_t)(b * 2 + 0) * alt.ns + alt.idx[j]) << LOGB, s1 = ((size_t)(b * 2 + 1) * alt.ns + alt.idx[j]) << LOGB;
            pa0 = alt.a + s0; pa1 = alt.a + s1; pb0 = alt.b + s0; pb1 = alt.b + s1;
        }
        fused_fwd_to_regs<A, LOGB, LOGT, false, (TFHE_TWL_CORE & 1) != 0>(lds, pa0, C, first, A0);
#pragma unroll
        for (int e = 0; e < E; e++) A0[e] = fp_reduce(A0[e], C.p, C.pinv);
        fused_fwd_to_regs<A, LOGB, LOGT, false, (TFHE_TWL_CORE & 1) != 0>(lds, pa1, C, first, A1);
#pragma unroll
        for (int e = 0; e < E; e++) A1[e] = fp_reduce(A1[e], C.p, C.pinv);
        fused_fwd_to_regs<A, LOGB, LOGT, false, (TFHE_TWL_CORE & 1) != 0>(lds, pb0, C, first, v);
        {
            const u32 tid = fresh_tid();
#pragma unroll
            for (int u = 0; u < G3::SETS; u++) {
                u32 c0, hi, base;
                G3::template coords<true>(tid, u, c0, hi, base);
#pragma unroll
                for (int r = 0; r < G3::R; r++) {
                    const int e = u * G3::R + r;
                    const u32 nat = (brev_bits((u32)r, K3) << (LOGB - K3)) + c0;
                    srow[nat] = A::to_lds(fp_mulmod_c(v[e], ftw_t{A1[e]}, C.p, C.pinv));               // a1 b0, parked as a lazy double
                    v[e] = fp_mulmod_c(v[e], ftw_t{A0[e]}, C.p, C.pinv);                                // a0 b0, in place
                }
            }
            fused_inv_from_regs<A, LOGB, LOGT, true, (TFHE_TWL_CORE & 2) != 0, AO>(lds, v, t0, C, nullptr);
        }
        fused_fwd_to_regs<A, LOGB, LOGT, false, (TFHE_TWL_CORE & 1) != 0>(lds, pb1, C, first, v);
        {
            const u32 tid = fresh_tid();
#pragma unroll
            for (int u = 0; u < G3::SETS; u++) {  // products in place: A0 <- a0 b1 + a1 b0, A1 <- a1 b1
                u32 c0, hi, base;
                G3::template coords<true>(tid, u, c0, hi, base);
#pragma unroll
                for (int r = 0; r < G3::R; r++) {
                    const int e = u * G3::R + r;
                    const u32 nat = (brev_bits((u32)r, K3) << (LOGB - K3)) + c0;
                    A0[e] = fp_mulmod_c(v[e], ftw_t{A0[e]}, C.p, C.pinv) + A::from_lds(srow[nat]);
                    A1[e] = fp_mulmod_c(v[e], ftw_t{A1[e]}, C.p, C.pinv);
                }
            }
            fused_inv_from_regs<A, LOGB, LOGT, true, (TFHE_TWL_CORE & 2) != 0, AO>(lds, A0, t1, C, nullptr);
            fused_inv_from_regs<A, LOGB, LOGT, true, (TFHE_TWL_CORE & 2) != 0, AO>(lds, A1, t2, C, nullptr);
        }
    }
}

// RNS digits as a separate pass (only for N > 2^14, where the lift is not fused into the NTT loads):
// dig [batch][level][nw][N]; digit i, limb j = centred([c_end]_{q_i}) mod q_w[j]  (rlwe_she.jl:326-329)
__global__ __launch_bounds__(256) void k_ks_digits(const u64* __restrict__ ct, u64* __restrict__ dig,
                                                    const ntt_limb_t* __restrict__ LT, ks_arg_t A, u32 n, u32 limb_mask) {
    const u32 row = blockIdx.x, j = row % (u32)A.nw, i = (row / (u32)A.nw) % (u32)A.level,
              b = row / ((u32)A.nw * (u32)A.level);
    if (limb_mask && !((limb_mask >> j) & 1u)) return;  // the lift-fused fp64 transforms take this working limb
    lift_t lf;
    lf.qi = LT[A.w.idx[i]].q; lf.half = lf.qi >> 1; lf.qj = LT[A.w.idx[j]].q; lf.bj = LT[A.w.idx[j]].br;
    const u64* c = ct + (((size_t)b * A.polys + (A.polys - 1)) * A.level + i) * n;
    u64* d = dig + (size_t)row * n;
    for (u32 k = blockIdx.y * blockDim.x + threadIdx.x; k < n; k += gridDim.y * blockDim.x) d[k] = lift_digit(c[k], lf);
}
// Base-2^w digits of the key switch (relin_window != 0, rlwe_she.jl:330-338): digit i of convert(Integer, x) for every
// coefficient x of c[end], x in [0, Q) reconstructed exactly from its residues (conv_core.h; a single limb is its own
// integer).  dig: [batch][nwin][level][N], the same small value in every limb.  One thread per coefficient.
__global__ __launch_bounds__(256) void k_ks_window_digits(const u64* __restrict__ ct, u64* __restrict__ dig,
                                                           const conv_tab_t* __restrict__ T, int level, int wbits, int nwin,
                                                           int polys, u32 n, u32 gx, int nw) {
    const u32 k = (blockIdx.x % gx) * 256 + threadIdx.x;
    const size_t b = blockIdx.x / gx;
    if (k >= n) return;
    // dig [batch][nwin][nw][N], nw = level (+ 1 with the special prime, modulusraising.jl:35-41)
    window_digits_coeff(T, ct + ((b * polys + polys - 1) * level) * n + k, n, level, wbits, nwin,
                        dig + (b * nwin * nw) * n + k, (size_t)nw * n, n, nw);
}

// out[b][s][j] += c[b][s][j] for the components that have one (N > 2^14 path); rows = batch*2*level
__global__ __launch_bounds__(256) void k_ks_add_ct(const u64* __restrict__ ct, u64* __restrict__ out,
                                                    const ntt_limb_t* __restrict__ LT, ks_arg_t A, u32 n, u32 add_s) {
    const u32 row = blockIdx.x, j = row % (u32)A.level, s = (row / (u32)A.level) & 1u, b = row / (2u * (u32)A.level);
    if (s >= add_s) return;
    const u64 q = LT[A.w.idx[j]].q;
    const u64* c = ct + (((size_t)b * A.polys + s) * A.level + j) * n;
    u64* o = out + (size_t)row * n;
    for (u32 k = blockIdx.y * blockDim.x + threadIdx.x; k < n; k += gridDim.y * blockDim.x) o[k] = addmod(o[k], c[k], q);
}

// ModulusRaised contraction fused with the "c +" of the key switch:  with x = P*c + S (limb-wise, special limb of
// P*c is 0) the reference's modswitch gives (x_j - [x_P]) P^-1 = c_j + (S_j - [S_P]_{q_j}) P^-1 (mod q_j)
// (modulusraising.jl:35-42, crt.jl:215-220).  T = INTT(S): [batch][2][nw][N]; out: [batch][2][level][N];
// rows = batch*2*level.  add_s: number of leading components s that have a c_s addend (2 for a 3-element input, else 1).
__global__ __launch_bounds__(256) void k_ks_rescale_add(const u64* __restrict__ T, const u64* __restrict__ ct,
                                                         u64* __restrict__ out, const ntt_limb_t* __restrict__ LT, ks_arg_t A,
                                                         rescale_arg_t ra, u32 n, u32 add_s) {
    const u32 row = blockIdx.x, j = row % (u32)A.level, s = (row / (u32)A.level) & 1u, b = row / (2u * (u32)A.level);
    const ntt_limb_t L = LT[A.w.idx[j]];
    const u64* tj = T + (((size_t)b * 2 + s) * A.nw + j) * n;
    const u64* tl = T + (((size_t)b * 2 + s) * A.nw + A.level) * n;
    const u64* c = s < add_s ? ct + (((size_t)b * A.polys + s) * A.level + j) * n : nullptr;
    u64* o = out + (size_t)row * n;
    for (u32 k = blockIdx.y * blockDim.x + threadIdx.x; k < n; k += gridDim.y * blockDim.x) {
        const u64 last = barrett_reduce128(tl[k], 0, L.br);
        u64 v = shoup_full(submod(tj[k], last, L.q), ra.qlinv[j], L.q);
        if (c) v = addmod(v, c[k], L.q);
        o[k] = v;
    }
}

// Hoisted rotations, the tail for ALL rotations of a diagonal product in one launch: T = INTT(S') [R][batch][2][nw][N] holds the
// inverse-transformed key sums of the unrotated digits; rotation r's result is sigma_g( . ) applied per limb in the coefficient
// domain (a signed permutation, BEFORE the floor of the ModulusRaised contraction, which does not commute with sign changes;
// see ks_finish), then out_j = sigma_g(c)_j + (T'_j - [T'_P]) P^-1 (special) or sigma_g(c)_j + T'_j, for s = 0 only the addend.
// Scatter form: the three source rows (T_j, T_P, c_j) are read in order -- coalesced -- and coefficient i goes to position
// g i mod N with the sign of floor(g i / N); the 8-byte stores of a row land in a 512 KiB window that stays in the L2 until its
// lines are complete.  (The gather form -- three scattered 8-byte reads per output -- ran at 0.85 TB/s and was 38 % of the
// encrypted-MNIST evaluation; a gather is bound by the texture addresser, ~8 B/clk/CU.)  out: [R][batch][2][level][N].
struct rot_tail_arg_t {
    u64 g[TFHE_DOT_MAX];
};
// Walk: a permutation of a row uses every cache line of its window 16 times over the WHOLE row, so the window must stay in
// the L2 until the row is done -- with one workgroup per row and a few thousand rows in flight the windows were evicted half
// written and every 8-byte store became a read-modify-write in HBM (4.0 ms per launch = 0.8 TB/s, scatter and gather alike).
// Here the workgroups of one XCD (blockIdx.x & 7 -- workgroups are dealt to the XCDs round-robin) share ONE row group
// (r, b, s) at a time, each taking a slice of the coefficients, so an XCD's 4 MiB L2 holds one or two groups' windows
// (level x N x 8 bytes each); T_P[i] is read once per coefficient for all limbs.
#ifndef TFHE_ROT_TAIL_SLOTS
#define TFHE_ROT_TAIL_SLOTS 128  // workgroups per XCD (measured on the encrypted-MNIST circuit at N = 2^16: 64 -> 43.9 k, 128 -> 46.0 k, 256 -> 41.7 k images/s)
#endif
__global__ __launch_bounds__(256) void k_ks_rot_tail(const u64* __restrict__ T, const u64* __restrict__ ct, u64* __restrict__ out,
                                                      const ntt_limb_t* __restrict__ LT, ks_arg_t A, rescale_arg_t ra, rot_tail_arg_t G,
                                                      u32 n, u32 batch, u32 ngroups) {
    __shared__ u64 lq[TFHE_MAX_LIMBS], lmu[TFHE_MAX_LIMBS], lw[TFHE_MAX_LIMBS], lwp[TFHE_MAX_LIMBS];
    __shared__ u32 lsh[TFHE_MAX_LIMBS];
    const u32 level = (u32)A.level, nw = (u32)A.nw;
    for (u32 j = threadIdx.x; j < level; j += blockDim.x) {
        const ntt_limb_t& L = LT[A.w.idx[j]];
        lq[j] = L.q; lmu[j] = L.br.mu; lsh[j] = L.br.sh; lw[j] = ra.qlinv[j].w; lwp[j] = ra.qlinv[j].wp;
    }
    __syncthreads();
    const u64 P = A.special ? LT[A.w.idx[level]].q : 0;
    const u32 xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const u64 mask2n = 2ull * n - 1;
    for (u32 grp = xcd; grp < ngroups; grp += 8u) {   // grp = (r * batch + b) * 2 + s
        const u32 s = grp & 1u, b = (grp >> 1) % batch, r = (grp >> 1) / batch;
        const u64 g = G.g[r];
        const u64* tg = T + (size_t)grp * nw * n;
        const u64* tl = tg + (size_t)level * n;
        const u64* cg = s == 0 ? ct + ((size_t)b * 2 * level) * n : nullptr;   // 2-element input: only c_1 has an addend (rlwe_she.jl:324)
        u64* og = out + (size_t)grp * level * n;
        for (u32 i = slot * blockDim.x + threadIdx.x; i < n; i += nslot * blockDim.x) {
            const u64 t = ((u64)i * g) & mask2n;
            const bool neg = t >= n;
            const u32 m = (u32)t & (n - 1);
            u64 last = 0;
            if (A.special) {
                last = tl[i];
                if (neg) last = negmod(last, P);
            }
            for (u32 j = 0; j < level; j++) {
                const u64 q = lq[j];
                u64 v = tg[(size_t)j * n + i];
                if (neg) v = negmod(v, q);
                if (A.special) {
                    const barrett_t br{q, lmu[j], lsh[j]};
                    v = shoup_full(submod(v, barrett_reduce128(last, 0, br), q), tw_t{lw[j], lwp[j]}, q);
                }
                if (cg) {
                    u64 a = cg[(size_t)j * n + i];
                    if (neg) a = negmod(a, q);
                    v = addmod(v, a, q);
                }
                og[(size_t)j * n + m] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tfhe_matmul_diag with a special prime, evaluation-domain form (infer.jl:140-149 over rlwe_she.jl:315-347 and
// modulusraising.jl:35-49).  With S' = sum_i NTT(digit_i) (.) key'_{r,i} the key sums of the UNROTATED digits against rotation
// r's prepared key (all nw working limbs, NTT domain), the rotated ciphertext is
//     rot_r = sigma_g( c_0 [s = 0] ) + floor-contraction( sigma_g( INTT(S') ) ),   contraction_j(x) = (x_j - [x_P]) P^-1 mod q_j
// (k_ks_rot_tail).  sigma_g is a ring automorphism, so in the evaluation domain it is the pure permutation pi_g of
// galois_ntt_pos -- no signs -- on every limb, and by linearity of NTT_j
//     NTT_j(rot_r)[k] = ( S'_j[pi k] P^-1 + X0_j[pi k] )  -  P^-1 NTT_j( [ INTT_P( S'_P o pi ) ] mod q_j )[k],     X0 = NTT(c_0):
// only the SPECIAL limb of every sum is inverse-transformed (after the permutation, so that the unsigned representative the
// floor takes is the rotated polynomial's), its lift costs `level` forward transforms per sum -- which the coefficient-domain
// path also spends, on the rotated ciphertext -- and the `level` inverse transforms per sum of that path are not needed.
// Every step is exact modular arithmetic on canonical residues: the accumulated product is bit-identical.
//   k_md_special_perm   P[grp]      = S'[grp][special] o pi_g            (XCD-cooperative gather, grp = (r * batch + b) * 2 + s; both s at once)
//   (inverse transform of P on the special limb)
//   k_md_lift           U[grp][j]   = ([P[grp]] mod q_j) P^-1             (unsigned representative, crt.jl:215-220)
//   (forward transforms of U; where md_lift_is_fused the lift rides on their loads -- ntt_io_t::lift_unsigned -- and k_md_acc
//   multiplies by P^-1 itself: USCALE)
//   (V[grp][j] = S'[grp][j] P^-1 + X0[b][j] [s = 0] is what the key-sum kernels store for j < level: k_ks_inner<.., EPI>, the two
//   components s = 0, 1 of a position interleaved -- epi_pair)
//   k_md_acc            out[b][s][j][k] = diag_0[j][k] X[b][s][j][k] + sum_r diag_{r+1}[j][k] (V[grp][j][pi_r k] - U[grp][j][k])
// The gathers are XCD-cooperative (one row per XCD at a time, as k_ks_rot_tail): a permutation uses every cache line of its
// 8 N-byte window 16 times, so the window has to stay in one L2 until the row is done.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_md_special_perm(const u64* __restrict__ S, u64* __restrict__ P, rot_tail_arg_t G, u32 n, u32 nw,
                                                          u32 level, u32 batch, u32 ngroups) {
    const u32 xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    // small rows: an XCD's workgroups split into teams of n / 256 (a row each) instead of idling beyond the row
    const u32 spt = n / blockDim.x >= nslot ? nslot : (n / blockDim.x ? n / blockDim.x : 1u), teams = nslot / spt, team = slot / spt, ts = slot % spt;
    if (team >= teams) return;
    for (u32 ci = team * 8u + xcd; ci < ngroups / 2u; ci += 8u * teams) {   // ciphertext ci = r * batch + b: both components together
        const u64 g = G.g[ci / batch];
        const u64x2_t* s = (const u64x2_t*)(S + epi_pair(ci, level, 0u, nw, n));   // (the EPI kernels' layout of the sums)
        u64 *d0 = P + (size_t)(2u * ci) * n, *d1 = d0 + n;
        for (u32 m = ts * blockDim.x + threadIdx.x; m < n; m += spt * blockDim.x) {
            const u64x2_t w = s[galois_ntt_pos(m, g, n)];
            d0[m] = w.x;
            d1[m] = w.y;
        }
    }
}
// rows = ngroups * level, row = grp * level + j
__global__ __launch_bounds__(256) void k_md_lift(const u64* __restrict__ P, u64* __restrict__ U, const ntt_limb_t* __restrict__ LT,
                                                  limb_sel_t sel, rescale_arg_t ra, u32 n) {
    const u32 row = blockIdx.x, level = (u32)sel.n, j = row % level, grp = row / level;
    const ntt_limb_t L = LT[sel.idx[j]];
    const u64* s = P + (size_t)grp * n;
    u64* d = U + (size_t)row * n;
    for (u32 i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x)
        d[i] = shoup_full(barrett_reduce128(s[i], 0, L.br), ra.qlinv[j], L.q);
}
// out rows (b, s, j) = (b * 2 + s) * level + j; diag: [R+1][level][N].  One XCD takes the two rows (b, 0, j), (b, 1, j) at a time:
// they share the diagonal words and the permuted positions, and the V windows in flight are the two rows of ONE rotation (a
// version that requested four rotations at a time per coefficient ran slower than the plain loop: eight windows are the whole
// L2).  The memory parallelism comes from KB coefficients per thread instead, with rotation t + 1 requested before rotation t
// is accumulated.
#ifndef TFHE_MD_KB
#define TFHE_MD_KB 8
#endif
#ifndef TFHE_MD_SLOTS
#define TFHE_MD_SLOTS 32   // workgroups per XCD (KB x SLOTS x 256 = 2^16: measured 63.7 k / 65.9 k / 66.9 k images/s at KB = 2 / 4 / 8)
#endif
template <bool USCALE>
__global__ __launch_bounds__(256) void k_md_acc(const u64* __restrict__ X, const u64* __restrict__ V, const u64* __restrict__ U,
                                                 const u64* __restrict__ diag, u64* __restrict__ out, const ntt_limb_t* __restrict__ LT,
                                                 limb_sel_t sel, rot_tail_arg_t G, rescale_arg_t ra, u32 n, u32 nw, u32 nrot, u32 batch) {
    constexpr int KB = TFHE_MD_KB;
    const u32 level = (u32)sel.n;
    const u32 xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    // small rows: an XCD's workgroups split into teams of n / (256 KB) workgroups, a row pair each (at N = 2^13 one team per
    // XCD had seven of its eight coefficient lanes clamped: half of the reference-size MNIST pass)
    const u32 per = blockDim.x * KB, spt = n / per >= nslot ? nslot : (n / per ? n / per : 1u), teams = nslot / spt, team = slot / spt, ts = slot % spt;
    if (team >= teams) return;
    const u32 stride = spt * blockDim.x;
    const size_t dstride = (size_t)level * n, gstride = (size_t)batch * 2;   // diagonal stride; groups per rotation
    struct term_t { u64 v0[KB], v1[KB], u0[KB], u1[KB], d[KB]; };
    for (u32 pr = team * 8u + xcd; pr < batch * level; pr += 8u * teams) {
        const u32 j = pr % level, b = pr / level;
        const ntt_limb_t L = LT[sel.idx[j]];
        const tw_t pinv = ra.qlinv[j];
        int bits = 0;
        while ((L.q >> bits) != 0) bits++;
        const u32 chunk = bits >= 62 ? 1u : (62 - bits >= 6 ? 64u : (1u << (62 - bits)));   // products summed between two reductions
        const size_t row0 = ((size_t)b * 2 * level + j) * n, row1 = row0 + (size_t)level * n;
        const u64* dj = diag + (size_t)j * n;
        for (u32 kb = ts * blockDim.x + threadIdx.x; kb < n; kb += stride * KB) {
            u32 k[KB];
#pragma unroll
            for (int i = 0; i < KB; i++) k[i] = kb + (u32)i * stride < n ? kb + (u32)i * stride : kb;   // (a clamped lane repeats kb; not stored)
            auto fetch = [&](u32 t, term_t& T) {
                const size_t g0 = (size_t)t * gstride + (size_t)b * 2;
                const u64 g = G.g[t];
                const u64x2_t* pv = (const u64x2_t*)(V + epi_pair((u32)(g0 >> 1), j, 0u, nw, n));   // both components of a position: one gather
                const u64 *pu0 = U + (g0 * level + j) * n, *pu1 = U + ((g0 + 1) * level + j) * n, *pd = dj + (size_t)(t + 1) * dstride;
#pragma unroll
                for (int i = 0; i < KB; i++) {
                    const u32 kk = galois_ntt_pos(k[i], g, n);
                    const u64x2_t vv = pv[kk];
                    T.v0[i] = vv.x; T.v1[i] = vv.y; T.u0[i] = pu0[k[i]]; T.u1[i] = pu1[k[i]]; T.d[i] = pd[k[i]];
                }
            };
            acc128 a0[KB], a1[KB];
#pragma unroll
            for (int i = 0; i < KB; i++) {
                const u64 d0 = dj[k[i]];
                a0[i] = acc128{0, 0}; a1[i] = acc128{0, 0};
                acc_mac(a0[i], X[row0 + k[i]], d0);
                acc_mac(a1[i], X[row1 + k[i]], d0);
            }
            u32 pend = 1;
            term_t cur;
            fetch(0, cur);
            for (u32 t = 0; t < nrot; t++) {
                term_t nxt;
                fetch(t + 1 < nrot ? t + 1 : t, nxt);
                if (pend == chunk) {
#pragma unroll
                    for (int i = 0; i < KB; i++) {
                        a0[i] = acc128{barrett_reduce128(a0[i].lo, a0[i].hi, L.br), 0};
                        a1[i] = acc128{barrett_reduce128(a1[i].lo, a1[i].hi, L.br), 0};
                    }
                    pend = 0;
                }
#pragma unroll
                for (int i = 0; i < KB; i++) {
                    u64 w0 = cur.u0[i], w1 = cur.u1[i];
                    if constexpr (USCALE) { w0 = shoup_full(w0, pinv, L.q); w1 = shoup_full(w1, pinv, L.q); }
                    acc_mac(a0[i], submod(cur.v0[i], w0, L.q), cur.d[i]);
                    acc_mac(a1[i], submod(cur.v1[i], w1, L.q), cur.d[i]);
                }
                pend++;
                cur = nxt;
            }
#pragma unroll
            for (int i = 0; i < KB; i++) {
                if (kb + (u32)i * stride < n) {
                    out[row0 + k[i]] = barrett_reduce128(a0[i].lo, a0[i].hi, L.br);
                    out[row1 + k[i]] = barrett_reduce128(a1[i].lo, a1[i].hi, L.br);
                }
            }
        }
    }
}

// Tail of a key switch whose sums are held as inverse-transformed sub-blocks (N = 2^(LOGB+X): k_ks_fused_sub at X = 1, the
// paired sub-block inverse at X = 2): the X inverse top stages (they need all sub-blocks of a row, ntt_inv_top_regs)
// together with what follows them -- "+ c" (k_ks_add_ct) or, with the special prime, the ModulusRaised contraction and
// "+ c" (k_ks_rescale_add) -- coefficient group by coefficient group, so the transformed sums are read once and nothing
// intermediate is written.
// T: [batch][2][nw][2^X][N >> X] (sub-block results, < 2q); out: [batch][2][level][N]; rows = batch*2*level.
template <int X>
__global__ __launch_bounds__(256) void k_ks_top_tail(const u64* __restrict__ T, const u64* __restrict__ ct, u64* __restrict__ out,
                                                      const ntt_limb_t* __restrict__ LT, ks_arg_t A, rescale_arg_t ra, u32 n,
                                                      u32 add_s) {
    constexpr int R = 1 << X;
    const u32 row = blockIdx.x, j = row % (u32)A.level, s = (row / (u32)A.level) & 1u, b = row / (2u * (u32)A.level);
    const ntt_limb_t L = LT[A.w.idx[j]];
    const u32 stride = n >> X;
    const u64* tj = T + (((size_t)b * 2 + s) * A.nw + j) * n;
    const u64* c = s < add_s ? ct + (((size_t)b * A.polys + s) * A.level + j) * n : nullptr;
    u64* o = out + (size_t)row * n;
    const u64 q = L.q;
    if (A.special) {
        const ntt_limb_t LP = LT[A.w.idx[A.level]];
        const u64* tl = T + (((size_t)b * 2 + s) * A.nw + A.level) * n;
        for (u32 k = blockIdx.y * blockDim.x + threadIdx.x; k < stride; k += gridDim.y * blockDim.x) {
            u64 v[R], p[R];
#pragma unroll
            for (int r = 0; r < R; r++) { v[r] = tj[k + (u32)r * stride]; p[r] = tl[k + (u32)r * stride]; }
            ntt_inv_top_regs<X>(v, L);
            ntt_inv_top_regs<X>(p, LP);
#pragma unroll
            for (int r = 0; r < R; r++) {
                u64 x = shoup_full(submod(v[r], barrett_reduce128(p[r], 0, L.br), q), ra.qlinv[j], q);
                if (c) x = addmod(x, c[k + (u32)r * stride], q);
                o[k + (u32)r * stride] = x;
            }
        }
    } else {
        for (u32 k = blockIdx.y * blockDim.x + threadIdx.x; k < stride; k += gridDim.y * blockDim.x) {
            u64 v[R];
#pragma unroll
            for (int r = 0; r < R; r++) v[r] = tj[k + (u32)r * stride];
            ntt_inv_top_regs<X>(v, L);
#pragma unroll
            for (int r = 0; r < R; r++) {
                u64 x = v[r];
                if (c) x = addmod(x, c[k + (u32)r * stride], q);
                o[k + (u32)r * stride] = x;
            }
        }
    }
}

// The same tail for a ROTATION whose key sums were formed from the UNrotated ciphertext against the prepared key (hoisting
// identity, tfhe_rotate_many): T holds sigma_g^-1 of the sums, so coefficient i of every limb -- and of the addend c_s, read
// straight from the unrotated input -- goes to position i g mod 2N with the sign of the wrap (pow2_cyc_rings.jl:321-329),
// BEFORE the ModulusRaised floor (it does not commute with the sign).  No rotated copy of the input is ever made: the separate
// automorphism pass over the polys x level input rows (k_galois_xcd: 9 % of a rotation at cfg#3) is gone; what it cost in
// scattered 8-byte stores moves into this kernel's final stores.  XCD-cooperative like k_galois_xcd: the workgroups of one XCD
// share one (ciphertext, component) -- its `level` output rows, one index map -- at a time, so the scattered stores land in
// windows that stay in that XCD's L2 until they are complete; the special limb's sub-block words are read once per
// coefficient, not once per limb.
template <int X>
__global__ __launch_bounds__(256) void k_ks_top_tail_rot(const u64* __restrict__ T, const u64* __restrict__ ct, u64* __restrict__ out,
                                                          const ntt_limb_t* __restrict__ LT, ks_arg_t A, rescale_arg_t ra, u32 n,
                                                          u32 add_s, u64 g, u32 npairs) {
    constexpr int R = 1 << X;
    const u32 xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const u32 stride = n >> X, level = (u32)A.level;
    // an XCD's workgroups beyond the stride / 256 that cover the coefficient groups once form further teams: team t takes the limbs
    // j = t, t + teams, ... (the special limb's words are re-read per team, from L2)
    const u32 spt = (stride + blockDim.x - 1) / blockDim.x < nslot ? (stride + blockDim.x - 1) / blockDim.x : nslot, teams = nslot / spt;
    const u32 team = slot / spt, ts = slot % spt;
    if (team >= teams) return;
    const u64 mask2n = 2ull * n - 1;
    const ntt_limb_t LP = LT[A.w.idx[A.special ? level : 0]];
    for (u32 pr = xcd; pr < npairs; pr += 8u) {   // pr = b * 2 + s
        const u32 s = pr & 1u, b = pr >> 1;
        const u64* tl = T + ((size_t)pr * A.nw + level) * n;
        for (u32 k = ts * blockDim.x + threadIdx.x; k < stride; k += spt * blockDim.x) {
            u64 p[R];
            u32 m[R];
            bool neg[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u64 t = ((u64)(k + (u32)r * stride) * g) & mask2n;
                neg[r] = t >= n;
                m[r] = (u32)t & (n - 1u);
            }
            if (A.special) {
#pragma unroll
                for (int r = 0; r < R; r++) p[r] = tl[k + (u32)r * stride];
                ntt_inv_top_regs<X>(p, LP);
#pragma unroll
                for (int r = 0; r < R; r++) p[r] = neg[r] ? negmod(p[r], LP.q) : p[r];   // the rotated polynomial's unsigned representative
            }
            // the team's limbs in blocks of JB: every word of a block is requested before the first is used
            constexpr u32 JB = X == 1 ? 4 : 2;
            const bool addc = s < add_s;
            for (u32 j0 = team; j0 < level; j0 += teams * JB) {
                u64 v[JB][R], cv[JB][R];
#pragma unroll
                for (u32 t = 0; t < JB; t++) {
                    const u32 j = j0 + t * teams < level ? j0 + t * teams : j0;
                    const u64* tj = T + ((size_t)pr * A.nw + j) * n;
                    const u64* c = ct + (((size_t)b * A.polys + (addc ? s : 0u)) * level + j) * n;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        v[t][r] = tj[k + (u32)r * stride];
                        cv[t][r] = addc ? c[k + (u32)r * stride] : 0;
                    }
                }
#pragma unroll
                for (u32 t = 0; t < JB; t++) {
                    const u32 j = j0 + t * teams;
                    if (j < level) {
                        const ntt_limb_t& L = LT[A.w.idx[j]];
                        const u64 q = L.q;
                        u64* o = out + ((size_t)pr * level + j) * n;
                        ntt_inv_top_regs<X>(v[t], L);
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            u64 x = neg[r] ? negmod(v[t][r], q) : v[t][r];
                            if (A.special) x = shoup_full(submod(x, barrett_reduce128(p[r], 0, L.br), q), ra.qlinv[j], q);
                            if (addc) x = addmod(x, neg[r] ? negmod(cv[t][r], q) : cv[t][r], q);
                            o[m[r]] = x;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BFV expand / contract: one thread per coefficient, per-thread scratch columns in LDS
// ------------------------------------------------------------------------------------------------
#define BFV_BS 128
__global__ __launch_bounds__(BFV_BS) void k_bfv_expand(const u64* __restrict__ src, u64* __restrict__ dst,
                                                        const bfv_tab_t* __restrict__ Bt, u32 n, u32 gx) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const bfv_tab_t& B = *Bt;
    const u32 k = (blockIdx.x % gx) * BFV_BS + threadIdx.x;
    const size_t p = blockIdx.x / gx;
    if (k >= n) return;
    bfv_expand_coeff(B, src + p * B.ns * n + k, n, dst + p * B.nb * n + k, n, lds + threadIdx.x, BFV_BS);
}
__global__ __launch_bounds__(BFV_BS) void k_bfv_contract(const u64* __restrict__ src, u64* __restrict__ dst,
                                                          const bfv_tab_t* __restrict__ Bt, u32 n, u32 gx) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const bfv_tab_t& B = *Bt;
    const u32 k = (blockIdx.x % gx) * BFV_BS + threadIdx.x;
    const size_t p = blockIdx.x / gx;
    if (k >= n) return;
    u64* xi = lds + threadIdx.x;
    u64* zb = xi + (size_t)B.nb * BFV_BS;
    u64* rb = zb + (size_t)B.nb * BFV_BS;
    bfv_contract_coeff(B, src + p * B.nb * n + k, n, dst + p * B.ns * n + k, n, xi, zb, rb, BFV_BS);
}

// register-resident fast path (bfv_fast.h): ℛbig = ℛ ∪ P with compile-time limb counts
// copy_shared = 0: the limbs ℛbig shares with ℛ are not written -- k_bfv_core_fused transforms them straight out of the
// input ciphertexts
// NARROW (all moduli below TFHE_FP_QMAX, known to the host) selects the body at compile time: each kernel carries one
// body and its register budget
template <int NS, int NP, bool NARROW>
__global__ __launch_bounds__(256) void k_bfv_expand_fast(const u64* __restrict__ src, u64* __restrict__ dst,
                                                          const bfv_fast_tab_t* __restrict__ Bt, u32 n, u32 gx, int copy_shared) {
    const u32 k = (blockIdx.x % gx) * 256 + threadIdx.x;
    const size_t p = blockIdx.x / gx;
    if (k >= n) return;
    if constexpr (NARROW) {
        __shared__ u64 col[(NS > NP ? NS : NP) * 256];  // scratch column of the rare exact-alpha decision (bfv_fast.h conv_alpha_fp)
        bfv_expand_narrow<NS, NP>(*Bt, src + p * NS * n + k, n, dst + p * (NS + NP) * n + k, n, col + threadIdx.x, 256, copy_shared != 0);
    }
    else bfv_expand_fast<NS, NP, false>(*Bt, src + p * NS * n + k, n, dst + p * (NS + NP) * n + k, n, copy_shared != 0);
}
// LIFT3 (narrow bodies, products: three polynomials per ciphertext): every third polynomial (c2) leaves as centred doubles
// for the fused key switch (bfv_contract_narrow, lifted_out: workgroup-uniform).
// TD (narrow bodies): the input rows are reduced doubles (k_bfv_core_fused<.., OUTD>) instead of canonical words
template <int NS, int NP, bool NARROW, bool LIFT3 = false, bool TD = false>
__global__ __launch_bounds__(256) void k_bfv_contract_fast(const u64* __restrict__ src, u64* __restrict__ dst,
                                                            const bfv_fast_tab_t* __restrict__ Bt, u32 n, u32 gx) {
    const u32 k = (blockIdx.x % gx) * 256 + threadIdx.x;
    const u32 b = blockIdx.x / gx;
    const size_t p = b;
    if (k >= n) return;
    if constexpr (NARROW) {
        __shared__ u64 col[(NS > NP ? NS : NP) * 256];  // scratch column of the rare exact-alpha decision (bfv_fast.h conv_alpha_fp)
        bfv_contract_narrow<NS, NP, TD>(*Bt, src + p * (NS + NP) * n + k, n, dst + p * NS * n + k, n, col + threadIdx.x, 256, LIFT3 && b % 3u == 2u);
    } else {
        bfv_contract_fast<NS, NP, false>(*Bt, src + p * (NS + NP) * n + k, n, dst + p * NS * n + k, n);
    }
}
