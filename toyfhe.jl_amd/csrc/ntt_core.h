// ntt_core.h -- per-thread bodies of the negacyclic NTT / inverse NTT kernels.
//
// Computes, per RNS limb, exactly the reference's
//     nntt :  â[k] = Σ_i a[i] ψ^{i(2k+1)} mod q           (src/pow2_cyc_rings.jl:278-303)
//     inntt:  a[i] = N⁻¹ ψ^{-i} Σ_k â[k] ω^{-ik} mod q     (src/pow2_cyc_rings.jl:308-318)
// natural order in and out (the NTT domain is user-visible: src/encoding.jl:35-43), dispatched
// per limb as src/crt.jl:247-267 does.  The *algorithm* is not the reference's (twist pass +
// FourierTransforms.jl CTPlan): it is the merged-twist Cooley-Tukey / Gentleman-Sande pair with
// Harvey lazy butterflies, blocked so that one 64-lane-wavefront workgroup keeps a 2^LOGB block
// in LDS and each thread does K consecutive stages on 2^K registers between LDS exchanges.
//
//   forward stage s (m = 2^s groups, t = N/2^(s+1)): (U,V) -> (U + W V, U - W V), W = Wtab[m + i],
//   Wtab[k] = ψ^{brv_logN(k)};  after all stages position j holds â[brv(j)].
//   inverse stage s: (A,B) -> (A + B, (A - B) Winv[m + i]); N⁻¹ is folded into the last stage.
//
// The functions are written against a flat `lds` array so that tests/emul/ can run the very same
// index logic on the CPU (looping over thread ids between barriers).
#pragma once
#include "modarith.h"

struct alignas(16) twd_t {  // twiddle (w, floor(w 2^64 / q)) -- one 16-byte load
    u64 w, wp;
};
TFHE_HD tw_t as_tw(const twd_t& t) { return tw_t{t.w, t.wp}; }
// Table pointers reach the kernels inside a struct loaded from memory, so the compiler only knows
// them as generic pointers and would emit flat_load (which ties up the LDS counter as well).  They
// always point to device global memory: say so.
TFHE_HD tw_t ld_tw(const twd_t* tab, u32 i) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(1))) const twd_t* gptr_t;
    const gptr_t g = (gptr_t)tab;
    return tw_t{g[i].w, g[i].wp};
#else
    return tw_t{tab[i].w, tab[i].wp};
#endif
}

// Padded LDS layout (word = 8 bytes).  Chosen with tools/lds_conflict_sim.py: for LOGB=14,
// 1024 threads, passes 4/4/4/2 every ds_read_b64/ds_write_b64 of every pass is conflict-free
// except the bit-reversed group access of the boundary pass (2-way on writes).
TFHE_HD u32 lds_phi(u32 j) { return j + 4u * (j >> 6) + (j >> 9); }
constexpr u32 lds_words(int logb) {
    return ((1u << logb) - 1) + 4u * (((1u << logb) - 1) >> 6) + (((1u << logb) - 1) >> 9) + 1;
}

// Harvey butterflies.  Forward keeps values in [0,4q); inverse keeps them in [0,2q).
TFHE_HD void bfly_fwd(u64& x, u64& y, tw_t w, u64 q) {
    u64 u = csub(x, 2 * q);
    u64 t = shoup_lazy(y, w, q);
    x = u + t;
    y = u - t + 2 * q;
}
TFHE_HD void bfly_inv(u64& x, u64& y, tw_t w, u64 q) {
    u64 a = csub(x + y, 2 * q);
    u64 d = x + 2 * q - y;
    x = a;
    y = shoup_lazy(d, w, q);
}

struct ntt_limb_t {   // per-limb constants (device copy lives in the context)
    u64 q;
    tw_t ninv;        // N^-1
    tw_t w1inv_ninv;  // Winv[1] * N^-1  (last inverse stage)
    barrett_t br;     // for products of two variable operands
    const twd_t* W;     // forward table, N entries (entry 0 unused)
    const twd_t* Winv;  // inverse table
};

// ---------------------------------------------------------------------------------------------
// One forward pass: stages S0 .. S0+K-1 of a 2^LOGB block on 2^K registers per set.
//   FIRST: operands come from global memory (block-local natural order, coalesced).
//   LAST : results go to global memory in natural NTT order; the thread->group map is bit-reversed
//          so that the 8-byte stores of a wavefront are contiguous.
//   pre = 2^x + sb, where the block is sub-block sb of a 2^(LOGB+x)-point transform (x = 0, sb = 0,
//   pre = 1 for N <= 2^LOGB); global stage index = x + local stage index.
// ---------------------------------------------------------------------------------------------
template <int LOGB, int LOGT, int S0, int K, bool FIRST, bool LAST>
TFHE_HD void ntt_fwd_pass(u64* lds, const u64* gsrc, u64* gdst, const twd_t* W, u64 q, u32 tid, u32 pre, int x,
                          u32 sb_rev) {
    constexpr int T = 1 << LOGT, E = 1 << (LOGB - LOGT), R = 1 << K, SETS = E >> K, LO = LOGB - S0 - K;
    static_assert(K >= 1 && (E >> K) >= 1, "pass wider than the per-thread register block");
    static_assert(!LAST || LO == 0, "LAST pass must end at stage LOGB-1");
#pragma unroll
    for (int u = 0; u < SETS; u++) {
        const u32 c0 = (u32)u * T + tid;
        const u32 c = LAST ? brev_bits(c0, LOGB - K) : c0;
        const u32 lo = c & ((1u << LO) - 1), hi = c >> LO;
        const u32 base = (hi << (LOGB - S0)) + lo;
        u64 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32 j = base + ((u32)r << LO);
            v[r] = FIRST ? gsrc[j] : lds[lds_phi(j)];
        }
#pragma unroll
        for (int d = 0; d < K; d++) {
            constexpr int dummy = 0;
            (void)dummy;
            const int half = 1 << (K - 1 - d);
#pragma unroll
            for (int g = 0; g < (1 << d); g++) {
                const tw_t w = ld_tw(W, (pre << (S0 + d)) + (hi << d) + (u32)g);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int r0 = (g << (K - d)) + i;
                    bfly_fwd(v[r0], v[r0 + half], w, q);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (LAST) {
                const u32 nat = (brev_bits((u32)r, K) << (LOGB - K)) + c0;  // brv_LOGB(block-local position)
                gdst[((u64)nat << x) + sb_rev] = csub(csub(v[r], 2 * q), q);
            } else {
                lds[lds_phi(base + ((u32)r << LO))] = v[r];
            }
        }
    }
}

// One inverse pass (mirror image).  FROM_GLOBAL: reads natural-order NTT values (S0+K == LOGB).
// TO_GLOBAL: writes block-local natural coefficient order (S0 == 0); SCALE folds N^-1 into the last
// stage (only when this block is the whole transform, x == 0).
template <int LOGB, int LOGT, int S0, int K, bool FROM_GLOBAL, bool TO_GLOBAL, bool SCALE>
TFHE_HD void ntt_inv_pass(u64* lds, const u64* gsrc, u64* gdst, const ntt_limb_t& L, u32 tid, u32 pre, int x,
                          u32 sb_rev) {
    constexpr int T = 1 << LOGT, E = 1 << (LOGB - LOGT), R = 1 << K, SETS = E >> K, LO = LOGB - S0 - K;
    static_assert(!FROM_GLOBAL || LO == 0, "FROM_GLOBAL pass must start at stage LOGB-1");
    static_assert(!TO_GLOBAL || S0 == 0, "TO_GLOBAL pass must end at stage 0");
    const u64 q = L.q;
#pragma unroll
    for (int u = 0; u < SETS; u++) {
        const u32 c0 = (u32)u * T + tid;
        const u32 c = FROM_GLOBAL ? brev_bits(c0, LOGB - K) : c0;
        const u32 lo = c & ((1u << LO) - 1), hi = c >> LO;
        const u32 base = (hi << (LOGB - S0)) + lo;
        u64 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (FROM_GLOBAL) {
                const u32 nat = (brev_bits((u32)r, K) << (LOGB - K)) + c0;
                v[r] = gsrc[((u64)nat << x) + sb_rev];
            } else {
                v[r] = lds[lds_phi(base + ((u32)r << LO))];
            }
        }
#pragma unroll
        for (int d = K - 1; d >= 0; d--) {
            const int half = 1 << (K - 1 - d);
#pragma unroll
            for (int g = 0; g < (1 << d); g++) {
                if (SCALE && TO_GLOBAL && d == 0) {
#pragma unroll
                    for (int i = 0; i < half; i++) {
                        const int r0 = i;
                        const u64 a = v[r0] + v[r0 + half], dd = v[r0] + 2 * q - v[r0 + half];
                        v[r0] = shoup_lazy(a, L.ninv, q);
                        v[r0 + half] = shoup_lazy(dd, L.w1inv_ninv, q);
                    }
                } else {
                    const tw_t w = ld_tw(L.Winv, (pre << (S0 + d)) + (hi << d) + (u32)g);
#pragma unroll
                    for (int i = 0; i < half; i++) {
                        const int r0 = (g << (K - d)) + i;
                        bfly_inv(v[r0], v[r0 + half], w, q);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u32 j = base + ((u32)r << LO);
            if (TO_GLOBAL)
                gdst[j] = SCALE ? csub(v[r], q) : v[r];  // unscaled blocks stay lazy in [0,2q) for the top kernel
            else
                lds[lds_phi(j)] = v[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Top stages of transforms larger than one LDS block (N = 2^(LOGB+X)): X stages directly on
// global memory, one column (2^X elements at stride N >> X) per thread.
// ---------------------------------------------------------------------------------------------
template <int X>
TFHE_HD void ntt_fwd_top(const u64* src, u64* dst, const twd_t* W, u64 q, u64 col, u64 stride) {
    constexpr int R = 1 << X;
    u64 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = src[col + (u64)r * stride];
#pragma unroll
    for (int d = 0; d < X; d++) {
        const int half = 1 << (X - 1 - d);
#pragma unroll
        for (int g = 0; g < (1 << d); g++) {
            const tw_t w = ld_tw(W, (1u << d) + (u32)g);
#pragma unroll
            for (int i = 0; i < half; i++) bfly_fwd(v[(g << (X - d)) + i], v[(g << (X - d)) + i + half], w, q);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) dst[col + (u64)r * stride] = v[r];  // lazy [0,4q): consumed by the block kernel
}

template <int X>
TFHE_HD void ntt_inv_top(const u64* src, u64* dst, const ntt_limb_t& L, u64 col, u64 stride) {
    constexpr int R = 1 << X;
    const u64 q = L.q;
    u64 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = src[col + (u64)r * stride];  // [0,2q)
#pragma unroll
    for (int d = X - 1; d >= 0; d--) {
        const int half = 1 << (X - 1 - d);
#pragma unroll
        for (int g = 0; g < (1 << d); g++) {
            if (d == 0) {
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const u64 a = v[i] + v[i + half], dd = v[i] + 2 * q - v[i + half];
                    v[i] = shoup_lazy(a, L.ninv, q);
                    v[i + half] = shoup_lazy(dd, L.w1inv_ninv, q);
                }
            } else {
                const tw_t w = ld_tw(L.Winv, (1u << d) + (u32)g);
#pragma unroll
                for (int i = 0; i < half; i++) bfly_inv(v[(g << (X - d)) + i], v[(g << (X - d)) + i + half], w, q);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) dst[col + (u64)r * stride] = csub(v[r], q);
}

// ---------------------------------------------------------------------------------------------
// Generic radix-2 kernel bodies for any N <= 2^14 (small rings of the reference's tests: N = 4..4096
// in test/*.jl and docs); one butterfly per thread per stage, plain LDS array, natural order I/O.
// `stage` loops and barriers live in the caller.
// ---------------------------------------------------------------------------------------------
TFHE_HD void ntt_generic_fwd_stage(u64* lds, const twd_t* W, u64 q, int logn, int s, u32 b) {
    const u32 tbits = (u32)(logn - 1 - s);
    const u32 i = b >> tbits, jl = b & ((1u << tbits) - 1);
    const u32 j = (i << (tbits + 1)) + jl;
    bfly_fwd(lds[j], lds[j + (1u << tbits)], ld_tw(W, (1u << s) + i), q);
}
TFHE_HD void ntt_generic_inv_stage(u64* lds, const ntt_limb_t& L, int logn, int s, u32 b) {
    const u32 tbits = (u32)(logn - 1 - s);
    const u32 i = b >> tbits, jl = b & ((1u << tbits) - 1);
    const u32 j = (i << (tbits + 1)) + jl;
    if (s == 0) {
        const u64 q = L.q;
        const u64 a = lds[j] + lds[j + (1u << tbits)], d = lds[j] + 2 * q - lds[j + (1u << tbits)];
        lds[j] = shoup_lazy(a, L.ninv, q);
        lds[j + (1u << tbits)] = shoup_lazy(d, L.w1inv_ninv, q);
    } else {
        bfly_inv(lds[j], lds[j + (1u << tbits)], ld_tw(L.Winv, (1u << s) + i), L.q);
    }
}
