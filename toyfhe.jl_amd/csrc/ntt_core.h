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
#include "fp64arith.h"
#include "modarith.h"

// compiler-level fence: keeps loads from being hoisted above it (register pressure control in the
// software-pipelined kernels); nothing at run time
#if defined(__HIP_DEVICE_COMPILE__)
#define TFHE_SCHED_FENCE() asm volatile("" ::: "memory")
#else
#define TFHE_SCHED_FENCE() ((void)0)
#endif

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

// Padded LDS layouts (word = 8 bytes), chosen with tools/lds_conflict_sim.py so that the ds_read_b64 /
// ds_write_b64 of every pass are bank-conflict free:
//   32 elements per thread (LOGB = 14: 512 threads, passes 5/5/4):  phi(j) = j + 2*(j>>6) + (j>>10)
//   16 elements per thread (LOGB <= 13, passes 4/4/..):              phi(j) = j + 4*(j>>6) + (j>>9)
template <int LOGB, int LOGT>
TFHE_HD u32 lds_phi(u32 j) {
    if (LOGB - LOGT == 5) return j + 2u * (j >> 6) + (j >> 10);
    return j + 4u * (j >> 6) + (j >> 9);
}
// phi is additive over disjoint bit fields (each term is a shift), so the address of element r of a register set
// is phi(base) + phi(r << LO) with the second term a compile-time constant: one address computation per set
// instead of one per access.
template <int LOGB, int LOGT>
constexpr u32 lds_phi_c(u32 j) {
    return (LOGB - LOGT == 5) ? (j + 2u * (j >> 6) + (j >> 10)) : (j + 4u * (j >> 6) + (j >> 9));
}
template <int LOGB, int LOGT>
constexpr u32 lds_words() {
    constexpr u32 m = (1u << LOGB) - 1;
    return (LOGB - LOGT == 5) ? (m + 2u * (m >> 6) + (m >> 10) + 1) : (m + 4u * (m >> 6) + (m >> 9) + 1);
}
// elements per thread and pass partition (K stages per pass, at most log2(E))
constexpr int logt_for(int logb) { return logb >= 13 ? logb - 5 : logb - 4; }
constexpr int pass_k_fwd(int logb, int logt, int s0) { return (logb - s0) >= (logb - logt) ? (logb - logt) : (logb - s0); }
constexpr int pass_k_inv(int logb, int logt, int s_end) { return (s_end % (logb - logt)) ? (s_end % (logb - logt)) : (logb - logt); }

// Harvey butterflies.  Forward keeps values in [0,4q); inverse keeps them in [0,2q).
TFHE_HD void bfly_fwd(u64& x, u64& y, tw_t w, u64 q) {
    u64 u = csub_s(x, 2 * q);    // x < 4q < 2^64, 2q < 2^63
    u64 t = shoup_lazy(y, w, q);
    x = u + t;
    y = u + 2 * q - t;
}
TFHE_HD void bfly_inv(u64& x, u64& y, tw_t w, u64 q) {
    u64 a = csub_s(x + y, 2 * q);  // x + y < 4q
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
    // fp64 variant (q < TFHE_FP_QMAX only; Wd == nullptr otherwise)
    double pd, pinvd;
    ftw_t ninv_d, w1inv_ninv_d;
    const ftwd_t* Wd;
    const ftwd_t* Winvd;
    // copies of the tables permuted for the boundary pass (forward last / inverse first pass), whose thread->group
    // map is bit-reversed: entry (2^s + (c0 << d) + g) of the copy = entry (2^s + (brv(c0) << d) + g) of the
    // table, so that the lanes of a wavefront read consecutive entries instead of 64 different cache lines.
    // Built for the geometry of the ring's degree: whole-transform blocks (N <= 2^14: pre = 1), and since r04 the 2^14-point
    // sub-blocks of larger transforms -- every sub-block prefix pre = 2^x + sb owns the index range [pre << s, (pre + 1) << s) of
    // stage block s and is permuted within it (ntt_tables.h permute_boundary_sub); nullptr when not built.
    const twd_t* Wb;
    const twd_t* Winvb;
    const ftwd_t* Wdb;
    const ftwd_t* Winvdb;
    // same idea for the split kernels (a 2^14 transform as two 2^13 half-problems, see k_ntt_fwd_split14): the
    // boundary pass of each half permuted within its half of every stage block
    // ---- N = 2^16, fp64 policy: the one-pass inverse with the two TOP stages FIRST (k_ntt_inv_quad2, decimation in frequency on
    // the natural-order input): a[4i'+c] = INTT'( psi^{-c(2k'+1)} sum_m A[k' + m N/4] I^{-cm} )[i'],  I = psi^{N/2},  INTT' the
    // 2^14-point inverse over psi^4 (its tables are the first quarter of this limb's).  nullptr / 0 unless built (ntt_tables.h).
    const ftwd_t* i2_t0;     // [3][2^LOGT]: psi^{-c(2t+1)} for c = 1, 2, 3 and t < 2^LOGT (the thread's first point; LOGT = 9)
    double i2_g[3];          // psi^{-c 2^(LOGT+1)}: from point k' to k' + 2^LOGT
    double i2_iinv;          // I^-1 = psi^{-N/2}
    const ftwd_t* i2_winvb;  // boundary-permuted copy (LOGB = 14 geometry) of the first 2^14 entries of Winvd
};

// (ArithFpWide, the fp64 policy for digit lifts out of source limbs above 2^52, is defined after ArithFp below.)
// Progress hook of the butterfly loops: called after every twiddle group with the number of butterflies of the
// pass done before / after it and the pass total.  The staged kernels use it to spread memory instructions
// (LDS-DMA of the next row, stores of the previous one) through the arithmetic instead of issuing them in a clump.
struct no_hook {
    TFHE_HD void operator()(int, int, int) const {}
};

// digit lift of the RNS key-switch decomposition (see ntt_io_t below)
struct lift_t {
    u64 qi, half, qj;
    barrett_t bj;
    double c32, qim;  // 2^32 mod q_j and q_i mod q_j (ArithFpWide: lifts out of source limbs above 2^52; see lift_wide_consts)
};
TFHE_HD u64 lift_digit(u64 x, const lift_t& f) {
    // |centred digit| <= q_i / 2: when that is below q_j (moduli of one size class, or a small limb lifted into a large one --
    // uniform over the item) the lift is a conditional subtraction, no reduction; otherwise (a 60-bit limb into a 40-bit one,
    // the unsigned lifts with q_i = 2^64 - 1) the 128-bit Barrett reduction
    if (f.half < f.qj) return x > f.half ? f.qj - (f.qi - x) : x;
    return x > f.half ? negmod(barrett_reduce128(f.qi - x, 0, f.bj), f.qj) : barrett_reduce128(x, 0, f.bj);
}

// ---------------------------------------------------------------------------------------------
// Arithmetic policies for the block passes: the index logic is shared, the element type and the
// butterfly differ.  ArithInt: u64 Harvey/Shoup (any q < 2^62).  ArithFp: exact integers in doubles
// (q < TFHE_FP_QMAX), see fp64arith.h.
// ---------------------------------------------------------------------------------------------
struct ArithInt {
    static constexpr bool prefetch_tw = false;
    static constexpr bool moddown = false;
    typedef u64 elem;
    typedef tw_t tw;
    struct ctx {
        u64 q;
        const twd_t *W, *Winv, *Wb, *Winvb;
        tw_t ninv, w1n;
    };
    static TFHE_HD ctx make(const ntt_limb_t& L) { return ctx{L.q, L.W, L.Winv, L.Wb, L.Winvb, L.ninv, L.w1inv_ninv}; }
    static TFHE_HD tw ld_fwd_b(const ctx& c, u32 i) { return ld_tw(c.Wb, i); }
    static TFHE_HD tw ld_inv_b(const ctx& c, u32 i) { return ld_tw(c.Winvb, i); }
    static TFHE_HD bool has_b(const ctx& c) { return c.Wb != nullptr; }
    static TFHE_HD elem from_global(u64 x, const ctx&) { return x; }
    static TFHE_HD elem from_global_plain(u64 x, const ctx&) { return x; }
    static TFHE_HD elem from_global_lift(u64 x, const ctx&, const lift_t& f, bool = false) { return lift_digit(x, f); }
    static TFHE_HD elem from_lds(u64 x) { return x; }
    static TFHE_HD u64 to_lds(elem v) { return v; }
    static TFHE_HD tw ld_fwd(const ctx& c, u32 i) { return ld_tw(c.W, i); }
    static TFHE_HD tw ld_inv(const ctx& c, u32 i) { return ld_tw(c.Winv, i); }
    static TFHE_HD void bf_fwd(elem& x, elem& y, tw w, const ctx& c) { bfly_fwd(x, y, w, c.q); }
    static TFHE_HD void bf_inv(elem& x, elem& y, tw w, const ctx& c) { bfly_inv(x, y, w, c.q); }
    static TFHE_HD void bf_inv_scaled(elem& x, elem& y, const ctx& c) {
        const u64 a = x + y, d = x + 2 * c.q - y;
        x = shoup_lazy(a, c.ninv, c.q);
        y = shoup_lazy(d, c.w1n, c.q);
    }
    static constexpr bool fwd_sweep_before(int, int) { return false; }  // Harvey butterflies keep [0, 4q) by themselves
    static constexpr double fp_a = TFHE_FP_A, fp_lim = TFHE_FP_LIMIT;   // (range plans: unused, range_inv is a no-op)
    template <int LOGB, int LOGT, int S0> static constexpr bool inv_lds_reduce() { return false; }
    template <int LOGB, int LOGT, int SEND> static constexpr double inv_entry() { return 0.502; }
    static TFHE_HD void range_fwd(elem&, const ctx&) {}
    static TFHE_HD void range_inv(elem&, const ctx&) {}
    static TFHE_HD u64 out_fwd(elem v, const ctx& c) { return csub(csub(v, 2 * c.q), c.q); }
    static TFHE_HD u64 out_inv_scaled(elem v, const ctx& c) { return csub(v, c.q); }
    static TFHE_HD u64 out_inv_lazy(elem v, const ctx&) { return v; }  // [0,2q) for the top kernel
};

// Forward range plan of the fp64 policy (fp64arith.h): a block's residues enter with |v| <= p (uncentred, or a loosely
// lifted digit), every stage maps the bound b to b (1 + 1.5 a) + 1/2 with a = TFHE_FP_A, and a sweep (reduce every element
// to |v| <= p/2) is placed before the first stage that would pass TFHE_FP_LIMIT p.  The plan runs over the block's stages
// regardless of the pass boundaries -- LDS holds lazy doubles: sweeps before stages 4 and 9 of a 2^14 block (bounds
// 1.88 3.08 4.73 7.01 | 1.19 2.13 3.43 5.22 7.68 | 1.19 ... 7.68 at the end; every consumer reduces or canonicalises).
constexpr bool fp_fwd_sweep_before(int nstages, int s) {
    double b = 1.0;
    for (int t = 0; t < nstages; t++) {
        const bool sweep = b * (1.0 + 1.5 * TFHE_FP_A) + 0.5 > TFHE_FP_LIMIT;
        if (t == s) return sweep;
        if (sweep) b = 0.5001;
        b = b * (1.0 + 1.5 * TFHE_FP_A) + 0.5;
    }
    return false;
}
struct ArithFp {
    static constexpr bool prefetch_tw = true;
    static constexpr bool moddown = false;   // ArithFpMD: the final store is the ModulusRaised contraction (inv_store)
    static constexpr bool fwd_sweep_before(int nstages, int s) { return fp_fwd_sweep_before(nstages, s); }
    // inverse range plan: every value stored to LDS is reduced (|v| <= p/2 at the start of every pass); ArithFpS relaxes this
    static constexpr double fp_a = TFHE_FP_A, fp_lim = TFHE_FP_LIMIT;
    template <int LOGB, int LOGT, int S0> static constexpr bool inv_lds_reduce() { return true; }
    template <int LOGB, int LOGT, int SEND> static constexpr double inv_entry() { return 0.502; }
    typedef double elem;
    typedef ftw_t tw;
    struct ctx {
        double p, pinv;
        const ftwd_t *W, *Winv, *Wb, *Winvb;
        ftw_t ninv, w1n;
        u64 q;
        // LDS copies of the first 2^(K1+K2) entries of W / Winv (the first and middle passes' twiddles), set by the fused
        // kernels; read through ArithFpL
        const ftwd_t *Wl = nullptr, *Winvl = nullptr;
        // ArithFpMD (k_ks_fused, SPMODE 2): P^-1 mod p and the addend row c (or a row of zeros) of the contraction
        double md_pinv = 0.0;
        const u64* md_c = nullptr;
        u32 md_g = 0;   // ArithFpMDR: the Galois element of a rotation finished in the store
    };
    static TFHE_HD ctx make(const ntt_limb_t& L) {
        return ctx{L.pd, L.pinvd, L.Wd, L.Winvd, L.Wdb, L.Winvdb, L.ninv_d, L.w1inv_ninv_d, L.q};
    }
    // centred representative: keeps |v| <= p/2 at the start of the first pass (range budget of a 5-stage pass)
    static TFHE_HD elem from_global(u64 x, const ctx& c) {
        const double d = fp_from_u64(x);
        return d + d > c.p ? d - c.p : d;
    }
    static TFHE_HD elem from_global_plain(u64 x, const ctx&) { return fp_from_u64(x); }
    // digit lift in fp64: centred residue of limb i (|d| <= q_i/2 < 2^51, exact) reduced mod p_j
    // loose: the caller's first pass accepts |v| <= p (5-stage passes with their mid-pass sweep): the centred digit,
    // |d| <= q_i / 2, needs no reduction when q_i <= 2 p (moduli of one size class)
    static TFHE_HD elem from_global_lift(u64 x, const ctx& c, const lift_t& f, bool loose = false) {
        double d = fp_from_u64(x);
        d = x > f.half ? d - (double)f.qi : d;
        return (loose && f.qi <= 2 * c.q) ? d : fp_reduce(d, c.p, c.pinv);
    }
    static TFHE_HD elem from_lds(u64 x) { double d; __builtin_memcpy(&d, &x, 8); return d; }
    static TFHE_HD u64 to_lds(elem v) { u64 b; __builtin_memcpy(&b, &v, 8); return b; }
    static TFHE_HD tw ld(const ftwd_t* tab, u32 i) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(1))) const ftwd_t* gptr_t;
        const gptr_t g = (gptr_t)tab;
        return ftw_t{g[i]};
#else
        return ftw_t{tab[i]};
#endif
    }
#ifndef TFHE_ABL_NOTW  // design aid: bit 0 -- first / middle pass twiddles all from one cache line, bit 1 -- boundary pass
#define TFHE_ABL_NOTW 0  // (wrong results; same instructions and registers, the loads hit the vector L1)
#endif
    static TFHE_HD tw ld_fwd(const ctx& c, u32 i) { return ld(c.W, (TFHE_ABL_NOTW & 1) ? (i & 7u) : i); }
    static TFHE_HD tw ld_inv(const ctx& c, u32 i) { return ld(c.Winv, (TFHE_ABL_NOTW & 1) ? (i & 7u) : i); }
    static TFHE_HD tw ld_fwd_b(const ctx& c, u32 i) { return ld(c.Wb, (TFHE_ABL_NOTW & 2) ? (i & 7u) : i); }
    static TFHE_HD tw ld_inv_b(const ctx& c, u32 i) { return ld(c.Winvb, (TFHE_ABL_NOTW & 2) ? (i & 7u) : i); }
    static TFHE_HD bool has_b(const ctx& c) { return c.Wb != nullptr; }
    static TFHE_HD void bf_fwd(elem& x, elem& y, tw w, const ctx& c) {
#ifdef TFHE_ABL_NOALU
        x += w.w; return;
#endif
        const double t = fp_mulmod_c(y, w, c.p, c.pinv);
        y = x - t;
        x = x + t;
    }
    static TFHE_HD void bf_inv(elem& x, elem& y, tw w, const ctx& c) {
#ifdef TFHE_ABL_NOALU
        x += w.w; return;
#endif
        const double a = x + y, d = x - y;
        x = a;
        y = fp_mulmod_c(d, w, c.p, c.pinv);
    }
    static TFHE_HD void bf_inv_scaled(elem& x, elem& y, const ctx& c) {
        const double a = x + y, d = x - y;
        x = fp_mulmod_c(a, c.ninv, c.p, c.pinv);
        y = fp_mulmod_c(d, c.w1n, c.p, c.pinv);
    }
    static TFHE_HD void range_fwd(elem& v, const ctx& c) { v = fp_reduce(v, c.p, c.pinv); }
    static TFHE_HD void range_inv(elem& v, const ctx& c) { v = fp_reduce(v, c.p, c.pinv); }
    // operand of a key / tensor product, output of a top stage formed at load: reduced (ArithFpS: left as they are)
    static TFHE_HD elem pre_product(elem v, const ctx& c) { return fp_reduce(v, c.p, c.pinv); }
    static TFHE_HD elem top_reduce(elem v, const ctx& c) { return fp_reduce(v, c.p, c.pinv); }
    static TFHE_HD u64 out_fwd(elem v, const ctx& c) { return fp_canon(v, c.p, c.pinv); }
    static TFHE_HD u64 out_inv_scaled(elem v, const ctx& c) { return fp_canon(v, c.p, c.pinv); }
    static TFHE_HD u64 out_inv_lazy(elem v, const ctx& c) { return fp_canon(v, c.p, c.pinv); }
};
// ArithFp with the middle pass's twiddles read from LDS (ctx::Wl / Winvl, filled once per item by the fused kernels).  In those
// kernels the middle pass's 31 twiddle words per thread were vector loads requested a few butterflies ahead of their use
// (256 registers leave no room to request them earlier): every stage began with an exposed L2 round trip in both waves of a
// SIMD, and the loads shared the in-order vmcnt with the row traffic.  The table is 8 KiB per direction (stages < K1 + K2:
// 2^10 entries) next to the 132 KiB row image; a 16-lane group reads one entry (broadcast).
// The table is padded by one word per 16 (entry i at i + (i >> 4)): the 16-lane groups of a wave read entries 2^d apart, which
// unpadded fall into the same banks for d = 4 (128 bytes apart) and d = 3.
TFHE_HD u32 tw_lds_pos(u32 i) { return i + (i >> 4); }
struct ArithFpL : ArithFp {
    static TFHE_HD tw ld_fwd(const ctx& c, u32 i) { return ftw_t{c.Wl[tw_lds_pos(i)]}; }
    static TFHE_HD tw ld_inv(const ctx& c, u32 i) { return ftw_t{c.Winvl[tw_lds_pos(i)]}; }
};
// ArithFp whose transform OUTPUT is the reduced lazy double (|r| <= p/2 + 1, as a bit pattern) instead of the canonical word:
// three instructions per element instead of nine.  For the tensor rows between k_bfv_core_fused and the narrow contraction
// (an internal buffer; the contraction's first step is an fp64 affine map anyway).
struct ArithFpD : ArithFp {
    static TFHE_HD u64 out_inv_scaled(elem v, const ctx& c) { return to_lds(fp_reduce(v, c.p, c.pinv)); }
    static TFHE_HD u64 out_inv_lazy(elem v, const ctx& c) { return to_lds(fp_reduce(v, c.p, c.pinv)); }
};
// ArithFp whose final store is the ModulusRaised contraction fused with the "+ c" of the key switch (modulusraising.jl:35-42,
// crt.jl:215-220; k_ks_rescale_add):  out = (v - [t_P]) P^-1 + c  (mod p), with t_P the special limb's coefficient (canonical,
// below 2^52: the unsigned representative, any residue of it mod p serves) arriving through the store phase's addend stream and c
// through ctx::md_c.  Exact: |reduce(v) - reduce(t)| <= p + 2, the product is <= 0.88 p, plus c < p: 1.9 p into the canonicalisation.
struct ArithFpMD : ArithFp {
    static constexpr bool moddown = true;
    static constexpr bool rot = false;
    static TFHE_HD u64 out_moddown(elem v, u64 tsp, u64 cw, const ctx& c) {
        const double x = fp_reduce(v, c.p, c.pinv) - fp_reduce(fp_from_u64(tsp), c.p, c.pinv);
        return fp_canon(fp_mulmod_c(x, ftw_t{c.md_pinv}, c.p, c.pinv) + fp_from_u64(cw), c.p, c.pinv);
    }
};
// ... of a ROTATION whose key sums came from the unrotated digits and the prepared key (hoisting identity): coefficient i goes to
// position i g mod 2N, negated on a wrap -- before the floor: with v' = -v, t' = P - t (t != 0), c' = -c
//     (v' - [t']) P^-1 + c' = -((v - t) P^-1 + c) - [t != 0]   (mod p),   since P P^-1 = 1
// (inv_store scatters the word; same bounds: one more unit into the canonicalisation)
struct ArithFpMDR : ArithFpMD {
    static constexpr bool rot = true;
    static TFHE_HD u64 out_moddown_rot(elem v, u64 tsp, u64 cw, bool neg, const ctx& c) {
        const double x = fp_reduce(v, c.p, c.pinv) - fp_reduce(fp_from_u64(tsp), c.p, c.pinv);
        const double y = fp_mulmod_c(x, ftw_t{c.md_pinv}, c.p, c.pinv) + fp_from_u64(cw);
        return fp_canon(neg ? -y - (tsp != 0 ? 1.0 : 0.0) : y, c.p, c.pinv);
    }
};
// The fp64 policy for digit lifts whose SOURCE limb may be above 2^52 (the 60-bit q0 of the reference's CKKS rings next to
// its 40-bit primes, infer.jl:98-107): such a residue does not fit a double, so that digit is centred and reduced in
// integers (lift_digit) and enters as the centred double of its canonical residue; the choice is uniform over the item (one
// source limb per row).  A separate policy type, so that the kernels of uniform fp64 rings keep their branch-free first
// pass (a run-time branch next to a load phase costs them a third of their rate).
struct ArithFpWide : ArithFp {
    // branch-free, any source width below 2^62:  x = xh 2^32 + xl, both halves exact doubles (v_cvt_f64_u32);
    //   centred digit = x - [x > q_i/2] q_i  =  xh (2^32 mod p) + xl - [x > q_i/2] (q_i mod p)   (mod p)
    // one exact modular product, two exact sums (|.| < 1.5 p + 2^32 < 2^53), one reduction to |.| <= p/2.
    static TFHE_HD elem from_global_lift(u64 x, const ctx& c, const lift_t& f, bool = false) {
        const double dl = (double)(u32)x, dh = (double)(u32)(x >> 32);
        double r = fp_mulmod_c(dh, ftw_t{f.c32}, c.p, c.pinv) + dl;
        r = x > f.half ? r - f.qim : r;
        return fp_reduce(r, c.p, c.pinv);
    }
};
// the two per-(i, j) constants of ArithFpWide's lift; a no-op for the other policies
template <class A>
TFHE_HD void lift_wide_consts(lift_t& f) {
    (void)f;
}
template <>
TFHE_HD void lift_wide_consts<ArithFpWide>(lift_t& f) {
    f.c32 = (double)barrett_reduce128(1ull << 32, 0, f.bj);
    f.qim = (double)barrett_reduce128(f.qi, 0, f.bj);
}

// ---- ArithFpS: the fp64 policy for moduli below 2^42 (fp64arith.h) -------------------------------------------------------------
// Bounds in units of p.  Forward: no sweep at all.  Inverse (Gentleman-Sande: a sum doubles the bound, a product returns to
// 1/2 + 0.0015 b): walking the passes from the first (entry <= 0.502: centred residues / reduced sums), a pass of K stages leaves
// <= 2^K b + K; the values are reduced at a pass's LDS store only when the NEXT pass would otherwise pass the limit.  For the
// 2^14-point geometry (passes of 4 / 5 / 5 stages): 0.502 -> 8.1 (stored as it is) -> 262 (reduced at the store) -> 0.502 -> 17.
constexpr double fps_inv_growth(double b, int K) {
    for (int k = 0; k < K; k++) b = 2.0 * b + 1.0;
    return b;
}
constexpr bool fps_inv_reduce(int logb, int logt, int s0) {
    double b = 0.502;
    for (int S = logb; S > 0;) {
        const int K = pass_k_inv(logb, logt, S), S0 = S - K;
        const double out = fps_inv_growth(b, K);
        const bool red = S0 > 0 && fps_inv_growth(out, pass_k_inv(logb, logt, S0)) > TFHE_FPS_LIMIT;
        if (S0 == s0) return red;
        b = red ? 0.502 : out;
        S = S0;
    }
    return true;
}
constexpr double fps_inv_entry(int logb, int logt, int send) {
    double b = 0.502;
    for (int S = logb; S > 0;) {
        if (S == send) return b;
        const int K = pass_k_inv(logb, logt, S), S0 = S - K;
        const double out = fps_inv_growth(b, K);
        const bool red = S0 > 0 && fps_inv_growth(out, pass_k_inv(logb, logt, S0)) > TFHE_FPS_LIMIT;
        b = red ? 0.502 : out;
        S = S0;
    }
    return b;
}
struct ArithFpS : ArithFp {
    static constexpr bool fwd_sweep_before(int, int) { return false; }
    static constexpr double fp_a = TFHE_FPS_A, fp_lim = TFHE_FPS_LIMIT;
    template <int LOGB, int LOGT, int S0> static constexpr bool inv_lds_reduce() { return fps_inv_reduce(LOGB, LOGT, S0); }
    template <int LOGB, int LOGT, int SEND> static constexpr double inv_entry() { return fps_inv_entry(LOGB, LOGT, SEND); }
    static TFHE_HD elem pre_product(elem v, const ctx&) { return v; }   // <= 10 p after a forward transform: terms <= 0.52 p
    static TFHE_HD elem top_reduce(elem v, const ctx&) { return v; }    // <= 3 p into the first forward pass
};

// Optional transforms fused into the block kernels' global I/O (key switching, src/rlwe_she.jl:326-344):
//   lift_t  : forward first pass reads limb i of a polynomial and lifts it, centred, into limb j --
//             digit i of the RNS decomposition (SignedMod(limb_i) re-reduced mod q_j, rlwe_she.jl:329)
//   addend  : inverse last pass adds a coefficient-domain polynomial to its result (c + INTT(S))
struct ntt_io_t {
    u32 mode;         // 0 plain (optional row groups), 1 digit-lift source (forward), 2 addend (inverse)
    u32 gsz;          // rows per group in the item numbering (0: identity mapping)
    u32 src_gstride;  // rows per group in the source buffer
    u32 dst_gstride;  // rows per group in the destination buffer
    u32 level, nw, polys;  // mode 1: item row = (b*level + i)*nw + j, source row = (b*polys + polys-1)*level + i
    u32 add_rows;     // mode 2: rows w < add_rows of each group have an addend ...
    u32 add_gstride;  //         ... at addend row g*add_gstride + w
    const u64* addend;
    u32 limb_mask;    // != 0: only the items whose limb position (index into the selection) has its bit set -- rings that mix
                      // fp64-size and larger moduli are transformed by two launches, one per arithmetic policy
    u32 lift_unsigned;  // mode 1: the source rows are residues of a modulus OUTSIDE the
                        // selection (the special prime) and are lifted as their unsigned representatives [x] mod q_j, the floor of
                        // modswitch (crt.jl:215-220), instead of centred digits; level = 1, polys = 1: source row = item / nw
};

// ---------------------------------------------------------------------------------------------
// Block passes.  A pass = stages S0 .. S0+K-1 of a 2^LOGB block, done by each thread on SETS register
// sets of R = 2^K elements.  Each pass is split into four phases so that the kernels can software-
// pipeline them (twiddles of pass p+1 and the data of the next polynomial are requested before the
// barrier that ends pass p); the CPU emulation and the simple wrappers call them back to back.
//   FIRST/FROM_GLOBAL: operands come from global memory.   LAST/TO_GLOBAL: results go to global memory.
//   The pass that touches natural-order NTT data (forward LAST, inverse FROM_GLOBAL) uses a bit-reversed
//   thread->group map so that the 8-byte accesses of a wavefront are contiguous (no bit-reversal pass).
//   pre = 2^x + sb, where the block is sub-block sb of a 2^(LOGB+x)-point transform (x = 0, sb = 0,
//   pre = 1 for N <= 2^LOGB); global stage index = x + local stage index.
// ---------------------------------------------------------------------------------------------
template <int LOGB, int LOGT, int S0, int K>
struct pgeom {
    static constexpr int T = 1 << LOGT, E = 1 << (LOGB - LOGT), R = 1 << K, SETS = E >> K, LO = LOGB - S0 - K;
    static constexpr int NTW = R - 1;  // twiddles per set: stage d uses slots (1<<d)-1 .. (2<<d)-2
    static_assert(K >= 1 && (E >> K) >= 1, "pass wider than the per-thread register block");
    // coordinates of register set u of thread tid; BREV selects the bit-reversed group map
    template <bool BREV>
    static TFHE_HD void coords(u32 tid, int u, u32& c0, u32& hi, u32& base) {
        c0 = (u32)u * T + tid;
        const u32 c = BREV ? brev_bits(c0, LOGB - K) : c0;
        const u32 lo = c & ((1u << LO) - 1);
        // one register set and LO >= LOGT: c = tid < 2^LO, so the twiddle index is workgroup-uniform (scalar loads)
        hi = (!BREV && SETS == 1 && LO >= LOGT) ? 0u : c >> LO;
        base = (hi << (LOGB - S0)) + lo;
    }
};

// ---- forward ----
// twiddles of stages d in [D0, D1) of the pass (the kernels request the early stages before the barrier
// that precedes the pass and the late ones, needed last, right after it)
template <class A, int LOGB, int LOGT, int S0, int K, bool LAST, int D0 = 0, int D1 = K>
TFHE_HD void fwd_load_tw(typename A::tw* tw, const typename A::ctx& C, u32 tid, u32 pre) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
#pragma unroll
    for (int u = 0; u < G::SETS; u++) {
        u32 c0, hi, base;
        G::template coords<LAST>(tid, u, c0, hi, base);
#pragma unroll
        for (int d = D0; d < D1; d++)
#pragma unroll
            for (int g = 0; g < (1 << d); g++)
                tw[u * G::NTW + (1 << d) - 1 + g] = (LAST && A::has_b(C))
                                                        ? A::ld_fwd_b(C, (pre << (S0 + d)) + (c0 << d) + (u32)g)
                                                        : A::ld_fwd(C, (pre << (S0 + d)) + (hi << d) + (u32)g);
    }
}
// Design aid (r06, wrong results by design): -DTFHE_ABL_NOXCHG=<bits> takes the pass-to-pass exchange out of every transform so that
// its share of a fused kernel's time can be MEASURED (profiles/LOG.md round 6): bit 0 -- __syncthreads() becomes a compiler barrier (no
// s_barrier, the waits on the LDS counters stay where the data dependences put them); bit 1 -- the LDS reads / writes between two
// passes are replaced by register pins (the address arithmetic stays: the "read" returns its own address); bit 2 -- only the writes;
// bit 3 -- only the reads.  Same butterflies, same global loads and stores.  (r06: with bit 1 or bit 3 the fused kernels leave their
// zero-scratch allocation -- 168-412 B -- and run 22 % SLOWER: the builds bound nothing; bit 0 and bit 2 keep the allocation.)
#ifndef TFHE_ABL_NOXCHG
#define TFHE_ABL_NOXCHG 0
#endif
#if (TFHE_ABL_NOXCHG & 1) && defined(__HIP_DEVICE_COMPILE__)
#define __syncthreads() asm volatile("" ::: "memory")
#endif
TFHE_HD u64 xchg_rd(const u64* lds, u32 i) {
#if (TFHE_ABL_NOXCHG & (2 | 8)) && defined(__HIP_DEVICE_COMPILE__)
    u64 r = (u64)i;   // (not the pointer: a generic address of the LDS array costs an aperture test per read)
    (void)lds;
    asm volatile("" : "+v"(r));
    return r;
#else
    return lds[i];
#endif
}
TFHE_HD void xchg_wr(u64* lds, u32 i, u64 v) {
#if (TFHE_ABL_NOXCHG & (2 | 4)) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v), "v"(i));
    (void)lds;
#else
    lds[i] = v;
#endif
}
// raw 64-bit words of the operands (global: residues; LDS: the policy's element bits)
template <int LOGB, int LOGT, int S0, int K, bool FIRST, bool LAST, int USEL = -1>
TFHE_HD void fwd_load_data(u64* raw, const u64* lds, const u64* gsrc, u32 tid, const lift_t* lift = nullptr) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
#pragma unroll
    for (int u = 0; u < G::SETS; u++) {
        if (USEL >= 0 && u != USEL) continue;
        u32 c0, hi, base;
        G::template coords<LAST>(tid, u, c0, hi, base);
        const u32 pb = FIRST ? 0u : lds_phi<LOGB, LOGT>(base);
#pragma unroll
        for (int r = 0; r < G::R; r++) {
            const u32 j = base + ((u32)r << G::LO);
            raw[u * G::R + r] = FIRST ? gsrc[j] : xchg_rd(lds, pb + lds_phi_c<LOGB, LOGT>((u32)r << G::LO));
        }
    }
    TFHE_SCHED_FENCE();  // every operand is requested before the first butterfly (no just-in-time read/wait pairs)
    (void)lift;
}
// Butterflies of the pass.  Twiddles of stages d < PF come from `twp` (requested ahead by the caller);
// the others are loaded here (the compiler schedules those loads).
// lds_early (passes that end in LDS; r05): every result is written to its LDS word right after the LAST stage's butterfly that
// produces it instead of in a store phase behind the pass (fwd_store) -- the 2^(LOGB-LOGT) ds_write of a thread then drain
// under the remaining butterflies.  A pass reads and writes the same LDS words (one geometry), so an early write cannot pass
// another thread's read of the same pass.
// Opt-in per kernel (ntt_fwd_pass<.., ES> / ntt_inv_pass<.., ES>): where the kernel has register headroom it is worth + 8 % (N = 2^16
// forward transform 2.75 -> 2.99 TB/s); in the fused kernels, which sit at the 256-VGPR cap, the changed schedule spills 13-20
// accumulator registers per digit (headline 61.2 k -> 56.4 k ciphertext-mul/s, cfg#3 40.1 k -> 30.4 k): off there.
template <class A, int LOGB, int LOGT, int S0, int K, bool FIRST, bool LAST, int PF, int USEL = -1, class HOOK = no_hook>
TFHE_HD void fwd_compute(typename A::elem* v, const u64* raw, const typename A::tw* twp, const typename A::ctx& C, u32 tid,
                         u32 pre, const lift_t* lift = nullptr, const HOOK& hook = HOOK(), u64* lds_early = nullptr) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
    const bool use_b = LAST && A::has_b(C);  // permuted boundary table (whole transforms, and since r04 the sub-blocks of larger ones)
    if (FIRST && lift) {  // digit lift: all conversions first, then the butterflies (register pressure)
#pragma unroll
        for (int i = 0; i < G::E; i++) v[i] = A::from_global_lift(raw[i], C, *lift, true);
        TFHE_SCHED_FENCE();
    }
#pragma unroll
    for (int u = 0; u < G::SETS; u++) {
        if (USEL >= 0 && u != USEL) continue;
        u32 c0, hi, base;
        G::template coords<LAST>(tid, u, c0, hi, base);
        typename A::elem* vv = v + u * G::R;
#pragma unroll
        for (int r = 0; r < G::R; r++) {
            if (FIRST && lift) continue;
            // residues enter uncentred (|v| < p); the policy's range plan places the sweeps (fp_fwd_sweep_before)
            vv[r] = FIRST ? A::from_global_plain(raw[u * G::R + r], C) : A::from_lds(raw[u * G::R + r]);
        }
#pragma unroll
        for (int d = 0; d < K; d++) {
            const int half = 1 << (K - 1 - d);
            if (A::fwd_sweep_before(LOGB, S0 + d)) {  // range control (fp64 budget; never for u64)
#pragma unroll
                for (int r = 0; r < G::R; r++) A::range_fwd(vv[r], C);
            }
#pragma unroll
            for (int g = 0; g < (1 << d); g++) {
                const typename A::tw w = d < PF ? twp[u * G::NTW + (1 << d) - 1 + g]
                                         : (LAST && use_b) ? A::ld_fwd_b(C, (pre << (S0 + d)) + (c0 << d) + (u32)g)
                                                           : A::ld_fwd(C, (pre << (S0 + d)) + (hi << d) + (u32)g);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int r0 = (g << (K - d)) + i;
                    A::bf_fwd(vv[r0], vv[r0 + half], w, C);
                    if (!LAST && lds_early && d == K - 1) {   // (half == 1: r0 and r0 + 1 are final)
                        const u32 pb = lds_phi<LOGB, LOGT>(base);
                        xchg_wr(lds_early, pb + lds_phi_c<LOGB, LOGT>((u32)r0 << G::LO), A::to_lds(vv[r0]));
                        xchg_wr(lds_early, pb + lds_phi_c<LOGB, LOGT>((u32)(r0 + 1) << G::LO), A::to_lds(vv[r0 + 1]));
                    }
                }
                hook(((u * K + d) * (G::R / 2)) + g * half, ((u * K + d) * (G::R / 2)) + (g + 1) * half, G::SETS * K * (G::R / 2));
            }
        }
    }
}
template <class A, int LOGB, int LOGT, int S0, int K, bool LAST, int USEL = -1>
TFHE_HD void fwd_store(typename A::elem* v, u64* lds, u64* gdst, const typename A::ctx& C, u32 tid, int x, u32 sb_rev) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
    static_assert(!LAST || G::LO == 0, "LAST pass must end at stage LOGB-1");
#pragma unroll
    for (int u = 0; u < G::SETS; u++) {
        if (USEL >= 0 && u != USEL) continue;
        u32 c0, hi, base;
        G::template coords<LAST>(tid, u, c0, hi, base);
        const u32 pb = LAST ? 0u : lds_phi<LOGB, LOGT>(base);
#pragma unroll
        for (int r = 0; r < G::R; r++) {
            typename A::elem e = v[u * G::R + r];
            if (LAST) {
                const u32 nat = (brev_bits((u32)r, K) << (LOGB - K)) + c0;  // brv_LOGB(block-local position)
                gdst[((u64)nat << x) + sb_rev] = A::out_fwd(e, C);
            } else {
                xchg_wr(lds, pb + lds_phi_c<LOGB, LOGT>((u32)r << G::LO), A::to_lds(e));  // lazy (the range plan spans the passes)
            }
        }
    }
}
template <class A, int LOGB, int LOGT, int S0, int K, bool FIRST, bool LAST, bool ES = false>
TFHE_HD void ntt_fwd_pass(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 tid, u32 pre, int x,
                          u32 sb_rev, const lift_t* lift = nullptr) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
    u64 raw[G::E];
    typename A::elem v[G::E];
    fwd_load_data<LOGB, LOGT, S0, K, FIRST, LAST>(raw, lds, gsrc, tid);
    if constexpr (!LAST && ES) {
        fwd_compute<A, LOGB, LOGT, S0, K, FIRST, LAST, 0>(v, raw, nullptr, C, tid, pre, FIRST ? lift : nullptr, no_hook(), lds);
    } else {
        fwd_compute<A, LOGB, LOGT, S0, K, FIRST, LAST, 0>(v, raw, nullptr, C, tid, pre, FIRST ? lift : nullptr);
        fwd_store<A, LOGB, LOGT, S0, K, LAST>(v, lds, gdst, C, tid, x, sb_rev);
    }
}

// ---- inverse (mirror image).  SCALE folds N^-1 into the last stage (whole-transform blocks, x == 0). ----
template <class A, int LOGB, int LOGT, int S0, int K, bool FROM_GLOBAL, int D0 = 0, int D1 = K>
TFHE_HD void inv_load_tw(typename A::tw* tw, const typename A::ctx& C, u32 tid, u32 pre) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
#pragma unroll
    for (int u = 0; u < G::SETS; u++) {
        u32 c0, hi, base;
        G::template coords<FROM_GLOBAL>(tid, u, c0, hi, base);
#pragma unroll
        for (int d = D0; d < D1; d++)
#pragma unroll
            for (int g = 0; g < (1 << d); g++)
                tw[u * G::NTW + (1 << d) - 1 + g] = A::ld_inv(C, (pre << (S0 + d)) + (hi << d) + (u32)g);
    }
}
template <int LOGB, int LOGT, int S0, int K, bool FROM_GLOBAL>
TFHE_HD void inv_load_data(u64* raw, const u64* lds, const u64* gsrc, u32 tid, int x, u32 sb_rev) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
    static_assert(!FROM_GLOBAL || G::LO == 0, "FROM_GLOBAL pass must start at stage LOGB-1");
#pragma unroll
    for (int u = 0; u < G::SETS; u++) {
        u32 c0, hi, base;
        G::template coords<FROM_GLOBAL>(tid, u, c0, hi, base);
#pragma unroll
        for (int r = 0; r < G::R; r++) {
            if (FROM_GLOBAL) {
                const u32 nat = (brev_bits((u32)r, K) << (LOGB - K)) + c0;
                raw[u * G::R + r] = gsrc[((u64)nat << x) + sb_rev];
            } else {
                raw[u * G::R + r] = xchg_rd(lds, lds_phi<LOGB, LOGT>(base) + lds_phi_c<LOGB, LOGT>((u32)r << G::LO));
            }
        }
    }
    TFHE_SCHED_FENCE();
}
// Range plan of an inverse pass (fp64 policy): sums double per Gentleman-Sande stage while products come back to
// <= 1/2 + 1.5 a b, so only the elements on long runs of sums ever approach the exactness limit.  Instead of sweeping
// all 2^K registers every third stage, reduce exactly the operands whose sum would exceed the limit (bounds in units of
// p for the worst admissible modulus, a = TFHE_FP_A, limit TFHE_FP_LIMIT < 2^53 / TFHE_FP_QMAX).
// mask[s] = registers reduced before processed stage s.
struct inv_plan_t {
    u32 mask[8];
};
template <int K>
constexpr inv_plan_t make_inv_plan(double a = TFHE_FP_A, double lim = TFHE_FP_LIMIT, double b0 = 0.502) {
    inv_plan_t P{};
    double b[1 << K] = {};
    for (int i = 0; i < (1 << K); i++) b[i] = b0;
    for (int st = 0; st < K; st++) {
        const int d = K - 1 - st, half = 1 << st;
        for (int g = 0; g < (1 << d); g++)
            for (int i = 0; i < half; i++) {
                const int r0 = (g << (K - d)) + i, r1 = r0 + half;
                while (b[r0] + b[r1] > lim) {
                    const int m = b[r0] >= b[r1] ? r0 : r1;
                    b[m] = 0.502;
                    P.mask[st] |= 1u << m;
                }
                const double sm = b[r0] + b[r1];
                b[r0] = sm;
                b[r1] = 0.5 + 1.5 * a * sm + 0.002;
            }
    }
    return P;
}

// Inverse butterflies (stage K-1 first).  Stages d >= K-PF come from `twp`; the others are loaded here.
// PRE: the operands are already in v as reduced elements (|v| <= p/2: the fused kernels' products) -- raw is not read
template <class A, int LOGB, int LOGT, int S0, int K, bool FROM_GLOBAL, bool SCALE, int PF, int USEL = -1, class HOOK = no_hook,
          bool PRE = false>
TFHE_HD void inv_compute(typename A::elem* v, const u64* raw, const typename A::tw* twp, const typename A::ctx& C, u32 tid,
                         u32 pre, const HOOK& hook = HOOK(), u64* lds_early = nullptr) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
    const bool use_b = FROM_GLOBAL && A::has_b(C);
#pragma unroll
    for (int u = 0; u < G::SETS; u++) {
        if (USEL >= 0 && u != USEL) continue;
        u32 c0, hi, base;
        G::template coords<FROM_GLOBAL>(tid, u, c0, hi, base);
        typename A::elem* vv = v + u * G::R;
#pragma unroll
        for (int r = 0; r < G::R; r++)
            if (!PRE) vv[r] = FROM_GLOBAL ? A::from_global(raw[u * G::R + r], C) : A::from_lds(raw[u * G::R + r]);
#pragma unroll
        for (int d = K - 1; d >= 0; d--) {
            const int half = 1 << (K - 1 - d);
            {
                constexpr inv_plan_t PLAN = make_inv_plan<K>(A::fp_a, A::fp_lim, A::template inv_entry<LOGB, LOGT, S0 + K>());
#pragma unroll
                for (int r = 0; r < G::R; r++)
                    if ((PLAN.mask[K - 1 - d] >> r) & 1u) A::range_inv(vv[r], C);
            }
#pragma unroll
            for (int g = 0; g < (1 << d); g++) {
                if (SCALE && S0 == 0 && d == 0) {
#pragma unroll
                    for (int i = 0; i < half; i++) A::bf_inv_scaled(vv[i], vv[i + half], C);
                } else {
                    const typename A::tw w = d >= K - PF ? twp[u * G::NTW + (1 << d) - 1 + g]
                                             : (FROM_GLOBAL && use_b) ? A::ld_inv_b(C, (pre << (S0 + d)) + (c0 << d) + (u32)g)
                                                                      : A::ld_inv(C, (pre << (S0 + d)) + (hi << d) + (u32)g);
#pragma unroll
                    for (int i = 0; i < half; i++) {
                        const int r0 = (g << (K - d)) + i;
                        A::bf_inv(vv[r0], vv[r0 + half], w, C);
                        if (S0 != 0 && lds_early && d == 0) {   // last processed stage: r0 and r0 + half are final (as inv_store)
                            const u32 pb = lds_phi<LOGB, LOGT>(base);
                            typename A::elem e0 = vv[r0], e1 = vv[r0 + half];
                            if (A::template inv_lds_reduce<LOGB, LOGT, S0>()) { A::range_inv(e0, C); A::range_inv(e1, C); }
                            xchg_wr(lds_early, pb + lds_phi_c<LOGB, LOGT>((u32)r0 << G::LO), A::to_lds(e0));
                            xchg_wr(lds_early, pb + lds_phi_c<LOGB, LOGT>((u32)(r0 + half) << G::LO), A::to_lds(e1));
                        }
                    }
                }
                hook(((u * K + (K - 1 - d)) * (G::R / 2)) + g * half, ((u * K + (K - 1 - d)) * (G::R / 2)) + (g + 1) * half,
                     G::SETS * K * (G::R / 2));
            }
        }
    }
}
template <class A, int LOGB, int LOGT, int S0, int K, bool FROM_GLOBAL, bool SCALE, int USEL = -1>
TFHE_HD void inv_store(typename A::elem* v, u64* lds, u64* gdst, const typename A::ctx& C, u32 tid,
                       const u64* addend = nullptr, u64* keep = nullptr) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
    constexpr bool TO_GLOBAL = (S0 == 0);
    if constexpr (TO_GLOBAL) {
        if (keep) {  // the caller stores: element e = u * R + r belongs to word base(u) + (r << LO) (k_ntt_inv_quad2: 16-byte pieces)
#pragma unroll
            for (int i = 0; i < G::E; i++) keep[i] = SCALE ? A::out_inv_scaled(v[i], C) : A::out_inv_lazy(v[i], C);
            return;
        }
        // element e = u * R + r of this thread lives at word base(u) + (r << LO)
        u32 pos[G::SETS];
#pragma unroll
        for (int u = 0; u < G::SETS; u++) {
            u32 c0, hi;
            G::template coords<FROM_GLOBAL>(tid, u, c0, hi, pos[u]);
        }
        if (addend) {
            // The addend words are requested in blocks, one block ahead of the results they are added to: the canonicalisation of
            // block q runs under the loads of block q + 1.  History: inside the store loop behind `if (addend)` each of the
            // 2^(LOGB-LOGT) loads was issued and awaited on its own (s_waitcnt vmcnt(0), which also drains the previous store) --
            // 32 serial round trips per inverse transform of the fused key switch, a fifth of that kernel's time; requested all
            // at once they left that kernel three registers short and three of them were spilled right behind their loads,
            // each with its own wait.  Four blocks keep two of them (E/2 words) in flight next to the results.
            // A::moddown (ArithFpMD): two streams -- `addend` is the special limb's row, ctx::md_c the ciphertext's -- in eight
            // blocks (the same words in flight), and the store is the contraction instead of the sum.
            constexpr bool MD = A::moddown;
            constexpr int NB = G::E >= 8 ? (MD && G::E >= 16 ? 8 : 4) : 1, BS = G::E / NB;
            u64 add[2][BS], add2[2][MD ? BS : 1];
            auto request = [&](int q) {
#pragma unroll
                for (int k = 0; k < BS; k++) {
                    const int e = q * BS + k, u = e / G::R, r = e % G::R;
                    if (USEL >= 0 && u != USEL) continue;
                    add[q & 1][k] = addend[pos[u] + ((u32)r << G::LO)];
                    if constexpr (MD) add2[q & 1][k] = C.md_c[pos[u] + ((u32)r << G::LO)];
                }
                TFHE_SCHED_FENCE();
            };
            request(0);
#pragma unroll
            for (int q = 0; q < NB; q++) {
                if (q + 1 < NB) request(q + 1);
                u64 o[BS];
#pragma unroll
                for (int k = 0; k < BS; k++) {
                    if constexpr (MD) {
                        if constexpr (A::rot) {   // (whole-transform blocks: N = 2^LOGB)
                            const int e = q * BS + k, u = e / G::R, r = e % G::R;
                            const u32 t = ((pos[u] + ((u32)r << G::LO)) * C.md_g) & ((2u << LOGB) - 1u);
                            o[k] = A::out_moddown_rot(v[q * BS + k], add[q & 1][k], add2[q & 1][k], (t >> LOGB) != 0, C);
                        } else {
                            o[k] = A::out_moddown(v[q * BS + k], add[q & 1][k], add2[q & 1][k], C);
                        }
                    } else o[k] = SCALE ? A::out_inv_scaled(v[q * BS + k], C) : A::out_inv_lazy(v[q * BS + k], C);
                }
                TFHE_SCHED_FENCE();
#pragma unroll
                for (int k = 0; k < BS; k++) {
                    const int e = q * BS + k, u = e / G::R, r = e % G::R;
                    if (USEL >= 0 && u != USEL) continue;
                    if constexpr (MD) {
                        if constexpr (A::rot) gdst[((pos[u] + ((u32)r << G::LO)) * C.md_g) & ((1u << LOGB) - 1u)] = o[k];
                        else gdst[pos[u] + ((u32)r << G::LO)] = o[k];
                    } else gdst[pos[u] + ((u32)r << G::LO)] = addmod(o[k], add[q & 1][k], C.q);
                }
                TFHE_SCHED_FENCE();
            }
        } else {
            // finish every result before the store phase starts (otherwise the scheduler interleaves the two
            // and the register allocator spills)
            u64 o[G::E];
#pragma unroll
            for (int i = 0; i < G::E; i++) o[i] = SCALE ? A::out_inv_scaled(v[i], C) : A::out_inv_lazy(v[i], C);
            TFHE_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < G::E; e++) {
                const int u = e / G::R, r = e % G::R;
                if (USEL >= 0 && u != USEL) continue;
                gdst[pos[u] + ((u32)r << G::LO)] = o[e];
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < G::SETS; u++) {
            if (USEL >= 0 && u != USEL) continue;
            u32 c0, hi, base;
            G::template coords<FROM_GLOBAL>(tid, u, c0, hi, base);
#pragma unroll
            for (int r = 0; r < G::R; r++) {
                typename A::elem e = v[u * G::R + r];
                if (A::template inv_lds_reduce<LOGB, LOGT, S0>()) A::range_inv(e, C);
                xchg_wr(lds, lds_phi<LOGB, LOGT>(base) + lds_phi_c<LOGB, LOGT>((u32)r << G::LO), A::to_lds(e));
            }
        }
    }
}
template <class A, int LOGB, int LOGT, int S0, int K, bool FROM_GLOBAL, bool TO_GLOBAL, bool SCALE, bool ES = false>
TFHE_HD void ntt_inv_pass(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 tid, u32 pre, int x,
                          u32 sb_rev, const u64* addend = nullptr) {
    typedef pgeom<LOGB, LOGT, S0, K> G;
    static_assert(TO_GLOBAL == (S0 == 0), "TO_GLOBAL pass is the one ending at stage 0");
    u64 raw[G::E];
    typename A::elem v[G::E];
    inv_load_data<LOGB, LOGT, S0, K, FROM_GLOBAL>(raw, lds, gsrc, tid, x, sb_rev);
    if constexpr (ES && !TO_GLOBAL) {
        inv_compute<A, LOGB, LOGT, S0, K, FROM_GLOBAL, SCALE, 0>(v, raw, nullptr, C, tid, pre, no_hook(), lds);
    } else {
        inv_compute<A, LOGB, LOGT, S0, K, FROM_GLOBAL, SCALE, 0>(v, raw, nullptr, C, tid, pre);
        inv_store<A, LOGB, LOGT, S0, K, FROM_GLOBAL, SCALE>(v, lds, gdst, C, tid, TO_GLOBAL ? addend : nullptr);
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
    for (int r = 0; r < R; r++) dst[col + (u64)r * stride] = csub(csub(v[r], 2 * q), q);  // canonical: either block policy can take it
}

// the X top stages of an inverse transform on 2^X values held in registers ([0, 2q) in, canonical out, N^-1 folded in)
template <int X>
TFHE_HD void ntt_inv_top_regs(u64 (&v)[1 << X], const ntt_limb_t& L) {
    constexpr int R = 1 << X;
    const u64 q = L.q;
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
    for (int r = 0; r < R; r++) v[r] = csub(v[r], q);
}
template <int X>
TFHE_HD void ntt_inv_top(const u64* src, u64* dst, const ntt_limb_t& L, u64 col, u64 stride) {
    constexpr int R = 1 << X;
    u64 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = src[col + (u64)r * stride];  // [0,2q)
    ntt_inv_top_regs<X>(v, L);
#pragma unroll
    for (int r = 0; r < R; r++) dst[col + (u64)r * stride] = v[r];
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
