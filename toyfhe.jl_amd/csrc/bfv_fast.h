// bfv_fast.h -- register-resident, constant-folded BFV expand / contract for ℛbig = ℛ ∪ P with compile-time
// limb counts (NS limbs of ℛ, NP limbs of P).  Same exact results as bfv_core.h (which stays as the
// general path and as the cross-check); what changes is the schedule:
//   * every per-limb scalar factor (t, (q)^-1, (A/a_j)^-1, the centring offsets) is folded into the
//     conversion matrices on the host, so a coefficient costs NS + NP Shoup products, 2*NS*NP 128-bit
//     MACs and NS + NP Barrett reductions instead of ~3x that many modular products;
//   * ξ lives in registers (loops are unrolled over NS / NP), constants are workgroup-uniform loads.
// Reference semantics: src/bfv.jl:34-40,172-226 via src/crt.jl:91-112 (see bfv_core.h for the derivation).
#pragma once
#include "conv_core.h"
#include "fp64arith.h"

template <int K>
TFHE_HD u32 conv_alpha_fast(const u64 (&xi)[K], const u64* rho, const u32* sh, u64& frac_out) {
    u64 frac = 0;
    u32 carries = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
        const u64 xb = xi[j] << sh[j];
        const u64 f = xb + mulhi64(xb, rho[j]);
        const u64 s = frac + f;
        carries += (s < f);
        frac = s;
    }
    frac_out = frac;
    return carries;
}
// exact decision of alpha (rare): is X = Σ ξ_j (A/a_j) >= (carries+1) A ?   (same algorithm as conv_prepare)
template <int K>
TFHE_HD u32 conv_alpha_exact(const u64 (&xi)[K], const u64* M, const u64* Aw, int nwords, u32 carries) {
    const u64 mult = (u64)carries + 1;
    u64 acc_lo = 0, acc_hi = 0, acc_ex = 0, mcarry = 0, borrow = 0;
    for (int w = 0; w <= nwords; w++) {
        if (w < nwords) {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const u64 mw = M[(size_t)j * nwords + w];
                const u64 lo = xi[j] * mw, hi = mulhi64(xi[j], mw);
                u64 s = acc_lo + lo;
                const u64 c = (s < lo);
                acc_lo = s;
                s = acc_hi + hi;
                u64 c2 = (s < hi);
                s += c;
                c2 += (s < c);
                acc_hi = s;
                acc_ex += c2;
            }
        }
        const u64 xw = acc_lo;
        acc_lo = acc_hi; acc_hi = acc_ex; acc_ex = 0;
        const u64 aw = w < nwords ? Aw[w] : 0;
        const u64 plo = aw * mult, phi = mulhi64(aw, mult);
        const u64 yw = plo + mcarry;
        mcarry = phi + (yw < plo);
        const u64 d = xw - yw;
        borrow = (u64)(xw < yw) | (u64)(d < borrow);
    }
    return carries + (borrow ? 0u : 1u);
}

// The same decision with the K words read from a scratch column (col[j * stride]) by ROLLED loops: a handful of live
// registers, whatever K is.  The narrow bodies take this form -- inlined and unrolled, the rare path tripled the register
// allocation of the whole kernel (66 -> 124 VGPRs in the contraction), which is the allocation every wave pays for.
TFHE_HD u32 conv_alpha_exact_col(int K, const u64* col, int stride, const u64* M, const u64* Aw, int nwords, u32 carries) {
    const u64 mult = (u64)carries + 1;
    u64 acc_lo = 0, acc_hi = 0, acc_ex = 0, mcarry = 0, borrow = 0;
#pragma unroll 1
    for (int w = 0; w <= nwords; w++) {
        if (w < nwords) {
#pragma unroll 1
            for (int j = 0; j < K; j++) {
                const u64 x = col[(size_t)j * stride], mw = M[(size_t)j * nwords + w];
                u64 lo, hi;
                mul64_full(x, mw, lo, hi);
                u64 s = acc_lo + lo;
                const u64 c = (s < lo);
                acc_lo = s;
                s = acc_hi + hi;
                u64 c2 = (s < hi);
                s += c;
                c2 += (s < c);
                acc_hi = s;
                acc_ex += c2;
            }
        }
        const u64 xw = acc_lo;
        acc_lo = acc_hi; acc_hi = acc_ex; acc_ex = 0;
        const u64 aw = w < nwords ? Aw[w] : 0;
        u64 plo, phi;
        mul64_full(aw, mult, plo, phi);
        const u64 yw = plo + mcarry;
        mcarry = phi + (yw < plo);
        const u64 d = xw - yw;
        borrow = (u64)(xw < yw) | (u64)(d < borrow);
    }
    return carries + (borrow ? 0u : 1u);
}

#define TFHE_FAST_MAX 12  // max NS / NP of the fast path tables

struct bfv_fast_tab_t {
    int ns, np, nb;
    // ---- expand: x (basis q, centred) -> the NP limbs of P; the NS shared limbs are copied ----
    int pos_s[TFHE_FAST_MAX], pos_p[TFHE_FAST_MAX];
    u64 q[TFHE_FAST_MAX];
    tw_t e_inv[TFHE_FAST_MAX];          // (q/q_i)^-1 mod q_i
    u64 e_half[TFHE_FAST_MAX];          // floor(q/2) mod q_i
    u64 rho_q[TFHE_FAST_MAX];
    u32 sh_q[TFHE_FAST_MAX];
    barrett_t pb[TFHE_FAST_MAX];        // P moduli
    u64 e_C[TFHE_FAST_MAX][TFHE_FAST_MAX];  // (q/q_i) mod p_j
    u64 e_A[TFHE_FAST_MAX];             // q mod p_j
    u64 e_halfT[TFHE_FAST_MAX];         // floor(q/2) mod p_j
    // ---- contract ----
    tw_t c_a1[TFHE_FAST_MAX];           // t (q/q_i)^-1 mod q_i
    u64 c_b1[TFHE_FAST_MAX];            // h (q/q_i)^-1 mod q_i,  h = (q-1)/2
    tw_t c_a2[TFHE_FAST_MAX];           // t q^-1 (P/p_j)^-1 mod p_j
    u64 c_b2[TFHE_FAST_MAX];            // (h q^-1 + floor(P/2)) (P/p_j)^-1 mod p_j
    u64 c_C1[TFHE_FAST_MAX][TFHE_FAST_MAX];  // (q/q_i) q^-1 (P/p_j)^-1 mod p_j
    u64 c_A1[TFHE_FAST_MAX];            // q q^-1 (P/p_j)^-1 = (P/p_j)^-1 mod p_j   (times alpha)
    u64 rho_p[TFHE_FAST_MAX];
    u32 sh_p[TFHE_FAST_MAX];
    barrett_t qb[TFHE_FAST_MAX];        // ℛ moduli
    u64 c_C2[TFHE_FAST_MAX][TFHE_FAST_MAX];  // (P/p_j) mod q_i
    u64 c_A2[TFHE_FAST_MAX];            // P mod q_i
    u64 c_halfT[TFHE_FAST_MAX];         // floor(P/2) mod q_i
    int lazy_q, lazy_p;                 // products that may be accumulated before a reduction (sources q / P)
    int narrow;                         // every modulus of ℛ and P is below 2^52: carry-free split accumulation
    // ---- narrow path (bfv_expand_narrow / bfv_contract_narrow): constants pre-split into 26-bit halves, packed
    //      lo26 | hi26 << 32, with every subtraction folded in as a negated constant so that a whole output is ONE
    //      carry-free product-sum followed by ONE Barrett reduction ----
    u64 n_eC[TFHE_FAST_MAX][TFHE_FAST_MAX];     // [i][j]  (q/q_i) mod p_j
    u64 n_eNegA[TFHE_FAST_MAX];                 // -(q mod p_j)                 (times alpha)
    u64 n_eNegHalf[TFHE_FAST_MAX];              // -(floor(q/2) mod p_j)        (plain value, added once)
    u64 n_cA2[TFHE_FAST_MAX];                   // t q^-1 (P/p_j)^-1 mod p_j    (times the P-limb input)
    u64 n_cNegC1[TFHE_FAST_MAX][TFHE_FAST_MAX]; // [i][j]  -(q/q_i) q^-1 (P/p_j)^-1 mod p_j
    u64 n_cA1[TFHE_FAST_MAX];                   // (P/p_j)^-1 mod p_j           (times alpha_1)
    u64 n_cC2[TFHE_FAST_MAX][TFHE_FAST_MAX];    // [j][i]  (P/p_j) mod q_i
    u64 n_cNegA2[TFHE_FAST_MAX];                // -(P mod q_i)                 (times alpha_2)
    u64 n_cNegHalf[TFHE_FAST_MAX];              // -(floor(P/2) mod q_i)        (plain)
    // exact-integer fp64 parts of the narrow path (every modulus below TFHE_FP_QMAX, fp64arith.h)
    mont26_t mq[TFHE_FAST_MAX], mp[TFHE_FAST_MAX];      // radix-2^26 Montgomery constants of q_i / p_j (acc52_redc); the n_* constants
                                                        // above carry the factor 2^78 of that reduction
    u64 n_cB2[TFHE_FAST_MAX];                           // c_b2 2^78 mod p_j            (plain, added once)
    // output-major rows of the three matrices (the constants of ONE output contiguous: one or two wide scalar loads per output,
    // six accumulator registers live at a time)
    u64 t_eC[TFHE_FAST_MAX][TFHE_FAST_MAX];             // [j][i] = n_eC[i][j]
    u64 t_cNegC1[TFHE_FAST_MAX][TFHE_FAST_MAX];         // [j][i] = n_cNegC1[i][j]
    u64 t_cC2[TFHE_FAST_MAX][TFHE_FAST_MAX];            // [i][j] = n_cC2[j][i]
    double f_q[TFHE_FAST_MAX], f_qinv[TFHE_FAST_MAX];   // q_i, 1/q_i
    double f_p[TFHE_FAST_MAX], f_pinv[TFHE_FAST_MAX];   // p_j, 1/p_j
    double f_ea[TFHE_FAST_MAX], f_eb[TFHE_FAST_MAX];    // expand:   xi_i = x_i ea_i + eb_i mod q_i  (ea = (q/q_i)^-1, eb = floor(q/2) ea)
    double f_ca[TFHE_FAST_MAX], f_cb[TFHE_FAST_MAX];    // contract: xi_i = y_i ca_i + cb_i mod q_i  (c_a1, c_b1)
    // exact-alpha tables (word arrays in global memory)
    const u64 *Mq, *Aq, *Mp, *Ap;
    int nwq, nwp;
};

template <int K, bool NARROW>
TFHE_HD u64 mac_reduce(const u64 (&xi)[K], const u64* col, int stride, const barrett_t& bt, int lazy) {
    if constexpr (NARROW) {  // every modulus below 2^52: carry-free 26-bit-split accumulation (modarith.h acc52)
        static_assert(K <= 16, "acc52 holds at most 16 terms");
        acc52 a{0, 0, 0};
#pragma unroll
        for (int j = 0; j < K; j++) {
            const u64 c = col[(size_t)j * stride];
            acc52_mac(a, (u32)xi[j] & 0x3ffffffu, (u32)(xi[j] >> 26), (u32)c & 0x3ffffffu, (u32)(c >> 26));
        }
        u64 lo, hi;
        acc52_fold(a, lo, hi);
        return barrett_reduce128(lo, hi, bt);
    } else {
        acc128 acc{0, 0};
        u64 sum = 0;
        int pending = 0;
#pragma unroll
        for (int j = 0; j < K; j++) {
            acc_mac(acc, xi[j], col[(size_t)j * stride]);
            if (++pending == lazy) {
                sum = addmod(sum, barrett_reduce128(acc.lo, acc.hi, bt), bt.q);
                acc = acc128{0, 0};
                pending = 0;
            }
        }
        if (pending) sum = addmod(sum, barrett_reduce128(acc.lo, acc.hi, bt), bt.q);
        return sum;
    }
}

// src: ℛ limb i of this coefficient at src[i*ls]; dst: ℛbig limb l at dst[l*ld]
template <int NS, int NP, bool NARROW = false>
TFHE_HD void bfv_expand_fast(const bfv_fast_tab_t& B, const u64* src, size_t ls, u64* dst, size_t ld, bool copy_shared = true) {
    u64 x[NS], xi[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        x[i] = src[(size_t)i * ls];
        xi[i] = shoup_full(addmod(x[i], B.e_half[i], B.q[i]), B.e_inv[i], B.q[i]);
    }
    u64 frac;
    u32 alpha = conv_alpha_fast<NS>(xi, B.rho_q, B.sh_q, frac);
    if (frac + 2ull * NS < frac) alpha = conv_alpha_exact<NS>(xi, B.Mq, B.Aq, B.nwq, alpha);
    if (copy_shared) {
#pragma unroll
        for (int i = 0; i < NS; i++) dst[(size_t)B.pos_s[i] * ld] = x[i];
    }
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const barrett_t& bt = B.pb[j];
        u64 r = mac_reduce<NS, NARROW>(xi, &B.e_C[0][j], TFHE_FAST_MAX, bt, B.lazy_q);
        r = submod(r, mulmod((u64)alpha, B.e_A[j], bt), bt.q);
        dst[(size_t)B.pos_p[j] * ld] = submod(r, B.e_halfT[j], bt.q);
    }
}

template <int NS, int NP, bool NARROW = false>
TFHE_HD void bfv_contract_fast(const bfv_fast_tab_t& B, const u64* src, size_t ls, u64* dst, size_t ld) {
    u64 xi[NS];
#pragma unroll
    for (int i = 0; i < NS; i++)  // ξ_i of r = (t y + h) mod q
        xi[i] = addmod(shoup_full(src[(size_t)B.pos_s[i] * ls], B.c_a1[i], B.q[i]), B.c_b1[i], B.q[i]);
    u64 frac;
    u32 a1 = conv_alpha_fast<NS>(xi, B.rho_q, B.sh_q, frac);
    if (frac + 2ull * NS < frac) a1 = conv_alpha_exact<NS>(xi, B.Mq, B.Aq, B.nwq, a1);
    u64 xp[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {  // ξ'_j of w + floor(P/2) in basis P
        const barrett_t& bt = B.pb[j];
        u64 v = addmod(shoup_full(src[(size_t)B.pos_p[j] * ls], B.c_a2[j], bt.q), B.c_b2[j], bt.q);
        v = submod(v, mac_reduce<NS, NARROW>(xi, &B.c_C1[0][j], TFHE_FAST_MAX, bt, B.lazy_q), bt.q);
        xp[j] = addmod(v, mulmod((u64)a1, B.c_A1[j], bt), bt.q);
    }
    u32 a2 = conv_alpha_fast<NP>(xp, B.rho_p, B.sh_p, frac);
    if (frac + 2ull * NP < frac) a2 = conv_alpha_exact<NP>(xp, B.Mp, B.Ap, B.nwp, a2);
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const barrett_t& bt = B.qb[i];
        u64 r = mac_reduce<NP, NARROW>(xp, &B.c_C2[0][i], TFHE_FAST_MAX, bt, B.lazy_p);
        r = submod(r, mulmod((u64)a2, B.c_A2[i], bt), bt.q);
        dst[(size_t)i * ld] = submod(r, B.c_halfT[i], bt.q);
    }
}

// ---- narrow path: every modulus below TFHE_FP_QMAX = 2^50 + 2^40 ----
TFHE_HD u64 pack26(u64 c) { return (c & 0x3ffffffull) | ((c >> 26) << 32); }
TFHE_HD void acc52_macp(acc52& a, u64 x, u64 cpacked) {
    acc52_mac(a, (u32)x & 0x3ffffffu, (u32)(x >> 26), (u32)cpacked, (u32)(cpacked >> 32));
}
TFHE_HD u64 acc52_reduce(const acc52& a, const barrett_t& bt) {
    u64 lo, hi;
    acc52_fold(a, lo, hi);
    return barrett_reduce128(lo, hi, bt);
}
// xi = (x a + b) mod p as an exact integer in a double, in [0, p)   (6-op fp64 modular product, fp64arith.h)
TFHE_HD double fp_affine(u64 x, double a, double b, double p, double pinv) {
    double r = fp_reduce(fp_mulmod_c(fp_from_u64(x), ftw_t{a}, p, pinv) + b, p, pinv);
    return r < 0.0 ? r + p : r;
}
// alpha = floor(Σ xi_j / a_j) from a double sum: |S' - S| < K^2 2^-52 <= 2^-44 for K <= 16, so the floor is certain
// unless the fraction is within 2^-40 of an integer -- then the exact multi-word comparison decides between the two
// candidates.  (All three conversions are offset by floor(A/2), so that only happens for values near +-A/2.)
// a 26-bit-split operand of the carry-free product sums, split ONCE where it is produced (left to itself the compiler re-derives
// the halves of a 64-bit word per use and drags the word's upper bits through every product: 324 v_mov_b32 and 148 spare
// v_mad_u64_u32 in the contraction's second conversion)
struct split26 {
    u32 lo, hi;
};
TFHE_HD split26 split_of(u64 x) {
    split26 r{(u32)x & 0x3ffffffu, (u32)(x >> 26)};
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(r.lo), "+v"(r.hi));  // opaque (NOT volatile: a volatile asm orders every load around it and serialises the kernel): the halves are what the products see, not the 64-bit word
#endif
    return r;
}
TFHE_HD u64 join_of(split26 x) { return ((u64)x.hi << 26) | x.lo; }
TFHE_HD void acc52_macs(acc52& a, split26 x, u64 cpacked) { acc52_mac(a, x.lo, x.hi, (u32)cpacked, (u32)(cpacked >> 32)); }
// small multiplier (the alpha counts, < 2^26): two partial products
TFHE_HD void acc52_mac_small(acc52& a, u32 x, u64 cpacked) {
    a.s0 += (u64)x * (u32)cpacked;
    a.s1 += (u64)x * (u32)(cpacked >> 32);
}

// alpha = floor(Σ xi_j / a_j) from a double sum: |S' - S| < K^2 2^-52 <= 2^-44 for K <= 16, so the floor is certain
// unless the fraction is within 2^-40 of an integer -- then the exact multi-word comparison decides between the two
// candidates.  (All three conversions are offset by floor(A/2), so that only happens for values near +-A/2.)
template <int K>
TFHE_HD u32 conv_alpha_fp(const double (&xd)[K], const split26 (&xs)[K], const double* ainv, const u64* M, const u64* Aw, int nwords,
                          u64* col, int cstride) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < K; j++) s = fp_fma(xd[j], ainv[j], s);
    const double fl = __builtin_floor(s), f = s - fl;
    u32 n = (u32)fl;
    if (!(f > 0x1p-40 && f < 1.0 - 0x1p-40)) {  // rare: the words go through the scratch column to the rolled exact decision
#pragma unroll
        for (int j = 0; j < K; j++) col[(size_t)j * cstride] = join_of(xs[j]);
        if (f <= 0x1p-40) n = n == 0 ? 0u : conv_alpha_exact_col(K, col, cstride, M, Aw, nwords, n - 1);
        else n = conv_alpha_exact_col(K, col, cstride, M, Aw, nwords, n);
    }
    return n;
}

template <int NS, int NP>
TFHE_HD void bfv_expand_narrow(const bfv_fast_tab_t& B, const u64* src, size_t ls, u64* dst, size_t ld, u64* col, int cstride,
                               bool copy_shared = true) {
    static_assert(NS + 2 <= 16 && NP + 2 <= 16, "acc52 term budget");
    u64 x[NS];
    split26 xs[NS];
    double xd[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        x[i] = src[(size_t)i * ls];
        xd[i] = fp_affine(x[i], B.f_ea[i], B.f_eb[i], B.f_q[i], B.f_qinv[i]);
        xs[i] = split_of(fp_to_u64(xd[i]));
    }
    const u32 alpha = conv_alpha_fp<NS>(xd, xs, B.f_qinv, B.Mq, B.Aq, B.nwq, col, cstride);
    if (copy_shared) {
#pragma unroll
        for (int i = 0; i < NS; i++) dst[(size_t)B.pos_s[i] * ld] = x[i];
    }
#pragma unroll
    for (int j = 0; j < NP; j++) {  // output-major: the constants of output j are one contiguous row
        acc52 a{B.n_eNegHalf[j], 0, 0};
#pragma unroll
        for (int i = 0; i < NS; i++) acc52_macs(a, xs[i], B.t_eC[j][i]);
        acc52_mac_small(a, alpha, B.n_eNegA[j]);
        dst[(size_t)B.pos_p[j] * ld] = acc52_redc(a, B.mp[j]);
    }
}

// centred residue r - q (r > q/2) or r of a canonical word, as the bit pattern of the exact double (|value| < 2^51):
// integer centring, then the 2^52 trick on value + 2^51
TFHE_HD u64 centred_double_bits(u64 r, u64 q) {
    const u64 s = r > (q >> 1) ? r - q : r;                       // two's complement of the centred value
    const u64 bits = (s + (1ull << 51)) | 0x4330000000000000ull;  // 2^52 + (value + 2^51), value + 2^51 in [0, 2^52)
    double d;
    __builtin_memcpy(&d, &bits, 8);
    d -= 6755399441055744.0;                                      // 2^52 + 2^51
    u64 out;
    __builtin_memcpy(&out, &d, 8);
    return out;
}
// lifted_out: the residues are written as CENTRED DOUBLES (bit patterns) instead of canonical words -- the form the fused key
// switch lifts its digit rows into (rlwe_she.jl:326-329); used for the c2 polynomial of a multiplication that is
// relinearised next (internal buffer of tfhe_bfv_mul_relin only)
// xi = (y a + b) mod p for a lazy double y, |y| <= p/2 + 1 (the reduced output of a transform)
TFHE_HD double fp_affine_d(double y, double a, double b, double p, double pinv) {
    double r = fp_reduce(fp_mulmod_c(y, ftw_t{a}, p, pinv) + b, p, pinv);
    return r < 0.0 ? r + p : r;
}
// TD: the inputs are reduced doubles y (bit patterns, |y| <= p/2 + 1) instead of canonical words -- the form
// k_bfv_core_fused<.., OUTD> leaves its result rows in.  ℛ-limbs go straight into the fp64 affine map; a P-limb enters its
// product sum as the non-negative integer y + p (< 1.5 p + 1: within the 26-bit-split budget, and congruent).
template <int NS, int NP, bool TD = false>
TFHE_HD void bfv_contract_narrow(const bfv_fast_tab_t& B, const u64* src, size_t ls, u64* dst, size_t ld, u64* col, int cstride,
                                 bool lifted_out = false) {
    split26 xs[NS];
    double xd[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {  // ξ_i of r = (t y + h) mod q
        const u64 w = src[(size_t)B.pos_s[i] * ls];
        if constexpr (TD) {
            double y;
            __builtin_memcpy(&y, &w, 8);
            xd[i] = fp_affine_d(y, B.f_ca[i], B.f_cb[i], B.f_q[i], B.f_qinv[i]);
        } else {
            xd[i] = fp_affine(w, B.f_ca[i], B.f_cb[i], B.f_q[i], B.f_qinv[i]);
        }
        xs[i] = split_of(fp_to_u64(xd[i]));
    }
    const u32 a1 = conv_alpha_fp<NS>(xd, xs, B.f_qinv, B.Mq, B.Aq, B.nwq, col, cstride);
    split26 xps[NP];
    double xpd[NP];
    // ξ'_j of w + floor(P/2) in basis P: one product-sum and one reduction per limb, output-major
#pragma unroll
    for (int j = 0; j < NP; j++) {
        acc52 a{B.n_cB2[j], 0, 0};
        u64 yp = src[(size_t)B.pos_p[j] * ls];
        if constexpr (TD) {
            double y;
            __builtin_memcpy(&y, &yp, 8);
            yp = fp_to_u64(y + B.f_p[j]);
        }
        acc52_macs(a, split_of(yp), B.n_cA2[j]);
#pragma unroll
        for (int i = 0; i < NS; i++) acc52_macs(a, xs[i], B.t_cNegC1[j][i]);
        acc52_mac_small(a, a1, B.n_cA1[j]);
        const u64 xpj = acc52_redc(a, B.mp[j]);
        xps[j] = split_of(xpj);
        xpd[j] = fp_from_u64(xpj);
    }
    const u32 a2 = conv_alpha_fp<NP>(xpd, xps, B.f_pinv, B.Mp, B.Ap, B.nwp, col, cstride);
#pragma unroll
    for (int i = 0; i < NS; i++) {
        acc52 a{B.n_cNegHalf[i], 0, 0};
#pragma unroll
        for (int j = 0; j < NP; j++) acc52_macs(a, xps[j], B.t_cC2[i][j]);
        acc52_mac_small(a, a2, B.n_cNegA2[i]);
        const u64 r = acc52_redc(a, B.mq[i]);
        dst[(size_t)i * ld] = lifted_out ? centred_double_bits(r, join_of(split26{B.mq[i].pl, B.mq[i].ph})) : r;
    }
}
