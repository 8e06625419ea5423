// conv_core.h -- exact RNS base conversion, per coefficient.
//
// The reference changes RNS basis by reconstructing a BigInt per coefficient
// (convert(Integer, ::CRTEncoded), src/crt.jl:98-112) and re-reducing it (src/crt.jl:91-95); the
// centred variant is SignedMod (src/signedmod.jl:12-19) / switchel (src/bfv.jl:202-220).  Results
// must therefore be *exact* -- a BEHZ/HPS "fast base conversion" that may be off by a multiple of
// the source modulus is not bit-compatible.  This file computes the same exact value without big
// integers on the common path:
//
//   x = Σ_j ξ_j (A/a_j) - α A,   ξ_j = x_j (A/a_j)^-1 mod a_j,   α = floor(Σ_j ξ_j / a_j)   (exactly)
//
// α is obtained from a 64-bit fixed-point sum that under-estimates Σ ξ_j/a_j by less than 2k ulp;
// when the fractional part is within 2k ulp of wrapping, α is decided by an exact multi-word
// comparison instead (rare: probability ~k 2^-63 for uniform x, but taken for structured inputs such
// as x = 1, and forced by tests).  The centred lift uses x' = x + floor(A/2) mod A, whose plain lift
// minus floor(A/2) is the centred representative with the reference's tie rule (n > A÷2 ⇒ n - A),
// and which keeps small |x| -- the FHE-typical case -- far away from the wrap.
#pragma once
#include "modarith.h"

#define TFHE_MAX_LIMBS 40

// Device-resident table for one (source basis -> targets) conversion.
struct conv_tab_t {
    int k, m;          // source limbs, target moduli
    int nwords;        // words of A (= k rounded to what A needs)
    int lazy;          // products that may be accumulated before a reduction
    // per source limb j
    u64 a[TFHE_MAX_LIMBS];        // modulus a_j
    tw_t inv[TFHE_MAX_LIMBS];     // (A/a_j)^-1 mod a_j
    u64 half[TFHE_MAX_LIMBS];     // floor(A/2) mod a_j
    u64 rho[TFHE_MAX_LIMBS];      // floor((2^128-1) / (a_j << sh_j)) - 2^64
    u32 sh[TFHE_MAX_LIMBS];       // normalisation shift (a_j << sh_j has bit 63 set)
    // per target i
    barrett_t t[TFHE_MAX_LIMBS];  // target modulus t_i
    u64 Amod[TFHE_MAX_LIMBS];     // A mod t_i
    u64 halfT[TFHE_MAX_LIMBS];    // floor(A/2) mod t_i
    int copy_from[TFHE_MAX_LIMBS];  // >= 0: t_i == a_{copy_from}, the residue is copied (centred lift ≡ x mod a_j)
    // C[j*m + i] = (A/a_j) mod t_i ; M[j*nwords + w] = word w of A/a_j ; Aw[w] = word w of A
    const u64* C;
    const u64* M;
    const u64* Aw;
};

// Step 1: xi[j*stride] holds residue x_j on entry and ξ_j on exit.  Returns α.
TFHE_HD u32 conv_prepare(const conv_tab_t& T, u64* xi, int stride, bool centred) {
    u64 frac = 0;
    u32 carries = 0;
    for (int j = 0; j < T.k; j++) {
        u64 x = xi[(size_t)j * stride];
        const u64 aj = T.a[j];
        if (centred) x = addmod(x, T.half[j], aj);
        const u64 xij = shoup_full(x, T.inv[j], aj);
        xi[(size_t)j * stride] = xij;
        // floor(ξ 2^64 / a) under-estimated by < 2: ξ̄ + mulhi(ξ̄, rho) with ξ̄ = ξ << sh
        const u64 xb = xij << T.sh[j];
        const u64 f = xb + mulhi64(xb, T.rho[j]);
        const u64 s = frac + f;
        carries += (s < f);
        frac = s;
    }
    const u64 slack = 2ull * (u64)T.k;
    if (frac + slack >= frac) return carries;  // no wrap possible: α is certain
    // Exact decision: is X = Σ ξ_j (A/a_j) >= (carries+1) A ?  Word-serial subtract, keep the borrow.
#if defined(TFHE_EMUL_COUNT_SLOW) && !defined(__HIP_DEVICE_COMPILE__)
    g_slow_hits++;
#endif
    const u64 mult = (u64)carries + 1;
    u64 acc_lo = 0, acc_hi = 0, acc_ex = 0;  // running column sum of X (192-bit)
    u64 mcarry = 0;                          // carry of mult * A
    u64 borrow = 0;
    for (int w = 0; w <= T.nwords; w++) {
        if (w < T.nwords) {
            for (int j = 0; j < T.k; j++) {
                const u64 xij = xi[(size_t)j * stride], mw = T.M[(size_t)j * T.nwords + w];
                const u64 lo = xij * mw, hi = mulhi64(xij, mw);
                u64 s = acc_lo + lo;
                u64 c = (s < lo);
                acc_lo = s;
                s = acc_hi + hi;
                u64 c2 = (s < hi);
                s += c;
                c2 += (s < c);
                acc_hi = s;
                acc_ex += c2;
            }
        }
        const u64 xw = acc_lo;  // word w of X
        acc_lo = acc_hi; acc_hi = acc_ex; acc_ex = 0;
        const u64 aw = w < T.nwords ? T.Aw[w] : 0;
        const u64 plo = aw * mult, phi = mulhi64(aw, mult);
        u64 yw = plo + mcarry;  // word w of mult*A
        mcarry = phi + (yw < plo);
        const u64 d = xw - yw;
        const u64 b1 = xw < yw;
        const u64 b2 = d < borrow;
        borrow = b1 | b2;
    }
    return carries + (borrow ? 0u : 1u);  // no final borrow  <=>  X >= (carries+1) A
}

// The exact plain lift itself, X - alpha A in [0, A), as T.nwords little-endian 64-bit words (xi, alpha from
// conv_prepare(…, centred = false)).  Used where the reference takes digits of convert(Integer, x)
// (rlwe_she.jl:334).  Word-serial: column sums of X in a 192-bit window, alpha A subtracted with a borrow chain.
TFHE_HD void conv_words(const conv_tab_t& T, const u64* xi, int stride, u32 alpha, u64* out) {
    u64 acc_lo = 0, acc_hi = 0, acc_ex = 0, mcarry = 0, borrow = 0;
    for (int w = 0; w < T.nwords; w++) {
        for (int j = 0; j < T.k; j++) {
            const u64 xij = xi[(size_t)j * stride], mw = T.M[(size_t)j * T.nwords + w];
            const u64 lo = xij * mw, hi = mulhi64(xij, mw);
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
        const u64 xw = acc_lo;  // word w of X
        acc_lo = acc_hi; acc_hi = acc_ex; acc_ex = 0;
        const u64 aw = T.Aw[w];
        const u64 plo = aw * (u64)alpha, phi = mulhi64(aw, (u64)alpha);
        const u64 yw = plo + mcarry;  // word w of alpha*A
        mcarry = phi + (yw < plo);
        const u64 d = xw - yw;
        out[w] = d - borrow;
        borrow = (u64)(xw < yw) | (u64)(d < borrow);
    }
}

// Base-2^w digits of one coefficient (key switch with relin_window = w, rlwe_she.jl:333-337): digit i of
// convert(Integer, x), x in [0, A) rebuilt exactly from the residues c[l*ls] (a single limb is its own integer);
// digit i is written to d[i*ds_digit + l*ds_limb] for every limb l (the same small value in each).
// nlw: limb rows written per digit (level, or level + 1 when the special prime's limb rides along: the digit is below every
// modulus, so its residue is the digit itself)
TFHE_HD void window_digits_coeff(const conv_tab_t* T, const u64* c, size_t ls, int level, int wbits, int nwin, u64* d,
                                 size_t ds_digit, size_t ds_limb, int nlw = -1) {
    if (nlw < 0) nlw = level;
    u64 words[TFHE_MAX_LIMBS + 1];
    int nwords = 1;
    if (level == 1) {
        words[0] = c[0];
    } else {
        u64 xi[TFHE_MAX_LIMBS];
        for (int l = 0; l < level; l++) xi[l] = c[(size_t)l * ls];
        const u32 alpha = conv_prepare(*T, xi, 1, false);
        conv_words(*T, xi, 1, alpha, words);
        nwords = T->nwords;
    }
    words[nwords] = 0;
    const u64 mask = (1ull << wbits) - 1;
    for (int i = 0; i < nwin; i++) {
        const int bit = i * wbits, wd = bit >> 6, off = bit & 63;
        u64 v = wd < nwords ? words[wd] >> off : 0;
        if (off + wbits > 64 && wd + 1 < nwords) v |= words[wd + 1] << (64 - off);
        v &= mask;
        for (int l = 0; l < nlw; l++) d[(size_t)i * ds_digit + (size_t)l * ds_limb] = v;
    }
}

// Step 2: the exact value (plain lift, or centred lift when `centred`) modulo target i.
TFHE_HD u64 conv_eval(const conv_tab_t& T, const u64* xi, int stride, int i, u32 alpha, bool centred) {
    const barrett_t& bt = T.t[i];
    u64 r;
    const int cf = T.copy_from[i];
    if (cf >= 0) {
        // target equals a source modulus: x mod a_j is the stored residue; recover it from ξ_j
        // (ξ_j = x'_j inv_j  =>  x'_j = ξ_j (A/a_j) mod a_j = ξ_j C[j][i] mod t_i)
        r = mulmod(xi[(size_t)cf * stride], T.C[(size_t)cf * T.m + i], bt);
        if (centred) r = submod(r, T.halfT[i], bt.q);
        return r;
    }
    acc128 acc{0, 0};
    u64 sum = 0;
    int pending = 0;
    for (int j = 0; j < T.k; j++) {
        acc_mac(acc, xi[(size_t)j * stride], T.C[(size_t)j * T.m + i]);
        if (++pending == T.lazy) {
            sum = addmod(sum, barrett_reduce128(acc.lo, acc.hi, bt), bt.q);
            acc = acc128{0, 0};
            pending = 0;
        }
    }
    if (pending) sum = addmod(sum, barrett_reduce128(acc.lo, acc.hi, bt), bt.q);
    r = submod(sum, mulmod((u64)alpha, T.Amod[i], bt), bt.q);
    if (centred) r = submod(r, T.halfT[i], bt.q);
    return r;
}
