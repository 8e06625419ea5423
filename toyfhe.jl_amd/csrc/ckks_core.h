// ckks_core.h -- per-coefficient pieces of the CKKS encoding (float; reference src/ckksencoding.jl:56-97 and the
// FixedRational conversions src/ckks.jl:35-59).  Host/device inline code, shared with the CPU emulation.
//
//   encode:  n = round(BigInt, big(x) * scale)   (ckks.jl:42; exact product, RoundNearest = ties to even),
//            stored as n mod q_l per limb (negative n: modulus + n, ckks.jl:43-45)
//   decode:  Float64(n / scale) with n the centred CRT reconstruction (ckks.jl:52-58)
// scale = smant * 2^sexp with a 64-bit integer smant (2^40, 2^80, ... exactly; any other positive scale to 2^-63).
#pragma once
#include "conv_core.h"

// z mod q for a 128-bit z (any size): three Barrett steps of at most 2^(bits+32)
TFHE_HD u64 mod128(u64 lo, u64 hi, const barrett_t& bt) {
    u64 r = barrett_reduce128(hi, 0, bt);
    r = barrett_reduce128((r << 32) | (lo >> 32), r >> 32, bt);
    r = barrett_reduce128((r << 32) | (lo & 0xffffffffull), r >> 32, bt);
    return r;
}
TFHE_HD u64 pow2mod(u32 e, const barrett_t& bt) {  // 2^e mod q
    u64 r = 1 % bt.q, b = 2 % bt.q;
    while (e) {
        if (e & 1u) r = mulmod(r, b, bt);
        b = mulmod(b, b, bt);
        e >>= 1;
    }
    return r;
}

// |n| = round_half_even(|x| * smant * 2^sexp) represented as mag * 2^lshift (mag < 2^128, lshift >= 0)
struct ckks_int_t {
    u64 lo, hi;
    u32 lshift;
    bool neg;
};
TFHE_HD ckks_int_t ckks_round_scaled(double x, u64 smant, int sexp) {
    ckks_int_t r{0, 0, 0, false};
    u64 bits;
    __builtin_memcpy(&bits, &x, 8);
    r.neg = (bits >> 63) != 0;
    const int ex = (int)((bits >> 52) & 0x7ff);
    u64 m = bits & 0x000fffffffffffffull;
    int e;
    if (ex == 0) e = -1074;                    // subnormal (or zero)
    else { m |= 1ull << 52; e = ex - 1075; }   // |x| = m * 2^e
    if (m == 0 || ex == 0x7ff) { r.neg = false; return r; }  // zero (inf / nan are the caller's problem: encoded as 0)
    u64 plo, phi;
    mul64_full(m, smant, plo, phi);            // exact 117-bit product
    const int s = e + sexp;
    if (s >= 0) { r.lo = plo; r.hi = phi; r.lshift = (u32)s; return r; }
    const int t = -s;
    if (t > 128) { r.neg = false; return r; }  // below 1/2: rounds to 0
    u64 nlo, nhi, rem_hi, rem_lo;              // n = P >> t, rem = P mod 2^t   (1 <= t <= 128)
    if (t < 64) { nlo = (plo >> t) | (phi << (64 - t)); nhi = phi >> t; rem_lo = plo & ((1ull << t) - 1); rem_hi = 0; }
    else if (t == 64) { nlo = phi; nhi = 0; rem_lo = plo; rem_hi = 0; }
    else if (t < 128) { nlo = phi >> (t - 64); nhi = 0; rem_lo = plo; rem_hi = phi & ((1ull << (t - 64)) - 1); }
    else { nlo = 0; nhi = 0; rem_lo = plo; rem_hi = phi; }
    const u64 half_lo = t <= 64 ? 1ull << (t - 1) : 0, half_hi = t > 64 ? 1ull << (t - 65) : 0;  // 2^(t-1)
    const bool gt = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
    const bool eq = rem_hi == half_hi && rem_lo == half_lo;
    if (gt || (eq && (nlo & 1ull))) { nlo += 1; nhi += (nlo == 0); }
    r.lo = nlo; r.hi = nhi;
    if ((nlo | nhi) == 0) r.neg = false;
    return r;
}
TFHE_HD u64 ckks_residue(const ckks_int_t& n, const barrett_t& bt) {
    u64 r = mod128(n.lo, n.hi, bt);
    if (n.lshift) r = mulmod(r, pow2mod(n.lshift, bt), bt);
    return n.neg ? negmod(r, bt.q) : r;
}

// magnitude words (little endian, nwords) and sign -> nearest double of  +-mag * 2^-sexp / smant
TFHE_HD double ckks_words_to_double(const u64* w, int nwords, bool neg, u64 smant, int sexp) {
    int top = nwords - 1;
    while (top >= 0 && w[top] == 0) top--;
    if (top < 0) return 0.0;
    const int lz = __builtin_clzll(w[top]);
    const int bitlen = (top + 1) * 64 - lz;
    // take the top 64 bits (value >> sh) and a sticky bit for everything below
    u64 hi64;
    bool sticky = false;
    const int sh = bitlen - 64;
    if (sh <= 0) hi64 = w[0];
    else {
        const int wd = sh >> 6, off = sh & 63;
        hi64 = w[wd] >> off;
        if (off && wd + 1 < nwords) hi64 |= w[wd + 1] << (64 - off);
        if (off && (w[wd] & ((1ull << off) - 1))) sticky = true;
        for (int i = 0; i < wd; i++) sticky = sticky || w[i] != 0;
    }
    // 64 -> 53 bits, round to nearest even with the sticky bit
    double d;
    if (sh <= 0 && bitlen <= 53) d = (double)hi64;
    else {
        const int drop = (sh <= 0 ? bitlen : 64) - 53;  // low bits of hi64 to drop
        u64 keep = hi64 >> drop;
        const u64 rem = hi64 & ((1ull << drop) - 1), halfv = 1ull << (drop - 1);
        if (rem > halfv || (rem == halfv && (sticky || (keep & 1ull)))) keep += 1;
        d = (double)keep;  // exact (<= 2^53)
        d = __builtin_ldexp(d, drop + (sh > 0 ? sh : 0));
    }
    d = __builtin_ldexp(d, -sexp);
    if (smant != 1) d = d / (double)smant;
    return neg ? -d : d;
}
