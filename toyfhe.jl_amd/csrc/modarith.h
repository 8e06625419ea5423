// modarith.h -- 64-bit modular arithmetic for RNS limbs q < 2^62 (the device stand-in for
// GaloisFields.PrimeField arithmetic, reference call sites src/pow2_cyc_rings.jl:3,15 and the
// limb-wise ops of src/crt.jl:120-134).
//
// Everything here is plain integer code usable from HIP device code and (for the CPU index-logic
// emulation under tests/emul/) from host C++.  No MFMA: this is integer modular work; the scarce
// resources are the 32-bit integer multiplier and HBM/LDS bandwidth.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define TFHE_HD __host__ __device__ __forceinline__
#else
#define TFHE_HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

struct tw_t {  // a multiplier constant w (< q) with its Shoup companion floor(w * 2^64 / q)
    u64 w, wp;
};

TFHE_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((u128)a * b) >> 64);
#endif
}

// full 64 x 64 -> 128 product as four 32 x 32 + 64 multiply-adds (v_mad_u64_u32); the builtin lo/hi pair costs six
TFHE_HD void mul64_full(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 t1 = (u64)a1 * b0 + (p00 >> 32);
    const u64 t2 = (u64)a0 * b1 + (u32)t1;
    hi = (u64)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    lo = (t2 << 32) | (u32)p00;
#else
    const u128 p = (u128)a * b;
    lo = (u64)p;
    hi = (u64)(p >> 64);
#endif
}

// x * w mod q for a precomputed (w, wp); x is ANY u64, result in [0, 2q)  (Harvey/Shoup lazy form)
// Device form (r04): r = lo64(x w + qh (2^64 - q)) accumulated by v_mad_u64_u32 on a running 64-bit sum -- the low products of
// x w and of qh q share one accumulator, so there is no 64-bit subtraction (v_sub_co / v_subb and the wait state between them)
// and four multiplier instructions carry their additions for free: 30 instructions per butterfly where the plain expression
// below compiled to 33.5 (hipcc -S of eight chained butterflies; profiles/LOG.md r04).  Same value: everything is mod 2^64.
TFHE_HD u64 shoup_lazy(u64 x, tw_t t, u64 q) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), p0 = (u32)t.wp, p1 = (u32)(t.wp >> 32);
    const u64 m = (u64)x1 * p0 + __umulhi(x0, p0);
    const u64 n = (u64)x0 * p1 + (u32)m;
    const u64 qh = (u64)x1 * p1 + (m >> 32) + (n >> 32);                     // mulhi64(x, wp)
    const u64 nq = 0 - q;
    const u32 w0 = (u32)t.w, w1 = (u32)(t.w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32), h0 = (u32)qh, h1 = (u32)(qh >> 32);
    u64 r = (u64)x0 * w0;
    r = (u64)h0 * n0 + r;
    const u32 hi = (u32)(r >> 32) + x0 * w1 + x1 * w0 + h0 * n1 + h1 * n0;
    return ((u64)hi << 32) | (u32)r;
#else
    return x * t.w - mulhi64(x, t.wp) * q;
#endif
}
// conditional subtraction for the butterflies' ranges (x < 2 m <= 2^63 + ..: x - m < 2^63 whenever x >= m, and wraps above 2^63
// otherwise): the sign of the difference selects -- one 64-bit add of the constant -m instead of a compare and a borrow chain
TFHE_HD u64 csub_s(u64 x, u64 m) {
    const u64 d = x + (0 - m);
    return (long long)d < 0 ? x : d;
}
// same, fully reduced to [0, q)
TFHE_HD u64 shoup_full(u64 x, tw_t t, u64 q) {
    u64 r = shoup_lazy(x, t, q);
    return r >= q ? r - q : r;
}

TFHE_HD u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }  // conditional subtract
TFHE_HD u64 addmod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
TFHE_HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
TFHE_HD u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// Per-modulus Barrett data for products of two variable operands.
//   k  = bit length of q (q < 2^62), sh = k - 2,  mu = floor(2^(k+62) / q) < 2^63.
// For z < 2^(k+62):  qh = mulhi64(z >> sh, mu) satisfies z/q - 2.5 < qh <= z/q, so
// z - qh*q lies in [0, 4q) and fits a u64; two conditional subtractions finish it.
struct barrett_t {
    u64 q, mu;
    u32 sh;
};

TFHE_HD u64 barrett_reduce128(u64 zlo, u64 zhi, const barrett_t& m) {
    // z >> sh, sh in [0, 60]; callers guarantee z < 2^(sh+64)
    u64 z1 = m.sh ? ((zlo >> m.sh) | (zhi << (64 - m.sh))) : zlo;
    u64 qh = mulhi64(z1, m.mu);
    u64 r = zlo - qh * m.q;
    r = csub(r, 2 * m.q);
    return csub(r, m.q);
}

TFHE_HD u64 mulmod(u64 a, u64 b, const barrett_t& m) {
    u64 lo, hi;
    mul64_full(a, b, lo, hi);
    return barrett_reduce128(lo, hi, m);
}

// 128-bit accumulator for lazy sums of products (reduced once by barrett_reduce128)
struct acc128 {
    u64 lo, hi;
};
TFHE_HD void acc_mac(acc128& a, u64 x, u64 y) {
    u64 lo, hi;
    mul64_full(x, y, lo, hi);
    u64 s = a.lo + lo;
    a.hi += hi + (s < lo);
    a.lo = s;
}

// Sums of products of operands below 2^52, both split into 26-bit halves: the four partial products accumulate in three
// independent u64 lanes with ONE v_mad_u64_u32 each and no carry chain (up to 16 terms: lane s1 stays below 2^57).
struct acc52 {
    u64 s0, s1, s2;  // value = s0 + s1 2^26 + s2 2^52
};
TFHE_HD void acc52_mac(acc52& a, u32 x0, u32 x1, u32 c0, u32 c1) {
    a.s0 += (u64)x0 * c0;
    a.s1 += (u64)x0 * c1;
    a.s1 += (u64)x1 * c0;
    a.s2 += (u64)x1 * c1;
}
TFHE_HD void acc52_fold(const acc52& a, u64& lo, u64& hi) {
    u64 l = a.s0, h = 0;
    u64 t = a.s1 << 26;
    l += t;
    h += (a.s1 >> 38) + (l < t);
    t = a.s2 << 52;
    l += t;
    h += (a.s2 >> 12) + (l < t);
    lo = l;
    hi = h;
}

// Montgomery reduction of a carry-free product sum in radix 2^26: z = s0 + s1 2^26 + s2 2^52 (s0, s1 < 2^58 and s2 < 2^53, which the
// 16-term budget of acc52 guarantees for operands below 2^50 + 2^40) -> z 2^-78 mod p, canonical.  Three rounds, each clearing the
// low 26 bits of the lowest live lane with a multiple of p = ph 2^26 + pl and carrying the rest (which fits 32 bits) upwards:
// mul_lo, two v_mad_u64_u32, one funnel shift and one 64-bit add per round -- about 20 instructions all in, against
// ~50 for acc52_fold + barrett_reduce128.  The factor 2^78 is folded into the constant operands of the sum on the host
// (bfv_tables.h mont26_fold), so the result is the plain residue.
struct mont26_t {
    u32 pl, ph;  // p = ph 2^26 + pl
    u32 pp;      // -p^-1 mod 2^26
    u32 pad_;
};
#define TFHE_MONT26_RBITS 78
TFHE_HD u64 acc52_redc(acc52 a, const mont26_t& M) {
    // rounds one and two take the multiplier unmasked (any m = -s p^-1 mod 2^26 clears the low 26 bits; a 32-bit m only makes
    // the carries larger: s1 < 2^57.1, s2 < 2^56.1, both carries still fit 32 bits); the last round masks, so that r < 2 p
    u32 m = (u32)a.s0 * M.pp;
    a.s0 += (u64)m * M.pl;  // low 26 bits are zero now
    a.s1 += (u64)m * M.ph;
    a.s1 += (u64)(u32)(a.s0 >> 26);
    m = (u32)a.s1 * M.pp;
    a.s1 += (u64)m * M.pl;
    a.s2 += (u64)m * M.ph;
    a.s2 += (u64)(u32)(a.s1 >> 26);
    m = ((u32)a.s2 * M.pp) & 0x3ffffffu;
    a.s2 += (u64)m * M.pl;
    const u64 r = (u64)m * M.ph + (u64)(u32)(a.s2 >> 26);  // < p + 2^31 < 2 p
    const u64 p = ((u64)M.ph << 26) | M.pl;
    unsigned long long d;
    return __builtin_usubll_overflow(r, p, &d) ? r : (u64)d;  // the borrow of r - p selects
}

// bit reversal of the low `bits` bits
TFHE_HD u32 brev_bits(u32 x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}
