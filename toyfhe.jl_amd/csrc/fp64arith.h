// fp64arith.h -- modular arithmetic on exact integers held in IEEE doubles, for primes p < 2^50 + 2^40.
//
// Why: on gfx950 a 64-bit Shoup butterfly costs ten 32-bit multiplier ops (v_mad_u64_u32 / v_mul_lo /
// v_mul_hi, 4 cycles per wave64 each) plus carry chains; v_fma_f64 / v_mul_f64 / v_rndne_f64 issue at
// the same 4 cycles, and a constant-operand modular product is 6 of them (gpurun_out/microbench.txt).
// All results are exact integers -- fma-based error-free transformations, no rounding anywhere that
// matters -- so the transform stays bit-identical to the integer path (checked against the oracle).
//
//   mulmod_c(y; w):  h = fl(w y), l = fma(w, y, -h)              (h + l = w y exactly)
//                     k = rint(fl(h * fl(1/p)))                   (|k - w y / p| <= 1/2 + 1.5 |y| 2^-52)
//                     r = fma(-k, p, h) + l                       (exact: |r| <= p (1/2 + 1.5 |y| 2^-52) < 2^53)
// A twiddle is ONE double (8 bytes in the table, 2 VGPRs).  Values are kept as signed lazy
// representatives; `reduce` brings |v| back to <= p/2 (+ tiny).
// Range bookkeeping (in units of p; a = p 2^-52 <= 0.25 (1 + 2^-10), exactness limit 2^53 / p >= 7.99; growth per forward
// stage b' = b (1 + 1.5 a) + 1/2 = 1.3754 b + 1/2).  The admissible range ends just above 2^50 on purpose: the moduli of
// this size class are the primes next to 2^50 (the reference's rule takes the first NTT-friendly primes above it), and
// with a <= 0.2503 five stages fit between two sweeps -- the old bound 1.125 * 2^50 allowed four.
//   forward: residues enter uncentred (|v| <= 1); a sweep (reduce everything to <= 1/2) is planned before the first stage
//        that would pass 7.9, across pass boundaries -- LDS holds lazy doubles (ntt_core.h fp_fwd_sweep_before):
//        1.88 3.08 4.73 7.01 | 1.19 2.13 3.43 5.22 7.68 | 1.19 ...  = two sweeps in a 14-stage block, 7.68 at its end
//   inverse: sums double per stage, products return to <= 1/2 + 1.5 a b; only the operands whose sum would pass 7.9 p are
//        reduced, by a compile-time plan (ntt_core.h make_inv_plan)
#pragma once
#include "modarith.h"

#define TFHE_FP_QMAX 1126999418470400ull /* 2^50 + 2^40 */
#ifndef TFHE_FP_A                   /* (overridable for design experiments: plans for a smaller size class of moduli) */
#define TFHE_FP_A 0.25025          /* >= TFHE_FP_QMAX 2^-52 */
#define TFHE_FP_LIMIT 7.9           /* < 2^53 / TFHE_FP_QMAX = 7.992 */
#endif

// A second size class (r04): moduli below 2^42 -- the 40-bit chains of the reference's CKKS parameter sets (test/ckks_*.jl,
// infer.jl:98-107).  a = p 2^-52 < 2^-10, so a stage adds 1/2 + 0.0015 b and the exactness limit is 2^53 / p > 2048: a forward
// transform of any supported size needs NO sweep (1 + 17 x 0.52 < 10), key-product terms need no reduced operand, and an inverse
// transform (sums double per stage) needs one sweep per ~11 stages instead of one per pass (ArithFpS, ntt_core.h).
#define TFHE_FPS_QMAX (1ull << 42)
#define TFHE_FPS_A 0.000977         /* >= 2^42 2^-52 */
#define TFHE_FPS_LIMIT 2000.0       /* < 2^53 / 2^42 = 2048 */

struct ftw_t {  // twiddle w, an exact integer < p
    double w;
};
typedef double ftwd_t;  // table entry

TFHE_HD double fp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
TFHE_HD double fp_rint(double x) { return __builtin_rint(x); }

#if defined(TFHE_EMUL_TRACK_RANGE) && !defined(__HIP_DEVICE_COMPILE__)
// CPU emulation only (tests/emul): largest |operand| / p that ever entered a modular product or a reduction
#define TFHE_TRACK(v, p) do { const double r_ = ((v) < 0 ? -(v) : (v)) / (p); if (r_ > g_fp_max_ratio) g_fp_max_ratio = r_; } while (0)
#else
#define TFHE_TRACK(v, p) ((void)0)
#endif
TFHE_HD double fp_mulmod_c(double y, ftw_t t, double p, double pinv) {
    TFHE_TRACK(y, p);
    const double h = t.w * y;
    const double l = fp_fma(t.w, y, -h);
    const double k = fp_rint(h * pinv);
    return fp_fma(-k, p, h) + l;
}
TFHE_HD double fp_reduce(double v, double p, double pinv) {
    TFHE_TRACK(v, p);
    return fp_fma(-fp_rint(v * pinv), p, v);
}

// exact conversions for 0 <= x < 2^52
TFHE_HD double fp_from_u64(u64 x) {
    const u64 bits = x | 0x4330000000000000ull;  // 2^52 + x
    double d;
    __builtin_memcpy(&d, &bits, 8);
    return d - 4503599627370496.0;
}
TFHE_HD u64 fp_to_u64(double v) {  // v an integer in [0, 2^52)
    const double d = v + 4503599627370496.0;
    u64 bits;
    __builtin_memcpy(&bits, &d, 8);
    return bits & 0x000fffffffffffffull;
}
// canonical residue in [0, p) of a lazy value
TFHE_HD u64 fp_canon(double v, double p, double pinv) {
    double r = fp_reduce(v, p, pinv);
    r = r < 0.0 ? r + p : r;
    return fp_to_u64(r);
}
