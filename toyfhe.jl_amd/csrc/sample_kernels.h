// sample_kernels.h -- device-side samplers for RingSampler (src/poly.jl:7-23, RNS variant src/crt.jl:277-279): uniform
// residues and rounded-Gaussian noise, so that keygen / encrypt can run without a host round trip.
// Randomness cannot match Julia's generator; the stream is defined here instead: Philox4x32-10
// (Salmon, Moraes, Dror, Shaw, SC'11), key = seed, counter = (coefficient index, polynomial index, attempt | limb, stream id):
// the polynomial counter and the coefficient index live in separate counter words, so draws for rings of different degree
// never share a counter.  A statistical generator for reproducible experiments, NOT a cryptographically secure one.
// oracle/spec.py carries the same definition; the uniform sampler is checked bit-for-bit against it.
#pragma once
#include <math.h>

#include "modarith.h"

struct philox_t {
    u32 c[4];
};
TFHE_HD philox_t philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1) {
    for (int r = 0; r < 10; r++) {
        const u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;
        const u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0, n1 = (u32)p1, n2 = (u32)(p0 >> 32) ^ c3 ^ k1, n3 = (u32)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return philox_t{{c0, c1, c2, c3}};
}
// uniform residue in [0, q): 64-bit draws, rejected above the largest multiple of q (exactly uniform); draw 2a and 2a+1 come
// from attempt a's Philox block.  counter = (idx lo, idx hi, attempt << 8 | limb, stream)
TFHE_HD u64 sample_uniform_mod(u64 idx, u32 limb, u32 stream, u64 seed, u64 q) {
    const u64 lim = (~0ull / q) * q;  // accept r < lim
    for (u32 a = 0;; a++) {
        const philox_t b = philox4x32_10((u32)idx, (u32)(idx >> 32), (a << 8) | limb, stream, (u32)seed, (u32)(seed >> 32));
        const u64 r0 = ((u64)b.c[1] << 32) | b.c[0], r1 = ((u64)b.c[3] << 32) | b.c[2];
        if (r0 < lim) return r0 % q;
        if (r1 < lim) return r1 % q;
    }
}
// rounded Gaussian integer: Box-Muller on two 53-bit uniforms of one Philox block, e = rint(sigma * z) (ties to even)
TFHE_HD long long sample_gauss_int(u64 idx, u32 stream, u64 seed, double sigma) {
    const philox_t b = philox4x32_10((u32)idx, (u32)(idx >> 32), 0xffffffffu, stream, (u32)seed, (u32)(seed >> 32));
    const u64 r0 = ((u64)b.c[1] << 32) | b.c[0], r1 = ((u64)b.c[3] << 32) | b.c[2];
    const double u1 = ((double)(r0 >> 11) + 1.0) * 0x1p-53, u2 = (double)(r1 >> 11) * 0x1p-53;  // (0,1], [0,1)
    const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    return (long long)rint(sigma * z);
}

#if defined(__HIPCC__)
#include "ntt_core.h"
struct limb_sel_t;
// out [count][level][N]: independent uniform residues per limb (crt.jl:146-148: a uniform CRT element is uniform limb-wise)
__global__ __launch_bounds__(256) void k_sample_uniform(u64* __restrict__ out, const ntt_limb_t* __restrict__ LT, int level, u64 seed,
                                                        u32 stream, u32 n, u64 first_poly) {
    const u32 k = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    const u64 p = blockIdx.z;
    if (k >= n) return;
    out[((size_t)p * level + l) * n + k] = sample_uniform_mod(((first_poly + p) << 32) | k, l, stream, seed, LT[l].q);
}
// out [count][level][N]: mult * e with e ~ round(N(0, sigma^2)), the same integer reduced into every limb
__global__ __launch_bounds__(256) void k_sample_gaussian(u64* __restrict__ out, const ntt_limb_t* __restrict__ LT, int level, double sigma,
                                                         u64 mult, u64 seed, u32 stream, u32 n, u64 first_poly) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    const u64 p = blockIdx.y;
    if (k >= n) return;
    const long long e = sample_gauss_int(((first_poly + p) << 32) | k, stream, seed, sigma);
    const u64 mag = (u64)(e < 0 ? -e : e);
    for (int l = 0; l < level; l++) {
        const barrett_t& bt = LT[l].br;
        u64 r = barrett_reduce128(mag, 0, bt);
        r = mulmod(r, mult % bt.q, bt);
        out[((size_t)p * level + l) * n + k] = e < 0 ? negmod(r, bt.q) : r;
    }
}
#endif
