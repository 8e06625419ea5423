// ckks_kernels.h -- CKKS encode / decode on the device (float; ckksencoding.jl:56-97, ckks.jl:35-59).
// Not a hot path: plain global-memory radix-2 FFT passes (complex double), one butterfly per thread.
//   encode: slots -> scatter (bit-reversed positions) -> DIT passes with conj roots -> x_k = Re(ip_k tw_k)/N ->
//           round(x * scale) -> residues per limb
//   decode: residues -> centred integer (exact CRT) -> double / scale -> * conj(tw_k) -> DIF passes -> gather
#pragma once
#include "ckks_core.h"

struct cplx_t {
    double re, im;
};
TFHE_HD cplx_t cmul(cplx_t a, cplx_t b) { return cplx_t{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// slots [batch][n/2] -> cm [batch][n] at bit-reversed positions; pos[2*i], pos[2*i+1] = brev(idx1_i), brev(idx2_i)
__global__ __launch_bounds__(256) void k_ckks_scatter(const cplx_t* __restrict__ slots, cplx_t* __restrict__ cm,
                                                      const u32* __restrict__ pos, u32 n) {
    const u32 n2 = n >> 1, b = blockIdx.y;
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    const cplx_t z = slots[(size_t)b * n2 + i];
    cplx_t* o = cm + (size_t)b * n;
    o[pos[2 * i]] = z;
    o[pos[2 * i + 1]] = cplx_t{z.re, -z.im};
}
// one radix-2 pass over [batch][n]; roots[t] = exp(-2 pi i t / n), t < n/2.  dif = 0: DIT butterfly (b * w first),
// dif = 1: DIF butterfly ((a - b) * w); conj_roots selects the inverse transform's roots.
__global__ __launch_bounds__(256) void k_fft_pass(cplx_t* __restrict__ data, const cplx_t* __restrict__ roots, u32 n, u32 h,
                                                  int dif, int conj_roots) {
    const u32 b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (n >> 1)) return;
    const u32 k = i & (h - 1), j = ((i - k) << 1) + k;
    cplx_t w = roots[(size_t)k * ((n >> 1) / h)];
    if (conj_roots) w.im = -w.im;
    cplx_t* d = data + (size_t)b * n;
    const cplx_t x = d[j], y = d[j + h];
    if (dif) {
        d[j] = cplx_t{x.re + y.re, x.im + y.im};
        d[j + h] = cmul(cplx_t{x.re - y.re, x.im - y.im}, w);
    } else {
        const cplx_t t = cmul(y, w);
        d[j] = cplx_t{x.re + t.re, x.im + t.im};
        d[j + h] = cplx_t{x.re - t.re, x.im - t.im};
    }
}
// encode tail: x_k = Re(cm_k * tw_k) / n, n_k = round(x_k * scale), out[b][l][k] = n_k mod q_l
__global__ __launch_bounds__(256) void k_ckks_encode_finish(const cplx_t* __restrict__ cm, const cplx_t* __restrict__ tw,
                                                            u64* __restrict__ out, const ntt_limb_t* __restrict__ LT,
                                                            limb_sel_t sel, u64 smant, int sexp, u32 n) {
    const u32 b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const cplx_t z = cm[(size_t)b * n + k], w = tw[k];
    const double x = (z.re * w.re - z.im * w.im) / (double)n;
    const ckks_int_t v = ckks_round_scaled(x, smant, sexp);
    for (int l = 0; l < sel.n; l++) out[((size_t)b * sel.n + l) * n + k] = ckks_residue(v, LT[sel.idx[l]].br);
}
// decode head: residues -> double value / scale -> * conj(tw_k) into the FFT buffer
__global__ __launch_bounds__(256) void k_ckks_decode_start(const u64* __restrict__ in, cplx_t* __restrict__ buf,
                                                           const cplx_t* __restrict__ tw, const conv_tab_t* __restrict__ T,
                                                           int level, u64 q0, u64 smant, int sexp, u32 n) {
    const u32 b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const u64* c = in + ((size_t)b * level) * n + k;
    double v;
    if (level == 1) {
        const u64 x = c[0];
        const bool neg = x > (q0 >> 1);  // n > div(modulus, 2) => n - modulus  (ckks.jl:54-56)
        const u64 mag = neg ? q0 - x : x;
        v = ckks_words_to_double(&mag, 1, neg, smant, sexp);
    } else {
        u64 xi[TFHE_MAX_LIMBS], w[TFHE_MAX_LIMBS + 1];
        for (int l = 0; l < level; l++) xi[l] = c[(size_t)l * n];
        const u32 alpha = conv_prepare(*T, xi, 1, true);   // xi of x + floor(A/2)
        conv_words(*T, xi, 1, alpha, w);                    // X' = x_centred + floor(A/2) in [0, A)
        // subtract floor(A/2) = (A - 1) / 2: words of A shifted right by one
        const int nw = T->nwords;
        u64 borrow = 0;
        for (int i = 0; i < nw; i++) {
            const u64 hw = (T->Aw[i] >> 1) | (i + 1 < nw ? T->Aw[i + 1] << 63 : 0);
            const u64 d = w[i] - hw;
            const u64 b1 = w[i] < hw;
            w[i] = d - borrow;
            borrow = b1 | (u64)(d < borrow);
        }
        const bool neg = borrow != 0;
        if (neg) {  // two's complement magnitude
            u64 carry = 1;
            for (int i = 0; i < nw; i++) { const u64 t = ~w[i] + carry; carry = (t < carry) ? 1 : 0; w[i] = t; }
        }
        v = ckks_words_to_double(w, nw, neg, smant, sexp);
    }
    const cplx_t t = tw[k];
    buf[(size_t)b * n + k] = cplx_t{v * t.re, -v * t.im};
}
// decode tail: slots[b][i] = buf[b][gpos[i]]  (gpos = bit-reversed gather positions)
__global__ __launch_bounds__(256) void k_ckks_gather(const cplx_t* __restrict__ buf, cplx_t* __restrict__ slots,
                                                     const u32* __restrict__ gpos, u32 n) {
    const u32 n2 = n >> 1, b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    slots[(size_t)b * n2 + i] = buf[(size_t)b * n + gpos[i]];
}
