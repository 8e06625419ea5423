// kernels.h -- __global__ kernels for gfx950 (MI355X).  Wave64; one workgroup per RNS limb-polynomial
// (or per block of it) for the NTTs, one workgroup per limb row for the streaming kernels, one
// thread per coefficient for the cross-limb kernels.  Integer work: no MFMA by design.
#pragma once
#include <hip/hip_runtime.h>
#include "bfv_core.h"
#include "bfv_fast.h"
#include "ntt_core.h"

struct limb_sel_t {  // which context modulus each buffer limb uses (crtselect, src/crt.jl:185-211)
    int n;
    int idx[TFHE_MAX_LIMBS];
};

// ------------------------------------------------------------------------------------------------
// NTT: register-blocked LDS kernel.  Block b = ((poly*limbs + j) << x) + sb handles sub-block sb of
// limb j of polynomial `poly`; x = log2(N) - LOGB (0 when the whole limb fits one LDS block).
// ------------------------------------------------------------------------------------------------

// ---- simple schedule: one workgroup per item, passes separated by barriers ----
template <class A, int LOGB, int LOGT, int S0>
__device__ __forceinline__ void fwd_schedule(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 tid,
                                             u32 pre, int x, u32 sbrev, const lift_t* lift) {
    constexpr int K = pass_k_fwd(LOGB, LOGT, S0);
    constexpr bool LAST = (S0 + K == LOGB);
    ntt_fwd_pass<A, LOGB, LOGT, S0, K, S0 == 0, LAST>(lds, gsrc, gdst, C, tid, pre, x, sbrev, lift);
    if constexpr (!LAST) {
        __syncthreads();
        fwd_schedule<A, LOGB, LOGT, S0 + K>(lds, gsrc, gdst, C, tid, pre, x, sbrev, lift);
    }
}
template <class A, int LOGB, int LOGT, int SEND, bool SCALE>
__device__ __forceinline__ void inv_schedule(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 tid,
                                             u32 pre, int x, u32 sbrev, const u64* addend) {
    constexpr int K = pass_k_inv(LOGB, LOGT, SEND);
    constexpr int S0 = SEND - K;
    ntt_inv_pass<A, LOGB, LOGT, S0, K, SEND == LOGB, S0 == 0, SCALE>(lds, gsrc, gdst, C, tid, pre, x, sbrev, addend);
    if constexpr (S0 != 0) {
        __syncthreads();
        inv_schedule<A, LOGB, LOGT, S0, SCALE>(lds, gsrc, gdst, C, tid, pre, x, sbrev, addend);
    }
}

// Workgroups loop over items i = blockIdx.x, blockIdx.x + gridDim.x, ... (item = (row << x) + sb, row = poly*limbs + j
// in the plain mode; see ntt_io_t for the grouped / digit-lift / addend row maps).
template <class A, int LOGB, int LOGT, int IOMODE>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_fwd_block(const u64* __restrict__ src, u64* __restrict__ dst,
                                                              const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int x,
                                                              u32 nitems, ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const size_t ntot = (size_t)1 << (LOGB + x);
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
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
            lift = &lf;
        }
        const typename A::ctx C = A::make(LT[sel.idx[j]]);
        if (item != blockIdx.x) __syncthreads();  // the previous item's last pass has read LDS
        fwd_schedule<A, LOGB, LOGT, 0>(lds, src + srow * ntot + ((size_t)sb << LOGB), dst + drow * ntot, C, threadIdx.x,
                                       (1u << x) + sb, x, brev_bits(sb, x), lift);
    }
}
template <class A, int LOGB, int LOGT, int IOMODE>
__global__ __launch_bounds__(1 << LOGT) void k_ntt_inv_block(const u64* __restrict__ src, u64* __restrict__ dst,
                                                              const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int x,
                                                              u32 nitems, ntt_io_t io) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const size_t ntot = (size_t)1 << (LOGB + x);
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
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
        const typename A::ctx C = A::make(LT[sel.idx[j]]);
        if (item != blockIdx.x) __syncthreads();
        if constexpr (A::whole_block_only) {  // the fp64 variant is only dispatched for x == 0
            inv_schedule<A, LOGB, LOGT, LOGB, true>(lds, src + srow * ntot, dst + drow * ntot, C, threadIdx.x, 1u, 0, 0u, addend);
        } else {
            if (x == 0)
                inv_schedule<A, LOGB, LOGT, LOGB, true>(lds, src + srow * ntot, dst + drow * ntot, C, threadIdx.x, 1u, 0, 0u, addend);
            else
                inv_schedule<A, LOGB, LOGT, LOGB, false>(lds, src + srow * ntot, dst + drow * ntot + ((size_t)sb << LOGB), C,
                                                         threadIdx.x, (1u << x) + sb, x, brev_bits(sb, x), nullptr);
        }
    }
}

// top stages of N > 2^LOGB transforms: one column per thread, rows = count*limbs
template <int X>
__global__ __launch_bounds__(256) void k_ntt_fwd_top(const u64* __restrict__ src, u64* __restrict__ dst,
                                                      const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int logn) {
    const u64 stride = (u64)1 << (logn - X);
    const u32 chunks = (u32)((stride + 255) / 256);
    const u32 row = blockIdx.x / chunks;
    const ntt_limb_t L = LT[sel.idx[row % (u32)sel.n]];
    const u64 col = (u64)(blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (col < stride) ntt_fwd_top<X>(src + ((size_t)row << logn), dst + ((size_t)row << logn), L.W, L.q, col, stride);
}
template <int X>
__global__ __launch_bounds__(256) void k_ntt_inv_top(const u64* __restrict__ src, u64* __restrict__ dst,
                                                      const ntt_limb_t* __restrict__ LT, limb_sel_t sel, int logn) {
    const u64 stride = (u64)1 << (logn - X);
    const u32 chunks = (u32)((stride + 255) / 256);
    const u32 row = blockIdx.x / chunks;
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
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
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

// tensor (rlwe_she.jl:255-258) in the NTT domain: a,b [batch][2][limbs][N] -> out [batch][3][limbs][N]
__global__ __launch_bounds__(256) void k_tensor(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out,
                                                 const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)sel.n, ct = row / (u32)sel.n;
    const ntt_limb_t L = LT[sel.idx[j]];
    const size_t ps = (size_t)sel.n * n;  // poly stride
    const u64 *a0 = a + (size_t)ct * 2 * ps + (size_t)j * n, *a1 = a0 + ps;
    const u64 *b0 = b + (size_t)ct * 2 * ps + (size_t)j * n, *b1 = b0 + ps;
    u64 *o0 = out + (size_t)ct * 3 * ps + (size_t)j * n, *o1 = o0 + ps, *o2 = o1 + ps;
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 x0 = a0[i], x1 = a1[i], y0 = b0[i], y1 = b1[i];
        o0[i] = mulmod(x0, y0, L.br);
        acc128 acc{0, 0};
        acc_mac(acc, x0, y1);
        if (L.br.sh <= 58) {  // two products fit the Barrett window when q < 2^61
            acc_mac(acc, x1, y0);
            o1[i] = barrett_reduce128(acc.lo, acc.hi, L.br);
        } else {
            o1[i] = addmod(barrett_reduce128(acc.lo, acc.hi, L.br), mulmod(x1, y0, L.br), L.q);
        }
        o2[i] = mulmod(x1, y1, L.br);
    }
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
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 last = barrett_reduce128(cl[i], 0, L.br);  // unsigned representative of c_last, mod q_j
        d[i] = shoup_full(submod(cj[i], last, L.q), ra.qlinv[j], L.q);
    }
}

__global__ __launch_bounds__(256) void k_select(const u64* __restrict__ src, u64* __restrict__ dst, limb_sel_t which,
                                                 int src_limbs, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)which.n, p = row / (u32)which.n;
    const u64* s = src + ((size_t)p * src_limbs + which.idx[j]) * n;
    u64* d = dst + (size_t)row * n;
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
}

// apply_galois_element (pow2_cyc_rings.jl:321-329) in gather form: out[r] = ± in[i], g*i ≡ r (mod N).
// ginv = g^-1 mod 2N.  i0 = r*ginv mod 2N; i0 < N: +in[i0]; else -in[i0-N].
__global__ __launch_bounds__(256) void k_galois(const u64* __restrict__ src, u64* __restrict__ dst,
                                                 const ntt_limb_t* __restrict__ LT, limb_sel_t sel, u64 ginv, u32 n) {
    const u32 row = blockIdx.x;
    const u64 q = LT[sel.idx[row % (u32)sel.n]].q;
    const size_t base = (size_t)row * n;
    const u64 mask2n = 2ull * n - 1;
    for (u32 r = threadIdx.x; r < n; r += blockDim.x) {
        const u64 i0 = ((u64)r * ginv) & mask2n;
        const u64 v = src[base + (i0 & (n - 1))];
        dst[base + r] = (i0 >= n) ? negmod(v, q) : v;
    }
}

// ------------------------------------------------------------------------------------------------
// keyswitch pieces (rlwe_she.jl:315-347, modulusraising.jl:35-49)
// ------------------------------------------------------------------------------------------------
struct ks_arg_t {
    int level, nw, special, polys;
    limb_sel_t w;                // working limbs: key limbs 0..level-1 (+ special prime)
    tw_t pmul[TFHE_MAX_LIMBS];   // P mod q_j (special) for j < level
};

// S[b][s][j] = Σ_i evk[i][s'][w[j]] * D[b][i][j]  (NTT domain); s = 0 (c1) uses masked, s = 1 (c2) uses mask
// (rlwe_she.jl:340-344).  One thread owns coefficient k of limb j for the WHOLE chunk of ciphertexts, so the
// evaluation-key values are loaded once into registers and reused across the batch (the key is shared by
// all ciphertexts; re-reading it per ciphertext was the dominant traffic).  grid = nw * ceil(N/256).
template <int DCH>
__global__ __launch_bounds__(256) void k_ks_inner(const u64* __restrict__ evk, const u64* __restrict__ dig,
                                                   u64* __restrict__ S, const ntt_limb_t* __restrict__ LT, ks_arg_t A,
                                                   int Lk, u32 n, u32 batch, u32 bsplit) {
    // blockIdx.x = (slice * nw + j) * gx + tile; slice = which part of the batch this workgroup owns
    const u32 gx = (n + 255) / 256, tile = blockIdx.x % gx, j = (blockIdx.x / gx) % (u32)A.nw, slice = blockIdx.x / (gx * (u32)A.nw);
    const u32 k = tile * 256 + threadIdx.x;
    const u32 per = (batch + bsplit - 1) / bsplit, b_lo = slice * per, b_hi = b_lo + per < batch ? b_lo + per : batch;
    if (k >= n) return;
    const ntt_limb_t L = LT[A.w.idx[j]];
    const int lazy = L.br.sh <= 50 ? (1 << 10) : (L.br.sh <= 58 ? (1 << (60 - L.br.sh)) : 1);
    for (int i0 = 0; i0 < A.level; i0 += DCH) {
        u64 mk[DCH], md[DCH];
#pragma unroll
        for (int ii = 0; ii < DCH; ii++) {
            const int i = i0 + ii < A.level ? i0 + ii : A.level - 1;
            mk[ii] = evk[(((size_t)i * 2 + 0) * Lk + A.w.idx[j]) * n + k];
            md[ii] = evk[(((size_t)i * 2 + 1) * Lk + A.w.idx[j]) * n + k];
        }
        for (u32 b = b_lo; b < b_hi; b++) {
            u64* s1p = S + (((size_t)b * 2 + 0) * A.nw + j) * n + k;
            u64* s2p = S + (((size_t)b * 2 + 1) * A.nw + j) * n + k;
            acc128 s1{0, 0}, s2{0, 0};
            u64 r1 = i0 ? *s1p : 0, r2 = i0 ? *s2p : 0;
            int pend = 0;
#pragma unroll
            for (int ii = 0; ii < DCH; ii++) {
                if (i0 + ii < A.level) {
                    const u64 d = dig[(((size_t)b * A.level + i0 + ii) * A.nw + j) * n + k];
                    acc_mac(s1, md[ii], d);
                    acc_mac(s2, mk[ii], d);
                    if (++pend == lazy) {
                        r1 = addmod(r1, barrett_reduce128(s1.lo, s1.hi, L.br), L.q);
                        r2 = addmod(r2, barrett_reduce128(s2.lo, s2.hi, L.br), L.q);
                        s1 = acc128{0, 0}; s2 = acc128{0, 0}; pend = 0;
                    }
                }
            }
            if (pend) {
                r1 = addmod(r1, barrett_reduce128(s1.lo, s1.hi, L.br), L.q);
                r2 = addmod(r2, barrett_reduce128(s2.lo, s2.hi, L.br), L.q);
            }
            *s1p = r1;
            *s2p = r2;
        }
    }
}

// RNS digits as a separate pass (only for N > 2^14, where the lift is not fused into the NTT loads):
// dig [batch][level][nw][N]; digit i, limb j = centred([c_end]_{q_i}) mod q_w[j]  (rlwe_she.jl:326-329)
__global__ __launch_bounds__(256) void k_ks_digits(const u64* __restrict__ ct, u64* __restrict__ dig,
                                                    const ntt_limb_t* __restrict__ LT, ks_arg_t A, u32 n) {
    const u32 row = blockIdx.x, j = row % (u32)A.nw, i = (row / (u32)A.nw) % (u32)A.level,
              b = row / ((u32)A.nw * (u32)A.level);
    lift_t lf;
    lf.qi = LT[A.w.idx[i]].q; lf.half = lf.qi >> 1; lf.qj = LT[A.w.idx[j]].q; lf.bj = LT[A.w.idx[j]].br;
    const u64* c = ct + (((size_t)b * A.polys + (A.polys - 1)) * A.level + i) * n;
    u64* d = dig + (size_t)row * n;
    for (u32 k = threadIdx.x; k < n; k += blockDim.x) d[k] = lift_digit(c[k], lf);
}
// out[b][s][j] += c[b][s][j] for the components that have one (N > 2^14 path); rows = batch*2*level
__global__ __launch_bounds__(256) void k_ks_add_ct(const u64* __restrict__ ct, u64* __restrict__ out,
                                                    const ntt_limb_t* __restrict__ LT, ks_arg_t A, u32 n, u32 add_s) {
    const u32 row = blockIdx.x, j = row % (u32)A.level, s = (row / (u32)A.level) & 1u, b = row / (2u * (u32)A.level);
    if (s >= add_s) return;
    const u64 q = LT[A.w.idx[j]].q;
    const u64* c = ct + (((size_t)b * A.polys + s) * A.level + j) * n;
    u64* o = out + (size_t)row * n;
    for (u32 k = threadIdx.x; k < n; k += blockDim.x) o[k] = addmod(o[k], c[k], q);
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
    for (u32 k = threadIdx.x; k < n; k += blockDim.x) {
        const u64 last = barrett_reduce128(tl[k], 0, L.br);
        u64 v = shoup_full(submod(tj[k], last, L.q), ra.qlinv[j], L.q);
        if (c) v = addmod(v, c[k], L.q);
        o[k] = v;
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
template <int NS, int NP>
__global__ __launch_bounds__(256) void k_bfv_expand_fast(const u64* __restrict__ src, u64* __restrict__ dst,
                                                          const bfv_fast_tab_t* __restrict__ Bt, u32 n, u32 gx) {
    const u32 k = (blockIdx.x % gx) * 256 + threadIdx.x;
    const size_t p = blockIdx.x / gx;
    if (k >= n) return;
    bfv_expand_fast<NS, NP>(*Bt, src + p * NS * n + k, n, dst + p * (NS + NP) * n + k, n);
}
template <int NS, int NP>
__global__ __launch_bounds__(256) void k_bfv_contract_fast(const u64* __restrict__ src, u64* __restrict__ dst,
                                                            const bfv_fast_tab_t* __restrict__ Bt, u32 n, u32 gx) {
    const u32 k = (blockIdx.x % gx) * 256 + threadIdx.x;
    const size_t p = blockIdx.x / gx;
    if (k >= n) return;
    bfv_contract_fast<NS, NP>(*Bt, src + p * (NS + NP) * n + k, n, dst + p * NS * n + k, n);
}
