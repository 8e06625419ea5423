// tools/ntt_exchange_ablate.hip -- where do the 18 us per row of the 2^14-point forward transform go?  (design aid, DESIGN.md
// section 9.)  Same rows, same butterflies, one ingredient removed at a time (results are WRONG in every reduced mode):
//   default:        full | no second LDS exchange | no exchange | no exchange + twiddles from arithmetic | arithmetic only |
//                   full without global stores | full without row loads
//   <rows> m [L]:   every combination of {row loads, global stores, twiddle loads} on / off, exchanges always on; "uniform" =
//                   twiddle loads kept but every lane reads the same entry.  L = number of limbs (twiddle footprint).
// Findings (MI355X, 8192 rows): arithmetic 10.4 us; the LDS exchanges cost nothing; twiddle loads ALONE cost nothing
// (10.3 us); row loads alone + 2.1, stores alone + 2.4, both + 5.6, all three 17.7-18 us; the twiddle footprint (1 limb or 8)
// does not matter; nontemporal hints on the row traffic change nothing; generating the per-thread twiddles from one base
// per stage (13 loads instead of 61 per thread, + 11 % arithmetic) is slower (18.7), prefetching those bases at row start
// slower still (20.4).  The row traffic at the copy rate (5.3 TB/s) would take 12.4 us.  Code size is not a factor: with the
// item loop unrolled 8 times (~170 KB of straight-line code per trip, the fused kernels' shape) the times are the same
// (10.4 without memory traffic, 17.9 complete); 128 more live VGPRs (two accumulator sets) change nothing either.
// build (repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value tools/ntt_exchange_ablate.hip -o tools/bin/ntt_exchange_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <type_traits>
#include "../toyfhe.jl_amd/csrc/kernels.h"
#include "../toyfhe.jl_amd/csrc/ntt_tables.h"

struct ArithFpNoTw : ArithFp {   // twiddles from arithmetic instead of memory (wrong values, no loads)
    static TFHE_HD tw ld_fwd(const ctx& c, u32 i) { return ftw_t{(double)(i | 1u) * 4097.0 + c.pinv}; }
    static TFHE_HD tw ld_fwd_b(const ctx& c, u32 i) { return ftw_t{(double)(i | 1u) * 4097.0 + c.pinv}; }
};
template <int MODE>
__global__ __launch_bounds__(512) void k_fwd(const u64* __restrict__ src, u64* __restrict__ dst, const ntt_limb_t* __restrict__ LT,
                                             limb_sel_t sel, u32 nitems) {
    typedef typename std::conditional<(MODE == 3 || MODE == 4), ArithFpNoTw, ArithFp>::type A;
    constexpr int LOGB = 14, LOGT = 9;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    constexpr int E = 1 << (LOGB - LOGT);
    bool first = true;
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
        const typename A::ctx C = A::make(LT[sel.idx[item % (u32)sel.n]]);
        const u32 tid = fresh_tid();
        const u64* g = src + ((size_t)item << LOGB);
        u64 raw[E];
        typename A::elem v[E];
        if (MODE == 4 || MODE == 7) {
#pragma unroll
            for (int i = 0; i < E; i++) { raw[i] = (u64)(tid * 131u + (u32)i * 7919u + item); pin_vgpr(raw[i]); }
        } else {
            fwd_load_data<LOGB, LOGT, 0, K1, true, false>(raw, lds, g, tid);
        }
        if (!first) __syncthreads();
        first = false;
        fwd_compute<A, LOGB, LOGT, 0, K1, true, false, 0>(v, raw, nullptr, C, tid, 1u);
        if (MODE <= 1 || MODE >= 6) {
            fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
            __syncthreads();
            fwd_load_data<LOGB, LOGT, K1, K2, false, false>(raw, lds, nullptr, tid);
        } else {
#pragma unroll
            for (int i = 0; i < E; i++) { raw[i] = A::to_lds(v[i]); pin_vgpr(raw[i]); }
        }
        fwd_compute<A, LOGB, LOGT, K1, K2, false, false, 0>(v, raw, nullptr, C, tid, 1u);
        if (MODE == 0 || MODE >= 6) {
            __syncthreads();
            fwd_store<A, LOGB, LOGT, K1, K2, false>(v, lds, nullptr, C, tid, 0, 0u);
            __syncthreads();
            fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(raw, lds, nullptr, tid);
        } else {
#pragma unroll
            for (int i = 0; i < E; i++) { raw[i] = A::to_lds(v[i]); pin_vgpr(raw[i]); }
        }
        fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, raw, nullptr, C, tid, 1u);
        if (MODE == 4 || MODE == 6) {
            double acc = 0;
#pragma unroll
            for (int i = 0; i < E; i++) acc += v[i];
            if (acc == 1.2345e301) dst[tid] = 1;   // keeps the arithmetic alive, never stores
        } else {
            fwd_store<A, LOGB, LOGT, K1 + K2, K3, true>(v, lds, dst + ((size_t)item << LOGB), C, tid, 0, 0u);
        }
    }
}


struct ArithFpUniTw : ArithFp {   // twiddle loads from one address per wavefront (same instruction count, one cache line each)
    static TFHE_HD tw ld_fwd(const ctx& c, u32 i) { return ld(c.W, (i >> 20) + 5u); }
    static TFHE_HD tw ld_fwd_b(const ctx& c, u32 i) { return ld(c.Wb, (i >> 20) + 5u); }
};
template <int MASK>
__global__ __launch_bounds__(512) void k_mask(const u64* __restrict__ src, u64* __restrict__ dst, const ntt_limb_t* __restrict__ LT,
                                              limb_sel_t sel, u32 nitems) {
    typedef typename std::conditional<(MASK & 4) != 0, ArithFpNoTw, typename std::conditional<(MASK & 8) != 0, ArithFpUniTw, ArithFp>::type>::type A;
    constexpr int LOGB = 14, LOGT = 9;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    constexpr int E = 1 << (LOGB - LOGT);
    bool first = true;
    double acc[2][(MASK & 16) ? E : 1];
    if (MASK & 16) {
#pragma unroll
        for (int i = 0; i < E; i++) acc[0][i] = acc[1][i] = 0.0;
    }
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
        const typename A::ctx C = A::make(LT[sel.idx[item % (u32)sel.n]]);
        const u32 tid = fresh_tid();
        const u64* g = src + ((size_t)item << LOGB);
        u64 raw[E];
        typename A::elem v[E];
        if (MASK & 1) {
#pragma unroll
            for (int i = 0; i < E; i++) { raw[i] = (u64)(tid * 131u + (u32)i * 7919u + item); pin_vgpr(raw[i]); }
        } else {
            fwd_load_data<LOGB, LOGT, 0, K1, true, false>(raw, lds, g, tid);
        }
        if (!first) __syncthreads();
        first = false;
        fwd_compute<A, LOGB, LOGT, 0, K1, true, false, 0>(v, raw, nullptr, C, tid, 1u);
        fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
        __syncthreads();
        fwd_load_data<LOGB, LOGT, K1, K2, false, false>(raw, lds, nullptr, tid);
        fwd_compute<A, LOGB, LOGT, K1, K2, false, false, 0>(v, raw, nullptr, C, tid, 1u);
        __syncthreads();
        fwd_store<A, LOGB, LOGT, K1, K2, false>(v, lds, nullptr, C, tid, 0, 0u);
        __syncthreads();
        fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(raw, lds, nullptr, tid);
        fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, raw, nullptr, C, tid, 1u);
        if (MASK & 16) {
#pragma unroll
            for (int i = 0; i < E; i++) {
                const double y = fp_reduce(v[i], C.p, C.pinv);
                acc[0][i] += fp_mulmod_c(y, ftw_t{C.pinv * 3.0 + (double)i}, C.p, C.pinv);
                acc[1][i] += fp_mulmod_c(y, ftw_t{C.pinv * 5.0 + (double)i}, C.p, C.pinv);
            }
        } else if (MASK & 2) {
            double a1 = 0;
#pragma unroll
            for (int i = 0; i < E; i++) a1 += v[i];
            if (a1 == 1.2345e301) dst[tid] = 1;
        } else {
            fwd_store<A, LOGB, LOGT, K1 + K2, K3, true>(v, lds, dst + ((size_t)item << LOGB), C, tid, 0, 0u);
        }
    }
    if (MASK & 16) {
        double a1 = 0;
#pragma unroll
        for (int i = 0; i < E; i++) a1 += acc[0][i] * acc[1][i];
        if (a1 == 1.2345e301) dst[threadIdx.x] = 1;
    }
}
// the same kernel with the item loop unrolled 8 times: ~170 KB of straight-line code per trip (the fused kernels' shape)
template <int MASK>
__global__ __launch_bounds__(512) void k_mask_bloat(const u64* __restrict__ src, u64* __restrict__ dst, const ntt_limb_t* __restrict__ LT,
                                              limb_sel_t sel, u32 nitems) {
    typedef typename std::conditional<(MASK & 4) != 0, ArithFpNoTw, typename std::conditional<(MASK & 8) != 0, ArithFpUniTw, ArithFp>::type>::type A;
    constexpr int LOGB = 14, LOGT = 9;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int K1 = pass_k_fwd(LOGB, LOGT, 0), K2 = pass_k_fwd(LOGB, LOGT, K1), K3 = LOGB - K1 - K2;
    constexpr int E = 1 << (LOGB - LOGT);
    bool first = true;
    double acc[2][(MASK & 16) ? E : 1];
    if (MASK & 16) {
#pragma unroll
        for (int i = 0; i < E; i++) acc[0][i] = acc[1][i] = 0.0;
    }
#pragma unroll 8
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
        const typename A::ctx C = A::make(LT[sel.idx[item % (u32)sel.n]]);
        const u32 tid = fresh_tid();
        const u64* g = src + ((size_t)item << LOGB);
        u64 raw[E];
        typename A::elem v[E];
        if (MASK & 1) {
#pragma unroll
            for (int i = 0; i < E; i++) { raw[i] = (u64)(tid * 131u + (u32)i * 7919u + item); pin_vgpr(raw[i]); }
        } else {
            fwd_load_data<LOGB, LOGT, 0, K1, true, false>(raw, lds, g, tid);
        }
        if (!first) __syncthreads();
        first = false;
        fwd_compute<A, LOGB, LOGT, 0, K1, true, false, 0>(v, raw, nullptr, C, tid, 1u);
        fwd_store<A, LOGB, LOGT, 0, K1, false>(v, lds, nullptr, C, tid, 0, 0u);
        __syncthreads();
        fwd_load_data<LOGB, LOGT, K1, K2, false, false>(raw, lds, nullptr, tid);
        fwd_compute<A, LOGB, LOGT, K1, K2, false, false, 0>(v, raw, nullptr, C, tid, 1u);
        __syncthreads();
        fwd_store<A, LOGB, LOGT, K1, K2, false>(v, lds, nullptr, C, tid, 0, 0u);
        __syncthreads();
        fwd_load_data<LOGB, LOGT, K1 + K2, K3, false, true>(raw, lds, nullptr, tid);
        fwd_compute<A, LOGB, LOGT, K1 + K2, K3, false, true, 0>(v, raw, nullptr, C, tid, 1u);
        if (MASK & 16) {
#pragma unroll
            for (int i = 0; i < E; i++) {
                const double y = fp_reduce(v[i], C.p, C.pinv);
                acc[0][i] += fp_mulmod_c(y, ftw_t{C.pinv * 3.0 + (double)i}, C.p, C.pinv);
                acc[1][i] += fp_mulmod_c(y, ftw_t{C.pinv * 5.0 + (double)i}, C.p, C.pinv);
            }
        } else if (MASK & 2) {
            double a1 = 0;
#pragma unroll
            for (int i = 0; i < E; i++) a1 += v[i];
            if (a1 == 1.2345e301) dst[tid] = 1;
        } else {
            fwd_store<A, LOGB, LOGT, K1 + K2, K3, true>(v, lds, dst + ((size_t)item << LOGB), C, tid, 0, 0u);
        }
    }
    if (MASK & 16) {
        double a1 = 0;
#pragma unroll
        for (int i = 0; i < E; i++) a1 += acc[0][i] * acc[1][i];
        if (a1 == 1.2345e301) dst[threadIdx.x] = 1;
    }
}
template <int MASK>
void runm(const ntt_limb_t* LT, int L, u64* a, u64* b, int rows) {
    limb_sel_t sel; sel.n = L; for (int j = 0; j < L; j++) sel.idx[j] = j;
    const size_t lds = (size_t)lds_words<14, 9>() * 8;
    hipFuncSetAttribute((const void*)k_mask<MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; it++) hipLaunchKernelGGL(k_mask<MASK>, dim3(256), dim3(512), lds, 0, a, b, LT, sel, (u32)rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5;
    }
    if (MASK & 16) printf("[128 accumulator VGPRs live, +product] ");
    printf("mask: rowloads %s  stores %s  twiddles %-7s  %.2f us per row per CU\n", (MASK & 1) ? "-" : "Y", (MASK & 2) ? "-" : "Y",
           (MASK & 4) ? "-" : (MASK & 8) ? "uniform" : "Y", best * 1e3 / (rows / 256.0));
}
template <int MASK>
void runb(const ntt_limb_t* LT, int L, u64* a, u64* b, int rows) {
    limb_sel_t sel; sel.n = L; for (int j = 0; j < L; j++) sel.idx[j] = j;
    const size_t lds = (size_t)lds_words<14, 9>() * 8;
    hipFuncSetAttribute((const void*)k_mask_bloat<MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; it++) hipLaunchKernelGGL(k_mask_bloat<MASK>, dim3(256), dim3(512), lds, 0, a, b, LT, sel, (u32)rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5;
    }
    if (MASK & 16) printf("[128 accumulator VGPRs live, +product] ");
    printf("[item loop unrolled x8] mask: rowloads %s  stores %s  twiddles %-7s  %.2f us per row per CU\n", (MASK & 1) ? "-" : "Y", (MASK & 2) ? "-" : "Y",
           (MASK & 4) ? "-" : (MASK & 8) ? "uniform" : "Y", best * 1e3 / (rows / 256.0));
}
template <int MODE>
void run(const char* name, const ntt_limb_t* LT, int L, u64* a, u64* b, int rows) {
    limb_sel_t sel; sel.n = L; for (int j = 0; j < L; j++) sel.idx[j] = j;
    const size_t lds = (size_t)lds_words<14, 9>() * 8;
    hipFuncSetAttribute((const void*)k_fwd<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; it++) hipLaunchKernelGGL(k_fwd<MODE>, dim3(256), dim3(512), lds, 0, a, b, LT, sel, (u32)rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5;
    }
    std::vector<u64> hb(16384 * 8); hipMemcpy(hb.data(), b, hb.size() * 8, hipMemcpyDeviceToHost);
    u64 cs = 0; for (u64 x : hb) cs = cs * 1099511628211ull + x;
    printf("[%016llx] ", (unsigned long long)cs);
    printf("%-34s rows=%d  %.3f ms  %.0f GB/s  %.2f us per row per CU\n", name, rows, best, (double)rows * 2 * 16384 * 8 / (best * 1e-3) / 1e9, best * 1e3 / (rows / 256.0));
}

int main(int argc, char** argv) {
    const int64_t N = 16384; const int L = argc > 3 ? atoi(argv[3]) : 8;
    const int rows = argc > 1 ? atoi(argv[1]) : 8192;
    std::vector<ntt_limb_t> LT(L);
    u64 q = (1ull << 50) + 1;
    for (int l = 0; l < L; l++) {
        do { q += 2 * N; } while (!hostmath::is_prime(q));
        ntt_host_tabs_t HT;
        build_ntt_tables_all(N, q, hostmath::minimal_primitive_root(q, 2 * N), HT, &LT[l]);
        auto up = [&](const void* h, size_t bytes) { void* d; hipMalloc(&d, bytes); hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); return d; };
        LT[l].W = (twd_t*)up(HT.W.data(), N * 16); LT[l].Winv = (twd_t*)up(HT.Wi.data(), N * 16);
        LT[l].Wb = (twd_t*)up(HT.Wb.data(), N * 16); LT[l].Winvb = (twd_t*)up(HT.Wib.data(), N * 16);
        LT[l].Wd = (ftwd_t*)up(HT.Wd.data(), N * 8); LT[l].Winvd = (ftwd_t*)up(HT.Wid.data(), N * 8);
        LT[l].Wdb = (ftwd_t*)up(HT.Wdb.data(), N * 8); LT[l].Winvdb = (ftwd_t*)up(HT.Widb.data(), N * 8);
    }
    ntt_limb_t* dLT; hipMalloc(&dLT, L * sizeof(ntt_limb_t)); hipMemcpy(dLT, LT.data(), L * sizeof(ntt_limb_t), hipMemcpyHostToDevice);
    u64 *a, *b; hipMalloc(&a, (size_t)rows * N * 8); hipMalloc(&b, (size_t)rows * N * 8);
    std::vector<u64> h((size_t)rows * N);
    u64 s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = s % ((1ull << 50) + 1); }
    hipMemcpy(a, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    if (argc > 2) {
        runm<0>(dLT, L, a, b, rows); runm<1>(dLT, L, a, b, rows); runm<2>(dLT, L, a, b, rows); runm<3>(dLT, L, a, b, rows);
        runm<4>(dLT, L, a, b, rows); runm<5>(dLT, L, a, b, rows); runm<6>(dLT, L, a, b, rows); runm<7>(dLT, L, a, b, rows);
        runm<16 + 2>(dLT, L, a, b, rows); runm<16 + 3>(dLT, L, a, b, rows); runm<16 + 7>(dLT, L, a, b, rows);
        runb<0>(dLT, L, a, b, rows); runb<3>(dLT, L, a, b, rows); runb<7>(dLT, L, a, b, rows);
        runm<8>(dLT, L, a, b, rows); runm<9>(dLT, L, a, b, rows); runm<10>(dLT, L, a, b, rows); runm<11>(dLT, L, a, b, rows);
        return 0;
    }
    for (int r = 0; r < 2; r++) {
        run<0>("full (two exchanges)", dLT, L, a, b, rows);
        run<1>("no second exchange", dLT, L, a, b, rows);
        run<2>("no exchange at all", dLT, L, a, b, rows);
        run<3>("no exchange, no twiddle loads", dLT, L, a, b, rows);
        run<4>("arithmetic only", dLT, L, a, b, rows);
        run<6>("full, no global stores", dLT, L, a, b, rows);
        run<7>("full, no row loads", dLT, L, a, b, rows);
    }
    return 0;
}
