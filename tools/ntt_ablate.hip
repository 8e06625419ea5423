// tools/ntt_ablate.hip -- standalone timing / ablation harness for the N = 2^14 NTT kernels (design aid).
// Build (from repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value [-DABL=n] \
//                         tools/ntt_ablate.hip -o tools/bin/ntt_ablate[_n]
// ABL: 0 full kernel; see ntt_core.h hooks (TFHE_ABL_*) for what each variant removes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#ifndef TFHE_TRACE
#define TFHE_TRACE 100
#endif
#include "../toyfhe.jl_amd/csrc/kernels.h"
#include "../toyfhe.jl_amd/csrc/ntt_tables.h"

template <class A>
void run(const char* name, const ntt_limb_t* LT, int L, u64* d_a, u64* d_b, int rows, bool inverse, bool staged = false) {
    limb_sel_t sel; sel.n = L; for (int j = 0; j < L; j++) sel.idx[j] = j;
    const size_t lds = (size_t)lds_words<14, logt_for(14)>() * 8;
    auto kf = k_ntt_fwd_block<A, 14, logt_for(14), 0>; auto ki = k_ntt_inv_block<A, 14, logt_for(14), 0>;
    auto si = k_ntt_inv_staged<ArithFp, 14, logt_for(14), 0>; auto sf = si;  // (forward staged kernel retired; `staged` is inverse-only)
    hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)sf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)si, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = rows < 256 ? rows : 256;
    ntt_io_t io; memset(&io, 0, sizeof io);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; it++) {
            if (staged) {
                if (inverse) hipLaunchKernelGGL(si, dim3(grid), dim3(1 << logt_for(14)), lds, 0, d_a, d_b, LT, sel, (u32)rows, io);
                else hipLaunchKernelGGL(si, dim3(grid), dim3(1 << logt_for(14)), lds, 0, d_a, d_b, LT, sel, (u32)rows, io);
            } else {
                if (inverse) hipLaunchKernelGGL(ki, dim3(grid), dim3(1 << logt_for(14)), lds, 0, d_a, d_b, LT, sel, 0, (u32)rows, io);
                else hipLaunchKernelGGL(kf, dim3(grid), dim3(1 << logt_for(14)), lds, 0, d_a, d_b, LT, sel, 0, (u32)rows, io);
            }
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double gbs = (double)rows * 2 * 16384 * 8 / (ms * 1e-3) / 1e9;
    printf("%-28s rows=%d  %.3f ms  %.0f GB/s  (%.2f us per workgroup-slot)\n", name, rows, ms, gbs, ms * 1e3 / (rows / 256.0));
    if (staged) {  // phase stamps of workgroup TFHE_TRACE (shader clock ticks, relative to stamp 0 of each item)
        static unsigned long long tr[64 * 16];
        hipMemcpyFromSymbol(tr, HIP_SYMBOL(tfhe_trace), sizeof tr);
        const int items = (rows + 255) / 256;
        double sum[9] = {0}; int cnt = 0;
        for (int it = 1; it + 1 < items && it < 63; it++) {
            for (int k = 0; k < 9; k++) sum[k] += (double)(long long)(tr[it * 16 + k] - tr[it * 16]);
            sum[0] += (double)(long long)(tr[(it + 1) * 16] - tr[it * 16]);
            cnt++;
        }
        if (cnt) {
            printf("    ticks from stamp 0: ");
            for (int k = 1; k < 9; k++) printf(" s%d=%.0f", k, sum[k] / cnt);
            printf("  next-item=%.0f  (items %d)\n", sum[0] / cnt, cnt);
            const double ticks = (double)(tr[(cnt) * 16] - tr[16]), real = (double)(tr[cnt * 16 + 15] - tr[16 + 15]);
            printf("    shader clock over those items: %.0f MHz (memtime ticks per 100 MHz realtime tick x 100)\n", ticks / real * 100.0);
        }
    }
}

int main(int argc, char** argv) {
    const int64_t N = 16384; const int L = 8;
    int rows = argc > 1 ? atoi(argv[1]) : 4096;
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
    u64 *d_a, *d_b; hipMalloc(&d_a, (size_t)rows * N * 8); hipMalloc(&d_b, (size_t)rows * N * 8);
    std::vector<u64> h((size_t)rows * N);
    u64 s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = s % ((1ull << 50) + 1); }
    hipMemcpy(d_a, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    run<ArithFp>("inv fp64 staged", dLT, L, d_a, d_b, rows, true, true);
    run<ArithFp>("fwd fp64", dLT, L, d_a, d_b, rows, false);
    run<ArithFp>("inv fp64", dLT, L, d_a, d_b, rows, true);
    run<ArithInt>("fwd u64", dLT, L, d_a, d_b, rows, false);
    run<ArithInt>("inv u64", dLT, L, d_a, d_b, rows, true);
    return 0;
}
