// tools/microbench.hip -- instruction-throughput probes for gfx950 integer / fp64 paths (design aid).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
#define ITERS 4096
#define ILP 8
template <int OP>
__global__ __launch_bounds__(256) void probe(u64* out, u64 seed) {
    u64 v[ILP]; double d[ILP];
    for (int i = 0; i < ILP; i++) { v[i] = seed + threadIdx.x * 977 + i * 131; d[i] = (double)(v[i] & 0xffff) + 1.5; }
    u64 k = seed | 1; double dk = 1.0000001;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) v[i] = (u64)((u32)v[i] * (u32)k);                              // v_mul_lo_u32
            else if (OP == 1) v[i] = (u64)__umulhi((u32)v[i], (u32)k);                   // v_mul_hi_u32
            else if (OP == 2) v[i] = (u64)(u32)v[i] * (u32)k + v[i];                     // v_mad_u64_u32
            else if (OP == 3) v[i] = __umul64hi(v[i], k) + 1;                            // 64x64 high
            else if (OP == 4) v[i] = v[i] * k + 1;                                       // 64x64 low
            else if (OP == 5) v[i] = v[i] + k;                                           // 64-bit add
            else if (OP == 6) d[i] = fma(d[i], dk, 0.5);                                 // v_fma_f64
            else if (OP == 7) d[i] = d[i] * dk;                                          // v_mul_f64
            else if (OP == 8) d[i] = (double)(u32)v[i] + d[i], v[i] += 3;                // cvt u32->f64 + add
            else if (OP == 9) { u64 lo = v[i] * k, hi = __umul64hi(v[i], k); v[i] = lo ^ hi; }  // full 64x64->128
            else if (OP == 10) v[i] = (u64)__builtin_rint(d[i]), d[i] += 1.0;            // f64 -> int path
            else if (OP == 11) { float f = (float)d[i]; f = __builtin_fmaf(f, 1.0001f, 0.5f); d[i] = f; }  // f32 fma ref
        }
    }
    u64 acc = 0; double da = 0;
    for (int i = 0; i < ILP; i++) { acc ^= v[i]; da += d[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = acc + (u64)da;
}
template <int OP> void run(const char* name, u64* d_out) {
    const int blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<OP><<<blocks, 256>>>(d_out, 12345); hipDeviceSynchronize();
    hipEventRecord(a); probe<OP><<<blocks, 256>>>(d_out, 12345); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * ITERS * ILP;
    double per_clk_cu = ops / (ms * 1e-3) / 2.4e9 / 256;
    printf("%-28s %8.3f ms  %7.2f Gop/s  %6.1f lane-ops/clk/CU (at 2.4 GHz)  => %.2f cyc per wave64-op per SIMD\n", name, ms, ops / ms / 1e6, per_clk_cu, 64.0 / (per_clk_cu / 4));
}
int main() {
    u64* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_mul_lo_u32", d); run<1>("v_mul_hi_u32", d); run<2>("v_mad_u64_u32", d); run<3>("umul64hi", d);
    run<4>("mul64 lo (+1)", d); run<5>("add64", d); run<6>("v_fma_f64", d); run<7>("v_mul_f64", d);
    run<8>("cvt_u32_f64+add", d); run<9>("mul64 full 128", d); run<10>("rint f64->u64", d); run<11>("f32 fma (+cvt)", d);
    return 0;
}
