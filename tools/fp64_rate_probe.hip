// Issue-rate probe for the fp64 instructions the butterflies use (gfx950): N independent chains per thread, many
// iterations; prints wave-instructions per clock per SIMD.  build: hipcc -O3 --offload-arch=gfx950 -o fp64_rate_probe fp64_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int iters) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = a * (threadIdx.x + i + 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) v[i] = __builtin_fma(v[i], b, a);
            else if (OP == 1) v[i] = v[i] + b;
            else if (OP == 2) v[i] = v[i] * b;
            else if (OP == 3) { v[i] = __builtin_rint(v[i]); asm volatile("" : "+v"(v[i])); }
            else if (OP == 4) { long long t = (long long)v[i]; asm volatile("" : "+v"(t)); v[i] = (double)t; }
            else if (OP == 5) { v[i] = __builtin_fabs(v[i]) ; asm volatile("" : "+v"(v[i])); }
            else if (OP == 6) { v[i] = (v[i] + 6755399441055744.0); asm volatile("" : "+v"(v[i])); v[i] -= 6755399441055744.0; }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, double* d) {
    const int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions: blocks*4 waves * iters*8
    const double winst = (double)blocks * 4 * iters * 8;
    printf("%-10s %8.3f ms  %.1f G wave-inst/s  -> %.2f clk/inst/SIMD @2.4GHz (1024 SIMDs)\n", name, ms, winst / ms / 1e6,
           2.4e9 * 1024 / (winst / (ms * 1e-3)));
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("fma", d); run<1>("add", d); run<2>("mul", d); run<3>("rndne", d); run<4>("cvt i64", d); run<5>("fabs", d); run<6>("magic2", d);
    run<0>("fma", d); run<3>("rndne", d);
    return 0;
}
