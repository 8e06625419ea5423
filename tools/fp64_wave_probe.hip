// Per-wave issue behaviour of fp64 VALU ops on gfx950: W waves per SIMD (W workgroups of 256 threads per CU), ILP independent
// dependency chains per thread; prints cycles per wave-instruction per SIMD measured with the shader clock (s_memtime).
// build: hipcc -O3 --offload-arch=gfx950 -o fp64_wave_probe tools/fp64_wave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, int OP>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, double a, double b, int iters) {
    double v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = a * (threadIdx.x + i + 1);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (OP == 0) v[i] = __builtin_fma(v[i], b, a);
                else { const double h = v[i] * b; const double l = __builtin_fma(v[i], b, -h); const double q = __builtin_rint(h * a); v[i] = __builtin_fma(-q, 3.0, h) + l; }  // the 6-op modular product
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP, int OP>
void run(int W, double* d, long long* c) {
    const int iters = 2000, blocks = 256 * W;
    hipLaunchKernelGGL((k<ILP, OP>), dim3(blocks), dim3(256), 0, 0, d, c, 1.0000001, 0.9999999, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ILP, OP>), dim3(blocks), dim3(256), 0, 0, d, c, 1.0000001, 0.9999999, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipDeviceSynchronize();
    static long long h[4096];
    hipMemcpy(h, c, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; i++) avg += (double)h[i]; avg /= blocks;
    const double per_wave = (double)iters * 8 * ILP * (OP == 0 ? 1 : 6);   // wave instructions per wave
    printf("%s  waves/SIMD %d  ILP %d : %.2f ticks per instruction per wave, %.2f per SIMD | kernel %.3f ms = %.2f ns per instruction per SIMD, counter %.3f GHz\n",
           OP == 0 ? "fma   " : "modmul", W, ILP, avg / per_wave, avg / per_wave / W, ms, ms * 1e6 / (per_wave * W), avg / (ms * 1e6));
}
int main() {
    double* d; long long* c; hipMalloc(&d, 4096 * 256 * 8); hipMalloc(&c, 4096 * 8);
    for (int W = 1; W <= 4; W *= 2) { run<1, 0>(W, d, c); run<2, 0>(W, d, c); run<4, 0>(W, d, c); run<8, 0>(W, d, c); }
    for (int W = 1; W <= 4; W *= 2) { run<1, 1>(W, d, c); run<2, 1>(W, d, c); run<4, 1>(W, d, c); run<8, 1>(W, d, c); }
    return 0;
}
