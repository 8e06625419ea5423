// Does the number of VGPR source operands change the fp64 issue rate on gfx950?  The butterflies' modular product with the
// twiddle in a VGPR (per-thread twiddles of the middle / last passes) has 2- and 3-VGPR-operand fma/mul; with the twiddle in an
// SGPR (uniform first pass) at most two.  One wave per SIMD ... eight; prints ns per wave-instruction per SIMD (wall clock).
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o fp64_operand_probe tools/fp64_operand_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* tw, double p, double pinv, int iters) {
    constexpr int ILP = 8;
    double v[ILP], w[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { v[i] = 1.0 + threadIdx.x + i; w[i] = MODE == 0 ? tw[i] : tw[(threadIdx.x * 8 + i) & 1023]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                const double ww = MODE == 0 ? __builtin_amdgcn_readfirstlane((int)0) + tw[i] * 0 + w[i] : w[i];
                if (MODE == 2) {            // 3-VGPR fma only
                    v[i] = __builtin_fma(w[i], v[i], w[(i + 1) % ILP]);
                } else if (MODE == 3) {     // 1-VGPR fma only
                    v[i] = __builtin_fma(v[i], p, pinv);
                } else {
                    const double h = ww * v[i];
                    const double l = __builtin_fma(ww, v[i], -h);
                    const double q = __builtin_rint(h * pinv);
                    v[i] = __builtin_fma(-q, p, h) + l;
                }
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int W, double* d, double* tw) {
    const int iters = 4000, blocks = 256 * W;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, tw, 1125899908022273.0, 1.0 / 1125899908022273.0, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, tw, 1125899908022273.0, 1.0 / 1125899908022273.0, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)iters * 4 * 8 * ((MODE == 2 || MODE == 3) ? 1 : 6);
    printf("%-34s waves/SIMD %d: %.3f ns per instruction per SIMD\n", name, W, ms * 1e6 / (per_wave * W));
}
int main() {
    double *d, *tw; hipMalloc(&d, 4096 * 256 * 8); hipMalloc(&tw, 1024 * 8);
    double h[1024]; for (int i = 0; i < 1024; i++) h[i] = 1000003.0 + 7919.0 * i;
    hipMemcpy(tw, h, sizeof h, hipMemcpyHostToDevice);
    for (int W = 1; W <= 4; W *= 2) {
        run<3>("fma 1 VGPR operand", W, d, tw);
        run<2>("fma 3 VGPR operands", W, d, tw);
        run<1>("modmul, twiddle in VGPRs", W, d, tw);
    }
    return 0;
}
