#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <set>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(u32* out) {
  if (threadIdx.x == 0) {
    u32 xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = (xcc & 0xf) << 16 | ((hw >> 8) & 0xf) << 8 | ((hw >> 13) & 0x7) << 4 | ((hw >> 12) & 1);   // xcc, cu_id, se_id, sh_id
  }
  // stay resident a while so that blocks spread over all allowed CUs
  u64 t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 20000) {}
}
// VALU-bound spinner: fp64 fma chains, `iters` per thread
__global__ __launch_bounds__(512) void k_valu(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0000001, c = 0.5, d = 0.25;
  for (int i = 0; i < iters; i++) { a = fma(a, b, c); c = fma(c, b, d); d = fma(d, b, a); b = fma(b, 1.0, 1e-9); }
  if (a + c + d == 12345.0) out[0] = a;
}
// HBM-bound copy
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("CUs %d\n", pr.multiProcessorCount);
  const int S = 24, NCU = pr.multiProcessorCount;
  std::vector<u32> mA(NCU / 32, 0), mB(NCU / 32, 0);
  for (int i = 0; i < NCU; i++) { if (i < NCU - S) mA[i / 32] |= 1u << (i % 32); else mB[i / 32] |= 1u << (i % 32); }
  hipStream_t sA, sB, s0;
  CK(hipExtStreamCreateWithCUMask(&sA, (uint32_t)mA.size(), mA.data()));
  CK(hipExtStreamCreateWithCUMask(&sB, (uint32_t)mB.size(), mB.data()));
  CK(hipStreamCreate(&s0));
  u32* w; CK(hipMalloc(&w, 4096 * 4));
  std::vector<u32> h(4096);
  for (int which = 0; which < 3; which++) {
    hipStream_t st = which == 0 ? s0 : which == 1 ? sA : sB;
    CK(hipMemset(w, 0xff, 4096 * 4));
    hipLaunchKernelGGL(k_where, dim3(2048), dim3(64), 0, st, w);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), w, 4096 * 4, hipMemcpyDeviceToHost));
    std::set<u32> cus; int per_xcc[16] = {0};
    for (int i = 0; i < 2048; i++) cus.insert(h[i]);
    for (u32 c : cus) per_xcc[(c >> 16) & 0xf]++;
    printf("stream %d: distinct (xcc,cu,se,sh) = %zu; per xcc:", which, cus.size());
    for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
    printf("\n");
  }
  // overlap test
  const size_t nb = (size_t)2 << 30; uint4 *a, *b; CK(hipMalloc(&a, nb)); CK(hipMalloc(&b, nb)); CK(hipMemset(a, 1, nb));
  double* o; CK(hipMalloc(&o, 8));
  hipEvent_t e0, e1, f0, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
  auto valu = [&](hipStream_t st, int grid) { hipLaunchKernelGGL(k_valu, dim3(grid), dim3(512), 0, st, o, 400000); };
  auto copy = [&](hipStream_t st, int grid) { for (int r = 0; r < 8; r++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, a, b, nb / 16); };
  float t;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0, s0); valu(s0, NCU); hipEventRecord(e1, s0); hipStreamSynchronize(s0); hipEventElapsedTime(&t, e0, e1); printf("valu alone full chip: %.2f ms\n", t);
    hipEventRecord(e0, s0); copy(s0, NCU * 8); hipEventRecord(e1, s0); hipStreamSynchronize(s0); hipEventElapsedTime(&t, e0, e1); printf("copy alone full chip: %.2f ms  (%.2f TB/s)\n", t, 8 * 2.0 * nb / t / 1e9);
    hipEventRecord(e0, sA); valu(sA, NCU - S); hipEventRecord(e1, sA); hipStreamSynchronize(sA); hipEventElapsedTime(&t, e0, e1); printf("valu alone on %d CUs: %.2f ms\n", NCU - S, t);
    hipEventRecord(f0, sB); copy(sB, S * 8); hipEventRecord(f1, sB); hipStreamSynchronize(sB); hipEventElapsedTime(&t, f0, f1); printf("copy alone on %d CUs: %.2f ms  (%.2f TB/s)\n", S, t, 8 * 2.0 * nb / t / 1e9);
    auto h0 = std::chrono::steady_clock::now();
    hipEventRecord(e0, sA); valu(sA, NCU - S); hipEventRecord(e1, sA);
    hipEventRecord(f0, sB); copy(sB, S * 8); hipEventRecord(f1, sB);
    hipStreamSynchronize(sA); hipStreamSynchronize(sB);
    auto h1 = std::chrono::steady_clock::now();
    float tv, tc; hipEventElapsedTime(&tv, e0, e1); hipEventElapsedTime(&tc, f0, f1);
    printf("together: valu %.2f ms, copy %.2f ms (%.2f TB/s), wall %.2f ms\n", tv, tc, 8 * 2.0 * nb / tc / 1e9, std::chrono::duration<double, std::milli>(h1 - h0).count());
  }
  return 0;
}
