// glds_probe.hip -- does LDS-DMA (global_load_lds_dwordx4) reach LDS byte addresses above 64 KiB on gfx950?
// Copies 128 KiB global -> LDS with 1 KiB wave-instructions, reads LDS back with ds_read and writes it out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ void glds16(const void* gsrc, u32 lds_byte_base) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_base) : "memory");
}

__global__ __launch_bounds__(512) void k_probe(const u64* __restrict__ src, u64* __restrict__ dst, int use_builtin) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) u64*)lds;
    for (int i = 0; i < 16; i++) {
        const u32 chunk = wave * 16 + i;  // 1 KiB chunks
        const char* g = (const char*)src + (size_t)chunk * 1024 + lane * 16;
        if (use_builtin) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)lds + chunk * 1024), 16, 0, 0);
        } else {
            glds16(g, __builtin_amdgcn_readfirstlane(lds0 + chunk * 1024));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int r = 0; r < 32; r++) dst[tid + 512 * r] = lds[tid + 512 * r];
}

int main() {
    const size_t n = 16384;
    std::vector<u64> h(n), o(n);
    for (size_t i = 0; i < n; i++) h[i] = 0x1000000000ull + i * 7919;
    u64 *d, *e;
    hipMalloc(&d, n * 8); hipMalloc(&e, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int ub = 0; ub < 2; ub++) {
        hipMemset(e, 0, n * 8);
        hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 8192);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(512), 131072 + 8192, 0, d, e, ub);
        hipError_t err = hipDeviceSynchronize();
        hipMemcpy(o.data(), e, n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0, first = n;
        for (size_t i = 0; i < n; i++) if (o[i] != h[i]) { bad++; if (first == n) first = i; }
        printf("%s: err=%d mismatches=%zu first_bad_word=%zu\n", ub ? "builtin" : "asm", (int)err, bad, first);
    }
    return 0;
}
