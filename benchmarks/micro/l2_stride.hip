// Microbenchmark: does the row stride of a K-contiguous GEMM operand (K * 2 bytes) and the lock-step walk of K hot-spot
// L2 channels?  Every wave instruction is an LDS-DMA of 8 rows x 128 B (what the GEMM kernels issue); the footprint is
// L2-resident (2 MiB).  "spread": every instruction picks a random 128-byte k-chunk; "lockstep": all waves of the chip read
// the same k-chunk index at the same iteration (tiles marching through K together); "skew": k-chunk = iteration + a
// per-workgroup offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void stride_kernel(const unsigned char* __restrict__ src, int stride, int nrowblk, int nchunk,
                                                     int mode, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wid = blockIdx.x * 4 + wave;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int vo = (lane >> 3) * stride + (lane & 7) * 16;
    unsigned char* wl = smem + wave * (U * 1024);
    unsigned seed = wid * 2654435761u + 12345u;
    unsigned kstep = 0, kskew = (blockIdx.x * 5u) % (unsigned)nchunk;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            seed = seed * 1664525u + 1013904223u;
            const unsigned rb = __umulhi(seed, (unsigned)nrowblk);             // no runtime division in the loop
            unsigned kc;
            if (mode == 0) kc = __umulhi(seed * 2246822519u, (unsigned)nchunk);
            else if (mode == 1) kc = kstep;
            else kc = kskew;
            const int soff = __builtin_amdgcn_readfirstlane((int)(rb * 8u * (unsigned)stride + kc * 128u));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wl + u * 1024), 16, vo, soff, 0, 0);
        }
        kstep = kstep + 1 == (unsigned)nchunk ? 0 : kstep + 1;
        kskew = kskew + 1 == (unsigned)nchunk ? 0 : kskew + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

int main() {
    unsigned char* d;
    CK(hipMalloc(&d, 64 << 20));
    CK(hipMemset(d, 1, 64 << 20));
    constexpr int U = 8;
    auto k = stride_kernel<U>;
    const int lds = 40 * 1024;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int strides[] = {1024, 1536, 2048, 3072, 4096, 4608, 6144, 8192, 1536 + 128, 4096 + 128, 6144 + 128};
    const char* mn[] = {"spread", "lockstep", "skew"};
    for (int s : strides) {
        const int nrows = (2 << 20) / s, nrowblk = nrows / 8, nchunk = s / 128;
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = 1000, grid = 1024;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d, s, nrowblk, nchunk, mode, iters);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d, s, nrowblk, nchunk, mode, iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double bytes = (double)grid * 4 * iters * U * 1024.0;
            printf("row stride %5d B (K = %4d bf16)  %-8s  %6.1f GB/s per CU  %5.2f TB/s\n", s, s / 2, mn[mode],
                   bytes / (best * 1e-3) / 1e9 / 256, bytes / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    }
    return 0;
}
