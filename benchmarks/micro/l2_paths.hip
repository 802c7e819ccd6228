// Microbenchmark: what one CU can pull out of L2 / Infinity Cache per mechanism, as a function of bytes in flight.
//   mode 0  LDS-DMA (buffer_load_dwordx4 .. lds), 1 KiB contiguous per wave instruction
//   mode 1  LDS-DMA, 8 rows x 128 B per instruction (row stride 1536 B: the K-contiguous GEMM operand, K = 768)
//   mode 2  VGPR loads, 1 KiB contiguous per instruction (what a fragment-packed weight copy would give)
//   mode 3  VGPR loads, 8 rows x 128 B
//   mode 4  VGPR loads, 16 rows x 64 B (MFMA-fragment shaped access of a row-major operand)
//   mode 5  half the instructions mode 1 (DMA), half mode 2 (VGPR): do the two paths add?
// Each wave issues U instructions, waits for all of them, repeats.  Build: hipcc --offload-arch=gfx950 -O3 l2_paths.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int U>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* __restrict__ src, unsigned footprint_mask, int iters,
                                                     unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wid = blockIdx.x * 4 + wave;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(footprint_mask + 1u + (1u << 20)), 0x00020000);
    // per-lane offset inside a piece
    int lin = lane * 16;
    int r128 = (lane >> 3) * 1536 + (lane & 7) * 16;
    int r64 = (lane & 15) * 1536 + (lane >> 4) * 16;
    u32x4 accv = {0, 0, 0, 0};
    unsigned char* wl = smem + wave * (U * 1024);
    unsigned seed = wid * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            seed = seed * 1664525u + 1013904223u;
            const unsigned piece = (seed >> 4) & footprint_mask & ~1023u;       // wave-uniform
            const int soff = __builtin_amdgcn_readfirstlane((int)piece);
            const bool dma = MODE == 0 || MODE == 1 || (MODE == 5 && (u & 1) == 0);
            if (dma) {
                const int vo = (MODE == 0) ? lin : r128;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wl + u * 1024), 16, vo, soff, 0, 0);
            } else {
                const int vo = (MODE == 2 || MODE == 5) ? lin : (MODE == 3 ? r128 : r64);
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool dma = MODE == 0 || MODE == 1 || (MODE == 5 && (u & 1) == 0);
            if (!dma) asm volatile("" ::"v"(v[u]));
        }
    }
    if (iters < 0) sink[tid] = accv[0] + smem[tid];
}

template <int MODE, int U>
double run(const unsigned char* d, unsigned mask, int wgpc, int iters, unsigned* sink) {
    auto k = stream_kernel<MODE, U>;
    int lds = 163840 / wgpc;            // forces exactly wgpc workgroups per CU
    lds &= ~1023;
    if (lds < 4 * U * 1024) return -1;  // each wave needs U KiB
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wgpc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d, mask, iters, sink);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d, mask, iters, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)grid * 4 * iters * U * 1024.0;
    return bytes / (best * 1e-3) / 1e9;   // GB/s chip-wide
}

template <int MODE>
void sweep(const char* name, const unsigned char* d, unsigned mask, unsigned* sink) {
    const int wg[] = {1, 2, 3, 4};
    for (int w : wg) {
        double r2 = run<MODE, 2>(d, mask, w, 4000, sink);
        double r4 = run<MODE, 4>(d, mask, w, 2000, sink);
        double r8 = run<MODE, 8>(d, mask, w, 1000, sink);
        double r16 = w <= 2 ? run<MODE, 16>(d, mask, w, 500, sink) : -1;
        printf("%-26s wg/CU %d  in flight/CU %3d %3d %3d %3d KiB   GB/s per CU: %6.1f %6.1f %6.1f %6.1f   chip TB/s: %5.2f %5.2f %5.2f %5.2f\n",
               name, w, w * 4 * 2, w * 4 * 4, w * 4 * 8, w * 4 * 16, r2 / 256, r4 / 256, r8 / 256, r16 / 256, r2 / 1e3, r4 / 1e3,
               r8 / 1e3, r16 / 1e3);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const size_t bufsz = (size_t)512 << 20;
    unsigned char* d;
    unsigned* sink;
    CK(hipMalloc(&d, bufsz + (2 << 20)));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(d, 1, bufsz + (2 << 20)));
    const unsigned masks[] = {(2u << 20) - 1, (32u << 20) - 1, (128u << 20) - 1};
    const char* mn[] = {"2 MiB (L2)", "32 MiB (L2 aggregate / MALL)", "128 MiB (MALL)"};
    for (int f = 0; f < 3; ++f) {
        printf("== footprint %s\n", mn[f]);
        sweep<0>("dma linear 1KiB", d, masks[f], sink);
        sweep<1>("dma 8 rows x 128B", d, masks[f], sink);
        sweep<2>("vgpr linear 1KiB", d, masks[f], sink);
        sweep<3>("vgpr 8 rows x 128B", d, masks[f], sink);
        sweep<4>("vgpr 16 rows x 64B", d, masks[f], sink);
        sweep<5>("mix dma rows + vgpr linear", d, masks[f], sink);
    }
    return 0;
}
