#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned char* out, int stride) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    const int lane = threadIdx.x;
    for (int pass = 0; pass < 2; ++pass) {
        for (int a = lane; a < 8192; a += 64) lds[a] = pass == 0 ? (a & 0xff) : ((a >> 8) & 0xff);
        __syncthreads();
        typedef __attribute__((address_space(3))) v2i lds_v2i;
        v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(lds + lane * stride));
        __syncthreads();
        for (int j = 0; j < 8; ++j) out[(pass * 64 + lane) * 8 + j] = (unsigned char)(r[j >> 2] >> (8 * (j & 3)));
    }
}
int main() {
    unsigned char* d; hipMalloc(&d, 1024);
    for (int stride : {8, 64, 128}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned char h[1024]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        printf("stride %d: result byte j of lane i <- (source lane, byte)\n", stride);
        for (int i = 0; i < 64; ++i) {
            printf("lane %2d:", i);
            for (int j = 0; j < 8; ++j) { int a = h[i * 8 + j] | (h[(64 + i) * 8 + j] << 8); printf(" (%2d,%d)", a / stride, a % stride); }
            printf("\n");
            if (i == 17) { printf("...\n"); i = 31; }
            if (i == 34) break;
        }
    }
    return 0;
}
