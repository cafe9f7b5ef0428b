// read + write pattern microbenchmark (no MFMA): per 32-pixel block read 32 x 256 B of x (fragment pattern) and write 32 x 1 KB of y
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
// mode bit 0: reads on, bit 1: writes on; bit 2: streaming (coalesced 1 KB per instruction) instead of the GEMM fragment / accumulator patterns
__global__ __launch_bounds__(512) void rw(const float* __restrict__ x, float* __restrict__ y, int M, int K, int N, int nblk, int mode, int wrap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((size_t)M * N * 4), 0x00020000);
    f32x4 s = {0, 0, 0, 0};
    for (int pbi = blockIdx.x; pbi < nblk; pbi += gridDim.x) {
        const int pb = wrap ? pbi % wrap : pbi;
        if (mode & 1) {
            if (mode & 4) {   // streaming: the 8 waves share the block's 8 KB: one 1 KB instruction each
                const unsigned off = (unsigned)pb * 32u * (unsigned)(K * 4) + wave * 1024u + lane * 16u;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)off, 0, 0));
                s += v;
            } else if (mode & 32) {   // staging pattern: the block's 8 KB once per workgroup, 8 lanes x 16 B = one 128-byte line per row piece
                const unsigned off = ((unsigned)pb * 32u + wave * 4 + (lane >> 4)) * (unsigned)(K * 4) + 16u * (lane & 15);
                s += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)off, 0, 0));
            } else {          // every wave reads the whole block in the fragment pattern (as the 8 channel-block waves of the GEMM do)
                const unsigned off = ((unsigned)pb * 32u + r) * (unsigned)(K * 4) + 16u * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) s += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(off + 32u * q), 0, 0));
            }
        }
        if (mode & 2) {
            if (mode & 4) {   // streaming: wave w writes rows 4w .. 4w+3 of the block, 1 KB per instruction
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned off = ((unsigned)pb * 32u + wave * 4 + i) * (unsigned)(N * 4) + lane * 16u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, s), rsY, (int)off, 0, 0);
                }
            } else if (mode & 16) {   // 8 lanes x 16 B = one full 128-byte line per row, 8 rows per instruction, 4 instructions = 32 rows x 128 B
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned off = ((unsigned)pb * 32u + 8 * g + (lane >> 3)) * (unsigned)(N * 4) + (unsigned)(wave * 128 + 16 * (lane & 7));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, s), rsY, (int)off, 0, 0);
                }
            } else if (mode & 8) {    // 4 lanes x 16 B = 64 B per row, 16 rows per instruction
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned off = ((unsigned)pb * 32u + 16 * (g >> 1) + (lane >> 2)) * (unsigned)(N * 4) + (unsigned)(wave * 128 + 64 * (g & 1) + 16 * (lane & 3));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, s), rsY, (int)off, 0, 0);
                }
            } else {
                const unsigned off = ((unsigned)pb * 32u + r) * (unsigned)(N * 4) + (unsigned)(wave * 128 + 16 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, s), rsY, (int)(off + 32u * g), 0, 0);
            }
        }
    }
    if (s[0] == 12345.f) y[0] = s[1];
}
int main() {
    const int M = 256 * 56 * 56, K = 64, N = 256;
    float *x, *y;
    CK(hipMalloc(&x, (size_t)M * K * 4)); CK(hipMalloc(&y, (size_t)M * N * 4));
    CK(hipMemset(x, 0, (size_t)M * K * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nblk = M / 32;
    for (int wrap : {0, 64})
    for (int mode : {35, 43, 51, 7})
        for (int grid : {512}) {
            float ms = 0;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(rw, dim3(grid), dim3(512), 0, 0, x, y, M, K, N, nblk, mode, wrap);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (rep >= 2) ms += t / 4;
            }
            const double bytes = ((mode & 1) ? (double)M * K * 4 : 0) + ((mode & 2) ? (double)M * N * 4 : 0);
            printf("wrap %2d mode %2d (%s%s%s%s%s) grid %4d: %.1f us  %.0f GB/s of HBM-side bytes\n", wrap, mode, mode & 1 ? "R" : "", mode & 2 ? "W" : "", mode & 4 ? " streaming" : "", mode & 32 ? " R:staged-lines" : (mode & 1 && !(mode & 4) ? " R:fragment x8" : ""), mode & 16 ? " W:128B" : (mode & 8 ? " W:64B" : (mode & 2 && !(mode & 4) ? " W:32B" : "")), grid, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
        }
    return 0;
}
