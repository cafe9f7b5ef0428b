// write-pattern microbenchmark: [M][256] fp32 output written by 64x64 tiles in the MFMA accumulator layout vs wider contiguous pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
// MODE 0: accumulator layout of v_mfma_f32_32x32x2 with (weights, activations) operand order: lane l: pixel l & 31, channels 8g + 4(l >> 5) .. +3, g = 0..3 -> 4 stores of 16 B, 32 B contiguous per pixel per instruction
// MODE 1: 64 B contiguous per pixel per instruction (4 lanes per pixel row, 16 pixels per instruction)
// MODE 2: 256 B contiguous per pixel (16 lanes per tile row, 4 rows per instruction)
// MODE 3: streaming: the workgroup owns 16 full rows (1 KB each): each instruction writes one full row
template <int MODE>
__global__ __launch_bounds__(256) void wk(float* __restrict__ y, int M, int N, int tiles_n, int total, int persistent) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (int t = blockIdx.x; t < total; t += persistent ? gridDim.x : total) {
        if (MODE == 3) {
            const long row0 = (long)t * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long r = row0 + wave * 4 + i;
                if (r < M) *reinterpret_cast<f32x4*>(y + r * N + lane * 4) = v;
            }
            continue;
        }
        const int tm = t / tiles_n, tn = t % tiles_n;
        const int wm = wave >> 1, wn = wave & 1;
        float* base = y + ((long)tm * 64 + wm * 32) * N + tn * 64 + wn * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int r, c;
            if (MODE == 0) { r = lane & 31; c = 8 * g + 4 * (lane >> 5); }
            else if (MODE == 1) { r = (lane >> 2) + 16 * (g >> 1); c = 4 * (lane & 3) + 16 * (g & 1); }
            else { r = (lane >> 3) + 8 * g; c = 4 * (lane & 7); }
            *reinterpret_cast<f32x4*>(base + (long)r * N + c) = v;
        }
    }
}
int main() {
    const int M = 256 * 56 * 56, N = 256;
    float* y;
    CK(hipMalloc(&y, (size_t)M * N * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int tiles_n = N / 64, total = (M / 64) * tiles_n, total3 = M / 16;
    for (int pers = 0; pers < 2; ++pers)
        for (int mode = 0; mode < 4; ++mode) {
            const int tot = mode == 3 ? total3 : total;
            const int grid = pers ? 256 * 5 : tot;
            float ms = 0;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(wk<0>, dim3(grid), dim3(256), 0, 0, y, M, N, tiles_n, tot, pers);
                if (mode == 1) hipLaunchKernelGGL(wk<1>, dim3(grid), dim3(256), 0, 0, y, M, N, tiles_n, tot, pers);
                if (mode == 2) hipLaunchKernelGGL(wk<2>, dim3(grid), dim3(256), 0, 0, y, M, N, tiles_n, tot, pers);
                if (mode == 3) hipLaunchKernelGGL(wk<3>, dim3(grid), dim3(256), 0, 0, y, M, N, tiles_n, tot, pers);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (rep >= 2) ms += t / 4;
            }
            printf("persistent %d mode %d: %.1f us  %.0f GB/s\n", pers, mode, ms * 1e3, (double)M * N * 4 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
