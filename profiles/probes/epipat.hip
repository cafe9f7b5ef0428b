// the epilogue streams of the fp32 dgrad + masked fan-in + BN-sums launch (NT_EPI_ACCRED) at 56 x 56, 256 channels, without arithmetic: per 32 x 32 block
// two tensors are read and one is written in the output's layout — in the accumulator layout (32 bytes per pixel and instruction) or as whole 128-byte lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
template <int LINE, int NREAD>
__global__ __launch_bounds__(256) void ep(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t bytes = (size_t)M * N * 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)bytes, 0x00020000);
    const int ntn = N / 64, total = (M / 64) * ntn;
    // XCD-aware tile order as the product
    const int tt = blockIdx.x, q = total >> 3, r8 = total & 7, xcd = tt & 7, idx = tt >> 3;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int tm = t / ntn, tn = t % ntn, wm = wave >> 1, wn = wave & 1;
    f32x4 s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned off;
        if (LINE) off = (((unsigned)tm * 64 + wm * 32 + 8 * g + (lane >> 3)) * N + tn * 64 + wn * 32 + 4 * (lane & 7)) * 4;
        else off = (((unsigned)tm * 64 + wm * 32 + (lane & 31)) * N + tn * 64 + wn * 32 + 8 * g + 4 * (lane >> 5)) * 4;
        s[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
        if (NREAD > 1) s[g] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)off, 0, 0));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned off;
        if (LINE) off = (((unsigned)tm * 64 + wm * 32 + 8 * g + (lane >> 3)) * N + tn * 64 + wn * 32 + 4 * (lane & 7)) * 4;
        else off = (((unsigned)tm * 64 + wm * 32 + (lane & 31)) * N + tn * 64 + wn * 32 + 8 * g + 4 * (lane >> 5)) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, s[g]), rsY, (int)off, 0, 0);
    }
}
template <int LINE, int NREAD>
static void run(const float* a, const float* b, float* y, int M, int N) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int total = (M / 64) * (N / 64);
    float ms = 0;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((ep<LINE, NREAD>), dim3(total), dim3(256), 0, 0, a, b, y, M, N);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 2) ms += t / 4;
    }
    const double bytes = (double)M * N * 4 * (NREAD + 1);
    printf("%d read stream(s) + 1 write stream, %s: %.1f us  %.0f GB/s\n", NREAD, LINE ? "128-byte lines " : "32-byte pieces ", ms * 1e3, bytes / (ms * 1e-3) / 1e9);
}
int main() {
    const int M = 256 * 56 * 56, N = 256;
    float *a, *b, *y;
    CK(hipMalloc(&a, (size_t)M * N * 4)); CK(hipMalloc(&b, (size_t)M * N * 4)); CK(hipMalloc(&y, (size_t)M * N * 4));
    CK(hipMemset(a, 0, (size_t)M * N * 4)); CK(hipMemset(b, 0, (size_t)M * N * 4));
    run<0, 1>(a, b, y, M, N); run<1, 1>(a, b, y, M, N);
    run<0, 2>(a, b, y, M, N); run<1, 2>(a, b, y, M, N);
    return 0;
}
