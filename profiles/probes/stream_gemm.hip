// prototype: barrier-free, LDS-free streaming GEMM for 1x1 convolutions with small K (fp32): y[M][N] = x[M][K] . W[N][K]^T
// one wave = one 32-channel block, walks 32-pixel blocks; W fragments live in registers; x fragments straight from global memory
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int K, int NWV, int D>
__global__ __launch_bounds__(NWV * 64, 1) void stream_gemm(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int M, int N, int nblk, int mode) {
    constexpr int Q = K / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int cb = wave;                                       // channel block of this wave (N = NWV * 32 in this prototype)
    f32x4 wf[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) wf[q] = *reinterpret_cast<const f32x4*>(w + (size_t)(cb * 32 + r) * K + 8 * q + 4 * h);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((size_t)M * N * 4), 0x00020000);
    // out-of-range pixel blocks: the buffer descriptor returns zeros / drops the stores -> no branches in the loop body
    auto load = [&](f32x4 (&xf)[Q], int pb) {
        const unsigned off = (mode & 2) ? 0x80000000u : ((unsigned)pb * 32u + r) * (unsigned)(K * 4) + 16u * h;
#pragma unroll
        for (int q = 0; q < Q; ++q) xf[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(off + 32u * q), 0, 0));
    };
    auto tile = [&](const f32x4 (&xf)[Q], int pb) {
        f32x16 acc = {0};
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q][e], xf[q][e], acc, 0, 0, 0);
        const unsigned off = ((pb < nblk && !(mode & 1)) ? ((unsigned)pb * 32u + r) * (unsigned)(N * 4) : 0x80000000u) + (unsigned)(cb * 128 + 16 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]}), rsY, (int)(off + 32u * g), 0, 0);
    };
    // prefetch distance D tiles: D + 1 register sets in rotation
    f32x4 xs[D + 1][Q];
    int pb = blockIdx.x;
    const int G = gridDim.x;
    const int niter = (nblk - pb + G - 1) / G;                 // uniform per workgroup
#pragma unroll
    for (int d = 0; d < D; ++d) load(xs[d], min(pb + d * G, nblk));
    for (int it = 0; it < niter; it += D + 1) {
#pragma unroll
        for (int u = 0; u <= D; ++u) {
            load(xs[(u + D) % (D + 1)], min(pb + (u + D) * G, nblk));
            __builtin_amdgcn_sched_barrier(0);
            tile(xs[u], pb + u * G);
            __builtin_amdgcn_sched_barrier(0);
        }
        pb += (D + 1) * G;
    }
}

int main(int argc, char** argv) {
    const int M = 256 * 56 * 56, K = 64, N = 256;
    std::vector<float> hx((size_t)4096 * K), hw((size_t)N * K);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
    float *x, *w, *y;
    CK(hipMalloc(&x, (size_t)M * K * 4)); CK(hipMalloc(&w, (size_t)N * K * 4)); CK(hipMalloc(&y, (size_t)M * N * 4));
    CK(hipMemset(x, 0, (size_t)M * K * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x + (size_t)(M - 4096) * K, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nblk = M / 32;
    for (int depth = 1; depth <= 4; ++depth) {
        const int mode = 0;
        const int grid = depth >= 3 ? 256 : 512;
        float ms = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (depth == 1) hipLaunchKernelGGL((stream_gemm<64, 8, 1>), dim3(grid), dim3(512), 0, 0, x, w, y, M, N, nblk, mode);
            if (depth == 2) hipLaunchKernelGGL((stream_gemm<64, 8, 2>), dim3(grid), dim3(512), 0, 0, x, w, y, M, N, nblk, mode);
            if (depth == 3) hipLaunchKernelGGL((stream_gemm<64, 8, 3>), dim3(grid), dim3(512), 0, 0, x, w, y, M, N, nblk, mode);
            if (depth == 4) hipLaunchKernelGGL((stream_gemm<64, 8, 4>), dim3(grid), dim3(512), 0, 0, x, w, y, M, N, nblk, mode);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep >= 2) ms += t / 4;
        }
        CK(hipGetLastError());
        printf("prefetch depth %d tiles, grid %d: %.1f us  %.1f TFLOP/s  %.0f GB/s\n", depth, grid, ms * 1e3, 2.0 * M * K * N / (ms * 1e-3) / 1e12, ((double)M * (K + N) * 4) / (ms * 1e-3) / 1e9);
    }
    // correctness on the first and last 4096 rows
    std::vector<float> hy((size_t)4096 * N);
    double worst = 0;
    for (int part = 0; part < 2; ++part) {
        CK(hipMemcpy(hy.data(), y + (part ? (size_t)(M - 4096) * N : 0), hy.size() * 4, hipMemcpyDeviceToHost));
        for (int m = 0; m < 4096; m += 37)
            for (int n = 0; n < N; n += 5) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hw[(size_t)n * K + k];
                worst = fmax(worst, fabs(s - hy[(size_t)m * N + n]));
            }
    }
    printf("max abs err %.3g\n", worst);
    return 0;
}
