// fp32 products on the bf16 matrix pipe: numerics + rate probe (round 5).
//
//   a = a1 + a2 + a3 exactly, each piece a bf16 (8 significant bits: 24 = 3 x 8); a_i * b_j is exact in fp32 (16-bit significand);
//   a * b = sum of the 9 piece products, accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (smallest terms first).
//
// One wave computes C[32][32] = A[32][K] * B[32][K]^T three ways — v_mfma_f32_32x32x2_f32 (what the fp32 kernels run today), 9 piece
// products, 6 piece products (a2*b3, a3*b2, a3*b3 dropped) — and the host compares each with the float64 product:
//   err = |c - c64| / sum_k |a_k * b_k|     (the quantity a forward error bound of a length-K dot product is stated in)
// Second part: the rate of the three inner loops with fragments re-read from LDS (one 64 x 64 wave tile per wave, operands split per wave).
//
//   hipcc --offload-arch=gfx950 -O3 -o emu_probe emu_probe.hip && ./emu_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Pieces { u32x4 p[3]; };        // three bf16x8 as raw dwords (element 2i in the low half of dword i)

// 8 floats (two 16-byte fragments) -> hi / mid / lo bf16 pieces; 4.5 vector instructions per element
__device__ __forceinline__ Pieces split8(const f32x4& a, const f32x4& b) {
    Pieces r;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f32x2 v = h < 2 ? f32x2{a[2 * h], a[2 * h + 1]} : f32x2{b[2 * h - 4], b[2 * h - 3]};
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        const f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
        const f32x2 r1 = v - hf;
        const unsigned mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
        const f32x2 mf = {__builtin_bit_cast(float, mid << 16), __builtin_bit_cast(float, mid & 0xffff0000u)};
        const f32x2 r2 = r1 - mf;
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
        r.p[0][h] = hi;
        r.p[1][h] = mid;
        r.p[2][h] = lo;
    }
    return r;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int TERMS>
__device__ __forceinline__ void emu_mma(f32x16& acc, const Pieces& a, const Pieces& b) {
    // smallest products first
    if (TERMS == 9) {
        acc = mfma_bf16(a.p[2], b.p[2], acc);
        acc = mfma_bf16(a.p[1], b.p[2], acc);
        acc = mfma_bf16(a.p[2], b.p[1], acc);
    }
    acc = mfma_bf16(a.p[0], b.p[2], acc);
    acc = mfma_bf16(a.p[2], b.p[0], acc);
    acc = mfma_bf16(a.p[1], b.p[1], acc);
    acc = mfma_bf16(a.p[0], b.p[1], acc);
    acc = mfma_bf16(a.p[1], b.p[0], acc);
    acc = mfma_bf16(a.p[0], b.p[0], acc);
}

// ---- numerics: one wave, A / B row-major [32][K]; lane (row = lane & 31, half = lane >> 5) owns k = 8 * (2 g + half) + 0..7 of group g (16 k)
template <int MODE>
__global__ void dot_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const float* pa = A + row * K + k0 + half * 8;
        const float* pb = B + row * K + k0 + half * 8;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa), a1 = *reinterpret_cast<const f32x4*>(pa + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(pb), b1 = *reinterpret_cast<const f32x4*>(pb + 4);
        if (MODE == 0) {
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc, 0, 0, 0);
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc, 0, 0, 0);
        } else {
            const Pieces pA = split8(a0, a1), pB = split8(b0, b1);
            emu_mma<MODE>(acc, pA, pB);
        }
    }
    // accumulator layout of the 32x32 MFMAs: column n = lane & 31 (the B row), rows m = 8 * q + 4 * half + e  (register 4 q + e)
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 4; ++e) C[(8 * q + 4 * half + e) * 32 + row] = acc[4 * q + e];
}

// ---- rate: 256 workgroups x 4 waves, each wave a 64 x 64 tile from an LDS image (fragments re-read every group, never re-staged)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* __restrict__ out, int groups) {
    __shared__ __attribute__((aligned(16))) float sA[128 * 32], sB[128 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 31, half = lane >> 5;
    for (int i = tid; i < 128 * 32; i += 256) {
        sA[i] = 1.f + 1e-3f * (float)((i * 37) % 101);
        sB[i] = 1.f - 1e-3f * (float)((i * 53) % 89);
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    for (int g = 0; g < groups; ++g) {
        const int kc = (g & 1) * 16 + half * 8;                 // two groups of 16 k per 32-float row
        f32x4 a[2][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* pa = sA + (wm * 64 + i * 32 + row) * 32 + kc;
            const float* pb = sB + (wn * 64 + i * 32 + row) * 32 + kc;
            a[i][0] = *reinterpret_cast<const f32x4*>(pa);
            a[i][1] = *reinterpret_cast<const f32x4*>(pa + 4);
            b[i][0] = *reinterpret_cast<const f32x4*>(pb);
            b[i][1] = *reinterpret_cast<const f32x4*>(pb + 4);
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s][e], b[j][s][e], acc[j][i], 0, 0, 0);
        } else {
            Pieces pA[2], pB[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pA[i] = split8(a[i][0], a[i][1]);
                pB[i] = split8(b[i][0], b[i][1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) emu_mma<MODE>(acc[j][i], pA[i], pB[j]);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 16; ++e) s += acc[j][i][e];
    out[blockIdx.x * 256 + tid] = s;
}

// the same loop, software-pipelined inside the wave: the fragments of group g + 1 are read and split in the issue slots between the MFMAs of group g
// (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles: 4 single-issue vector instructions per gap are nearly free)
template <int MODE>
__global__ __launch_bounds__(256, 2) void rate_kernel_pipe(float* __restrict__ out, int groups) {
    __shared__ __attribute__((aligned(16))) float sA[128 * 32], sB[128 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 31, half = lane >> 5;
    for (int i = tid; i < 128 * 32; i += 256) {
        sA[i] = 1.f + 1e-3f * (float)((i * 37) % 101);
        sB[i] = 1.f - 1e-3f * (float)((i * 53) % 89);
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    Pieces pA[2][2], pB[2][2];
    auto fetch_split = [&](int g, int set) {
        const int kc = (g & 1) * 16 + half * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* pa = sA + (wm * 64 + i * 32 + row) * 32 + kc;
            const float* pb = sB + (wn * 64 + i * 32 + row) * 32 + kc;
            pA[set][i] = split8(*reinterpret_cast<const f32x4*>(pa), *reinterpret_cast<const f32x4*>(pa + 4));
            pB[set][i] = split8(*reinterpret_cast<const f32x4*>(pb), *reinterpret_cast<const f32x4*>(pb + 4));
        }
    };
    fetch_split(0, 0);
    for (int g = 0; g < groups; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            fetch_split(g + u + 1, (u + 1) & 1);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) emu_mma<MODE>(acc[j][i], pA[u & 1][i], pB[u & 1][j]);
            // 4 * MODE MFMAs, each followed by 4 vector instructions and (first eight gaps) one LDS read
#pragma unroll
            for (int m = 0; m < 4 * MODE; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (m < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, MODE == 9 ? 4 : 6, 0);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 16; ++e) s += acc[j][i][e];
    out[blockIdx.x * 256 + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double frand() { return (double)rand() / RAND_MAX; }
int main() {
    srand(1234);
    const int Ks[3] = {64, 512, 4608};
    const char* names[3] = {"fp32 mfma 32x32x2", "bf16 x 9", "bf16 x 6"};
    for (int dist = 0; dist < 3; ++dist) {
        for (int K : Ks) {
            std::vector<float> A(32 * K), B(32 * K), C(32 * 32);
            for (int i = 0; i < 32 * K; ++i) {
                // 0: uniform (-1, 1); 1: activations after ReLU x weights (half zeros, positive); 2: wide dynamic range (2^-20 .. 2^20)
                double a = 2 * frand() - 1, b = 2 * frand() - 1;
                if (dist == 1) { a = a > 0 ? a * 3 : 0; b *= 0.05; }
                if (dist == 2) { a *= std::ldexp(1.0, (int)(40 * frand()) - 20); b *= std::ldexp(1.0, (int)(40 * frand()) - 20); }
                A[i] = (float)a;
                B[i] = (float)b;
            }
            float *dA, *dB, *dC;
            CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
            CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            for (int mode = 0; mode < 3; ++mode) {
                if (mode == 0) dot_kernel<0><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 1) dot_kernel<9><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 2) dot_kernel<6><<<1, 64>>>(dA, dB, dC, K);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
                double emax = 0, esum = 0;
                for (int m = 0; m < 32; ++m)
                    for (int n = 0; n < 32; ++n) {
                        double ref = 0, mag = 0;
                        for (int k = 0; k < K; ++k) {
                            const double t = (double)A[m * K + k] * (double)B[n * K + k];
                            ref += t;
                            mag += std::fabs(t);
                        }
                        const double e = std::fabs((double)C[m * 32 + n] - ref) / (mag > 0 ? mag : 1);
                        emax = e > emax ? e : emax;
                        esum += e;
                    }
                printf("dist %d K %4d  %-18s  max err %.3e  mean err %.3e   (2^-24 = 5.96e-08)\n", dist, K, names[mode], emax, esum / 1024);
            }
            CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
        }
    }
    // rate
    float* out;
    CK(hipMalloc(&out, 2048 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int groups = 4096;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) rate_kernel<0><<<1024, 256>>>(out, groups);
            if (mode == 1) rate_kernel<9><<<1024, 256>>>(out, groups);
            if (mode == 2) rate_kernel<6><<<1024, 256>>>(out, groups);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = 2.0 * 1024 * 128 * 128 * 16.0 * groups;
            if (rep) printf("rate  %-18s  %.3f ms  %.1f TFLOP/s (fp32-equivalent; 1024 workgroups of 128 x 128, %d groups of 16 k)\n", names[mode], ms, flop / ms * 1e-9, groups);
        }
    }
    for (int mode = 1; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 1) rate_kernel_pipe<9><<<1024, 256>>>(out, groups);
            if (mode == 2) rate_kernel_pipe<6><<<1024, 256>>>(out, groups);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = 2.0 * 1024 * 128 * 128 * 16.0 * groups;
            if (rep) printf("rate  %-18s  %.3f ms  %.1f TFLOP/s (split of group g + 1 in the MFMA gaps of group g, two workgroups per CU)\n", names[mode], ms, flop / ms * 1e-9);
        }
    }
    return 0;
}
