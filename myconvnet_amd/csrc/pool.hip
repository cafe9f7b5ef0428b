// pool.hip — max / average / global-average pooling on NHWC activations (HBM-bound).
// One thread per (output pixel, VEC channels): 16-byte coalesced accesses along C.
#include "common.h"

template <typename T, int VEC>
__device__ __forceinline__ void pld(const T* p, float (&o)[VEC]) {
    if constexpr (VEC == 1) o[0] = to_f32(p[0]);
    else {
        const Chunk<T> c = load_chunk<T>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = c.get(i);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void pst(T* p, const float (&v)[VEC]) {
    if constexpr (VEC == 1) p[0] = from_f32<T>(v[0]);
    else {
        Chunk<T> c;
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.set(i, v[i]);
        store_chunk<T>(p, c);
    }
}

struct PoolParams {
    int N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW;
};
static inline unsigned pool_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// AFF: the input is read through y0 = relu(x * scale[c] + shift[c]) rounded to the storage type — the batch-norm apply pass of the
// layer in front (the ResNet stem: conv -> BN -> ReLU -> 3x3/2 pool) folded into the pool's loads, so that the normalised
// tensor is never written or read.  Rounding before the comparison keeps value, ties and arg-max those of the unfused pair.
// KK / SS: compile-time window / stride (0 = run time).  With the stem's 3x3 / 2 as constants the tap loops unroll and the nine loads of a
// window are issued back to back (run-time bounds made every tap a load -> wait -> compare round trip: 3.2 TB/s on the stem).
template <typename T, int VEC, bool AFF = false, int KK = 0, int SS = 0>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int8_t* __restrict__ arg, PoolParams p,
                                                          const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr) {
    const int KH = KK ? KK : p.KH, KW = KK ? KK : p.KW, SH = SS ? SS : p.SH, SW = SS ? SS : p.SW;
    const unsigned cv = (unsigned)(p.C / VEC);
    const unsigned total = (unsigned)((long)p.N * p.OH * p.OW * cv);       // (host: < 2^32 — 32-bit index arithmetic: four 64-bit divisions per item cost more than the window)
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        unsigned r = idx / cv;
        const int c = (int)(idx - r * cv) * VEC;
        const unsigned r1 = r / (unsigned)p.OW;
        const int ox = (int)(r - r1 * (unsigned)p.OW);
        const int n = (int)(r1 / (unsigned)p.OH), oy = (int)(r1 - (unsigned)n * (unsigned)p.OH);
        float best[VEC];
        int bi[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { best[i] = -INFINITY; bi[i] = 0; }
        float sc[VEC], sh[VEC];
        if constexpr (AFF) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
        }
        if constexpr (KK > 0) {
            // all KK x KK loads first (clamped addresses, no branches), then the affine / compare chain
            float v[KK][KK][VEC];
            bool okk[KK][KK];
#pragma unroll
            for (int kr = 0; kr < KK; ++kr) {
                const int iy = oy * SH + kr - p.padT;
                const bool yok = (unsigned)iy < (unsigned)p.H;
#pragma unroll
                for (int ks = 0; ks < KK; ++ks) {
                    const int ix = ox * SW + ks - p.padL;
                    okk[kr][ks] = yok && (unsigned)ix < (unsigned)p.W;
                    pld<T, VEC>(x + (((long)n * p.H + (okk[kr][ks] ? iy : 0)) * p.W + (okk[kr][ks] ? ix : 0)) * p.C + c, v[kr][ks]);
                }
            }
#pragma unroll
            for (int kr = 0; kr < KK; ++kr)
#pragma unroll
                for (int ks = 0; ks < KK; ++ks) {
                    if constexpr (AFF) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[kr][ks][i] = to_f32(from_f32<T>(fmaxf(fmaf(v[kr][ks][i], sc[i], sh[i]), 0.f)));
                    }
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        if (okk[kr][ks] && v[kr][ks][i] > best[i]) { best[i] = v[kr][ks][i]; bi[i] = kr * KK + ks; }
                }
        } else
        for (int kr = 0; kr < KH; ++kr) {
            const int iy = oy * SH + kr - p.padT;
            const bool yok = (unsigned)iy < (unsigned)p.H;
            for (int ks = 0; ks < KW; ++ks) {
                const int ix = ox * SW + ks - p.padL;
                const bool ok = yok && (unsigned)ix < (unsigned)p.W;     // clamped address + select: no branch per load
                float v[VEC];
                pld<T, VEC>(x + (((long)n * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * p.C + c, v);
                if constexpr (AFF) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) v[i] = to_f32(from_f32<T>(fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f)));
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (ok && v[i] > best[i]) { best[i] = v[i]; bi[i] = kr * KW + ks; }   // strict '>' : first maximum wins
            }
        }
        const long o = (((long)n * p.OH + oy) * p.OW + ox) * p.C + c;
        pst<T, VEC>(y + o, best);
        signed char a[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) a[i] = (signed char)bi[i];
        if constexpr (VEC == 8) *reinterpret_cast<unsigned long long*>(arg + o) = *reinterpret_cast<const unsigned long long*>(a);   // one store, not VEC
        else if constexpr (VEC == 4) *reinterpret_cast<unsigned*>(arg + o) = *reinterpret_cast<const unsigned*>(a);
        else arg[o] = a[0];
    }
}

// KK / SS: compile-time window / stride (0 = run time).  The 3x3 / 2 pool of the ResNet stem gets constants: its divisions and
// modulos by the stride become shifts and the tap loops unroll.
template <typename T, int VEC, int KK = 0, int SS = 0>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const int8_t* __restrict__ arg, T* __restrict__ dx, PoolParams p) {
    const int KH = KK ? KK : p.KH, KW = KK ? KK : p.KW, SH = SS ? SS : p.SH, SW = SS ? SS : p.SW;
    const unsigned cv = (unsigned)(p.C / VEC);
    const unsigned total = (unsigned)((long)p.N * p.H * p.W * cv);          // (host: < 2^32)
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        unsigned r = idx / cv;
        const int c = (int)(idx - r * cv) * VEC;
        const unsigned r2 = r / (unsigned)p.W;
        const int ix = (int)(r - r2 * (unsigned)p.W);
        const int n = (int)(r2 / (unsigned)p.H), iy = (int)(r2 - (unsigned)n * (unsigned)p.H);
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        if constexpr (KK == 3 && SS == 2) {
            // an input pixel is seen by at most 2 x 2 outputs: (oy0, kr0) and (oy0 - 1, kr0 + 2) per axis; clamped loads + masks
            const int ty0 = iy + p.padT, tx0 = ix + p.padL;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int oy = (ty0 >> 1) - a, kr = (ty0 & 1) + 2 * a;
                const bool yok = kr < 3 && (unsigned)oy < (unsigned)p.OH;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ox = (tx0 >> 1) - b, ks = (tx0 & 1) + 2 * b;
                    const bool ok = yok && ks < 3 && (unsigned)ox < (unsigned)p.OW;
                    const long o = (((long)n * p.OH + (ok ? oy : 0)) * p.OW + (ok ? ox : 0)) * p.C + c;
                    float g[VEC];
                    pld<T, VEC>(dy + o, g);
                    signed char ac[VEC];
                    if constexpr (VEC == 8) *reinterpret_cast<unsigned long long*>(ac) = *reinterpret_cast<const unsigned long long*>(arg + o);
                    else if constexpr (VEC == 4) *reinterpret_cast<unsigned*>(ac) = *reinterpret_cast<const unsigned*>(arg + o);
                    else ac[0] = arg[o];
                    const int code = ok ? kr * 3 + ks : -1;
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        if (ac[i] == code) acc[i] += g[i];
                }
            }
            pst<T, VEC>(dx + (((long)n * p.H + iy) * p.W + ix) * p.C + c, acc);
            continue;
        }
        // (no unroll request here: with a run-time window the optimizer cannot honour it — nine -Wpass-failed notes per build — and the
        // instantiations with a constant window unroll by themselves)
        for (int kr = 0; kr < KH; ++kr) {
            const int ty = iy + p.padT - kr;
            if (ty < 0 || ty % SH) continue;
            const int oy = ty / SH;
            if (oy >= p.OH) continue;
            for (int ks = 0; ks < KW; ++ks) {
                const int tx = ix + p.padL - ks;
                if (tx < 0 || tx % SW) continue;
                const int ox = tx / SW;
                if (ox >= p.OW) continue;
                const long o = (((long)n * p.OH + oy) * p.OW + ox) * p.C + c;
                float g[VEC];
                pld<T, VEC>(dy + o, g);
                const int code = kr * KW + ks;
                // the VEC arg-max codes of this pixel chunk in ONE load (byte loads: VEC memory instructions per tap)
                signed char a[VEC];
                if constexpr (VEC == 8) *reinterpret_cast<unsigned long long*>(a) = *reinterpret_cast<const unsigned long long*>(arg + o);
                else if constexpr (VEC == 4) *reinterpret_cast<unsigned*>(a) = *reinterpret_cast<const unsigned*>(arg + o);
                else a[0] = arg[o];
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (a[i] == code) acc[i] += g[i];
            }
        }
        pst<T, VEC>(dx + (((long)n * p.H + iy) * p.W + ix) * p.C + c, acc);
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolParams p) {
    const int cv = p.C / VEC;
    const long total = (long)p.N * p.OH * p.OW * cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * VEC;
        long r = idx / cv;
        const int ox = (int)(r % p.OW);
        r /= p.OW;
        const int oy = (int)(r % p.OH), n = (int)(r / p.OH);
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        int cnt = 0;
        for (int kr = 0; kr < p.KH; ++kr) {
            const int iy = oy * p.SH + kr - p.padT;
            if (iy < 0 || iy >= p.H) continue;
            for (int ks = 0; ks < p.KW; ++ks) {
                const int ix = ox * p.SW + ks - p.padL;
                if (ix < 0 || ix >= p.W) continue;
                float v[VEC];
                pld<T, VEC>(x + (((long)n * p.H + iy) * p.W + ix) * p.C + c, v);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += v[i];
                ++cnt;
            }
        }
        const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] *= inv;
        pst<T, VEC>(y + (((long)n * p.OH + oy) * p.OW + ox) * p.C + c, acc);
    }
}

__device__ __forceinline__ int valid_count(int o, int S, int K, int pad, int L) {
    int lo = o * S - pad, hi = lo + K;
    if (lo < 0) lo = 0;
    if (hi > L) hi = L;
    return hi - lo;
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, PoolParams p) {
    const int cv = p.C / VEC;
    const long total = (long)p.N * p.H * p.W * cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * VEC;
        long r = idx / cv;
        const int ix = (int)(r % p.W);
        r /= p.W;
        const int iy = (int)(r % p.H), n = (int)(r / p.H);
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        for (int kr = 0; kr < p.KH; ++kr) {
            const int ty = iy + p.padT - kr;
            if (ty < 0 || ty % p.SH) continue;
            const int oy = ty / p.SH;
            if (oy >= p.OH) continue;
            for (int ks = 0; ks < p.KW; ++ks) {
                const int tx = ix + p.padL - ks;
                if (tx < 0 || tx % p.SW) continue;
                const int ox = tx / p.SW;
                if (ox >= p.OW) continue;
                const int cnt = valid_count(oy, p.SH, p.KH, p.padT, p.H) * valid_count(ox, p.SW, p.KW, p.padL, p.W);
                float g[VEC];
                pld<T, VEC>(dy + (((long)n * p.OH + oy) * p.OW + ox) * p.C + c, g);
                const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += g[i] * inv;
            }
        }
        pst<T, VEC>(dx + (((long)n * p.H + iy) * p.W + ix) * p.C + c, acc);
    }
}

// global average over HW: [N][HW][C] -> [N][C].  One block per (image, group of TX channel chunks): TY rows of the
// image are summed in parallel and folded through LDS (the squeeze-excite means of EfficientNet reduce up to 112x112
// pixels per image; a thread per output would serialise them).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int HW, int C, int TX, int TY) {
    extern __shared__ float gap_red[];                           // [TY][TX*VEC]
    const int cv = C / VEC;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int chunk = blockIdx.x * TX + tx;
    const long n = blockIdx.y;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (ty < TY && chunk < cv) {
        const T* base = x + n * HW * C + (long)chunk * VEC;
        for (int q = ty; q < HW; q += TY) {
            float v[VEC];
            pld<T, VEC>(base + (long)q * C, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += v[i];
        }
    }
    const int cols = TX * VEC;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) gap_red[ty * cols + tx * VEC + i] = acc[i];
    }
    __syncthreads();
    if (ty == 0 && chunk < cv) {
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float s = 0.f;
            for (int k = 0; k < TY; ++k) s += gap_red[k * cols + tx * VEC + i];
            acc[i] = s * inv;
        }
        pst<T, VEC>(y + n * C + (long)chunk * VEC, acc);
    }
}
template <typename T, int VEC, bool ACC = false>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int HW, int C) {
    const unsigned cv = (unsigned)(C / VEC);
    const unsigned total = (unsigned)((long)N * HW * cv);                   // (host: < 2^32)
    const float inv = 1.f / (float)HW;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned pqu = idx / cv;
        const int c = (int)(idx - pqu * cv) * VEC;
        const long pq = pqu;
        const long n = pqu / (unsigned)HW;
        float g[VEC];
        pld<T, VEC>(dy + n * C + c, g);
        if (ACC) {                                               // dx += : the second gradient contribution of an SE block's input
            float o[VEC];
            pld<T, VEC>(dx + pq * C + c, o);
#pragma unroll
            for (int i = 0; i < VEC; ++i) g[i] = fmaf(g[i], inv, o[i]);       // (explicit fma: mcn_bn_bwd_se composes the same value)
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) g[i] *= inv;
        }
        pst<T, VEC>(dx + pq * C + c, g);
    }
}

// ---- host --------------------------------------------------------------------------------------------------
static int pool_check(const void* a, const void* b, int N, int H, int W, int C, int KH, int KW, int SH, int SW, int padT, int padL, int OH,
                      int OW, const char* name) {
    if (!a || !b || N < 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || SH <= 0 || SW <= 0 || padT < 0 || padL < 0 || OH <= 0 || OW <= 0 ||
        KH * KW > 127)
        MCN_FAIL(MCN_E_BADARG, "%s: bad argument", name);
    if ((long)N * H * W * C >= 0xffffffffl || (long)N * OH * OW * C >= 0xffffffffl) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: more than 2^32 elements (32-bit index arithmetic)", name);
    return MCN_OK;
}
#define POOL_DISPATCH(KERNEL, TOTAL, ...)                                                                        \
    do {                                                                                                         \
        if (dtype == MCN_F32) {                                                                                  \
            if (C % 4 == 0) hipLaunchKernelGGL((KERNEL<float, 4>), dim3(pool_blocks((TOTAL) / 4)), dim3(256), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<float, 1>), dim3(pool_blocks(TOTAL)), dim3(256), 0, st, __VA_ARGS__);  \
        } else if (dtype == MCN_BF16) {                                                                          \
            if (C % 8 == 0) hipLaunchKernelGGL((KERNEL<bf16_t, 8>), dim3(pool_blocks((TOTAL) / 8)), dim3(256), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<bf16_t, 1>), dim3(pool_blocks(TOTAL)), dim3(256), 0, st, __VA_ARGS__); \
        } else if (dtype == MCN_F16) {                                                                           \
            if (C % 8 == 0) hipLaunchKernelGGL((KERNEL<f16_t, 8>), dim3(pool_blocks((TOTAL) / 8)), dim3(256), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<f16_t, 1>), dim3(pool_blocks(TOTAL)), dim3(256), 0, st, __VA_ARGS__); \
        } else MCN_FAIL(MCN_E_UNSUPPORTED, "pool: dtype %d unsupported", (int)dtype);                            \
        MCN_CHECK_LAUNCH();                                                                                      \
    } while (0)

template <typename T> static const T* cp(const void* p) { return (const T*)p; }

extern "C" int mcn_maxpool_fwd(const void* x, void* y, int8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                               int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype, void* stream) {
    int rc = pool_check(x, y, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW, "maxpool_fwd");
    if (rc) return rc;
    if (!argmax) MCN_FAIL(MCN_E_BADARG, "maxpool_fwd: null argmax");
    if (N == 0) return MCN_OK;
    const PoolParams p = {N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW};
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * OH * OW * C;
    if (dtype == MCN_F32) {
        if (C % 4 == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<float, 4>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)x, (float*)y, argmax, p);
        else hipLaunchKernelGGL((maxpool_fwd_kernel<float, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, argmax, p);
    } else if (dtype == MCN_BF16) {
        if (C % 8 == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, p);
        else hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, p);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<f16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, argmax, p);
        else hipLaunchKernelGGL((maxpool_fwd_kernel<f16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, argmax, p);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "maxpool_fwd: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
// max-pool over relu(x * scale + shift) (see maxpool_fwd_kernel<AFF>); scale / shift: fp32 [C] on the device
extern "C" int mcn_maxpool_fwd_affine_relu(const void* x, const float* scale, const float* shift, void* y, int8_t* argmax, int32_t N, int32_t H, int32_t W,
                                           int32_t C, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW,
                                           mcn_dtype dtype, void* stream) {
    int rc = pool_check(x, y, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW, "maxpool_fwd_affine_relu");
    if (rc) return rc;
    if (!argmax || !scale || !shift) MCN_FAIL(MCN_E_BADARG, "maxpool_fwd_affine_relu: null pointer");
    if (N == 0) return MCN_OK;
    const PoolParams p = {N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW};
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * OH * OW * C;
    const bool k3s2 = KH == 3 && KW == 3 && SH == 2 && SW == 2;      // the stem: window and stride as compile-time constants
    if (dtype == MCN_F32) {
        if (C % 4 == 0 && k3s2) hipLaunchKernelGGL((maxpool_fwd_kernel<float, 4, true, 3, 2>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)x, (float*)y, argmax, p, scale, shift);
        else if (C % 4 == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<float, 4, true>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)x, (float*)y, argmax, p, scale, shift);
        else hipLaunchKernelGGL((maxpool_fwd_kernel<float, 1, true>), dim3(pool_blocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, argmax, p, scale, shift);
    } else if (dtype == MCN_BF16) {
        if (C % 8 == 0 && k3s2) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 8, true, 3, 2>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, p, scale, shift);
        else if (C % 8 == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 8, true>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, p, scale, shift);
        else hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 1, true>), dim3(pool_blocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, p, scale, shift);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0 && k3s2) hipLaunchKernelGGL((maxpool_fwd_kernel<f16_t, 8, true, 3, 2>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, argmax, p, scale, shift);
        else if (C % 8 == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<f16_t, 8, true>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, argmax, p, scale, shift);
        else hipLaunchKernelGGL((maxpool_fwd_kernel<f16_t, 1, true>), dim3(pool_blocks(total)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, argmax, p, scale, shift);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "maxpool_fwd_affine_relu: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
// 3x3 / 2 with no leading padding on an even map (the ResNet stem pool: 112 -> 56, pads (0, 1)): one thread per 2x2 block of
// dx and channel chunk.  The block (rows 2a, 2a+1; columns 2b, 2b+1) is seen by exactly the four windows (a - da, b - db),
// da, db in {0, 1}: their dy / arg-max chunks are loaded ONCE and routed to the four pixels (the per-pixel form loads four
// candidate windows per pixel: 8 loads per store instead of 2).  Window (a - da, b - db) covers block row r with filter row
// kr = r + 2 da (valid when < 3), the same for columns.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_3x3s2_block_kernel(const T* __restrict__ dy, const int8_t* __restrict__ arg, T* __restrict__ dx,
                                                                      PoolParams p) {
    const unsigned cv = (unsigned)(p.C / VEC);
    const unsigned total = (unsigned)((long)p.N * p.OH * p.OW * cv);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        unsigned r = idx / cv;
        const int c = (int)(idx - r * cv) * VEC;
        const unsigned r2 = r / (unsigned)p.OW;
        const int b = (int)(r - r2 * (unsigned)p.OW);
        const int n = (int)(r2 / (unsigned)p.OH), a = (int)(r2 - (unsigned)n * (unsigned)p.OH);
        float acc[2][2][VEC];
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[y][x][i] = 0.f;
#pragma unroll
        for (int da = 0; da < 2; ++da)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bool ok = a - da >= 0 && b - db >= 0;
                const long o = (((long)n * p.OH + (ok ? a - da : 0)) * p.OW + (ok ? b - db : 0)) * p.C + c;
                float g[VEC];
                pld<T, VEC>(dy + o, g);
                signed char ac[VEC];
                if constexpr (VEC == 8) *reinterpret_cast<unsigned long long*>(ac) = *reinterpret_cast<const unsigned long long*>(arg + o);
                else *reinterpret_cast<unsigned*>(ac) = *reinterpret_cast<const unsigned*>(arg + o);
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    const int kr = y + 2 * da;
                    if (kr >= 3) continue;
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        const int ks = x + 2 * db;
                        if (ks >= 3) continue;
                        const int code = ok ? kr * 3 + ks : -1;
#pragma unroll
                        for (int i = 0; i < VEC; ++i)
                            if (ac[i] == code) acc[y][x][i] += g[i];
                    }
                }
            }
        T* o0 = dx + (((long)n * p.H + 2 * a) * p.W + 2 * b) * p.C + c;
        pst<T, VEC>(o0, acc[0][0]);
        pst<T, VEC>(o0 + p.C, acc[0][1]);
        pst<T, VEC>(o0 + (long)p.W * p.C, acc[1][0]);
        pst<T, VEC>(o0 + (long)p.W * p.C + p.C, acc[1][1]);
    }
}

extern "C" int mcn_maxpool_bwd(const void* dy, const int8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH,
                               int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype,
                               void* stream) {
    int rc = pool_check(dy, dx, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW, "maxpool_bwd");
    if (rc) return rc;
    if (!argmax) MCN_FAIL(MCN_E_BADARG, "maxpool_bwd: null argmax");
    if (N == 0) return MCN_OK;
    const PoolParams p = {N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW};
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * H * W * C;
    // every dx pixel lies in exactly one 2x2 block and every window is one of the four a block sees
    const bool blockable = KH == 3 && KW == 3 && SH == 2 && SW == 2 && padT == 0 && padL == 0 && H % 2 == 0 && W % 2 == 0 && OH == H / 2 && OW == W / 2;
    if (dtype == MCN_F32) {
        if (total >= 0xffffffffL) MCN_FAIL(MCN_E_UNSUPPORTED, "maxpool_bwd: more than 2^32 elements");
        if (C % 4 == 0 && blockable) hipLaunchKernelGGL((maxpool_bwd_3x3s2_block_kernel<float, 4>), dim3(pool_blocks(total / 16)), dim3(256), 0, st, (const float*)dy, argmax, (float*)dx, p);
        else if (C % 4 == 0 && KH == 3 && KW == 3 && SH == 2 && SW == 2) hipLaunchKernelGGL((maxpool_bwd_kernel<float, 4, 3, 2>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)dy, argmax, (float*)dx, p);
        else if (C % 4 == 0) hipLaunchKernelGGL((maxpool_bwd_kernel<float, 4>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)dy, argmax, (float*)dx, p);
        else hipLaunchKernelGGL((maxpool_bwd_kernel<float, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const float*)dy, argmax, (float*)dx, p);
    } else if (dtype == MCN_BF16) {
        if (total >= 0xffffffffL) MCN_FAIL(MCN_E_UNSUPPORTED, "maxpool_bwd: more than 2^32 elements");
        if (C % 8 == 0 && blockable) hipLaunchKernelGGL((maxpool_bwd_3x3s2_block_kernel<bf16_t, 8>), dim3(pool_blocks(total / 32)), dim3(256), 0, st, (const bf16_t*)dy, argmax, (bf16_t*)dx, p);
        else if (C % 8 == 0 && KH == 3 && KW == 3 && SH == 2 && SW == 2) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 8, 3, 2>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)dy, argmax, (bf16_t*)dx, p);
        else if (C % 8 == 0) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)dy, argmax, (bf16_t*)dx, p);
        else hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const bf16_t*)dy, argmax, (bf16_t*)dx, p);
    } else if (dtype == MCN_F16) {
        if (total >= 0xffffffffL) MCN_FAIL(MCN_E_UNSUPPORTED, "maxpool_bwd: more than 2^32 elements");
        if (C % 8 == 0 && blockable) hipLaunchKernelGGL((maxpool_bwd_3x3s2_block_kernel<f16_t, 8>), dim3(pool_blocks(total / 32)), dim3(256), 0, st, (const f16_t*)dy, argmax, (f16_t*)dx, p);
        else if (C % 8 == 0 && KH == 3 && KW == 3 && SH == 2 && SW == 2) hipLaunchKernelGGL((maxpool_bwd_kernel<f16_t, 8, 3, 2>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)dy, argmax, (f16_t*)dx, p);
        else if (C % 8 == 0) hipLaunchKernelGGL((maxpool_bwd_kernel<f16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)dy, argmax, (f16_t*)dx, p);
        else hipLaunchKernelGGL((maxpool_bwd_kernel<f16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const f16_t*)dy, argmax, (f16_t*)dx, p);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "maxpool_bwd: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_avgpool_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW, int32_t SH,
                               int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype, void* stream) {
    int rc = pool_check(x, y, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW, "avgpool_fwd");
    if (rc) return rc;
    if (N == 0) return MCN_OK;
    const PoolParams p = {N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW};
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * OH * OW * C;
    if (dtype == MCN_F32) {
        if (C % 4 == 0) hipLaunchKernelGGL((avgpool_fwd_kernel<float, 4>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)x, (float*)y, p);
        else hipLaunchKernelGGL((avgpool_fwd_kernel<float, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, p);
    } else if (dtype == MCN_BF16) {
        if (C % 8 == 0) hipLaunchKernelGGL((avgpool_fwd_kernel<bf16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, p);
        else hipLaunchKernelGGL((avgpool_fwd_kernel<bf16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, p);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) hipLaunchKernelGGL((avgpool_fwd_kernel<f16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, p);
        else hipLaunchKernelGGL((avgpool_fwd_kernel<f16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, p);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "avgpool_fwd: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW, int32_t SH,
                               int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype, void* stream) {
    int rc = pool_check(dy, dx, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW, "avgpool_bwd");
    if (rc) return rc;
    if (N == 0) return MCN_OK;
    const PoolParams p = {N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW};
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * H * W * C;
    if (dtype == MCN_F32) {
        if (C % 4 == 0) hipLaunchKernelGGL((avgpool_bwd_kernel<float, 4>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)dy, (float*)dx, p);
        else hipLaunchKernelGGL((avgpool_bwd_kernel<float, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const float*)dy, (float*)dx, p);
    } else if (dtype == MCN_BF16) {
        if (C % 8 == 0) hipLaunchKernelGGL((avgpool_bwd_kernel<bf16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, p);
        else hipLaunchKernelGGL((avgpool_bwd_kernel<bf16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, p);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) hipLaunchKernelGGL((avgpool_bwd_kernel<f16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, p);
        else hipLaunchKernelGGL((avgpool_bwd_kernel<f16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, p);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "avgpool_bwd: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_global_avgpool_fwd(const void* x, void* y, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void* stream) {
    if (!x || !y || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "global_avgpool_fwd: bad argument");
    if (N == 0) return MCN_OK;
    hipStream_t st = (hipStream_t)stream;
#define GAP_FWD(T, VEC)                                                                                                        \
    do {                                                                                                                       \
        const int cv = C / VEC;                                                                                                \
        int TX = cv < 32 ? cv : 32;                                                                                            \
        int TY = 256 / TX;                                                                                                     \
        if (TY > HW) TY = HW;                                                                                                  \
        hipLaunchKernelGGL((gap_fwd_kernel<T, VEC>), dim3((unsigned)((cv + TX - 1) / TX), (unsigned)N), dim3(256),           \
                           (size_t)TY * TX * VEC * sizeof(float), st, (const T*)x, (T*)y, N, HW, C, TX, TY);                   \
    } while (0)
    if (dtype == MCN_F32) {
        if (C % 4 == 0) GAP_FWD(float, 4); else GAP_FWD(float, 1);
    } else if (dtype == MCN_BF16) {
        if (C % 8 == 0) GAP_FWD(bf16_t, 8); else GAP_FWD(bf16_t, 1);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) GAP_FWD(f16_t, 8); else GAP_FWD(f16_t, 1);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "global_avgpool_fwd: dtype %d unsupported", (int)dtype);
#undef GAP_FWD
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_global_avgpool_bwd_acc(const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void* stream) {
    if (!dy || !dx || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "global_avgpool_bwd_acc: bad argument");
    if (N == 0) return MCN_OK;
    if ((long)N * HW * C >= 0xffffffffl) MCN_FAIL(MCN_E_UNSUPPORTED, "global_avgpool_bwd_acc: more than 2^32 elements (32-bit index arithmetic)");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * HW * C;
    if (dtype == MCN_F32) {
        if (C % 4 == 0) hipLaunchKernelGGL((gap_bwd_kernel<float, 4, true>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)dy, (float*)dx, N, HW, C);
        else hipLaunchKernelGGL((gap_bwd_kernel<float, 1, true>), dim3(pool_blocks(total)), dim3(256), 0, st, (const float*)dy, (float*)dx, N, HW, C);
    } else if (dtype == MCN_BF16) {
        if (C % 8 == 0) hipLaunchKernelGGL((gap_bwd_kernel<bf16_t, 8, true>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, HW, C);
        else hipLaunchKernelGGL((gap_bwd_kernel<bf16_t, 1, true>), dim3(pool_blocks(total)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, HW, C);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) hipLaunchKernelGGL((gap_bwd_kernel<f16_t, 8, true>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, N, HW, C);
        else hipLaunchKernelGGL((gap_bwd_kernel<f16_t, 1, true>), dim3(pool_blocks(total)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, N, HW, C);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "global_avgpool_bwd_acc: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_global_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void* stream) {
    if (!dy || !dx || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "global_avgpool_bwd: bad argument");
    if (N == 0) return MCN_OK;
    if ((long)N * HW * C >= 0xffffffffl) MCN_FAIL(MCN_E_UNSUPPORTED, "global_avgpool_bwd: more than 2^32 elements (32-bit index arithmetic)");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * HW * C;
    if (dtype == MCN_F32) {
        if (C % 4 == 0) hipLaunchKernelGGL((gap_bwd_kernel<float, 4>), dim3(pool_blocks(total / 4)), dim3(256), 0, st, (const float*)dy, (float*)dx, N, HW, C);
        else hipLaunchKernelGGL((gap_bwd_kernel<float, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const float*)dy, (float*)dx, N, HW, C);
    } else if (dtype == MCN_BF16) {
        if (C % 8 == 0) hipLaunchKernelGGL((gap_bwd_kernel<bf16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, HW, C);
        else hipLaunchKernelGGL((gap_bwd_kernel<bf16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, HW, C);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) hipLaunchKernelGGL((gap_bwd_kernel<f16_t, 8>), dim3(pool_blocks(total / 8)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, N, HW, C);
        else hipLaunchKernelGGL((gap_bwd_kernel<f16_t, 1>), dim3(pool_blocks(total)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, N, HW, C);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "global_avgpool_bwd: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
