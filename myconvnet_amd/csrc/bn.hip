// bn.hip — batch normalisation (tf.nn.fused_batch_norm semantics) on [M][C] NHWC activations.
//
// HBM-bound.  Every kernel uses one column layout: a thread owns one 16-byte channel chunk
// (8 bf16 / 4 fp32 channels) and walks rows, so loads are coalesced along C, per-channel
// parameters live in registers, and reductions over M are per-thread sums (shifted by a
// per-channel pivot x[0][c] to avoid E[x^2]-E[x]^2 cancellation) folded across the block through
// LDS, then across blocks by a deterministic second-stage kernel in double precision.
//   fwd : stats (1 read of x) -> finalize -> apply (1 read of x [+skip], 1 write of y), ReLU and
//         the residual add fused into apply.
//   bwd : reduce (reads dy, x [, y]) -> finalize -> apply (reads dy, x [, y]; writes dx [, dskip]).
#include "common.h"

template <typename T, int VEC>
__device__ __forceinline__ void ldv(const T* p, float (&o)[VEC]) {
    if constexpr (VEC == 1) {
        o[0] = to_f32(p[0]);
    } else {
        const Chunk<T> c = load_chunk<T>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = c.get(i);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void stv(T* p, const float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        p[0] = from_f32<T>(v[0]);
    } else {
        Chunk<T> c;
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.set(i, v[i]);
        store_chunk<T>(p, c);
    }
}

struct ColLayout {
    int cols;   // number of VEC-wide columns = C / VEC
    int TX;     // columns per block
    int TY;     // row lanes per block
    int gx;     // column blocks
    int gy;     // row blocks
    long rpb;   // rows per row-block
};
static ColLayout make_layout(long M, int C, int vec, int target_blocks) {
    ColLayout L;
    L.cols = C / vec;
    L.TX = L.cols < 256 ? L.cols : 256;
    L.TY = 256 / L.TX;
    L.gx = (L.cols + L.TX - 1) / L.TX;
    long gy = target_blocks / L.gx;
    const long maxgy = (M + (long)L.TY * 8 - 1) / ((long)L.TY * 8);   // at least ~8 rows per thread
    if (gy > maxgy) gy = maxgy;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    L.rpb = (M + gy - 1) / gy;
    L.rpb = (L.rpb + L.TY - 1) / L.TY * L.TY;
    if (L.rpb < 1) L.rpb = L.TY;
    L.gy = (int)((M + L.rpb - 1) / L.rpb);
    if (L.gy < 1) L.gy = 1;
    return L;
}

// VEC per-channel fp32 parameters starting at p (16-byte aligned when VEC % 4 == 0: one or two 16-byte loads, not VEC dwords);
// dflt when p is null
template <int VEC>
__device__ __forceinline__ void ldc(const float* __restrict__ p, float (&o)[VEC], float dflt = 0.f) {
    if (!p) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = dflt;
        return;
    }
    if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
            o[4 * q] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = p[i];
    }
}

// ---- forward statistics --------------------------------------------------------------------------
// part[(rb*2 + {0,1})*C + c] = sum over the block's rows of (x - pivot), (x - pivot)^2
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, long M, int C, int TX, int TY,
                                                       long rpb) {
    extern __shared__ float red[];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    const bool active = ty < TY && col * VEC < C;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float piv[VEC];
        ldv<T, VEC>(x + (long)col * VEC, piv);
        const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
        constexpr int U = 4;                  // rows per trip, loads first (one 16-byte load in flight per thread is latency bound)
        long r = r0 + ty;
        for (; r + (long)(U - 1) * TY < r1; r += (long)U * TY) {
            float v[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) ldv<T, VEC>(x + (r + (long)u * TY) * C + (long)col * VEC, v[u]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float d = v[u][i] - piv[i];
                    s1[i] += d;
                    s2[i] = fmaf(d, d, s2[i]);
                }
        }
        for (; r < r1; r += TY) {
            float v[VEC];
            ldv<T, VEC>(x + r * C + (long)col * VEC, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float d = v[i] - piv[i];
                s1[i] += d;
                s2[i] = fmaf(d, d, s2[i]);
            }
        }
    }
    // fold the TY row lanes
    float* r1p = red;                       // [TY][TX*VEC]
    float* r2p = red + 256 * VEC;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            r1p[(ty * TX + tx) * VEC + i] = s1[i];
            r2p[(ty * TX + tx) * VEC + i] = s2[i];
        }
    }
    __syncthreads();
    if (ty == 0 && col * VEC < C) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < TY; ++k) {
                a += r1p[(k * TX + tx) * VEC + i];
                b += r2p[(k * TX + tx) * VEC + i];
            }
            part[((long)blockIdx.y * 2 + 0) * C + col * VEC + i] = a;
            part[((long)blockIdx.y * 2 + 1) * C + col * VEC + i] = b;
        }
    }
}

// Second reduction stage shared by the forward / backward finalize kernels: a block owns FIN_CH channels and
// FIN_LANES threads per channel stride over the row-block partials (coalesced FIN_CH-float segments, several
// independent loads in flight per thread), then fold through LDS in double precision.  (A one-thread-per-channel
// loop over ~2000 partials is a serial chain of dependent L2 round trips: it cost 0.45 ms per layer.)
#define FIN_CH 8
#define FIN_LANES 64
// fold of per-thread (a, b) over the FIN_LANES lanes of a channel: thread = lane * FIN_CH + channel, so a wave holds 8 lanes x 8 channels —
// three xor-shuffles (lane bits), then the waves' sums through LDS; fixed order, result in the threads of lane 0 (threadIdx.x < FIN_CH).
// (round 3: these kernels are pure latency chains — 7-12 us per launch, ~160 launches per step; the serial LDS walk of lane 0 over every
// lane's pair and the 4-deep load batches were most of it)
__device__ __forceinline__ void fin_fold(double* sh, double& a, double& b) {
#pragma unroll
    for (int o = FIN_CH; o < 64; o <<= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
    constexpr int NWAVES = FIN_CH * FIN_LANES / 64;
    if (wl < FIN_CH) {
        sh[(wave * FIN_CH + wl) * 2 + 0] = a;
        sh[(wave * FIN_CH + wl) * 2 + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x < FIN_CH) {
        a = 0.0;
        b = 0.0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            a += sh[(w * FIN_CH + threadIdx.x) * 2 + 0];
            b += sh[(w * FIN_CH + threadIdx.x) * 2 + 1];
        }
    }
}
__device__ __forceinline__ void fin_reduce(const float* __restrict__ part, int nparts, int C, int c, int lane, double* sh, double& a, double& b) {
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        // U rows per trip, all 2U loads issued before the first use (rows past the end re-read the lane's first row and are zeroed:
        // an `if` per row makes every load its own basic block with its own wait)
        constexpr int U = 8;
        for (int k = lane; k < nparts; k += U * FIN_LANES) {
            float v0[U], v1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = k + u * FIN_LANES;
                const long row = kk < nparts ? kk : k;
                v0[u] = part[(row * 2 + 0) * C + c];
                v1[u] = part[(row * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = k + u * FIN_LANES < nparts;
                s0 += ok ? (double)v0[u] : 0.0;
                s1 += ok ? (double)v1[u] : 0.0;
            }
        }
    }
    a = s0;
    b = s1;
    fin_fold(sh, a, b);
}

// scale/shift for the apply pass + saved statistics + running-statistics update
template <typename T>
__global__ __launch_bounds__(FIN_CH * FIN_LANES) void bn_fwd_finalize_kernel(
    const T* __restrict__ x, const float* __restrict__ part, int nparts, long M, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    float* __restrict__ batch_mean, float* __restrict__ batch_var, float* __restrict__ running_mean, float* __restrict__ running_var,
    float momentum, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double sh[FIN_CH * FIN_LANES / 64 * FIN_CH * 2];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    const int lane = threadIdx.x / FIN_CH;
    double a, b;
    fin_reduce(part, nparts, C, c, lane, sh, a, b);
    if (lane != 0 || c >= C) return;
    const double piv = (double)to_f32(x[c]);
    const double inv_m = 1.0 / (double)M;
    const double dm = a * inv_m;
    const double mean = piv + dm;
    double var = b * inv_m - dm * dm;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float fmean = (float)mean, finv = (float)invstd;
    save_mean[c] = fmean;
    save_invstd[c] = finv;
    const double ub = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
    if (batch_mean) batch_mean[c] = fmean;
    if (batch_var) batch_var[c] = (float)ub;
    if (running_mean) running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * fmean;
    if (running_var) running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)ub;
    const float sc = g * finv;
    scale[c] = sc;
    shift[c] = bt - fmean * sc;
}

// y = act(x*scale + shift [+ skip]); ACT: 0 none, 1 relu, 2 swish
template <typename T, int VEC, bool SKIP, int ACT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, long M, int C,
                                                       int TX, int TY, long rpb, unsigned char* __restrict__ mask = nullptr) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float sc[VEC], sh[VEC];
    ldc<VEC>(scale + col * VEC, sc);
    ldc<VEC>(shift + col * VEC, sh);
    const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    for (long r = r0 + ty; r < r1; r += TY) {
        const long off = r * C + (long)col * VEC;
        float v[VEC];
        ldv<T, VEC>(x + off, v);
        float s[VEC];
        if (SKIP) ldv<T, VEC>(skip + off, s);
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float o = fmaf(v[i], sc[i], sh[i]);
            if (SKIP) o += s[i];
            if (ACT == 1) {
                bits |= (o > 0.f ? 1u : 0u) << i;
                o = fmaxf(o, 0.f);
            }
            if (ACT == 2) o = o * fast_sigmoid(o);          // swish = z*sigmoid(z) (convnet.py:2553)
            v[i] = o;
        }
        stv<T, VEC>(y + off, v);
        // ReLU mask, one byte per 16-byte chunk: the backward of a BN with a fused residual reads it instead of y
        if (ACT == 1 && VEC > 1 && mask) mask[r * (C / VEC) + col] = (unsigned char)bits;
    }
}

// y = relu(x*scale + shift + round_T(xs*sscale + sshift)): the residual unit with a projection shortcut (conv -> BN on the skip
// path).  The skip branch's BN apply pass is folded in: its normalised tensor, rounded to the storage type exactly as the stored
// one would have been, is formed here from the shortcut conv's raw output and never written or read.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_apply_affskip_kernel(const T* __restrict__ x, const T* __restrict__ xs, T* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ sscale, const float* __restrict__ sshift,
                                                               long M, int C, int TX, int TY, long rpb, unsigned char* __restrict__ mask) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float sc[VEC], sh[VEC], ssc[VEC], ssh[VEC];
    ldc<VEC>(scale + col * VEC, sc);
    ldc<VEC>(shift + col * VEC, sh);
    ldc<VEC>(sscale + col * VEC, ssc);
    ldc<VEC>(sshift + col * VEC, ssh);
    const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    for (long r = r0 + ty; r < r1; r += TY) {
        const long off = r * C + (long)col * VEC;
        float v[VEC], s[VEC];
        ldv<T, VEC>(x + off, v);
        ldv<T, VEC>(xs + off, s);
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float o = fmaf(v[i], sc[i], sh[i]) + to_f32(from_f32<T>(fmaf(s[i], ssc[i], ssh[i])));
            bits |= (o > 0.f ? 1u : 0u) << i;
            v[i] = fmaxf(o, 0.f);
        }
        stv<T, VEC>(y + off, v);
        if (VEC > 1 && mask) mask[r * (C / VEC) + col] = (unsigned char)bits;
    }
}

// ---- backward --------------------------------------------------------------------------------------
__device__ __forceinline__ float swish_grad(float z) {
    const float sg = fast_sigmoid(z);
    return sg * (1.f + z * (1.f - sg));
}
// part[(rb*2+0)*C + c] = sum dy', part[(rb*2+1)*C + c] = sum dy' * xhat   (dy' = dy*[y>0] if RELU)
// RELU: 0 = no activation, 1 = mask from the stored forward output y, 2 = mask RECOMPUTED from x as
// [fma(x, gamma*invstd, beta - mean*gamma*invstd) > 0] — the same fp32 expression the forward apply pass evaluated, so
// the mask is the forward's; saves one full read of y in each backward pass (only valid without a fused residual).
// RELU == 3: swish, dz = dy * (s + z*s*(1-s)), s = sigmoid(z), z recomputed from x the same way (no residual).
// RELU == 4: mask from the bit mask the forward apply pass wrote (one byte per 16-byte chunk; `y` then points to it):
//            1/16 of the bytes of reading y itself — used for the BNs with a fused residual.
// (4 waves per SIMD: the grid is 1024 blocks = 4 per CU; the 2-byte variants with a y-based mask / swish had compiled to 132-155
// registers, i.e. 3 blocks per CU and a second, third-full round)
template <typename T, int VEC, int RELU>
__global__ __launch_bounds__(256, 4) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ part, long M, int C, int TX, int TY, long rpb) {
    extern __shared__ float red[];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    const bool active = ty < TY && col * VEC < C;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], is[VEC], sc[VEC], sh[VEC];
        ldc<VEC>(mean + col * VEC, mu);
        ldc<VEC>(invstd + col * VEC, is);
        if (RELU >= 2) {
            ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
            ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sc[i] *= is[i];
                sh[i] -= mu[i] * sc[i];
            }
        }
        const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
        // U rows per trip with all their loads issued first: two 16-byte loads in flight per thread kept these kernels at
        // 3.8-4.2 TB/s (the apply kernels, whose stores need no wait, reach 5.7)
        constexpr int U = 4;                                    // (swish spills 60 bytes inside the loop at 128 registers; U = 3 is spill-free and 0.3 % slower per step)
        auto row = [&](const float* g, const float* v, const float* o, unsigned bits) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float gg = g[i];
                if (RELU == 1) gg = o[i] > 0.f ? gg : 0.f;
                if (RELU == 4) gg = (bits >> i) & 1u ? gg : 0.f;
                if (RELU == 2) gg = fmaf(v[i], sc[i], sh[i]) > 0.f ? gg : 0.f;
                if (RELU == 3) gg *= swish_grad(fmaf(v[i], sc[i], sh[i]));
                s1[i] += gg;
                s2[i] = fmaf(gg, (v[i] - mu[i]) * is[i], s2[i]);
            }
        };
        long r = r0 + ty;
        for (; r + (long)(U - 1) * TY < r1; r += (long)U * TY) {
            float g[U][VEC], v[U][VEC], o[U][VEC];
            unsigned bits[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long off = (r + (long)u * TY) * C + (long)col * VEC;
                ldv<T, VEC>(dy + off, g[u]);
                ldv<T, VEC>(x + off, v[u]);
                if (RELU == 1) ldv<T, VEC>(y + off, o[u]);
                bits[u] = 0;
                if (RELU == 4) bits[u] = reinterpret_cast<const unsigned char*>(y)[(r + (long)u * TY) * (C / VEC) + col];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) row(g[u], v[u], o[u], bits[u]);
        }
        for (; r < r1; r += TY) {
            const long off = r * C + (long)col * VEC;
            float g[VEC], v[VEC], o[VEC];
            ldv<T, VEC>(dy + off, g);
            ldv<T, VEC>(x + off, v);
            if (RELU == 1) ldv<T, VEC>(y + off, o);
            unsigned bits = 0;
            if (RELU == 4) bits = reinterpret_cast<const unsigned char*>(y)[r * (C / VEC) + col];
            row(g, v, o, bits);
        }
    }
    float* r1p = red;
    float* r2p = red + 256 * VEC;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            r1p[(ty * TX + tx) * VEC + i] = s1[i];
            r2p[(ty * TX + tx) * VEC + i] = s2[i];
        }
    }
    __syncthreads();
    if (ty == 0 && col * VEC < C) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < TY; ++k) {
                a += r1p[(k * TX + tx) * VEC + i];
                b += r2p[(k * TX + tx) * VEC + i];
            }
            part[((long)blockIdx.y * 2 + 0) * C + col * VEC + i] = a;
            part[((long)blockIdx.y * 2 + 1) * C + col * VEC + i] = b;
        }
    }
}

// coef[0][c] = gamma*invstd, coef[1][c] = dbeta/M, coef[2][c] = dgamma/M
// raw_mean != nullptr: the second partial is sum dy' * x (conv-epilogue partials, mcn_conv2d_dgrad_bnred), not sum dy' * xhat:
// sum dy' * xhat = invstd * (sum dy' * x - mean * sum dy'), in double
__global__ __launch_bounds__(FIN_CH * FIN_LANES) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, long M, int C,
                                                                             const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                             float grad_scale, float* __restrict__ coef, int frozen,
                                                                             const float* __restrict__ raw_mean = nullptr) {
    __shared__ double sh[FIN_CH * FIN_LANES / 64 * FIN_CH * 2];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH;
    const int lane = threadIdx.x / FIN_CH;
    double a, b;
    fin_reduce(part, nparts, C, c, lane, sh, a, b);
    if (lane != 0 || c >= C) return;
    if (raw_mean) b = (double)invstd[c] * (b - (double)raw_mean[c] * a);
    if (dbeta) dbeta[c] = (float)a * grad_scale;
    if (dgamma) dgamma[c] = (float)b * grad_scale;
    coef[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
    // frozen statistics (is_training=False): mean / variance are constants, the two correction terms of dx vanish
    coef[C + c] = frozen ? 0.f : (float)(a / (double)M);
    coef[2 * C + c] = frozen ? 0.f : (float)(b / (double)M);
}

template <typename T, int VEC, int RELU, bool DSKIP>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ coef, T* __restrict__ dx, T* __restrict__ dskip,
                                                           long M, int C, int TX, int TY, long rpb) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float mu[VEC], is[VEC], ca[VEC], cb[VEC], cc[VEC], sc[VEC], sh[VEC];
    ldc<VEC>(mean + col * VEC, mu);
    ldc<VEC>(invstd + col * VEC, is);
    ldc<VEC>(coef + col * VEC, ca);
    ldc<VEC>(coef + C + col * VEC, cb);
    ldc<VEC>(coef + 2 * C + col * VEC, cc);
    if (RELU >= 2) {
        ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
        ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] *= is[i];
            sh[i] -= mu[i] * sc[i];
        }
    }
    const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    for (long r = r0 + ty; r < r1; r += TY) {
        const long off = r * C + (long)col * VEC;
        float g[VEC], v[VEC], o[VEC];
        ldv<T, VEC>(dy + off, g);
        ldv<T, VEC>(x + off, v);
        if (RELU == 1) ldv<T, VEC>(y + off, o);
        unsigned bits = 0;
        if (RELU == 4) bits = reinterpret_cast<const unsigned char*>(y)[r * (C / VEC) + col];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if (RELU == 1) g[i] = o[i] > 0.f ? g[i] : 0.f;
            if (RELU == 4) g[i] = (bits >> i) & 1u ? g[i] : 0.f;
            if (RELU == 2) g[i] = fmaf(v[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
            if (RELU == 3) g[i] *= swish_grad(fmaf(v[i], sc[i], sh[i]));
            const float xh = (v[i] - mu[i]) * is[i];
            v[i] = ca[i] * (g[i] - cb[i] - xh * cc[i]);
        }
        stv<T, VEC>(dx + off, v);
        if (DSKIP) stv<T, VEC>(dskip + off, g);
    }
}

// ---- backward of BN + ReLU whose output gradient is the gradient of a 3x3 / 2 max-pool (the ResNet stem) ---------------------
// mcn_maxpool_bwd would write the full-resolution gradient (the largest gradient tensor of the network) for the two BN passes to
// read it back; here both passes build their dy chunk themselves: pixel (iy, ix) is seen by at most 2 x 2 pool windows, each
// contributes its pooled gradient where its arg-max points at the pixel; the sum is rounded to the storage type exactly as the
// stored tensor would have been.  ReLU mask recomputed from x (forward without a fused residual), as in RELU == 2 above.
struct PoolRoute { int H, W, OH, OW, padT, padL; };
template <typename T, int VEC>
__device__ __forceinline__ void pool_route_dy(const T* __restrict__ dp, const int8_t* __restrict__ arg, const PoolRoute& P, unsigned r, int C, int col,
                                              float (&g)[VEC]) {
    static_assert(VEC == 4 || VEC == 8, "one 16-byte chunk per thread");
    const unsigned r2 = r / (unsigned)P.W;
    const int ix = (int)(r - r2 * (unsigned)P.W);
    const int n = (int)(r2 / (unsigned)P.H), iy = (int)(r2 - (unsigned)n * (unsigned)P.H);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
    const int ty0 = iy + P.padT, tx0 = ix + P.padL;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = (ty0 >> 1) - a, kr = (ty0 & 1) + 2 * a;
        const bool yok = kr < 3 && (unsigned)oy < (unsigned)P.OH;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = (tx0 >> 1) - b, ks = (tx0 & 1) + 2 * b;
            const bool ok = yok && ks < 3 && (unsigned)ox < (unsigned)P.OW;
            const long o = (((long)n * P.OH + (ok ? oy : 0)) * P.OW + (ok ? ox : 0)) * C + (long)col * VEC;
            float gp[VEC];
            ldv<T, VEC>(dp + o, gp);
            signed char ac[VEC];
            if constexpr (VEC == 8) *reinterpret_cast<unsigned long long*>(ac) = *reinterpret_cast<const unsigned long long*>(arg + o);
            else *reinterpret_cast<unsigned*>(ac) = *reinterpret_cast<const unsigned*>(arg + o);
            const int code = ok ? kr * 3 + ks : -1;
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (ac[i] == code) g[i] += gp[i];
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = to_f32(from_f32<T>(g[i]));           // the value mcn_maxpool_bwd stores
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const T* __restrict__ dp, const int8_t* __restrict__ arg, const PoolRoute P, const T* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ part, long M, int C, int TX, int TY, long rpb) {
    extern __shared__ float red[];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    const bool active = ty < TY && col * VEC < C;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], is[VEC], sc[VEC], sh[VEC];
        ldc<VEC>(mean + col * VEC, mu);
        ldc<VEC>(invstd + col * VEC, is);
        ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
        ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] *= is[i];
            sh[i] -= mu[i] * sc[i];
        }
        const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
        for (long r = r0 + ty; r < r1; r += TY) {
            float g[VEC], v[VEC];
            ldv<T, VEC>(x + r * C + (long)col * VEC, v);
            pool_route_dy<T, VEC>(dp, arg, P, (unsigned)r, C, col, g);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float gg = fmaf(v[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
                s1[i] += gg;
                s2[i] = fmaf(gg, (v[i] - mu[i]) * is[i], s2[i]);
            }
        }
    }
    float* r1p = red;
    float* r2p = red + 256 * VEC;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            r1p[(ty * TX + tx) * VEC + i] = s1[i];
            r2p[(ty * TX + tx) * VEC + i] = s2[i];
        }
    }
    __syncthreads();
    if (ty == 0 && col * VEC < C) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < TY; ++k) {
                a += r1p[(k * TX + tx) * VEC + i];
                b += r2p[(k * TX + tx) * VEC + i];
            }
            part[((long)blockIdx.y * 2 + 0) * C + col * VEC + i] = a;
            part[((long)blockIdx.y * 2 + 1) * C + col * VEC + i] = b;
        }
    }
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const T* __restrict__ dp, const int8_t* __restrict__ arg, const PoolRoute P, const T* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ coef, T* __restrict__ dx, long M, int C,
                                                                int TX, int TY, long rpb) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float mu[VEC], is[VEC], ca[VEC], cb[VEC], cc[VEC], sc[VEC], sh[VEC];
    ldc<VEC>(mean + col * VEC, mu);
    ldc<VEC>(invstd + col * VEC, is);
    ldc<VEC>(coef + col * VEC, ca);
    ldc<VEC>(coef + C + col * VEC, cb);
    ldc<VEC>(coef + 2 * C + col * VEC, cc);
    ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
    ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sc[i] *= is[i];
        sh[i] -= mu[i] * sc[i];
    }
    const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    for (long r = r0 + ty; r < r1; r += TY) {
        const long off = r * C + (long)col * VEC;
        float g[VEC], v[VEC];
        ldv<T, VEC>(x + off, v);
        pool_route_dy<T, VEC>(dp, arg, P, (unsigned)r, C, col, g);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float gg = fmaf(v[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
            const float xh = (v[i] - mu[i]) * is[i];
            v[i] = ca[i] * (gg - cb[i] - xh * cc[i]);
        }
        stv<T, VEC>(dx + off, v);
    }
}

// ---- backward of BN + swish whose output x_se feeds a squeeze-excite block (models/efficientnet.py:152-163) --------------------
// x_se has two readers, the global average pool of the SE branch and the channel scale y = x_se * m[n,c]; its gradient is
//   g = round_T(round_T(dy * m[n,c]) + dgap[n,c] / HW)
// (what mcn_channel_scale_bwd writes, then mcn_global_avgpool_bwd_acc adds to).  Both BN passes form g themselves from the gradient
// of the scaled tensor, the SE mask and the pooled branch's gradient: the channel-scale backward keeps only its reduction
// (mcn_channel_scale_bwd_dm) and the gradient tensor of x_se — three passes over the widest activations of the network — is gone.
template <typename T, int VEC>
__device__ __forceinline__ void se_route_dy(const T* __restrict__ dy, const T* __restrict__ m, const T* __restrict__ dgap, long off, long nc, float inv_hw, float (&g)[VEC]) {
    float mm[VEC], q[VEC];
    ldv<T, VEC>(dy + off, g);
    ldv<T, VEC>(m + nc, mm);
    ldv<T, VEC>(dgap + nc, q);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = to_f32(from_f32<T>(fmaf(q[i], inv_hw, to_f32(from_f32<T>(g[i] * mm[i])))));      // (explicit fma: as gap_bwd_kernel<ACC>)
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_se_kernel(const T* __restrict__ dy, const T* __restrict__ m, const T* __restrict__ dgap, long HW, const T* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ part, long M, int C, int TX, int TY, long rpb) {
    extern __shared__ float red[];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    const bool active = ty < TY && col * VEC < C;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], is[VEC], sc[VEC], sh[VEC];
        ldc<VEC>(mean + col * VEC, mu);
        ldc<VEC>(invstd + col * VEC, is);
        ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
        ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] *= is[i];
            sh[i] -= mu[i] * sc[i];
        }
        const float inv_hw = 1.f / (float)HW;
        const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
        for (long r = r0 + ty; r < r1; r += TY) {
            const long off = r * C + (long)col * VEC;
            float g[VEC], v[VEC];
            ldv<T, VEC>(x + off, v);
            se_route_dy<T, VEC>(dy, m, dgap, off, (long)((unsigned)r / (unsigned)HW) * C + (long)col * VEC, inv_hw, g);       // (32-bit division: the host checks M < 2^32)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float gg = g[i] * swish_grad(fmaf(v[i], sc[i], sh[i]));
                s1[i] += gg;
                s2[i] = fmaf(gg, (v[i] - mu[i]) * is[i], s2[i]);
            }
        }
    }
    float* r1p = red;
    float* r2p = red + 256 * VEC;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            r1p[(ty * TX + tx) * VEC + i] = s1[i];
            r2p[(ty * TX + tx) * VEC + i] = s2[i];
        }
    }
    __syncthreads();
    if (ty == 0 && col * VEC < C) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < TY; ++k) {
                a += r1p[(k * TX + tx) * VEC + i];
                b += r2p[(k * TX + tx) * VEC + i];
            }
            part[((long)blockIdx.y * 2 + 0) * C + col * VEC + i] = a;
            part[((long)blockIdx.y * 2 + 1) * C + col * VEC + i] = b;
        }
    }
}
// ---- the reduction pass of mcn_bn_bwd_se folded into the channel scale's own reduction (round 4) -------------------------------------
// The BN-backward sums over g = dy * m[n,c] + dgap[n,c] / HW separate per image:
//   sum g * s'       = sum_n ( m[n,c] * A[n,c]  + dgap[n,c] / HW * B[n,c]  ),   A = sum_hw dy * s',        B = sum_hw s'
//   sum g * s' * xh  = sum_n ( m[n,c] * A2[n,c] + dgap[n,c] / HW * B2[n,c] ),   A2 = sum_hw dy * s' * xh,  B2 = sum_hw s' * xh
// (s' = swish'(bn(x)), xh = (x - mean) * invstd), and A, A2, B, B2 need neither the SE mask nor the pooled branch's gradient: the pass that
// computes dm[n,c] = sum_hw dy * x_se — which must run BEFORE the SE branch's backward — produces them too, reading the BN's INPUT x and
// rebuilding x_se = round_T(swish(bn(x))) from it (same two tensor reads as before), and the pass bn_bwd_reduce_se_kernel (two more reads of the widest
// activations of the network per block) becomes a loop over [N * HS][5][C] floats (HS = pixel slices per image, se_bwd_layout; plane 0 = the dm partial).  The sums are those of the unrounded g (the apply pass still
// forms the rounded g per element): dgamma / dbeta agree with the three-step composition to fp32 summation accuracy, not bit for bit.
#ifndef MCN_SE_PRE_U
#define MCN_SE_PRE_U 1       // rows per trip; 2 (four loads in flight) needs 132 registers = 3 waves per SIMD: 26.0 -> 26.3 ms per EfficientNet-B0 step
#endif
template <typename T, int VEC>
__global__ __launch_bounds__(256) void se_bwd_pre_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ sums, long HW, int C,
                                                         int TX, int TY, int HS, long rps) {
    // grid.y = image x slice of its HW pixels (HS slices of rps rows: enough workgroups to fill the chip at every layer shape); sums: [N * HS][5][C], plane 0 = the dm partial
    extern __shared__ float red[];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    const long n = blockIdx.y / HS;
    const long r_lo = (long)(blockIdx.y % HS) * rps, r_hi = min(HW, r_lo + rps);
    const bool active = ty < TY && col * VEC < C;
    float acc[5][VEC];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;
    if (active) {
        float mu[VEC], is[VEC], sc[VEC], sh[VEC];
        ldc<VEC>(mean + col * VEC, mu);
        ldc<VEC>(invstd + col * VEC, is);
        ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
        ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] *= is[i];
            sh[i] -= mu[i] * sc[i];
        }
        // packed fp32 pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): with scalar math the pass is VALU bound (~24 instructions per element), not HBM bound
        typedef float se_f32x2 __attribute__((ext_vector_type(2)));
        auto row = [&](const float* g, const float* v) {
#pragma unroll
            for (int i = 0; i < VEC; i += 2) {
                const se_f32x2 v2 = {v[i], v[i + 1]}, g2 = {g[i], g[i + 1]};
                const se_f32x2 z = __builtin_elementwise_fma(v2, se_f32x2{sc[i], sc[i + 1]}, se_f32x2{sh[i], sh[i + 1]});
                const se_f32x2 sig = {fast_sigmoid(z[0]), fast_sigmoid(z[1])};
                const se_f32x2 sw = z * sig;
                const se_f32x2 xse = {to_f32(from_f32<T>(sw[0])), to_f32(from_f32<T>(sw[1]))};      // the BN + swish output as the forward stored it
                const se_f32x2 one = {1.f, 1.f};
                const se_f32x2 sp = sig * __builtin_elementwise_fma(z, one - sig, one);               // swish'(z)
                const se_f32x2 xh = (v2 - se_f32x2{mu[i], mu[i + 1]}) * se_f32x2{is[i], is[i + 1]};
                const se_f32x2 gs = g2 * sp;
                const se_f32x2 a0 = __builtin_elementwise_fma(g2, xse, se_f32x2{acc[0][i], acc[0][i + 1]});
                const se_f32x2 a1 = se_f32x2{acc[1][i], acc[1][i + 1]} + gs;
                const se_f32x2 a2 = __builtin_elementwise_fma(gs, xh, se_f32x2{acc[2][i], acc[2][i + 1]});
                const se_f32x2 a3 = se_f32x2{acc[3][i], acc[3][i + 1]} + sp;
                const se_f32x2 a4 = __builtin_elementwise_fma(sp, xh, se_f32x2{acc[4][i], acc[4][i + 1]});
                acc[0][i] = a0[0]; acc[0][i + 1] = a0[1];
                acc[1][i] = a1[0]; acc[1][i + 1] = a1[1];
                acc[2][i] = a2[0]; acc[2][i + 1] = a2[1];
                acc[3][i] = a3[0]; acc[3][i + 1] = a3[1];
                acc[4][i] = a4[0]; acc[4][i + 1] = a4[1];
            }
        };
        constexpr int U = MCN_SE_PRE_U;
        long r = r_lo + ty;
        for (; r + (long)(U - 1) * TY < r_hi; r += (long)U * TY) {
            float g[U][VEC], v[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long off = (n * HW + r + (long)u * TY) * C + (long)col * VEC;
                ldv<T, VEC>(dy + off, g[u]);
                ldv<T, VEC>(x + off, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) row(g[u], v[u]);
        }
        for (; r < r_hi; r += TY) {
            const long off = (n * HW + r) * C + (long)col * VEC;
            float g[VEC], v[VEC];
            ldv<T, VEC>(dy + off, g);
            ldv<T, VEC>(x + off, v);
            row(g, v);
        }
    }
    const int cols = TX * VEC;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (ty < TY) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) red[ty * cols + tx * VEC + i] = acc[k][i];
        }
        __syncthreads();
        if (ty == 0 && col * VEC < C) {
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float t = 0.f;
                for (int q = 0; q < TY; ++q) t += red[q * cols + tx * VEC + i];
                o[i] = t;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) sums[((long)blockIdx.y * 5 + k) * C + (long)col * VEC + i] = o[i];
        }
        __syncthreads();
    }
}
// dm[n,c] = round_T(sum over the image's slices of plane 0)
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_dm_fold_kernel(const float* __restrict__ sums, T* __restrict__ dm, long NC, int C, int HS) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    const long n = i / C;
    const int c = (int)(i - n * C);
    float t = 0.f;
    for (int h = 0; h < HS; ++h) t += sums[((n * HS + h) * 5) * C + c];
    dm[i] = from_f32<T>(t);
}
// partial rows part[nb][0][c] = sum g * s', part[nb][1][c] = sum g * s' * xh over the images [nb * NB, nb * NB + NB) from the per-image(-slice) sums, the SE
// mask and the pooled branch's gradient (bn_bwd_finalize_kernel adds the rows)
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_sums_kernel(const float* __restrict__ sums, const T* __restrict__ m, const T* __restrict__ dgap, float* __restrict__ part, int N, long HW, int C, int HS,
                                                          int NB) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float inv_hw = 1.f / (float)HW;
    float s1 = 0.f, s2 = 0.f;
    const int n0 = blockIdx.y * NB, n1 = min(N, n0 + NB);
    for (int n = n0; n < n1; ++n) {
        const float mm = to_f32(m[(long)n * C + c]), q = to_f32(dgap[(long)n * C + c]) * inv_hw;
        float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
        for (int h = 0; h < HS; ++h) {
            const float* const sp = sums + ((long)n * HS + h) * 5 * C + c;
            a1 += sp[(long)C];
            a2 += sp[2 * (long)C];
            b1 += sp[3 * (long)C];
            b2 += sp[4 * (long)C];
        }
        s1 += fmaf(mm, a1, q * b1);
        s2 += fmaf(mm, a2, q * b2);
    }
    part[((long)blockIdx.y * 2 + 0) * C + c] = s1;
    part[((long)blockIdx.y * 2 + 1) * C + c] = s2;
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_se_kernel(const T* __restrict__ dy, const T* __restrict__ m, const T* __restrict__ dgap, long HW, const T* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ coef, T* __restrict__ dx, long M, int C, int TX,
                                                              int TY, long rpb) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float mu[VEC], is[VEC], ca[VEC], cb[VEC], cc[VEC], sc[VEC], sh[VEC];
    ldc<VEC>(mean + col * VEC, mu);
    ldc<VEC>(invstd + col * VEC, is);
    ldc<VEC>(coef + col * VEC, ca);
    ldc<VEC>(coef + C + col * VEC, cb);
    ldc<VEC>(coef + 2 * C + col * VEC, cc);
    ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
    ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sc[i] *= is[i];
        sh[i] -= mu[i] * sc[i];
    }
    const float inv_hw = 1.f / (float)HW;
    const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    for (long r = r0 + ty; r < r1; r += TY) {
        const long off = r * C + (long)col * VEC;
        float g[VEC], v[VEC];
        ldv<T, VEC>(x + off, v);
        se_route_dy<T, VEC>(dy, m, dgap, off, (long)((unsigned)r / (unsigned)HW) * C + (long)col * VEC, inv_hw, g);       // (32-bit division: the host checks M < 2^32)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float gg = g[i] * swish_grad(fmaf(v[i], sc[i], sh[i]));
            const float xh = (v[i] - mu[i]) * is[i];
            v[i] = ca[i] * (gg - cb[i] - xh * cc[i]);
        }
        stv<T, VEC>(dx + off, v);
    }
}

// ---- host ------------------------------------------------------------------------------------------
#define BN_TARGET_BLOCKS 1024       /* upper bound of the row blocks (workspace sizing) */
// row blocks per kernel by element size (same-box A/B of the whole ResNet-50 step: fp32 1024 > 768 > 2048; bf16, since the chunk loads
// are streaming: 1024 21.56-21.58 ms, 768 21.62-21.70, 512 21.82-21.85 — before that 768 > 512 ~ 1024)
#define BN_TGT_BF16 1024
template <typename T>
static int bn_target() {
    static const int env = [] { const char* e = getenv("MCN_BN_TGT"); const int v = e ? atoi(e) : 0; return v > BN_TARGET_BLOCKS ? BN_TARGET_BLOCKS : v; }();   // experiments
    return env > 0 ? env : (sizeof(T) == 2 ? BN_TGT_BF16 : 1024);
}
static size_t bn_parts_bytes(long M, int C) {
    // worst case over vector widths: gy <= BN_TARGET_BLOCKS
    return align_up((size_t)BN_TARGET_BLOCKS * 2 * C * sizeof(float), 256);
}
extern "C" size_t mcn_bn_workspace_bytes(int64_t M, int32_t C) {
    if (M < 0 || C <= 0) return 0;
    return bn_parts_bytes(M, C) + align_up((size_t)4 * C * sizeof(float), 256);   // partials + coef[3][C] + invstd[C] (frozen backward)
}

template <typename T, int VEC>
static int bn_fwd_train_t(const void* x, const float* gamma, const float* beta, const void* skip, void* y, unsigned char* relu_mask, float* save_mean,
                          float* save_invstd, float* batch_mean, float* batch_var, float* running_mean, float* running_var,
                          float momentum, long M, int C, float eps, mcn_act act, void* ws, hipStream_t st) {
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    float* part = (float*)ws;
    float* scale = (float*)((char*)ws + bn_parts_bytes(M, C));
    float* shift = scale + C;
    const dim3 grid(L.gx, L.gy), block(256);
    hipLaunchKernelGGL((bn_stats_kernel<T, VEC>), grid, block, 2 * 256 * VEC * sizeof(float), st, (const T*)x, part, M, C, L.TX, L.TY, L.rpb);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_fwd_finalize_kernel<T>), dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, st, (const T*)x, (const float*)part, L.gy, M, C, gamma,
                       beta, eps, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, scale, shift);
    MCN_CHECK_LAUNCH();
    const int a = (int)act;
#define BN_APPLY(SK, RL)                                                                                                   \
    hipLaunchKernelGGL((bn_apply_kernel<T, VEC, SK, RL>), grid, block, 0, st, (const T*)x, (const T*)skip, (T*)y, (const float*)scale, \
                       (const float*)shift, M, C, L.TX, L.TY, L.rpb, relu_mask)
    if (skip) { if (a == 1) BN_APPLY(true, 1); else if (a == 2) BN_APPLY(true, 2); else BN_APPLY(true, 0); }
    else { if (a == 1) BN_APPLY(false, 1); else if (a == 2) BN_APPLY(false, 2); else BN_APPLY(false, 0); }
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// y = act(x*scale + shift) AND the per-image channel means of the STORED y in one pass (round 4): the squeeze-excite block reads the global
// average of its input right after the BN + swish that produces it (models/efficientnet.py:183 -> tf.reduce_mean(x, axis=[1, 2])), which
// used to be a second full read of that tensor (gap_fwd_kernel).  One block per (image, group of TX channel chunks): TY row lanes walk the
// image's HW pixels, the rounded outputs are summed in fp32 per thread and folded over the row lanes through LDS, gap[n][c] = sum / HW.
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(256) void bn_apply_gap_kernel(const T* __restrict__ x, T* __restrict__ y, T* __restrict__ gap, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int HW, int C, int TX, int TY) {
    // y == nullptr (round 4): the output itself is not stored — a squeeze-excite block whose channel scale rebuilds it from x (mcn_bn_act_scale_fwd) needs the means only
    extern __shared__ float gapred[];                              // [TY][TX * VEC]
    const int cv = C / VEC;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    const long n = blockIdx.y;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (ty < TY && col < cv) {
        float sc[VEC], sh[VEC];
        ldc<VEC>(scale + col * VEC, sc);
        ldc<VEC>(shift + col * VEC, sh);
        const long base = n * HW * C + (long)col * VEC;
        auto row = [&](float (&v)[VEC], int q) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float o = fmaf(v[i], sc[i], sh[i]);
                if (ACT == 1) o = fmaxf(o, 0.f);
                if (ACT == 2) o = o * fast_sigmoid(o);
                v[i] = to_f32(from_f32<T>(o));                     // the value as stored: what a separate pooling pass would read back
                acc[i] += v[i];
            }
            if (y) stv<T, VEC>(y + base + (long)q * C, v);                      // (uniform)
        };
        // U rows per trip, their loads issued first: with one 16-byte load in flight per thread the means-only form (y == nullptr) read at 3.2 TB/s
        constexpr int U = 4;
        int q = ty;
        for (; q + (U - 1) * TY < HW; q += U * TY) {
            float v[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) ldv<T, VEC>(x + base + (long)(q + u * TY) * C, v[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) row(v[u], q + u * TY);
        }
        for (; q < HW; q += TY) {
            float v[VEC];
            ldv<T, VEC>(x + base + (long)q * C, v);
            row(v, q);
        }
    }
    const int cols = TX * VEC;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) gapred[ty * cols + tx * VEC + i] = acc[i];
    }
    __syncthreads();
    if (ty == 0 && col < cv) {
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float s2 = 0.f;
            for (int k = 0; k < TY; ++k) s2 += gapred[k * cols + tx * VEC + i];
            acc[i] = s2 * inv;
        }
        stv<T, VEC>(gap + n * C + (long)col * VEC, acc);
    }
}
template <typename T, int VEC>
static int bn_fwd_train_gap_t(const void* x, const float* gamma, const float* beta, void* y, void* gap, float* save_mean, float* save_invstd, float* batch_mean,
                              float* batch_var, float* running_mean, float* running_var, float momentum, int N, int HW, int C, float eps, mcn_act act, void* ws,
                              hipStream_t st) {
    const long M = (long)N * HW;
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    float* part = (float*)ws;
    float* scale = (float*)((char*)ws + bn_parts_bytes(M, C));
    float* shift = scale + C;
    hipLaunchKernelGGL((bn_stats_kernel<T, VEC>), dim3(L.gx, L.gy), dim3(256), 2 * 256 * VEC * sizeof(float), st, (const T*)x, part, M, C, L.TX, L.TY, L.rpb);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_fwd_finalize_kernel<T>), dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, st, (const T*)x, (const float*)part, L.gy, M, C, gamma,
                       beta, eps, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, scale, shift);
    MCN_CHECK_LAUNCH();
    // apply + pooled means: image-aligned blocks (the statistics pass above keeps the row-block layout)
    const int cv = C / VEC;
    int TX = cv < 32 ? cv : 32;
    int TY = 256 / TX;
    if (TY > HW) TY = HW;
    const dim3 grid((unsigned)((cv + TX - 1) / TX), (unsigned)N);
    const size_t lds = (size_t)TY * TX * VEC * sizeof(float);
    const int a = (int)act;
#define BN_APPLY_GAP(RL) hipLaunchKernelGGL((bn_apply_gap_kernel<T, VEC, RL>), grid, dim3(256), lds, st, (const T*)x, (T*)y, (T*)gap, (const float*)scale, (const float*)shift, HW, C, TX, TY)
    if (a == 1) BN_APPLY_GAP(1); else if (a == 2) BN_APPLY_GAP(2); else BN_APPLY_GAP(0);
#undef BN_APPLY_GAP
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_bn_fwd_train_gap(const void* x, const float* gamma, const float* beta, void* y, void* gap, float* save_mean, float* save_invstd,
                                    float* batch_mean, float* batch_var, float* running_mean, float* running_var, float momentum, int32_t N, int32_t HW,
                                    int32_t C, float eps, mcn_act act, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !gap || !save_mean || !save_invstd || N <= 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_gap: bad argument (N=%d HW=%d C=%d)", N, HW, C);      // (y may be NULL: means only)
    if (N > 65535) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train_gap: more than 65535 images (one block row per image)");
    if ((int)act < 0 || (int)act > 2) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train_gap: activation %d (none / relu / swish only)", (int)act);
    const int64_t M = (int64_t)N * HW;
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_fwd_train_gap: workspace too small");
    hipStream_t st = (hipStream_t)stream;
#define BN_GAP_T(TT, VV) return bn_fwd_train_gap_t<TT, VV>(x, gamma, beta, y, gap, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, N, HW, C, eps, act, ws, st)
    if (dtype == MCN_F32) { if (C % 4 == 0) BN_GAP_T(float, 4); BN_GAP_T(float, 1); }
    if (dtype == MCN_BF16) { if (C % 8 == 0) BN_GAP_T(bf16_t, 8); BN_GAP_T(bf16_t, 1); }
    if (dtype == MCN_F16) { if (C % 8 == 0) BN_GAP_T(f16_t, 8); BN_GAP_T(f16_t, 1); }
#undef BN_GAP_T
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train_gap: dtype %d unsupported", (int)dtype);
}

// y = round(round(act(bn(x))) * m[n,c]): the channel scale of a squeeze-excite block applied to the BN + swish output REBUILT from the BN's input (the value
// mcn_bn_fwd_train_gap would have stored, up to the last bit of scale / shift: here gamma * invstd and beta - mean * gamma * invstd in fp32, the expression the
// backward passes use) — with mcn_bn_fwd_train_gap(y = NULL) the BN's output is never written: one pass fewer over the expanded activations of an MBConv block
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(256) void bn_act_scale_kernel(const T* __restrict__ x, const T* __restrict__ m, T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, long HW, long M, int C, int TX, int TY, long rpb) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float mu[VEC], is[VEC], sc[VEC], sh[VEC];
    ldc<VEC>(mean + col * VEC, mu);
    ldc<VEC>(invstd + col * VEC, is);
    ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
    ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sc[i] *= is[i];
        sh[i] -= mu[i] * sc[i];
    }
    const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    for (long r = r0 + ty; r < r1; r += TY) {
        const long off = r * C + (long)col * VEC;
        float v[VEC], mm[VEC];
        ldv<T, VEC>(x + off, v);
        ldv<T, VEC>(m + (long)((unsigned)r / (unsigned)HW) * C + (long)col * VEC, mm);      // (32-bit division: the host checks M < 2^32)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float o = fmaf(v[i], sc[i], sh[i]);
            if (ACT == 1) o = fmaxf(o, 0.f);
            if (ACT == 2) o = o * fast_sigmoid(o);
            v[i] = to_f32(from_f32<T>(o)) * mm[i];
        }
        stv<T, VEC>(y + off, v);
    }
}
template <typename T, int VEC>
static int bn_act_scale_t(const void* x, const void* m, void* y, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, long HW, long M, int C, int act,
                          hipStream_t st) {
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    const dim3 grid(L.gx, L.gy), block(256);
#define BN_ACT_SCALE(AA) hipLaunchKernelGGL((bn_act_scale_kernel<T, VEC, AA>), grid, block, 0, st, (const T*)x, (const T*)m, (T*)y, save_mean, save_invstd, gamma, beta, HW, M, C, L.TX, L.TY, L.rpb)
    if (act == 1) BN_ACT_SCALE(1); else if (act == 2) BN_ACT_SCALE(2); else BN_ACT_SCALE(0);
#undef BN_ACT_SCALE
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_bn_act_scale_fwd(const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, const void* m, void* y, int32_t N, int64_t HW,
                                    int32_t C, mcn_act act, mcn_dtype dtype, void* stream) {
    const int64_t M = (int64_t)N * HW;
    if (!x || !m || !y || !save_mean || !save_invstd || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_act_scale_fwd: bad argument");
    if ((int)act < 0 || (int)act > 2) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_act_scale_fwd: activation %d (none, relu, swish)", (int)act);
    if (M >= 0xffffffffll) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_act_scale_fwd: more than 2^32 pixels");
    if (N == 0) return MCN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32 && C % 4 == 0) return bn_act_scale_t<float, 4>(x, m, y, gamma, beta, save_mean, save_invstd, (long)HW, (long)M, C, (int)act, st);
    if (dtype == MCN_BF16 && C % 8 == 0) return bn_act_scale_t<bf16_t, 8>(x, m, y, gamma, beta, save_mean, save_invstd, (long)HW, (long)M, C, (int)act, st);
    if (dtype == MCN_F16 && C % 8 == 0) return bn_act_scale_t<f16_t, 8>(x, m, y, gamma, beta, save_mean, save_invstd, (long)HW, (long)M, C, (int)act, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_act_scale_fwd: dtype %d / C = %d (a multiple of the 16-byte chunk) unsupported", (int)dtype, C);
}

// ---- statistics from conv-epilogue partials -----------------------------------------------------------------------
// part[(k*3 + {0,1,2})*C + c] = sum(y - p), sum((y - p)^2), p over the rows [k*rpp, min(M, (k+1)*rpp)) of y.
// Each partial is an accurate fp32 (count, mean, M2) thanks to its own pivot; partials are merged in double as
// sum n*mean and sum (M2 + n*mean^2).
#define BN_FOLD_ROWS 512
// rpp == 0: "counted" rows part[(k*4 + {0,1,2,3})*C + c] — the fourth plane holds the number of pixel rows the row sums (a
// persistent conv workgroup's share of the tensor, not a contiguous row range; rows with count 0 are skipped)
// A partial row contributes sum y = s1 + n p and sum y^2 = s2 + 2 p s1 + n p^2 (n = its pixel rows), merged in double.
// four rows k, k + stride, ... at once: all their loads are issued before the first use (one thread walking 16-64 rows one at a time is
// a chain of dependent L2 round trips — 13 us for the 512-row finalize, with or without the two double divisions per row of the
// first formulation n * mean, M2 + n * mean^2); rows >= kend and rows without pixels contribute nothing
__device__ __forceinline__ void fused_partial4(const float* __restrict__ part, int k, int stride, int kend, int C, int c, long M, int rpp, double& sm, double& sq) {
    const int planes = rpp > 0 ? 3 : 4;
    // counts first (counted rows of a wide layer are mostly rows of OTHER channel blocks: count 0, nothing else to read) ...
    long n[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int kk = k + u * stride;
        if (rpp > 0) {
            n[u] = M - (long)kk * rpp;
            if (n[u] > rpp) n[u] = rpp;
        } else {
            n[u] = kk < kend ? (long)part[((long)kk * 4 + 3) * C + c] : 0;
        }
        if (kk >= kend) n[u] = 0;
    }
    // ... then the three sums of the rows that have pixels, again all loads before the first use
    float s1[4], s2[4], pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long row = (long)(k + u * stride) * planes;
        s1[u] = s2[u] = pv[u] = 0.f;
        if (n[u] > 0) {
            s1[u] = part[(row + 0) * C + c];
            s2[u] = part[(row + 1) * C + c];
            pv[u] = part[(row + 2) * C + c];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (n[u] > 0) {
            const double a = (double)s1[u], b = (double)s2[u], p0 = (double)pv[u];
            sm += a + (double)n[u] * p0;
            sq += b + 2.0 * p0 * a + (double)n[u] * p0 * p0;
        }
}
// stage 1 (only for many partials): groups of rpg partial rows -> fold[(g*2 + {0,1})*C + c] (double)
__global__ __launch_bounds__(256) void bn_fold_partials_kernel(const float* __restrict__ part, double* __restrict__ fold, int nparts, int C, long M, int rpp,
                                                               int rpg, int TX) {
    // TX channel lanes x 256/TX row lanes (narrow layers — 16-96 channels with 50 000 partial rows in EfficientNet's first
    // blocks — left most of a channel-per-thread block idle and one thread walking 100+ rows); lanes fold in a fixed order
    __shared__ double red[256 * 2];
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = blockIdx.x * TX + tx;
    const int r0 = blockIdx.y * rpg, r1 = min(nparts, r0 + rpg);
    double sm = 0.0, sq = 0.0;
    if (c < C)
        for (int r = r0 + ty; r < r1; r += 4 * TY) fused_partial4(part, r, TY, r1, C, c, M, rpp, sm, sq);
    red[threadIdx.x * 2] = sm;
    red[threadIdx.x * 2 + 1] = sq;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < TY; ++k) {
        a += red[(k * TX + tx) * 2];
        b += red[(k * TX + tx) * 2 + 1];
    }
    fold[((long)blockIdx.y * 2 + 0) * C + c] = a;
    fold[((long)blockIdx.y * 2 + 1) * C + c] = b;
}
// COMPACT counted rows (rpp < 0, round 3): a persistent conv workgroup owns ONE block of BNB = -rpp output channels, so its row is keyed by
// that block — part[((nb * RPB + j) * 4 + {0,1,2,3}) * BNB + cl], nb = c / BNB, cl = c % BNB, j < RPB = nparts / blocks — and a channel's rows
// are exactly the RPB rows of its block: no scan over the other blocks' rows (and no zero counts written for them), all four planes of
// U rows in flight at once, no fold launch.
__device__ __forceinline__ void compact_partials(const float* __restrict__ part, int nparts, int BNB, int C, int c, int lane, double& sm, double& sq) {
    const int nblk = (C + BNB - 1) / BNB, RPB = nparts / nblk;
    const int nb = c / BNB, cl = c - nb * BNB;
    const float* base = part + (size_t)nb * RPB * 4 * BNB + cl;
    constexpr int U = 4;
    for (int j = lane; j < RPB; j += U * FIN_LANES) {
        float s1[U], s2[U], pv[U], n[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = j + u * FIN_LANES;
            const float* r = base + (size_t)(jj < RPB ? jj : j) * 4 * BNB;
            s1[u] = r[0];
            s2[u] = r[BNB];
            pv[u] = r[2 * BNB];
            n[u] = r[3 * BNB];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = j + u * FIN_LANES < RPB && n[u] > 0.f;
            const double a = (double)s1[u], b = (double)s2[u], p0 = (double)pv[u], nn = (double)n[u];
            sm += ok ? a + nn * p0 : 0.0;
            sq += ok ? b + 2.0 * p0 * a + nn * p0 * p0 : 0.0;
        }
    }
}
// stage 2: FOLDED reads the doubles of stage 1, otherwise the raw partials
template <bool FOLDED>
__global__ __launch_bounds__(FIN_CH * FIN_LANES) void bn_fwd_finalize_fused_kernel(
    const float* __restrict__ part, const double* __restrict__ fold, int nrows, int rpp, long M, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ batch_mean,
    float* __restrict__ batch_var, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float* __restrict__ scale,
    float* __restrict__ shift) {
    __shared__ double sh[FIN_CH * FIN_LANES / 64 * FIN_CH * 2];
    const int cl = threadIdx.x % FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    const int lane = threadIdx.x / FIN_CH;
    double sm = 0.0, sq = 0.0;
    if (c < C) {
        if (FOLDED) {
            constexpr int U = 4;                            // (all 2U loads of a trip before the first add, as fin_reduce)
            for (int k = lane; k < nrows; k += U * FIN_LANES) {
                double v0[U], v1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long row = k + u * FIN_LANES < nrows ? k + u * FIN_LANES : k;
                    v0[u] = fold[(row * 2 + 0) * C + c];
                    v1[u] = fold[(row * 2 + 1) * C + c];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = k + u * FIN_LANES < nrows;
                    sm += ok ? v0[u] : 0.0;
                    sq += ok ? v1[u] : 0.0;
                }
            }
        } else if (rpp < 0) {
            compact_partials(part, nrows, -rpp, C, c, lane, sm, sq);
        } else {
            for (int k = lane; k < nrows; k += 4 * FIN_LANES) fused_partial4(part, k, FIN_LANES, nrows, C, c, M, rpp, sm, sq);
        }
    }
    double a = sm, b = sq;
    fin_fold(sh, a, b);
    if (lane != 0 || c >= C) return;
    const double inv_m = 1.0 / (double)M;
    const double mean = a * inv_m;
    double var = b * inv_m - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float fmean = (float)mean, finv = (float)invstd;
    save_mean[c] = fmean;
    save_invstd[c] = finv;
    const double ub = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
    if (batch_mean) batch_mean[c] = fmean;
    if (batch_var) batch_var[c] = (float)ub;
    if (running_mean) running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * fmean;
    if (running_var) running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)ub;
    const float sc = g * finv;
    scale[c] = sc;
    shift[c] = bt - fmean * sc;
}

template <typename T, int VEC>
static int bn_fwd_fused_t(const void* x, const float* parts, int nparts, int rpp, const float* gamma, const float* beta, const void* skip, void* y,
                          unsigned char* relu_mask, float* save_mean, float* save_invstd, float* batch_mean, float* batch_var, float* running_mean, float* running_var,
                          float momentum, long M, int C, float eps, mcn_act act, void* ws, hipStream_t st) {
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    double* fold = (double*)ws;                                  // BN_FOLD_ROWS*2*C doubles fit the partial area of the workspace
    float* scale = (float*)((char*)ws + bn_parts_bytes(M, C));
    float* shift = scale + C;
    const dim3 grid(L.gx, L.gy), block(256);
    const dim3 fgrid((C + FIN_CH - 1) / FIN_CH), fblock(FIN_CH * FIN_LANES);
    if (nparts > BN_FOLD_ROWS && rpp >= 0) {
        const int rpg = (nparts + BN_FOLD_ROWS - 1) / BN_FOLD_ROWS;
        const int n = (nparts + rpg - 1) / rpg;
        int TX = 8;
        while (TX < C && TX < 256) TX *= 2;
        hipLaunchKernelGGL(bn_fold_partials_kernel, dim3((C + TX - 1) / TX, n), dim3(256), 0, st, parts, fold, nparts, C, M, rpp, rpg, TX);
        MCN_CHECK_LAUNCH();
        hipLaunchKernelGGL((bn_fwd_finalize_fused_kernel<true>), fgrid, fblock, 0, st, parts, (const double*)fold, n, rpp, M, C, gamma, beta, eps, save_mean,
                           save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, scale, shift);
    } else {
        hipLaunchKernelGGL((bn_fwd_finalize_fused_kernel<false>), fgrid, fblock, 0, st, parts, (const double*)nullptr, nparts, rpp, M, C, gamma, beta, eps,
                           save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, scale, shift);
    }
    MCN_CHECK_LAUNCH();
    const int a = (int)act;
#define BN_APPLY_F(SK, RL)                                                                                                   \
    hipLaunchKernelGGL((bn_apply_kernel<T, VEC, SK, RL>), grid, block, 0, st, (const T*)x, (const T*)skip, (T*)y, (const float*)scale, \
                       (const float*)shift, M, C, L.TX, L.TY, L.rpb, relu_mask)
    if (skip) { if (a == 1) BN_APPLY_F(true, 1); else if (a == 2) BN_APPLY_F(true, 2); else BN_APPLY_F(true, 0); }
    else { if (a == 1) BN_APPLY_F(false, 1); else if (a == 2) BN_APPLY_F(false, 2); else BN_APPLY_F(false, 0); }
#undef BN_APPLY_F
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
// rows_per_partial of mcn_conv2d_bnstats_rows: > 0 pixel rows per partial (they must cover M), 0 counted rows, < 0 compact counted rows of
// -rows_per_partial channels per block (a multiple of 4 that divides the row count into whole blocks)
static bool bad_rpp(int rpp, int nparts, int64_t M, int C) {
    if (rpp > 0) return (int64_t)nparts * rpp < M;
    if (rpp == 0) return false;
    const int bnb = -rpp, nblk = (C + bnb - 1) / bnb;
    return bnb % 4 != 0 || nblk <= 0 || nparts % nblk != 0;
}
// statistics from the conv-epilogue partials + (scale, shift) in the workspace; returns them through the out-pointers
static int bn_fused_finalize(const float* parts, int nparts, int rpp, const float* gamma, const float* beta, float* save_mean, float* save_invstd, float* batch_mean,
                             float* batch_var, float* running_mean, float* running_var, float momentum, long M, int C, float eps, void* ws, hipStream_t st,
                             float** scale_out, float** shift_out, float* dst = nullptr) {
    double* fold = (double*)ws;
    float* scale = dst ? dst : (float*)((char*)ws + bn_parts_bytes(M, C));         // dst: caller-owned [2][C] that outlives the workspace's next user
    float* shift = scale + C;
    const dim3 fgrid((C + FIN_CH - 1) / FIN_CH), fblock(FIN_CH * FIN_LANES);
    if (nparts > BN_FOLD_ROWS && rpp >= 0) {
        const int rpg = (nparts + BN_FOLD_ROWS - 1) / BN_FOLD_ROWS;
        const int n = (nparts + rpg - 1) / rpg;
        int TX = 8;
        while (TX < C && TX < 256) TX *= 2;
        hipLaunchKernelGGL(bn_fold_partials_kernel, dim3((C + TX - 1) / TX, n), dim3(256), 0, st, parts, fold, nparts, C, M, rpp, rpg, TX);
        MCN_CHECK_LAUNCH();
        hipLaunchKernelGGL((bn_fwd_finalize_fused_kernel<true>), fgrid, fblock, 0, st, parts, (const double*)fold, n, rpp, M, C, gamma, beta, eps, save_mean,
                           save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, scale, shift);
    } else {
        hipLaunchKernelGGL((bn_fwd_finalize_fused_kernel<false>), fgrid, fblock, 0, st, parts, (const double*)nullptr, nparts, rpp, M, C, gamma, beta, eps,
                           save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, scale, shift);
    }
    MCN_CHECK_LAUNCH();
    *scale_out = scale;
    *shift_out = shift;
    return MCN_OK;
}
extern "C" int mcn_maxpool_fwd_affine_relu(const void* x, const float* scale, const float* shift, void* y, int8_t* argmax, int32_t N, int32_t H, int32_t W,
                                           int32_t C, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW,
                                           mcn_dtype dtype, void* stream);
// conv -> BN(train) -> ReLU -> max-pool with the BN statistics from the conv epilogue: finalize, then ONE pass that normalises,
// rectifies and pools (the normalised tensor is never stored; its gradient path does not need it: the BN backward recomputes the
// ReLU mask from x, the pool backward routes by arg-max)
extern "C" int mcn_bn_fwd_train_fused_maxpool(const void* x, const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma,
                                              const float* beta, void* pooled, int8_t* argmax, float* save_mean, float* save_invstd, float* batch_mean,
                                              float* batch_var, float* running_mean, float* running_var, float momentum, int32_t N, int32_t H, int32_t W,
                                              int32_t C, float eps, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH,
                                              int32_t OW, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    const int64_t M = (int64_t)N * H * W;
    if (bad_rpp(rows_per_partial, nparts, M, C)) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused_maxpool: partials do not cover M rows");
    if (!x || !pooled || !argmax || !stats_partials || nparts <= 0 || !save_mean || !save_invstd || M <= 0 || C <= 0 || C % 4) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused_maxpool: bad argument");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_fwd_train_fused_maxpool: workspace too small");
    float *scale, *shift;
    if (int rc = bn_fused_finalize(stats_partials, nparts, rows_per_partial, gamma, beta, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum,
                                   (long)M, C, eps, ws, (hipStream_t)stream, &scale, &shift))
        return rc;
    return mcn_maxpool_fwd_affine_relu(x, scale, shift, pooled, argmax, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW, dtype, stream);
}

// statistics of a training-mode BN from the conv-epilogue partials WITHOUT the apply pass: saved / batch / running statistics and
// the affine (scale = gamma*invstd, shift = beta - mean*scale) into scale_shift [2][C] (caller-owned) for a consumer that folds
// the normalisation into its own pass (mcn_bn_fwd_train_fused_affskip)
extern "C" int mcn_bn_fwd_train_fused_stats(const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma, const float* beta, float* save_mean,
                                            float* save_invstd, float* batch_mean, float* batch_var, float* running_mean, float* running_var, float momentum,
                                            int64_t M, int32_t C, float eps, float* scale_shift, void* ws, size_t ws_bytes, void* stream) {
    if (bad_rpp(rows_per_partial, nparts, M, C)) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused_stats: partials do not cover M rows");
    if (!stats_partials || nparts <= 0 || !save_mean || !save_invstd || !scale_shift || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused_stats: bad argument");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_fwd_train_fused_stats: workspace too small");
    float *sc, *sh;
    return bn_fused_finalize(stats_partials, nparts, rows_per_partial, gamma, beta, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, (long)M, C,
                             eps, ws, (hipStream_t)stream, &sc, &sh, scale_shift);
}
// y = relu(bn(x) + bn_s(xs)): mcn_bn_fwd_train_fused(act = ReLU) whose residual input is the OUTPUT of another training-mode BN given
// as that BN's input xs and affine skip_scale_shift [2][C] (from mcn_bn_fwd_train_fused_stats) — same y, bit for bit, as applying
// the skip BN to a tensor first
extern "C" int mcn_bn_fwd_train_fused_affskip(const void* x, const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma, const float* beta,
                                              const void* skip_x, const float* skip_scale_shift, void* y, uint8_t* relu_mask, float* save_mean, float* save_invstd,
                                              float* batch_mean, float* batch_var, float* running_mean, float* running_var, float momentum, int64_t M, int32_t C,
                                              float eps, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (bad_rpp(rows_per_partial, nparts, M, C)) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused_affskip: partials do not cover M rows");
    if (!x || !y || !skip_x || !skip_scale_shift || !stats_partials || nparts <= 0 || !save_mean || !save_invstd || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused_affskip: bad argument");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_fwd_train_fused_affskip: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *scale, *shift;
    if (int rc = bn_fused_finalize(stats_partials, nparts, rows_per_partial, gamma, beta, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, (long)M, C,
                                   eps, ws, st, &scale, &shift))
        return rc;
    const int vec = dtype == MCN_F32 ? 4 : 8;
    if (!mcn_dtype_ok(dtype) || C % vec) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train_fused_affskip: dtype %d / C = %d (a multiple of the 16-byte chunk) unsupported", (int)dtype, C);
    const ColLayout L = make_layout((long)M, C, vec, dtype == MCN_F32 ? bn_target<float>() : bn_target<bf16_t>());
    const dim3 grid(L.gx, L.gy), block(256);
#define BN_AFFSKIP(TT, VV) hipLaunchKernelGGL((bn_apply_affskip_kernel<TT, VV>), grid, block, 0, st, (const TT*)x, (const TT*)skip_x, (TT*)y, (const float*)scale, (const float*)shift, \
                                              skip_scale_shift, skip_scale_shift + C, (long)M, C, L.TX, L.TY, L.rpb, relu_mask)
    if (dtype == MCN_F32) BN_AFFSKIP(float, 4);
    else if (dtype == MCN_BF16) BN_AFFSKIP(bf16_t, 8);
    else BN_AFFSKIP(f16_t, 8);
#undef BN_AFFSKIP
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

extern "C" int mcn_bn_fwd_train_fused(const void* x, const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma, const float* beta,
                                      const void* skip, void* y, uint8_t* relu_mask, float* save_mean, float* save_invstd, float* batch_mean,
                                      float* batch_var, float* running_mean, float* running_var, float momentum, int64_t M, int32_t C, float eps,
                                      mcn_act act, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (bad_rpp(rows_per_partial, nparts, M, C)) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused: partials do not cover M rows");
    if (!x || !y || !stats_partials || nparts <= 0 || !save_mean || !save_invstd || M <= 0 || C <= 0 || C % 4) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train_fused: bad argument");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_fwd_train_fused: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return bn_fwd_fused_t<float, 4>(x, stats_partials, nparts, rows_per_partial, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
    if (dtype == MCN_BF16) {
        if (C % 8 == 0) return bn_fwd_fused_t<bf16_t, 8>(x, stats_partials, nparts, rows_per_partial, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
        MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train_fused: bf16 needs C %% 8 == 0");
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) return bn_fwd_fused_t<f16_t, 8>(x, stats_partials, nparts, rows_per_partial, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
        MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train_fused: bf16 needs C %% 8 == 0");
    }
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train_fused: dtype %d unsupported", (int)dtype);
}

extern "C" size_t mcn_bn_relu_mask_bytes(int64_t M, int32_t C, mcn_dtype dtype) {
    const int vec = dtype == MCN_F32 ? 4 : 8;
    if (M <= 0 || C <= 0 || C % vec) return 0;
    return (size_t)M * (C / vec);
}
extern "C" int mcn_bn_fwd_train(const void* x, const float* gamma, const float* beta, const void* skip, void* y, uint8_t* relu_mask, float* save_mean,
                                float* save_invstd, float* batch_mean, float* batch_var, float* running_mean, float* running_var,
                                float momentum, int64_t M, int32_t C, float eps, mcn_act act, mcn_dtype dtype, void* ws, size_t ws_bytes,
                                void* stream) {
    if (!x || !y || !save_mean || !save_invstd || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_fwd_train: bad argument (M=%ld C=%d)", (long)M, C);
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_fwd_train: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) {
        if (C % 4 == 0) return bn_fwd_train_t<float, 4>(x, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
        return bn_fwd_train_t<float, 1>(x, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
    }
    if (dtype == MCN_BF16) {
        if (C % 8 == 0) return bn_fwd_train_t<bf16_t, 8>(x, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
        return bn_fwd_train_t<bf16_t, 1>(x, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
    } else if (dtype == MCN_F16) {
        if (C % 8 == 0) return bn_fwd_train_t<f16_t, 8>(x, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
        return bn_fwd_train_t<f16_t, 1>(x, gamma, beta, skip, y, relu_mask, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, M, C, eps, act, ws, st);
    }
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_train: dtype %d unsupported", (int)dtype);
}


template <typename T, int VEC>
static int channel_affine_t(const void* x, const float* scale, const float* shift, void* y, long M, int C, hipStream_t st) {
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    const dim3 grid(L.gx, L.gy), block(256);
    const void* skip = nullptr;
    unsigned char* relu_mask = nullptr;
    BN_APPLY(false, 0);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
#undef BN_APPLY
extern "C" int mcn_channel_affine(const void* x, const float* scale, const float* shift, void* y, int64_t M, int32_t C, mcn_dtype dtype,
                                  void* stream) {
    if (!x || !y || !scale || !shift || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "channel_affine: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return C % 4 == 0 ? channel_affine_t<float, 4>(x, scale, shift, y, M, C, st) : channel_affine_t<float, 1>(x, scale, shift, y, M, C, st);
    if (dtype == MCN_BF16) return C % 8 == 0 ? channel_affine_t<bf16_t, 8>(x, scale, shift, y, M, C, st) : channel_affine_t<bf16_t, 1>(x, scale, shift, y, M, C, st);
    else if (dtype == MCN_F16) return C % 8 == 0 ? channel_affine_t<f16_t, 8>(x, scale, shift, y, M, C, st) : channel_affine_t<f16_t, 1>(x, scale, shift, y, M, C, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "channel_affine: dtype %d unsupported", (int)dtype);
}

// inference: the caller owns all memory and passes no workspace, so the per-channel scale/shift are
// computed in the kernel prologue (registers) instead of a separate finalize launch.
template <typename T, int VEC, bool SKIP, int ACT>
__global__ __launch_bounds__(256) void bn_infer_kernel(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ var, float eps, long M,
                                                       int C, int TX, int TY, long rpb) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int c = col * VEC + i;
        const float inv = 1.f / sqrtf(var[c] + eps);
        sc[i] = (gamma ? gamma[c] : 1.f) * inv;
        sh[i] = (beta ? beta[c] : 0.f) - mean[c] * sc[i];
    }
    const long r0 = (long)blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    for (long r = r0 + ty; r < r1; r += TY) {
        const long off = r * C + (long)col * VEC;
        float v[VEC], s[VEC];
        ldv<T, VEC>(x + off, v);
        if (SKIP) ldv<T, VEC>(skip + off, s);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float o = fmaf(v[i], sc[i], sh[i]);
            if (SKIP) o += s[i];
            if (ACT == 1) o = fmaxf(o, 0.f);
            if (ACT == 2) o = o * fast_sigmoid(o);          // swish = z*sigmoid(z) (convnet.py:2553)
            v[i] = o;
        }
        stv<T, VEC>(y + off, v);
    }
}
template <typename T, int VEC>
static int bn_infer_t(const void* x, const float* gamma, const float* beta, const float* mean, const float* var, const void* skip, void* y,
                      long M, int C, float eps, mcn_act act, hipStream_t st) {
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    const dim3 grid(L.gx, L.gy), block(256);
    const int a = (int)act;
#define BN_INFER(SK, RL)                                                                                                     \
    hipLaunchKernelGGL((bn_infer_kernel<T, VEC, SK, RL>), grid, block, 0, st, (const T*)x, (const T*)skip, (T*)y, gamma, beta, mean, var, \
                       eps, M, C, L.TX, L.TY, L.rpb)
    if (skip) { if (a == 1) BN_INFER(true, 1); else if (a == 2) BN_INFER(true, 2); else BN_INFER(true, 0); }
    else { if (a == 1) BN_INFER(false, 1); else if (a == 2) BN_INFER(false, 2); else BN_INFER(false, 0); }
#undef BN_INFER
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_bn_fwd_infer(const void* x, const float* gamma, const float* beta, const float* mean, const float* var, const void* skip,
                                void* y, int64_t M, int32_t C, float eps, mcn_act act, mcn_dtype dtype, void* stream) {
    if (!x || !y || !mean || !var || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_fwd_infer: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return C % 4 == 0 ? bn_infer_t<float, 4>(x, gamma, beta, mean, var, skip, y, M, C, eps, act, st)
                                            : bn_infer_t<float, 1>(x, gamma, beta, mean, var, skip, y, M, C, eps, act, st);
    if (dtype == MCN_BF16) return C % 8 == 0 ? bn_infer_t<bf16_t, 8>(x, gamma, beta, mean, var, skip, y, M, C, eps, act, st)
                                             : bn_infer_t<bf16_t, 1>(x, gamma, beta, mean, var, skip, y, M, C, eps, act, st);
    else if (dtype == MCN_F16) return C % 8 == 0 ? bn_infer_t<f16_t, 8>(x, gamma, beta, mean, var, skip, y, M, C, eps, act, st)
                                             : bn_infer_t<f16_t, 1>(x, gamma, beta, mean, var, skip, y, M, C, eps, act, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_fwd_infer: dtype %d unsupported", (int)dtype);
}

template <typename T, int VEC>
static int bn_bwd_t(const void* dy, const void* x, const void* y_in, const unsigned char* relu_mask, const float* gamma, const float* beta, const float* save_mean,
                    const float* save_invstd, void* dx, void* dskip, float* dgamma, float* dbeta, float grad_scale, long M, int C, mcn_act act,
                    void* ws, hipStream_t st, bool frozen = false) {
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    float* part = (float*)ws;
    float* coef = (float*)((char*)ws + bn_parts_bytes(M, C));
    const dim3 grid(L.gx, L.gy), block(256);
    const bool use_mask = act == MCN_ACT_RELU && relu_mask && VEC > 1;
    const void* y = use_mask ? (const void*)relu_mask : y_in;
    const int relu = act == MCN_ACT_SWISH ? 3 : (act != MCN_ACT_RELU ? 0 : (use_mask ? 4 : (y ? 1 : 2)));
#define BN_BWD_REDUCE(RL)                                                                                                          \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC, RL>), grid, block, 2 * 256 * VEC * sizeof(float), st, (const T*)dy, (const T*)x, \
                       (const T*)y, save_mean, save_invstd, gamma, beta, part, M, C, L.TX, L.TY, L.rpb)
    if (relu == 0) BN_BWD_REDUCE(0); else if (relu == 1) BN_BWD_REDUCE(1); else if (relu == 2) BN_BWD_REDUCE(2); else if (relu == 3) BN_BWD_REDUCE(3); else BN_BWD_REDUCE(4);
#undef BN_BWD_REDUCE
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, st, (const float*)part, L.gy, M, C, gamma, save_invstd, dgamma,
                       dbeta, grad_scale, coef, frozen ? 1 : 0);
    MCN_CHECK_LAUNCH();
#define BN_BWD_APPLY(RL, DS)                                                                                                    \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, VEC, RL, DS>), grid, block, 0, st, (const T*)dy, (const T*)x, (const T*)y, save_mean, \
                       save_invstd, gamma, beta, (const float*)coef, (T*)dx, (T*)dskip, M, C, L.TX, L.TY, L.rpb)
    if (relu == 1) { if (dskip) BN_BWD_APPLY(1, true); else BN_BWD_APPLY(1, false); }
    else if (relu == 2) { if (dskip) BN_BWD_APPLY(2, true); else BN_BWD_APPLY(2, false); }
    else if (relu == 3) BN_BWD_APPLY(3, false);
    else if (relu == 4) { if (dskip) BN_BWD_APPLY(4, true); else BN_BWD_APPLY(4, false); }
    else { if (dskip) BN_BWD_APPLY(0, true); else BN_BWD_APPLY(0, false); }
#undef BN_BWD_APPLY
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_bn_bwd(const void* dy, const void* x, const void* y, const uint8_t* relu_mask, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_invstd, void* dx, void* dskip, float* dgamma, float* dbeta, float grad_scale, int64_t M,
                          int32_t C, mcn_act act, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !dx || !save_mean || !save_invstd || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_bwd: bad argument");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_bwd: workspace too small");
    if (act == MCN_ACT_SWISH && dskip) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd: swish with a fused residual is not built");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return C % 4 == 0 ? bn_bwd_t<float, 4>(dy, x, y, relu_mask, gamma, beta, save_mean, save_invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st)
                                            : bn_bwd_t<float, 1>(dy, x, y, relu_mask, gamma, beta, save_mean, save_invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st);
    if (dtype == MCN_BF16) return C % 8 == 0 ? bn_bwd_t<bf16_t, 8>(dy, x, y, relu_mask, gamma, beta, save_mean, save_invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st)
                                             : bn_bwd_t<bf16_t, 1>(dy, x, y, relu_mask, gamma, beta, save_mean, save_invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st);
    else if (dtype == MCN_F16) return C % 8 == 0 ? bn_bwd_t<f16_t, 8>(dy, x, y, relu_mask, gamma, beta, save_mean, save_invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st)
                                             : bn_bwd_t<f16_t, 1>(dy, x, y, relu_mask, gamma, beta, save_mean, save_invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd: dtype %d unsupported", (int)dtype);
}

// ---- the same two passes in BLOCK form (round 3) for the stem geometry: H, W even, no leading pad (TF SAME on an even map) ----------------
// A thread owns one 16-byte chunk of the 2 x 2 input pixels (2a + dy, 2b + dx): the only windows that see them are (a, b), (a-1, b),
// (a, b-1), (a-1, b-1), so their pooled gradients and arg-max codes are loaded ONCE for the four pixels (the per-pixel form gathers
// four windows per pixel — 96 bytes of L1 / L2 traffic per 16-byte chunk of x — and pays two integer divisions per pixel: 2.1-3.5 TB/s
// on the largest tensor of the network).  Contributions are summed in the per-pixel form's order: the routed gradient is bit-identical.
// raw form: the four windows' chunks stay packed (16 + 8 bytes each) and a pixel's gradient is unpacked when that pixel is processed — the
// first build unpacked everything up front (2 x 2 x 8 floats each of x, the routed gradient and the window gradients): 175 VGPRs in
// the 2-byte apply kernel = two waves per SIMD, slower than the per-pixel form.
template <typename T, int VEC>
struct PoolBlock {
    Chunk<T> gp[2][2];
    unsigned long long ac[2][2];          // arg-max codes, one byte per channel (all 0xff: window outside the map)
    long xoff[2][2];
};
template <typename T, int VEC>
__device__ __forceinline__ void pool_block_load(const T* __restrict__ dp, const int8_t* __restrict__ arg, const PoolRoute& P, unsigned q, int C, int col, PoolBlock<T, VEC>& B) {
    const unsigned W2 = (unsigned)P.W >> 1, H2 = (unsigned)P.H >> 1;
    const unsigned q2 = q / W2;
    const int b = (int)(q - q2 * W2);
    const int n = (int)(q2 / H2), a = (int)(q2 - (unsigned)n * H2);
#pragma unroll
    for (int al = 0; al < 2; ++al)
#pragma unroll
        for (int be = 0; be < 2; ++be) {
            const int oy = a - al, ox = b - be;
            const bool ok = oy >= 0 && ox >= 0 && oy < P.OH && ox < P.OW;
            const long o = (((long)n * P.OH + (ok ? oy : 0)) * P.OW + (ok ? ox : 0)) * C + (long)col * VEC;
            B.gp[al][be] = load_chunk<T>(dp + o);
            unsigned long long c8;
            if constexpr (VEC == 8) c8 = *reinterpret_cast<const unsigned long long*>(arg + o);
            else c8 = *reinterpret_cast<const unsigned*>(arg + o);
            B.ac[al][be] = ok ? c8 : ~0ull;
        }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) B.xoff[dy][dx] = (((long)n * P.H + 2 * a + dy) * P.W + 2 * b + dx) * C + (long)col * VEC;
}
// routed gradient of pixel (dy, dx) of the block, contributions summed in the per-pixel form's order, rounded to the storage type
template <typename T, int VEC>
__device__ __forceinline__ void pool_block_pixel(const PoolBlock<T, VEC>& B, int dy, int dx, float (&g)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
#pragma unroll
    for (int al = 0; al < 2; ++al)
#pragma unroll
        for (int be = 0; be < 2; ++be) {
            if (dy + 2 * al > 2 || dx + 2 * be > 2) continue;      // window (a - al, b - be) holds the pixel at filter position (dy + 2 al, dx + 2 be)
            const unsigned code = (unsigned)((dy + 2 * al) * 3 + (dx + 2 * be));
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (((unsigned)(B.ac[al][be] >> (8 * i)) & 0xffu) == code) g[i] += B.gp[al][be].get(i);
        }
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = to_f32(from_f32<T>(g[i]));
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_poolblk_kernel(const T* __restrict__ dp, const int8_t* __restrict__ arg, const PoolRoute P, const T* __restrict__ x,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ part, long MB, int C, int TX, int TY, long rpb) {
    extern __shared__ float red[];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    const bool active = ty < TY && col * VEC < C;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], is[VEC], sc[VEC], sh[VEC];
        ldc<VEC>(mean + col * VEC, mu);
        ldc<VEC>(invstd + col * VEC, is);
        ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
        ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc[i] *= is[i];
            sh[i] -= mu[i] * sc[i];
        }
        const long q0 = (long)blockIdx.y * rpb, q1 = min(MB, q0 + rpb);
        for (long q = q0 + ty; q < q1; q += TY) {
            PoolBlock<T, VEC> B;
            pool_block_load<T, VEC>(dp, arg, P, (unsigned)q, C, col, B);
            Chunk<T> xr[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) xr[dy][dx] = load_chunk<T>(x + B.xoff[dy][dx]);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float g[VEC];
                    pool_block_pixel<T, VEC>(B, dy, dx, g);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float v = xr[dy][dx].get(i);
                        const float gg = fmaf(v, sc[i], sh[i]) > 0.f ? g[i] : 0.f;
                        s1[i] += gg;
                        s2[i] = fmaf(gg, (v - mu[i]) * is[i], s2[i]);
                    }
                }
        }
    }
    float* r1p = red;
    float* r2p = red + 256 * VEC;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            r1p[(ty * TX + tx) * VEC + i] = s1[i];
            r2p[(ty * TX + tx) * VEC + i] = s2[i];
        }
    }
    __syncthreads();
    if (ty == 0 && col * VEC < C) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < TY; ++k) {
                a += r1p[(k * TX + tx) * VEC + i];
                b += r2p[(k * TX + tx) * VEC + i];
            }
            part[((long)blockIdx.y * 2 + 0) * C + col * VEC + i] = a;
            part[((long)blockIdx.y * 2 + 1) * C + col * VEC + i] = b;
        }
    }
}
template <typename T, int VEC>
__global__ __launch_bounds__(256, 4) void bn_bwd_apply_poolblk_kernel(const T* __restrict__ dp, const int8_t* __restrict__ arg, const PoolRoute P, const T* __restrict__ x,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ coef, T* __restrict__ dx, long MB, int C,
                                                                   int TX, int TY, long rpb) {
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int col = blockIdx.x * TX + tx;
    if (ty >= TY || col * VEC >= C) return;
    float mu[VEC], is[VEC], ca[VEC], cb[VEC], cc[VEC], sc[VEC], sh[VEC];
    ldc<VEC>(mean + col * VEC, mu);
    ldc<VEC>(invstd + col * VEC, is);
    ldc<VEC>(coef + col * VEC, ca);
    ldc<VEC>(coef + C + col * VEC, cb);
    ldc<VEC>(coef + 2 * C + col * VEC, cc);
    ldc<VEC>(gamma ? gamma + col * VEC : nullptr, sc, 1.f);
    ldc<VEC>(beta ? beta + col * VEC : nullptr, sh, 0.f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sc[i] *= is[i];
        sh[i] -= mu[i] * sc[i];
    }
    const long q0 = (long)blockIdx.y * rpb, q1 = min(MB, q0 + rpb);
    for (long q = q0 + ty; q < q1; q += TY) {
        PoolBlock<T, VEC> B;
        pool_block_load<T, VEC>(dp, arg, P, (unsigned)q, C, col, B);
        Chunk<T> xr[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) xr[dy][dxx] = load_chunk<T>(x + B.xoff[dy][dxx]);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                float g[VEC];
                pool_block_pixel<T, VEC>(B, dy, dxx, g);
                Chunk<T> o;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float v = xr[dy][dxx].get(i);
                    const float gg = fmaf(v, sc[i], sh[i]) > 0.f ? g[i] : 0.f;
                    const float xh = (v - mu[i]) * is[i];
                    o.set(i, ca[i] * (gg - cb[i] - xh * cc[i]));
                }
                store_chunk<T>(dx + B.xoff[dy][dxx], o);
                __builtin_amdgcn_sched_barrier(0);          // one pixel at a time: the scheduler would otherwise unpack all four up front
            }
    }
}

template <typename T, int VEC>
static int bn_bwd_pool_t(const void* dp, const int8_t* arg, const PoolRoute& P, const void* x, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, void* dx, float* dgamma, float* dbeta, float grad_scale, long M, int C, void* ws, hipStream_t st) {
    float* part = (float*)ws;
    float* coef = (float*)((char*)ws + bn_parts_bytes(M, C));
    // MCN_BN_POOL_BLOCK=0: the per-pixel form everywhere
    static const int blk = [] { const char* e = getenv("MCN_BN_POOL_BLOCK"); return e ? atoi(e) : 1; }();
    if (blk && P.padT == 0 && P.padL == 0 && P.H % 2 == 0 && P.W % 2 == 0 && 2 * P.OH == P.H && 2 * P.OW == P.W) {
        const long MB = M / 4;                                       // 2 x 2 pixel blocks
        const ColLayout L = make_layout(MB, C, VEC, bn_target<T>());
        const dim3 grid(L.gx, L.gy), block(256);
        hipLaunchKernelGGL((bn_bwd_reduce_poolblk_kernel<T, VEC>), grid, block, 2 * 256 * VEC * sizeof(float), st, (const T*)dp, arg, P, (const T*)x, save_mean, save_invstd, gamma,
                           beta, part, MB, C, L.TX, L.TY, L.rpb);
        MCN_CHECK_LAUNCH();
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, st, (const float*)part, L.gy, M, C, gamma, save_invstd, dgamma,
                           dbeta, grad_scale, coef, 0);
        MCN_CHECK_LAUNCH();
        hipLaunchKernelGGL((bn_bwd_apply_poolblk_kernel<T, VEC>), grid, block, 0, st, (const T*)dp, arg, P, (const T*)x, save_mean, save_invstd, gamma, beta, (const float*)coef,
                           (T*)dx, MB, C, L.TX, L.TY, L.rpb);
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    const dim3 grid(L.gx, L.gy), block(256);
    hipLaunchKernelGGL((bn_bwd_reduce_pool_kernel<T, VEC>), grid, block, 2 * 256 * VEC * sizeof(float), st, (const T*)dp, arg, P, (const T*)x, save_mean, save_invstd, gamma, beta,
                       part, M, C, L.TX, L.TY, L.rpb);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, st, (const float*)part, L.gy, M, C, gamma, save_invstd, dgamma,
                       dbeta, grad_scale, coef, 0);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<T, VEC>), grid, block, 0, st, (const T*)dp, arg, P, (const T*)x, save_mean, save_invstd, gamma, beta, (const float*)coef, (T*)dx,
                       M, C, L.TX, L.TY, L.rpb);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
// mcn_bn_bwd(act = ReLU, mask recomputed from x) for a BN whose output feeds ONLY a 3x3 / stride-2 max-pool: takes the POOLED
// gradient dy_pooled [N,OH,OW,C] and the pool's arg-max instead of the full-resolution dy (same result as mcn_maxpool_bwd
// followed by mcn_bn_bwd: the routed gradient itself is bit-identical; dgamma / dbeta and with them dx agree up to the order of the fp32 sums —
// the 2 x 2 block form used for even maps without a leading pad sums the pixels in another order than mcn_bn_bwd).
extern "C" int mcn_bn_bwd_maxpool(const void* dy_pooled, const int8_t* argmax, const void* x, const float* gamma, const float* beta, const float* save_mean,
                                  const float* save_invstd, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int32_t H, int32_t W, int32_t C,
                                  int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype, void* ws,
                                  size_t ws_bytes, void* stream) {
    const int64_t M = (int64_t)N * H * W;
    if (!dy_pooled || !argmax || !x || !dx || !save_mean || !save_invstd || M <= 0 || C <= 0 || OH <= 0 || OW <= 0) MCN_FAIL(MCN_E_BADARG, "bn_bwd_maxpool: bad argument");
    if (KH != 3 || KW != 3 || SH != 2 || SW != 2 || padT < 0 || padL < 0 || padT > 2 || padL > 2) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_maxpool: 3x3 / stride-2 pools only");
    if (M >= 0xffffffffll) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_maxpool: more than 2^32 pixels");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_bwd_maxpool: workspace too small");
    const PoolRoute P = {H, W, OH, OW, padT, padL};
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32 && C % 4 == 0) return bn_bwd_pool_t<float, 4>(dy_pooled, argmax, P, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st);
    if (dtype == MCN_BF16 && C % 8 == 0) return bn_bwd_pool_t<bf16_t, 8>(dy_pooled, argmax, P, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st);
    if (dtype == MCN_F16 && C % 8 == 0) return bn_bwd_pool_t<f16_t, 8>(dy_pooled, argmax, P, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_maxpool: dtype %d / C = %d (a multiple of the 16-byte chunk) unsupported", (int)dtype, C);
}

// launch shape of se_bwd_pre_kernel, shared with the consumers of its sums [N * HS][5][C] (mcn_se_bwd_sums_floats)
struct SeBwdLayout { int TX, TY, gx, HS; long rps; };
static SeBwdLayout se_bwd_layout(int N, long HW, int C, int vec) {
    SeBwdLayout L;
    const int cols = C / vec;
    L.TX = 1;
    for (int d = 1; d <= 32 && d <= cols; ++d)
        if (cols % d == 0) L.TX = d;
    if (L.TX < 8 && cols > 32) L.TX = 32;                         // awkward chunk counts: accept idle lanes
    L.TY = 256 / L.TX;
    L.gx = (cols + L.TX - 1) / L.TX;
    // ~4 workgroups per CU, every slice at least 4 row passes of the block
    static const int per_cu = [] { const char* e = getenv("MCN_SE_PRE_WGS"); return e ? atoi(e) : 4; }();      // (same-box sweep 2 / 4 / 8 / 16 / 32: 27.49 / 27.28 / 27.51 / 28.11 / 28.78 ms per step)
    long hs = ((long)per_cu * MCN_NUM_CU + (long)L.gx * N - 1) / ((long)L.gx * (N > 0 ? N : 1));
    const long cap = (HW + 4L * L.TY - 1) / (4L * L.TY);
    if (hs > cap) hs = cap;
    if (hs < 1) hs = 1;
    L.rps = (HW + hs - 1) / hs;
    L.HS = (int)((HW + L.rps - 1) / L.rps);
    return L;
}
extern "C" size_t mcn_se_bwd_sums_floats(int32_t N, int64_t HW, int32_t C, mcn_dtype dtype) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    const int vec = dtype == MCN_F32 ? 4 : 8;
    if (C % vec) return 0;
    return (size_t)N * se_bwd_layout(N, (long)HW, C, vec).HS * 5 * C;
}
template <typename T, int VEC>
static int bn_bwd_se_t(const void* dy, const void* m, const void* dgap, long HW, const void* x, const float* gamma, const float* beta, const float* save_mean,
                       const float* save_invstd, void* dx, float* dgamma, float* dbeta, float grad_scale, long M, int C, void* ws, hipStream_t st, const float* sums = nullptr) {
    const ColLayout L = make_layout(M, C, VEC, bn_target<T>());
    float* part = (float*)ws;
    float* coef = (float*)((char*)ws + bn_parts_bytes(M, C));
    const dim3 grid(L.gx, L.gy), block(256);
    int nparts = L.gy;
    if (sums) {                                                   // the per-image sums of mcn_channel_scale_bwd_dm_bnsums: no pass over the activations
        const int N = (int)(M / HW);
        const int rows = N < 64 ? N : 64;                         // partial rows: the workspace holds L.gy of them
        const int use = rows < L.gy ? rows : L.gy;
        const int NB = (N + use - 1) / use;
        nparts = (N + NB - 1) / NB;
        hipLaunchKernelGGL((se_bwd_sums_kernel<T>), dim3((C + 255) / 256, nparts), dim3(256), 0, st, sums, (const T*)m, (const T*)dgap, part, N, HW, C,
                           se_bwd_layout(N, HW, C, VEC).HS, NB);
    } else {
        hipLaunchKernelGGL((bn_bwd_reduce_se_kernel<T, VEC>), grid, block, 2 * 256 * VEC * sizeof(float), st, (const T*)dy, (const T*)m, (const T*)dgap, HW, (const T*)x, save_mean,
                           save_invstd, gamma, beta, part, M, C, L.TX, L.TY, L.rpb);
    }
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, st, (const float*)part, nparts, M, C, gamma, save_invstd, dgamma,
                       dbeta, grad_scale, coef, 0);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_bwd_apply_se_kernel<T, VEC>), grid, block, 0, st, (const T*)dy, (const T*)m, (const T*)dgap, HW, (const T*)x, save_mean, save_invstd, gamma, beta,
                       (const float*)coef, (T*)dx, M, C, L.TX, L.TY, L.rpb);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
// mcn_bn_bwd(act = swish) for the BN in front of a squeeze-excite block: dy is the gradient of the SCALED tensor, se_mask [N,C] the SE
// factors, dgap [N,C] the gradient of the SE branch's global average pool; the gradient of the BN's output is composed inside the
// two passes (bit-identical to mcn_channel_scale_bwd + mcn_global_avgpool_bwd_acc + mcn_bn_bwd)
extern "C" int mcn_bn_bwd_se(const void* dy, const void* se_mask, const void* dgap, const void* x, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype,
                             void* ws, size_t ws_bytes, void* stream) {
    const int64_t M = (int64_t)N * HW;
    if (!dy || !se_mask || !dgap || !x || !dx || !save_mean || !save_invstd || N <= 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_bwd_se: bad argument");
    if (M >= 0xffffffffll) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_se: more than 2^32 pixels");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_bwd_se: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32 && C % 4 == 0) return bn_bwd_se_t<float, 4>(dy, se_mask, dgap, (long)HW, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st);
    if (dtype == MCN_BF16 && C % 8 == 0) return bn_bwd_se_t<bf16_t, 8>(dy, se_mask, dgap, (long)HW, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st);
    if (dtype == MCN_F16 && C % 8 == 0) return bn_bwd_se_t<f16_t, 8>(dy, se_mask, dgap, (long)HW, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_se: dtype %d / C = %d (a multiple of the 16-byte chunk) unsupported", (int)dtype, C);
}

// mcn_bn_bwd_se with the reduction pass replaced by the per-image sums of mcn_channel_scale_bwd_dm_bnsums (sums: fp32 [N * HS][5][C], mcn_se_bwd_sums_floats() of them)
extern "C" int mcn_bn_bwd_se_sums(const void* dy, const void* se_mask, const void* dgap, const void* x, const float* gamma, const float* beta, const float* save_mean,
                                  const float* save_invstd, const float* sums, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int64_t HW, int32_t C,
                                  mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    const int64_t M = (int64_t)N * HW;
    if (!dy || !se_mask || !dgap || !x || !dx || !save_mean || !save_invstd || !sums || N <= 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_bwd_se_sums: bad argument");
    if (M >= 0xffffffffll) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_se_sums: more than 2^32 pixels");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_bwd_se_sums: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32 && C % 4 == 0) return bn_bwd_se_t<float, 4>(dy, se_mask, dgap, (long)HW, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st, sums);
    if (dtype == MCN_BF16 && C % 8 == 0) return bn_bwd_se_t<bf16_t, 8>(dy, se_mask, dgap, (long)HW, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st, sums);
    if (dtype == MCN_F16 && C % 8 == 0) return bn_bwd_se_t<f16_t, 8>(dy, se_mask, dgap, (long)HW, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, (long)M, C, ws, st, sums);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_se_sums: dtype %d / C = %d (a multiple of the 16-byte chunk) unsupported", (int)dtype, C);
}
template <typename T, int VEC>
static int se_bwd_pre_t(const void* dy, const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dm, float* sums, int N, long HW, int C,
                        hipStream_t st) {
    const SeBwdLayout L = se_bwd_layout(N, HW, C, VEC);
    const dim3 grid((unsigned)L.gx, (unsigned)(N * L.HS)), block(256);
    hipLaunchKernelGGL((se_bwd_pre_kernel<T, VEC>), grid, block, (size_t)L.TY * L.TX * VEC * sizeof(float), st, (const T*)dy, (const T*)x, save_mean, save_invstd, gamma, beta, sums, HW, C,
                       L.TX, L.TY, L.HS, L.rps);
    MCN_CHECK_LAUNCH();
    const long NC = (long)N * C;
    hipLaunchKernelGGL((se_bwd_dm_fold_kernel<T>), dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, st, (const float*)sums, (T*)dm, NC, C, L.HS);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
// dm[n,c] = sum_hw dy * x_se (what mcn_channel_scale_bwd_dm gives on the stored x_se) from the BN's INPUT x — x_se = round(swish(bn(x))) is rebuilt on
// the fly — plus the per-image-slice sums [N * HS][5][C] (dm partial, sum dy s', sum dy s' xh, sum s', sum s' xh) that let mcn_bn_bwd_se_sums skip its reduction pass
extern "C" int mcn_channel_scale_bwd_dm_bnsums(const void* dy, const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dm,
                                               float* sums, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void* stream) {
    if (!dy || !x || !dm || !sums || !save_mean || !save_invstd || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "channel_scale_bwd_dm_bnsums: bad argument");
    if (N == 0) return MCN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32 && C % 4 == 0) return se_bwd_pre_t<float, 4>(dy, x, gamma, beta, save_mean, save_invstd, dm, sums, N, (long)HW, C, st);
    if (dtype == MCN_BF16 && C % 8 == 0) return se_bwd_pre_t<bf16_t, 8>(dy, x, gamma, beta, save_mean, save_invstd, dm, sums, N, (long)HW, C, st);
    if (dtype == MCN_F16 && C % 8 == 0) return se_bwd_pre_t<f16_t, 8>(dy, x, gamma, beta, save_mean, save_invstd, dm, sums, N, (long)HW, C, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_bwd_dm_bnsums: dtype %d / C = %d (a multiple of the 16-byte chunk) unsupported", (int)dtype, C);
}

// groups of rpg rows of [nparts][2][C] sums -> fold[(g*2 + {0,1})*C + c] (summed in double, fixed order)
__global__ __launch_bounds__(256) void bn_bwd_fold_partials_kernel(const float* __restrict__ part, float* __restrict__ fold, int nparts, int C, int rpg, int TX) {
    __shared__ double red[256 * 2];
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = blockIdx.x * TX + tx;
    const int r0 = blockIdx.y * rpg, r1 = min(nparts, r0 + rpg);
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int r = r0 + ty; r < r1; r += TY) {
            a += (double)part[((long)r * 2 + 0) * C + c];
            b += (double)part[((long)r * 2 + 1) * C + c];
        }
    red[threadIdx.x * 2] = a;
    red[threadIdx.x * 2 + 1] = b;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    a = 0.0, b = 0.0;
    for (int k = 0; k < TY; ++k) {
        a += red[(k * TX + tx) * 2];
        b += red[(k * TX + tx) * 2 + 1];
    }
    fold[((long)blockIdx.y * 2 + 0) * C + c] = (float)a;
    fold[((long)blockIdx.y * 2 + 1) * C + c] = (float)b;
}

// BN + ReLU backward whose reduction pass ran in the epilogue of the dgrad that produced dy (mcn_conv2d_dgrad_bnred): finalize from
// those partial rows ([nparts][2][C]: sum dy', sum dy' * x), then the apply pass with the forward's ReLU byte mask.
extern "C" int mcn_bn_bwd_from_partials(const void* dy, const void* x, const uint8_t* relu_mask, const float* gamma, const float* beta, const float* save_mean,
                                        const float* save_invstd, const float* red_partials, int32_t nparts, void* dx, float* dgamma, float* dbeta,
                                        float grad_scale, int64_t M, int32_t C, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !relu_mask || !dx || !save_mean || !save_invstd || !red_partials || nparts <= 0 || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_bwd_from_partials: bad argument");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_bwd_from_partials: workspace too small");
    const int vec = dtype == MCN_F32 ? 4 : 8;
    if (!mcn_dtype_ok(dtype) || C % vec) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_from_partials: dtype %d / C = %d (a multiple of the 16-byte chunk) unsupported", (int)dtype, C);
    hipStream_t st = (hipStream_t)stream;
    float* coef = (float*)((char*)ws + bn_parts_bytes(M, C));
    const float* rows = red_partials;
    if (nparts > BN_FOLD_ROWS) {
        // a dgrad leaves one row per (M tile, wave row) — 12 000-25 000 for the 56 x 56 layers: fold them with the whole chip first (the
        // finalize kernel has C / FIN_CH workgroups)
        const int rpg = (nparts + BN_FOLD_ROWS - 1) / BN_FOLD_ROWS;
        const int n = (nparts + rpg - 1) / rpg;
        int TX = 8;
        while (TX < C && TX < 256) TX *= 2;
        hipLaunchKernelGGL(bn_bwd_fold_partials_kernel, dim3((C + TX - 1) / TX, n), dim3(256), 0, st, red_partials, (float*)ws, nparts, C, rpg, TX);
        MCN_CHECK_LAUNCH();
        rows = (const float*)ws;
        nparts = n;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_LANES), 0, st, rows, nparts, (long)M, C, gamma, save_invstd, dgamma,
                       dbeta, grad_scale, coef, 0, save_mean);
    MCN_CHECK_LAUNCH();
    const ColLayout L = make_layout((long)M, C, vec, dtype == MCN_F32 ? bn_target<float>() : bn_target<bf16_t>());
    const dim3 grid(L.gx, L.gy), block(256);
#define BN_BWD_APPLY_M(TT, VV) hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, VV, 4, false>), grid, block, 0, st, (const TT*)dy, (const TT*)x, (const TT*)relu_mask, save_mean, save_invstd, \
                                                  gamma, beta, (const float*)coef, (TT*)dx, (TT*)nullptr, (long)M, C, L.TX, L.TY, L.rpb)
    if (dtype == MCN_F32) BN_BWD_APPLY_M(float, 4);
    else if (dtype == MCN_BF16) BN_BWD_APPLY_M(bf16_t, 8);
    else BN_BWD_APPLY_M(f16_t, 8);
#undef BN_BWD_APPLY_M
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// ---- backward of the frozen-statistics BN (fused_batch_norm(is_training=False) inside a training graph) ----------------
__global__ void bn_invstd_kernel(const float* __restrict__ var, float eps, float* __restrict__ out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) out[c] = 1.f / sqrtf(var[c] + eps);
}
extern "C" int mcn_bn_bwd_frozen(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* mean, const float* var,
                                 float eps, void* dx, void* dskip, float* dgamma, float* dbeta, float grad_scale, int64_t M, int32_t C, mcn_act act,
                                 mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !dx || !mean || !var || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bn_bwd_frozen: bad argument");
    if (!ws || ws_bytes < mcn_bn_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bn_bwd_frozen: workspace too small");
    if (act == MCN_ACT_RELU && !y) MCN_FAIL(MCN_E_BADARG, "bn_bwd_frozen: the ReLU mask is taken from the stored output y");
    if (act == MCN_ACT_SWISH && dskip) MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_frozen: swish with a fused residual is not built");
    hipStream_t st = (hipStream_t)stream;
    float* invstd = (float*)((char*)ws + bn_parts_bytes(M, C)) + 3 * (size_t)C;
    hipLaunchKernelGGL(bn_invstd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, var, eps, invstd, C);
    MCN_CHECK_LAUNCH();
    if (dtype == MCN_F32) return C % 4 == 0 ? bn_bwd_t<float, 4>(dy, x, y, nullptr, gamma, beta, mean, invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st, true)
                                            : bn_bwd_t<float, 1>(dy, x, y, nullptr, gamma, beta, mean, invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st, true);
    if (dtype == MCN_BF16) return C % 8 == 0 ? bn_bwd_t<bf16_t, 8>(dy, x, y, nullptr, gamma, beta, mean, invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st, true)
                                             : bn_bwd_t<bf16_t, 1>(dy, x, y, nullptr, gamma, beta, mean, invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st, true);
    else if (dtype == MCN_F16) return C % 8 == 0 ? bn_bwd_t<f16_t, 8>(dy, x, y, nullptr, gamma, beta, mean, invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st, true)
                                             : bn_bwd_t<f16_t, 1>(dy, x, y, nullptr, gamma, beta, mean, invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, ws, st, true);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bn_bwd_frozen: dtype %d unsupported", (int)dtype);
}
