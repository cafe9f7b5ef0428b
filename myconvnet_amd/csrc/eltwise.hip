// eltwise.hip — HBM-bound element-wise kernels: ReLU, residual add(+ReLU), gradient accumulation,
// casts, input preparation, one-hot labels.  16-byte vector accesses (8 bf16 / 4 fp32 per lane),
// grid-stride loops capped at 256 CUs x 8 blocks.
#include "common.h"

static inline unsigned ew_blocks(long nvec) {
    long b = (nvec + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

enum { EW_RELU = 0, EW_RELU_BWD = 1, EW_ADD = 2, EW_ADD_RELU = 3, EW_ACC = 4, EW_SWISH = 5, EW_SIGMOID = 6, EW_SWISH_BWD = 7, EW_SIGMOID_BWD = 8,
       EW_RELU6 = 9, EW_LRELU = 10, EW_TANH = 11, EW_RELU6_BWD = 12, EW_LRELU_BWD = 13, EW_TANH_BWD = 14 };

template <int OP>
__device__ __forceinline__ float ew_apply(float va, float vb, float p = 0.f) {
    if (OP == EW_RELU) return fmaxf(va, 0.f);
    if (OP == EW_RELU6) return fminf(fmaxf(va, 0.f), 6.f);                   // tf.nn.relu6, convnet.py:2539-2540
    if (OP == EW_LRELU) return va > 0.f ? va : p * va;                       // tf.nn.leaky_relu(alpha), convnet.py:2542-2545
    if (OP == EW_TANH) return tanhf(va);                                     // tf.nn.tanh, convnet.py:2547
    if (OP == EW_RELU6_BWD) return (vb > 0.f && vb < 6.f) ? va : 0.f;        // Relu6Grad: a = dy, b = y (0 < y < 6 <=> 0 < x < 6)
    if (OP == EW_LRELU_BWD) return vb > 0.f ? va : p * va;                   // LeakyReluGrad: a = dy, b = x
    if (OP == EW_TANH_BWD) return va * (1.f - vb * vb);                      // TanhGrad: a = dy, b = y
    if (OP == EW_RELU_BWD) return vb > 0.f ? va : 0.f;                       // a = dy, b = y
    if (OP == EW_ADD || OP == EW_ACC) return va + vb;
    if (OP == EW_ADD_RELU) return fmaxf(va + vb, 0.f);
    if (OP == EW_SWISH) return va / (1.f + expf(-va));                       // x*sigmoid(x), convnet.py:2553-2556
    if (OP == EW_SIGMOID) return 1.f / (1.f + expf(-va));                    // convnet.py:2550
    if (OP == EW_SWISH_BWD) {                                                // a = dy, b = x
        const float sg = 1.f / (1.f + expf(-vb));
        return va * sg * (1.f + vb * (1.f - sg));
    }
    return va * vb * (1.f - vb);                                             // EW_SIGMOID_BWD: a = dy, b = y
}
template <int OP>
struct EwUnary { static constexpr bool value = OP == EW_RELU || OP == EW_SWISH || OP == EW_SIGMOID || OP == EW_RELU6 || OP == EW_LRELU || OP == EW_TANH; };

template <typename T, int OP>
__global__ __launch_bounds__(256) void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n, float p) {
    constexpr int CE = VecTraits<T>::CE;
    const long nvec = n / CE;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        Chunk<T> ca = load_chunk<T>(a + i * CE), cb, co;
        if (!EwUnary<OP>::value) cb = load_chunk<T>(b + i * CE);
#pragma unroll
        for (int k = 0; k < CE; ++k) co.set(k, ew_apply<OP>(ca.get(k), EwUnary<OP>::value ? 0.f : cb.get(k), p));
        store_chunk<T>(y + i * CE, co);
    }
    // scalar tail
    const long t0 = nvec * CE;
    for (long i = t0 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = from_f32<T>(ew_apply<OP>(to_f32(a[i]), EwUnary<OP>::value ? 0.f : to_f32(b[i]), p));
}

template <int OP>
static int ew_dispatch(const void* a, const void* b, void* y, long n, mcn_dtype dt, hipStream_t st, const char* name, float p = 0.f) {
    if (n < 0 || !a || !y) MCN_FAIL(MCN_E_BADARG, "%s: bad argument", name);
    if (n == 0) return MCN_OK;
    if (dt == MCN_F32)
        hipLaunchKernelGGL((ew_kernel<float, OP>), dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)y, n, p);
    else if (dt == MCN_BF16)
        hipLaunchKernelGGL((ew_kernel<bf16_t, OP>), dim3(ew_blocks(n / 8 + 1)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n, p);
    else if (dt == MCN_F16)
        hipLaunchKernelGGL((ew_kernel<f16_t, OP>), dim3(ew_blocks(n / 8 + 1)), dim3(256), 0, st, (const f16_t*)a, (const f16_t*)b, (f16_t*)y, n, p);
    else
        MCN_FAIL(MCN_E_UNSUPPORTED, "%s: dtype %d unsupported", name, (int)dt);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

extern "C" int mcn_relu_fwd(const void* x, void* y, int64_t n, mcn_dtype dtype, void* stream) {
    return ew_dispatch<EW_RELU>(x, nullptr, y, n, dtype, (hipStream_t)stream, "relu_fwd");
}
extern "C" int mcn_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, mcn_dtype dtype, void* stream) {
    if (!y) MCN_FAIL(MCN_E_BADARG, "relu_bwd: null y");
    return ew_dispatch<EW_RELU_BWD>(dy, y, dx, n, dtype, (hipStream_t)stream, "relu_bwd");
}
extern "C" int mcn_add_relu_fwd(const void* a, const void* b, void* y, int64_t n, mcn_act act, mcn_dtype dtype, void* stream) {
    if (!b) MCN_FAIL(MCN_E_BADARG, "add_relu_fwd: null b");
    if (act == MCN_ACT_RELU) return ew_dispatch<EW_ADD_RELU>(a, b, y, n, dtype, (hipStream_t)stream, "add_relu_fwd");
    return ew_dispatch<EW_ADD>(a, b, y, n, dtype, (hipStream_t)stream, "add_relu_fwd");
}
extern "C" int mcn_add_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, mcn_act act, mcn_dtype dtype, void* stream) {
    if (act == MCN_ACT_RELU) {
        if (!y) MCN_FAIL(MCN_E_BADARG, "add_relu_bwd: null y");
        return ew_dispatch<EW_RELU_BWD>(dy, y, dx, n, dtype, (hipStream_t)stream, "add_relu_bwd");
    }
    if (dx != dy) {
        if (hipMemcpyAsync(dx, dy, (size_t)n * mcn_dtype_size(dtype), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            MCN_FAIL(MCN_E_LAUNCH, "add_relu_bwd: copy failed");
    }
    return MCN_OK;
}
/* element-wise activations (the reference's dispatcher, convnet.py:2514-2556): relu / relu6 / lrelu(alpha) / tanh / sigmoid / swish */
extern "C" int mcn_act_fwd_p(const void* x, void* y, int64_t n, mcn_act act, float param, mcn_dtype dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (act == MCN_ACT_RELU) return ew_dispatch<EW_RELU>(x, nullptr, y, n, dtype, st, "act_fwd");
    if (act == MCN_ACT_SWISH) return ew_dispatch<EW_SWISH>(x, nullptr, y, n, dtype, st, "act_fwd");
    if (act == MCN_ACT_SIGMOID) return ew_dispatch<EW_SIGMOID>(x, nullptr, y, n, dtype, st, "act_fwd");
    if (act == MCN_ACT_RELU6) return ew_dispatch<EW_RELU6>(x, nullptr, y, n, dtype, st, "act_fwd");
    if (act == MCN_ACT_LRELU) return ew_dispatch<EW_LRELU>(x, nullptr, y, n, dtype, st, "act_fwd", param);
    if (act == MCN_ACT_TANH) return ew_dispatch<EW_TANH>(x, nullptr, y, n, dtype, st, "act_fwd");
    MCN_FAIL(MCN_E_UNSUPPORTED, "act_fwd: activation %d unsupported", (int)act);
}
extern "C" int mcn_act_fwd(const void* x, void* y, int64_t n, mcn_act act, mcn_dtype dtype, void* stream) {
    return mcn_act_fwd_p(x, y, n, act, 0.2f, dtype, stream);                 // alpha = 0.2: the reference's default (convnet.py:2543-2544)
}
/* dx = dy * act'(.)  — relu, relu6, tanh and sigmoid differentiate through the stored output y, swish and lrelu through the input x */
extern "C" int mcn_act_bwd_p(const void* dy, const void* x, const void* y, void* dx, int64_t n, mcn_act act, float param, mcn_dtype dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (act == MCN_ACT_RELU || act == MCN_ACT_RELU6 || act == MCN_ACT_TANH || act == MCN_ACT_SIGMOID) {
        if (!y) MCN_FAIL(MCN_E_BADARG, "act_bwd: activation %d needs y", (int)act);
        if (act == MCN_ACT_RELU) return ew_dispatch<EW_RELU_BWD>(dy, y, dx, n, dtype, st, "act_bwd");
        if (act == MCN_ACT_RELU6) return ew_dispatch<EW_RELU6_BWD>(dy, y, dx, n, dtype, st, "act_bwd");
        if (act == MCN_ACT_TANH) return ew_dispatch<EW_TANH_BWD>(dy, y, dx, n, dtype, st, "act_bwd");
        return ew_dispatch<EW_SIGMOID_BWD>(dy, y, dx, n, dtype, st, "act_bwd");
    }
    if (act == MCN_ACT_SWISH || act == MCN_ACT_LRELU) {
        if (!x) MCN_FAIL(MCN_E_BADARG, "act_bwd: activation %d needs x", (int)act);
        if (act == MCN_ACT_SWISH) return ew_dispatch<EW_SWISH_BWD>(dy, x, dx, n, dtype, st, "act_bwd");
        return ew_dispatch<EW_LRELU_BWD>(dy, x, dx, n, dtype, st, "act_bwd", param);
    }
    MCN_FAIL(MCN_E_UNSUPPORTED, "act_bwd: activation %d unsupported", (int)act);
}
extern "C" int mcn_act_bwd(const void* dy, const void* x, const void* y, void* dx, int64_t n, mcn_act act, mcn_dtype dtype, void* stream) {
    return mcn_act_bwd_p(dy, x, y, dx, n, act, 0.2f, dtype, stream);
}
extern "C" int mcn_accumulate(void* a, const void* b, int64_t n, mcn_dtype dtype, void* stream) {
    if (!b) MCN_FAIL(MCN_E_BADARG, "accumulate: null b");
    return ew_dispatch<EW_ACC>(a, b, a, n, dtype, (hipStream_t)stream, "accumulate");
}

// ---- cast ---------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = to_f32(s[i * 4 + k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[i * 4 + k] = from_f32<D>(v[k]);
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = from_f32<D>(to_f32(s[i]));
}
extern "C" int mcn_cast(const void* src, mcn_dtype sd, void* dst, mcn_dtype dd, int64_t n, void* stream) {
    if (!src || !dst || n < 0) MCN_FAIL(MCN_E_BADARG, "cast: bad argument");
    if (n == 0) return MCN_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ew_blocks(n / 4 + 1)), block(256);
    if (sd == MCN_F32 && dd == MCN_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, st, (const float*)src, (bf16_t*)dst, (long)n);
    else if (sd == MCN_BF16 && dd == MCN_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)src, (float*)dst, (long)n);
    else if (sd == MCN_F32 && dd == MCN_F16) hipLaunchKernelGGL((cast_kernel<float, f16_t>), grid, block, 0, st, (const float*)src, (f16_t*)dst, (long)n);
    else if (sd == MCN_F16 && dd == MCN_F32) hipLaunchKernelGGL((cast_kernel<f16_t, float>), grid, block, 0, st, (const f16_t*)src, (float*)dst, (long)n);
    else if (sd == dd && mcn_dtype_ok(sd)) {
        if (hipMemcpyAsync(dst, src, (size_t)n * mcn_dtype_size(sd), hipMemcpyDeviceToDevice, st) != hipSuccess) MCN_FAIL(MCN_E_LAUNCH, "cast: copy failed");
        return MCN_OK;
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "cast: %d -> %d unsupported", (int)sd, (int)dd);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// ---- input preparation ------------------------------------------------------------------------------------
// one thread per output pixel: reads C fp32 channels (NHWC: contiguous; NCHW: plane-strided), writes
// out_cs channels of the compute dtype, zero padded.
template <typename T, bool NCHW>
__global__ __launch_bounds__(256) void input_prep_kernel(const float* __restrict__ x, T* __restrict__ y, long npix, int HW, int C, int out_cs,
                                                         float mean, float scale) {
    const unsigned stride = gridDim.x * blockDim.x;
    constexpr int CE = (int)(16 / sizeof(T));
    for (unsigned pu = blockIdx.x * blockDim.x + threadIdx.x; pu < (unsigned)npix; pu += stride) {       // (host: npix < 2^32)
        const long p = pu;
        T* o = y + p * out_cs;
        const long n = pu / (unsigned)HW, r = p - n * HW;
        if (out_cs == CE) {                                  // the padded pixel is one 16-byte chunk: one store, not out_cs
            Chunk<T> ch;
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                float v = 0.f;
                if (c < C) v = ((NCHW ? x[(n * C + c) * HW + r] : x[p * C + c]) - mean) * scale;
                ch.set(c, v);
            }
            store_chunk<T>(o, ch);
            continue;
        }
        if (sizeof(T) == 2 && out_cs == 4) {                 // pixel-pair layout of the stem (4 channels per pixel): one 8-byte store
            T v4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v4[c] = from_f32<T>(c < C ? ((NCHW ? x[(n * C + c) * HW + r] : x[p * C + c]) - mean) * scale : 0.f);
            *reinterpret_cast<uint64_t*>(o) = *reinterpret_cast<const uint64_t*>(v4);
            continue;
        }
        for (int c = 0; c < out_cs; ++c) {
            float v = 0.f;
            if (c < C) {
                const float in = NCHW ? x[(n * C + c) * HW + r] : x[p * C + c];
                v = (in - mean) * scale;
            }
            o[c] = from_f32<T>(v);
        }
    }
}
extern "C" int mcn_input_prep(const float* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t out_cs, float image_mean,
                              float scale_factor, mcn_layout src_layout, mcn_dtype dtype, void* stream) {
    if (!x || !y || N < 0 || H <= 0 || W <= 0 || C <= 0 || out_cs < C) MCN_FAIL(MCN_E_BADARG, "input_prep: bad argument");
    const long npix = (long)N * H * W;
    if (npix == 0) return MCN_OK;
    if (npix >= 0xffffffffl) MCN_FAIL(MCN_E_UNSUPPORTED, "input_prep: more than 2^32 pixels");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ew_blocks(npix)), block(256);
    const bool nchw = src_layout == MCN_NCHW;
    if (dtype == MCN_F32) {
        if (nchw) hipLaunchKernelGGL((input_prep_kernel<float, true>), grid, block, 0, st, x, (float*)y, npix, H * W, C, out_cs, image_mean, scale_factor);
        else hipLaunchKernelGGL((input_prep_kernel<float, false>), grid, block, 0, st, x, (float*)y, npix, H * W, C, out_cs, image_mean, scale_factor);
    } else if (dtype == MCN_BF16) {
        if (nchw) hipLaunchKernelGGL((input_prep_kernel<bf16_t, true>), grid, block, 0, st, x, (bf16_t*)y, npix, H * W, C, out_cs, image_mean, scale_factor);
        else hipLaunchKernelGGL((input_prep_kernel<bf16_t, false>), grid, block, 0, st, x, (bf16_t*)y, npix, H * W, C, out_cs, image_mean, scale_factor);
    } else if (dtype == MCN_F16) {
        if (nchw) hipLaunchKernelGGL((input_prep_kernel<f16_t, true>), grid, block, 0, st, x, (f16_t*)y, npix, H * W, C, out_cs, image_mean, scale_factor);
        else hipLaunchKernelGGL((input_prep_kernel<f16_t, false>), grid, block, 0, st, x, (f16_t*)y, npix, H * W, C, out_cs, image_mean, scale_factor);
    } else MCN_FAIL(MCN_E_UNSUPPORTED, "input_prep: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

__global__ void one_hot_kernel(const float* __restrict__ labels, float* __restrict__ onehot, int B, int C) {
    const long total = (long)B * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / C), c = (int)(i - (long)b * C);
        float l = labels[b];
        if (l != l) l = -1.f;                 // NaN -> -1 (convnet.py:441-443)
        const int li = (int)l;                // tf.cast(float -> int32) truncates toward zero
        onehot[i] = (li == c && l > -1.f) ? 1.f : 0.f;
    }
}
extern "C" int mcn_one_hot(const float* labels, float* onehot, int32_t B, int32_t C, void* stream) {
    if (!labels || !onehot || B < 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "one_hot: bad argument");
    if (B == 0) return MCN_OK;
    hipLaunchKernelGGL(one_hot_kernel, dim3(ew_blocks((long)B * C)), dim3(256), 0, (hipStream_t)stream, labels, onehot, B, C);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}


// ---- depthwise channel multiplier (convnet.py:1634-1650: tf.nn.depthwise_conv2d with a [kh, kw, cin, mult] filter, output channel c * mult + q) ----
// A depthwise convolution with multiplier `mult` IS the multiplier-1 convolution of the input with every channel repeated `mult` times against the
// same filter buffer read as [kh][kw][cin * mult]: y_rep[m][c * mult + q] = x[m][c] here, then mcn_dwconv2d_* on cin * mult channels; backwards the
// data gradient of the repeated tensor is summed over each channel's `mult` copies (fp32, rounded once), the filter gradient needs nothing extra.
// One thread per element, consecutive threads on consecutive (output / input) channels: coalesced on the wide side of either pass.
template <typename T>
__global__ __launch_bounds__(256) void channel_repeat_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n_out, int C, int mult) {
    const long stride = (long)gridDim.x * blockDim.x;
    const int CO = C * mult;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        const long m = i / CO;
        const int j = (int)(i - m * CO);
        y[i] = x[m * C + j / mult];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void channel_repeat_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, long n_in, int mult) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += stride) {
        float a = 0.f;
        for (int q = 0; q < mult; ++q) a += to_f32(dy[i * mult + q]);
        dx[i] = from_f32<T>(a);
    }
}
extern "C" int mcn_channel_repeat_fwd(const void* x, void* y, int64_t M, int32_t C, int32_t mult, mcn_dtype dtype, void* stream) {
    if (!x || !y || M <= 0 || C <= 0 || mult < 1) MCN_FAIL(MCN_E_BADARG, "channel_repeat_fwd: bad argument (M=%ld C=%d mult=%d)", (long)M, C, mult);
    const long n = (long)M * C * mult;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) hipLaunchKernelGGL((channel_repeat_fwd_kernel<float>), dim3(ew_blocks(n)), dim3(256), 0, st, (const float*)x, (float*)y, n, C, mult);
    else if (dtype == MCN_BF16) hipLaunchKernelGGL((channel_repeat_fwd_kernel<bf16_t>), dim3(ew_blocks(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n, C, mult);
    else if (dtype == MCN_F16) hipLaunchKernelGGL((channel_repeat_fwd_kernel<f16_t>), dim3(ew_blocks(n)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, n, C, mult);
    else MCN_FAIL(MCN_E_UNSUPPORTED, "channel_repeat_fwd: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_channel_repeat_bwd(const void* dy, void* dx, int64_t M, int32_t C, int32_t mult, mcn_dtype dtype, void* stream) {
    if (!dy || !dx || M <= 0 || C <= 0 || mult < 1) MCN_FAIL(MCN_E_BADARG, "channel_repeat_bwd: bad argument (M=%ld C=%d mult=%d)", (long)M, C, mult);
    const long n = (long)M * C;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) hipLaunchKernelGGL((channel_repeat_bwd_kernel<float>), dim3(ew_blocks(n)), dim3(256), 0, st, (const float*)dy, (float*)dx, n, mult);
    else if (dtype == MCN_BF16) hipLaunchKernelGGL((channel_repeat_bwd_kernel<bf16_t>), dim3(ew_blocks(n)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, n, mult);
    else if (dtype == MCN_F16) hipLaunchKernelGGL((channel_repeat_bwd_kernel<f16_t>), dim3(ew_blocks(n)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, n, mult);
    else MCN_FAIL(MCN_E_UNSUPPORTED, "channel_repeat_bwd: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
