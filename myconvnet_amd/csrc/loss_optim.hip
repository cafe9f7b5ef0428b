// loss_optim.hip — softmax cross-entropy (fused forward + gradient), L2 regulariser value,
// fused Nesterov-momentum + L2 + EMA update, EMA / running-statistics helpers.  All fp32.
#include "common.h"

// one 256-thread block per row; wave shuffles + a 4-entry LDS stage for the row reductions
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
    v = wave_reduce_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
    v = wave_reduce_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, sh[i]);
    return t;
}

// focal factors of one row: fs = F * S (the factor on the row's cross-entropy), dfs = S * dF/dp_t (S is a constant under tf.stop_gradient)
struct FocalTerms { float fs, dfs; };
__device__ __forceinline__ FocalTerms focal_terms(float pt, float fgamma, float salpha) {
    float F = 1.f, dF = 0.f, S = 1.f;
    if (fgamma > 0.f) {
        const float om = fmaxf(1.f - pt, 0.f);
        F = powf(om, fgamma);
        dF = om > 0.f ? -fgamma * powf(om, fgamma - 1.f) : 0.f;
    }
    if (salpha > 0.f) S = (1.f - 1.f / (1.f + expf(-salpha * (pt - 0.5f)))) / (1.f - 1.f / (1.f + expf(0.5f * salpha)));
    return FocalTerms{F * S, dF * S};
}
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                           const float* __restrict__ class_w, float* __restrict__ pred,
                                                           float* __restrict__ ce, float* __restrict__ coef, float* __restrict__ dlogits,
                                                           int B, int C, float ls, float loss_scale, float fgamma, float salpha) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const float* z = logits + (long)b * C;
    const float* yv = labels + (long)b * C;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += blockDim.x) mx = fmaxf(mx, z[c]);
    mx = block_reduce_max(mx, sh);
    float se = 0.f, sy = 0.f, bw = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        se += expf(z[c] - mx);
        const float y = yv[c];
        sy += y;
        bw += y * (class_w ? class_w[c] : 1.f);
    }
    se = block_reduce_sum(se, sh);
    sy = block_reduce_sum(sy, sh);
    bw = block_reduce_sum(bw, sh);
    const float lse = logf(se);
    const float valid = (sy > 1.f - 1e-5f && sy < 1.f + 1e-5f) ? 1.f : 0.f;
    const float cf = bw * valid;
    // smoothed labels: y*(1-ls) + ls/C ; their sum = sy*(1-ls) + ls
    const float lab_sum = ls > 0.f ? sy * (1.f - ls) + ls : sy;
    const float gscale = cf * loss_scale / (float)B;
    float cel = 0.f;
    if (fgamma > 0.f || salpha > 0.f) {
        // focal factors (convnet.py:581-592): F = (1 - p_t)^gamma with p_t = sum_c Y_c p_c — differentiated THROUGH the softmax, as
        // tf.gradients does —, S = (1 - sigmoid(alpha (p_t - 0.5))) / (1 - sigmoid(-alpha / 2)) under tf.stop_gradient; the row's
        // loss is coef * F * S * CE.  Both need the whole row's CE and p_t before the gradient: two passes over the row.
        float pt = 0.f;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float lsm = z[c] - mx - lse;
            const float lab = ls > 0.f ? yv[c] * (1.f - ls) + ls / (float)C : yv[c];
            cel -= lab * lsm;
            pt += yv[c] * expf(lsm);
        }
        cel = block_reduce_sum(cel, sh);
        pt = block_reduce_sum(pt, sh);
        const FocalTerms ft = focal_terms(pt, fgamma, salpha);
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float lsm = z[c] - mx - lse;
            const float p = expf(lsm);
            const float lab = ls > 0.f ? yv[c] * (1.f - ls) + ls / (float)C : yv[c];
            if (pred) pred[(long)b * C + c] = p;
            if (dlogits) dlogits[(long)b * C + c] = (ft.fs * (p * lab_sum - lab) + cel * ft.dfs * (yv[c] * p - pt * p)) * gscale;
        }
        if (threadIdx.x == 0) {
            ce[b] = cel * ft.fs;
            coef[b] = cf;
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float lsm = z[c] - mx - lse;
        const float p = expf(lsm);
        const float lab = ls > 0.f ? yv[c] * (1.f - ls) + ls / (float)C : yv[c];
        cel -= lab * lsm;
        if (pred) pred[(long)b * C + c] = p;
        if (dlogits) dlogits[(long)b * C + c] = (p * lab_sum - lab) * gscale;
    }
    cel = block_reduce_sum(cel, sh);
    if (threadIdx.x == 0) {
        ce[b] = cel;
        coef[b] = cf;
    }
}
// few classes, many rows (per-pixel loss of the segmentation path, B = N*H*W): a block stages XR_ROWS rows of logits and
// labels through LDS with coalesced loads (row pitch C|1, odd => conflict-free), one thread owns one row in LDS, results
// (pred, dlogits) go back through the same LDS tiles and leave coalesced.  Same arithmetic as the block-per-row kernel.
#define XR_ROWS 256
__global__ __launch_bounds__(XR_ROWS) void softmax_xent_rows_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                                    const float* __restrict__ class_w, float* __restrict__ pred,
                                                                    float* __restrict__ ce, float* __restrict__ coef, float* __restrict__ dlogits,
                                                                    long B, int C, float ls, float loss_scale, const float* __restrict__ avg, float fgamma,
                                                                    float salpha) {
    extern __shared__ float xs[];
    const int P = C | 1;
    float* zt = xs;
    float* yt = xs + XR_ROWS * P;
    for (long b0 = (long)blockIdx.x * XR_ROWS; b0 < B; b0 += (long)gridDim.x * XR_ROWS) {
        const int rows = (int)min((long)XR_ROWS, B - b0);
        const int n = rows * C;                                   // (32-bit: a 64-bit division per element cost more than the exps)
        const float* zsrc = logits + b0 * C;
        const float* ysrc = labels + b0 * C;
        for (int i = threadIdx.x; i < n; i += XR_ROWS) {
            const int r = i / C, c = i - r * C;
            zt[r * P + c] = zsrc[i];
            yt[r * P + c] = ysrc[i];
        }
        __syncthreads();
        if ((int)threadIdx.x < rows) {
            float* z = zt + threadIdx.x * P;
            float* yv = yt + threadIdx.x * P;
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
            // one exponential per class: e = exp(z - max) is kept in the row's LDS slot and pred = e / sum(e) (the second
            // exp(log-softmax) per class made this kernel VALU bound at 21 classes x 4.2 M pixels); the loss uses
            // -sum(lab * lsm) = lse * sum(lab) - sum(lab * (z - max))
            // soft-label target of the smoothing: 1/C (convnet.py:603-607) or the 5x5 average of the label map (segnet.py:117-122;
            // a rarely used option: the row's averages are read straight from global memory)
            const float* av = avg ? avg + (b0 + threadIdx.x) * C : nullptr;
            float se = 0.f, sy = 0.f, bw = 0.f, slz = 0.f, sl = 0.f, sye = 0.f;
            for (int c = 0; c < C; ++c) {
                const float d = z[c] - mx;
                const float e = expf(d);
                se += e;
                const float y = yv[c];
                sy += y;
                sye += y * e;
                bw += y * (class_w ? class_w[c] : 1.f);
                const float lab = ls > 0.f ? y * (1.f - ls) + ls * (av ? av[c] : 1.f / (float)C) : y;
                sl += lab;
                slz += lab * d;
                z[c] = e;
            }
            const float lse = logf(se);
            const float inv = 1.f / se;
            const float valid = (sy > 1.f - 1e-5f && sy < 1.f + 1e-5f) ? 1.f : 0.f;
            const float cf = bw * valid;
            const float lab_sum = av ? sl : (ls > 0.f ? sy * (1.f - ls) + ls : sy);
            const float gscale = cf * loss_scale / (float)B;
            float cel = lse * lab_sum - slz;
            const float pt = sye * inv;
            FocalTerms ft = {1.f, 0.f};
            if (fgamma > 0.f || salpha > 0.f) ft = focal_terms(pt, fgamma, salpha);
            for (int c = 0; c < C; ++c) {
                const float pr = z[c] * inv;
                const float lab = ls > 0.f ? yv[c] * (1.f - ls) + ls * (av ? av[c] : 1.f / (float)C) : yv[c];
                const float yc = yv[c];
                z[c] = pr;
                yv[c] = (ft.fs * (pr * lab_sum - lab) + cel * ft.dfs * (yc * pr - pt * pr)) * gscale;
            }
            cel *= ft.fs;
            ce[b0 + threadIdx.x] = cel;
            coef[b0 + threadIdx.x] = cf;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += XR_ROWS) {
            const int r = i / C, c = i - r * C;
            if (pred) pred[b0 * C + i] = zt[r * P + c];
            if (dlogits) dlogits[b0 * C + i] = yt[r * P + c];
        }
        __syncthreads();
    }
}
// two-stage mean for many rows: 1024 block partials (written over the first entries of `part`), then one block
__global__ __launch_bounds__(256) void xent_partial_kernel(const float* __restrict__ ce, const float* __restrict__ coef, float* __restrict__ part, long B) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < B; i += (long)gridDim.x * 256) s += ce[i] * coef[i];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void xent_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ loss, long B) {
    __shared__ double shd[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)part[i];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(shd[0] / (double)B);
}
__global__ __launch_bounds__(256) void xent_mean_kernel(const float* __restrict__ ce, const float* __restrict__ coef, float* __restrict__ loss, int B) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) s += ce[i] * coef[i];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}
extern "C" int mcn_softmax_xent_focal_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce, float* coef,
                                              float* dlogits, float* loss, int32_t B, int32_t C, float label_smoothing, float loss_scale,
                                              float focal_gamma, float sigmoid_focal_alpha, void* stream) {
    if (!logits || !labels || !ce || !coef || B <= 0 || C <= 0 || focal_gamma < 0.f || sigmoid_focal_alpha < 0.f) MCN_FAIL(MCN_E_BADARG, "softmax_xent: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(B), dim3(256), 0, st, logits, labels, class_w, pred, ce, coef, dlogits, B, C, label_smoothing,
                       loss_scale, focal_gamma, sigmoid_focal_alpha);
    MCN_CHECK_LAUNCH();
    if (loss) {
        hipLaunchKernelGGL(xent_mean_kernel, dim3(1), dim3(256), 0, st, (const float*)ce, (const float*)coef, loss, B);
        MCN_CHECK_LAUNCH();
    }
    return MCN_OK;
}

extern "C" int mcn_softmax_xent_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce, float* coef,
                                        float* dlogits, float* loss, int32_t B, int32_t C, float label_smoothing, float loss_scale,
                                        void* stream) {
    return mcn_softmax_xent_focal_fwd_bwd(logits, labels, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale, 0.f, 0.f, stream);
}

/* per-pixel variant (segmentation: rows = N*H*W pixels, few classes): one thread per row, two-stage mean through the
 * caller's workspace (>= 1024 floats) */
extern "C" int mcn_softmax_xent_rows_focal_fwd_bwd(const float* logits, const float* labels, const float* avg_labels, const float* class_w, float* pred,
                                                   float* ce, float* coef, float* dlogits, float* loss, int64_t B, int32_t C, float label_smoothing,
                                                   float loss_scale, float focal_gamma, float sigmoid_focal_alpha, void* ws, size_t ws_bytes, void* stream) {
    if (focal_gamma < 0.f || sigmoid_focal_alpha < 0.f) MCN_FAIL(MCN_E_BADARG, "softmax_xent_rows: negative focal factor");
    if (!logits || !labels || !ce || !coef || B <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "softmax_xent_rows: bad argument");
    if (C > 64) MCN_FAIL(MCN_E_UNSUPPORTED, "softmax_xent_rows: C=%d > 64 classes: use mcn_softmax_xent_fwd_bwd", C);
    if (loss && (!ws || ws_bytes < 1024 * sizeof(float))) MCN_FAIL(MCN_E_WORKSPACE, "softmax_xent_rows: workspace needs 4096 bytes");
    hipStream_t st = (hipStream_t)stream;
    long blocks = ((long)B + XR_ROWS - 1) / XR_ROWS;
    if (blocks > 4096) blocks = 4096;
    const size_t lds = (size_t)2 * XR_ROWS * (C | 1) * sizeof(float);
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_xent_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * XR_ROWS * 65 * 4), true);
    (void)once;
    hipLaunchKernelGGL(softmax_xent_rows_kernel, dim3((unsigned)blocks), dim3(XR_ROWS), lds, st, logits, labels, class_w, pred, ce, coef, dlogits, (long)B, C,
                       label_smoothing, loss_scale, label_smoothing > 0.f ? avg_labels : (const float*)nullptr, focal_gamma, sigmoid_focal_alpha);
    MCN_CHECK_LAUNCH();
    if (loss) {
        hipLaunchKernelGGL(xent_partial_kernel, dim3(1024), dim3(256), 0, st, (const float*)ce, (const float*)coef, (float*)ws, (long)B);
        MCN_CHECK_LAUNCH();
        hipLaunchKernelGGL(xent_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, 1024, loss, (long)B);
        MCN_CHECK_LAUNCH();
    }
    return MCN_OK;
}
extern "C" int mcn_softmax_xent_rows_soft_fwd_bwd(const float* logits, const float* labels, const float* avg_labels, const float* class_w, float* pred,
                                                  float* ce, float* coef, float* dlogits, float* loss, int64_t B, int32_t C, float label_smoothing,
                                                  float loss_scale, void* ws, size_t ws_bytes, void* stream) {
    return mcn_softmax_xent_rows_focal_fwd_bwd(logits, labels, avg_labels, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale, 0.f, 0.f, ws,
                                               ws_bytes, stream);
}
extern "C" int mcn_softmax_xent_rows_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce, float* coef,
                                             float* dlogits, float* loss, int64_t B, int32_t C, float label_smoothing, float loss_scale, void* ws,
                                             size_t ws_bytes, void* stream) {
    return mcn_softmax_xent_rows_soft_fwd_bwd(logits, labels, nullptr, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale, ws, ws_bytes,
                                              stream);
}

// ---- L2 regulariser value ---------------------------------------------------------------------------------
#define L2_BLOCKS 1024
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ w, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    const long n4 = n / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += w[i] * w[i];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void l2_final_kernel(const float* __restrict__ part, int nparts, float factor, float* __restrict__ out) {
    __shared__ double shd[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += (double)part[i];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += (float)(0.5 * (double)factor * shd[0]);
}
extern "C" int mcn_l2_loss(const float* w, int64_t n, float factor, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !out || n < 0) MCN_FAIL(MCN_E_BADARG, "l2_loss: bad argument");
    if (!ws || ws_bytes < L2_BLOCKS * sizeof(float)) MCN_FAIL(MCN_E_WORKSPACE, "l2_loss: workspace needs %zu bytes", (size_t)L2_BLOCKS * sizeof(float));
    if (((uintptr_t)w & 15) != 0) MCN_FAIL(MCN_E_BADARG, "l2_loss: w must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(l2_partial_kernel, dim3(L2_BLOCKS), dim3(256), 0, st, w, (long)n, (float*)ws);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(l2_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, L2_BLOCKS, factor, out);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// ---- L1 regulariser (convnet.py:553-557: l1_factor * sum_w sum |w|): value and gradient ---------------------------------------------------
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ w, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += fabsf(w[i]);
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, int nparts, float factor, float* __restrict__ out) {
    __shared__ double shd[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += (double)part[i];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += (float)((double)factor * shd[0]);
}
extern "C" int mcn_l1_loss(const float* w, int64_t n, float factor, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !out || n < 0) MCN_FAIL(MCN_E_BADARG, "l1_loss: bad argument");
    if (!ws || ws_bytes < L2_BLOCKS * sizeof(float)) MCN_FAIL(MCN_E_WORKSPACE, "l1_loss: workspace needs %zu bytes", (size_t)L2_BLOCKS * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(l1_partial_kernel, dim3(L2_BLOCKS), dim3(256), 0, st, w, (long)n, (float*)ws);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, L2_BLOCKS, factor, out);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
// g += (l1 / grad_scale) * sign(w): launched in front of the update, whose kernel multiplies g by grad_scale = hyper[3] (1 / towers)
__global__ __launch_bounds__(256) void l1_grad_kernel(float* __restrict__ g, const float* __restrict__ w, long n, float l1, const float* __restrict__ hyper) {
    const float f = l1 / hyper[3];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = w[i];
        g[i] += v > 0.f ? f : (v < 0.f ? -f : 0.f);
    }
}
extern "C" int mcn_l1_grad_h(float* g, const float* w, int64_t n, float l1, const float* hyper, void* stream) {
    if (!g || !w || !hyper || n < 0) MCN_FAIL(MCN_E_BADARG, "l1_grad: bad argument");
    if (n == 0 || l1 == 0.f) return MCN_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(l1_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, w, (long)n, l1, hyper);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// ---- gradient clipping by global norm (tf.clip_by_global_norm, optimizers.py:112-113) ------------------------
// The reference clips the gradient of the FULL loss (cross-entropy + L2 term), per tower, before the tower mean; here
// the L2 gradient normally rides in the optimizer kernel, so with clipping it is folded into g first.
__global__ __launch_bounds__(256) void clip_prepare_kernel(float* __restrict__ g, const float* __restrict__ w, long n, long n_l2, float l2,
                                                           float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = g[i];
        if (i < n_l2) {
            v = fmaf(l2, w[i], v);
            g[i] = v;
        }
        s = fmaf(v, v, s);
    }
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void clip_final_kernel(const float* __restrict__ part, int nparts, float thr, float* __restrict__ out) {
    __shared__ double shd[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += (double)part[i];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(shd[0]);
        out[0] = thr / fmaxf(norm, thr);
        out[1] = norm;
    }
}
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, long n, const float* __restrict__ sc, float* __restrict__ norm_out) {
    const float f = sc[0];
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = sc[1];
    if (f == 1.f) return;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= f;
}
extern "C" int mcn_clip_by_global_norm(float* g, const float* w, int64_t n, int64_t n_l2, float l2, float threshold, float* norm_out, void* ws,
                                       size_t ws_bytes, void* stream) {
    if (!g || n < 0 || n_l2 < 0 || n_l2 > n || !(threshold > 0.f) || (n_l2 > 0 && l2 != 0.f && !w)) MCN_FAIL(MCN_E_BADARG, "clip_by_global_norm: bad argument");
    if (!ws || ws_bytes < (L2_BLOCKS + 4) * sizeof(float)) MCN_FAIL(MCN_E_WORKSPACE, "clip_by_global_norm: workspace needs %zu bytes", (size_t)(L2_BLOCKS + 4) * sizeof(float));
    if (n == 0) return MCN_OK;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    float* sc = part + L2_BLOCKS;
    hipLaunchKernelGGL(clip_prepare_kernel, dim3(L2_BLOCKS), dim3(256), 0, st, g, w, (long)n, (long)(l2 != 0.f ? n_l2 : 0), l2, part);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(clip_final_kernel, dim3(1), dim3(256), 0, st, (const float*)part, L2_BLOCKS, threshold, sc);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(clip_scale_kernel, dim3(L2_BLOCKS), dim3(256), 0, st, g, (long)n, (const float*)sc, norm_out);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

extern "C" int mcn_clip_by_global_norm_runs(float* g, const float* w, const int64_t* runs, int32_t nruns, float l2, float threshold, float* norm_out,
                                            void* ws, size_t ws_bytes, void* stream) {
    if (!g || !runs || nruns < 0 || !(threshold > 0.f)) MCN_FAIL(MCN_E_BADARG, "clip_by_global_norm_runs: bad argument");
    for (int r = 0; r < nruns; ++r) {
        const int64_t s = runs[3 * r], e = runs[3 * r + 1], le = runs[3 * r + 2];
        if (s < 0 || e < s || le < s || le > e || (le > s && l2 != 0.f && !w)) MCN_FAIL(MCN_E_BADARG, "clip_by_global_norm_runs: bad run %d", r);
    }
    const size_t need = ((size_t)nruns * L2_BLOCKS + 4) * sizeof(float);
    if (!ws || ws_bytes < need) MCN_FAIL(MCN_E_WORKSPACE, "clip_by_global_norm_runs: workspace needs %zu bytes", need);
    if (nruns == 0) return MCN_OK;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    float* sc = part + (size_t)nruns * L2_BLOCKS;
    for (int r = 0; r < nruns; ++r) {                       // every slot of the run's partial row is written (empty blocks write 0)
        const int64_t s = runs[3 * r], e = runs[3 * r + 1], le = runs[3 * r + 2];
        hipLaunchKernelGGL(clip_prepare_kernel, dim3(L2_BLOCKS), dim3(256), 0, st, g + s, w ? w + s : w, (long)(e - s), (long)(l2 != 0.f ? le - s : 0), l2,
                           part + (size_t)r * L2_BLOCKS);
        MCN_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(clip_final_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nruns * L2_BLOCKS, threshold, sc);
    MCN_CHECK_LAUNCH();
    for (int r = 0; r < nruns; ++r) {
        const int64_t s = runs[3 * r], e = runs[3 * r + 1];
        hipLaunchKernelGGL(clip_scale_kernel, dim3(L2_BLOCKS), dim3(256), 0, st, g + s, (long)(e - s), (const float*)sc, r == 0 ? norm_out : (float*)nullptr);
        MCN_CHECK_LAUNCH();
    }
    return MCN_OK;
}

// ---- fused Nesterov momentum + L2 + EMA + decoupled decay ----------------------------------------------------
template <bool EMA>
__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ a,
                                                           float* __restrict__ ema, long n, float lr, float mom, float l2, float wd,
                                                           float d, float gs, const float* __restrict__ hy) {
    if (hy) {                                          // per-step scalars from device memory (the launch is then replayable: hipGraph)
        lr = hy[0];
        wd = hy[1] * wd;                               // wd argument = 0 / 1 switch of the decay for this range
        d = hy[2];
        gs = hy[3];
    }
    const long n4 = n / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 wv = *reinterpret_cast<f32x4*>(w + i * 4);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
        f32x4 av = *reinterpret_cast<f32x4*>(a + i * 4);
        if (EMA) {
            f32x4 ev = *reinterpret_cast<f32x4*>(ema + i * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) ev[k] = d * ev[k] + (1.f - d) * wv[k];
            *reinterpret_cast<f32x4*>(ema + i * 4) = ev;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gt = gs * gv[k] + l2 * wv[k];
            av[k] = mom * av[k] + gt;
            float nw = wv[k] - lr * gt - lr * mom * av[k];
            if (wd > 0.f) nw -= wd * nw;
            wv[k] = nw;
        }
        *reinterpret_cast<f32x4*>(w + i * 4) = wv;
        *reinterpret_cast<f32x4*>(a + i * 4) = av;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float wv = w[i];
        if (EMA) ema[i] = d * ema[i] + (1.f - d) * wv;
        const float gt = gs * g[i] + l2 * wv;
        const float av = mom * a[i] + gt;
        float nw = wv - lr * gt - lr * mom * av;
        if (wd > 0.f) nw -= wd * nw;
        w[i] = nw;
        a[i] = av;
    }
}
static int sgd_launch(float* w, const float* g, float* accum, float* ema, int64_t n, float lr, float momentum, float l2, float wd, float ema_decay,
                      float grad_scale, const float* hyper, void* stream);
extern "C" int mcn_sgd_nesterov_fused(float* w, const float* g, float* accum, float* ema, int64_t n, float lr, float momentum, float l2,
                                      float wd, float ema_decay, float grad_scale, void* stream) {
    return sgd_launch(w, g, accum, ema, n, lr, momentum, l2, wd, ema_decay, grad_scale, nullptr, stream);
}
extern "C" int mcn_sgd_nesterov_fused_h(float* w, const float* g, float* accum, float* ema, int64_t n, const float* hyper, float momentum, float l2,
                                        int32_t use_wd, void* stream) {
    if (!hyper) MCN_FAIL(MCN_E_BADARG, "sgd_nesterov_fused_h: null hyper-parameter buffer");
    return sgd_launch(w, g, accum, ema, n, 0.f, momentum, l2, use_wd ? 1.f : 0.f, 0.f, 1.f, hyper, stream);
}
static int sgd_launch(float* w, const float* g, float* accum, float* ema, int64_t n, float lr, float momentum, float l2, float wd, float ema_decay,
                      float grad_scale, const float* hyper, void* stream) {
    if (!w || !g || !accum || n < 0) MCN_FAIL(MCN_E_BADARG, "sgd_nesterov_fused: bad argument");
    if (n == 0) return MCN_OK;
    if ((((uintptr_t)w | (uintptr_t)g | (uintptr_t)accum | (uintptr_t)ema) & 15) != 0) MCN_FAIL(MCN_E_BADARG, "sgd_nesterov_fused: buffers must be 16-byte aligned");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (ema) hipLaunchKernelGGL((sgd_nesterov_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, w, g, accum, ema, (long)n, lr, momentum, l2, wd, ema_decay, grad_scale, hyper);
    else hipLaunchKernelGGL((sgd_nesterov_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, w, g, accum, ema, (long)n, lr, momentum, l2, wd, ema_decay, grad_scale, hyper);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// ---- decoupled weight decay variants (optimizers.py:163-170) --------------------------------------------------------
// MODE 0: w -= wd*w ; 1: w -= wd*sign(w) ; 2 (pseudo-Huber): w -= wd*w / sqrt(1 + (w/delta)^2)
template <int MODE>
__global__ __launch_bounds__(256) void decoupled_decay_kernel(float* __restrict__ w, long n, float wd, float delta, const float* __restrict__ hy) {
    if (hy) wd = hy[1];
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = w[i];
        float d;
        if (MODE == 0) d = v;
        else if (MODE == 1) d = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
        else { const float r = v / delta; d = v / sqrtf(1.f + r * r); }
        w[i] = v - wd * d;
    }
}
static int decay_launch(float* w, int64_t n, float wd, int32_t mode, float delta, const float* hyper, void* stream);
extern "C" int mcn_decoupled_decay(float* w, int64_t n, float wd, int32_t mode, float delta, void* stream) {
    return decay_launch(w, n, wd, mode, delta, nullptr, stream);
}
extern "C" int mcn_decoupled_decay_h(float* w, int64_t n, const float* hyper, int32_t mode, float delta, void* stream) {
    if (!hyper) MCN_FAIL(MCN_E_BADARG, "decoupled_decay_h: null hyper-parameter buffer");
    return decay_launch(w, n, 1.f, mode, delta, hyper, stream);
}
static int decay_launch(float* w, int64_t n, float wd, int32_t mode, float delta, const float* hyper, void* stream) {
    if (!w || n < 0 || mode < 0 || mode > 2 || (mode == 2 && !(delta > 0.f))) MCN_FAIL(MCN_E_BADARG, "decoupled_decay: bad argument (mode %d, delta %g)", (int)mode, (double)delta);
    if (n == 0 || (wd == 0.f && !hyper)) return MCN_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL((decoupled_decay_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, st, w, (long)n, wd, delta, hyper);
    else if (mode == 1) hipLaunchKernelGGL((decoupled_decay_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, w, (long)n, wd, delta, hyper);
    else hipLaunchKernelGGL((decoupled_decay_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, w, (long)n, wd, delta, hyper);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

__global__ void ema_kernel(float* __restrict__ s, const float* __restrict__ v, long n, float d, const float* __restrict__ hy) {
    if (hy) d = hy[2];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s[i] = d * s[i] + (1.f - d) * v[i];
}
static int ema_launch(float* shadow, const float* v, int64_t n, float decay, const float* hyper, void* stream);
extern "C" int mcn_ema_update(float* shadow, const float* v, int64_t n, float decay, void* stream) { return ema_launch(shadow, v, n, decay, nullptr, stream); }
extern "C" int mcn_ema_update_h(float* shadow, const float* v, int64_t n, const float* hyper, void* stream) {
    if (!hyper) MCN_FAIL(MCN_E_BADARG, "ema_update_h: null hyper-parameter buffer");
    return ema_launch(shadow, v, n, 0.f, hyper, stream);
}
static int ema_launch(float* shadow, const float* v, int64_t n, float decay, const float* hyper, void* stream) {
    if (!shadow || !v || n < 0) MCN_FAIL(MCN_E_BADARG, "ema_update: bad argument");
    if (n == 0) return MCN_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, shadow, v, (long)n, decay, hyper);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
__global__ void bn_chain_kernel(float* __restrict__ run, const float* __restrict__ batch, int towers, long n, long stride, float m) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float r = run[i];
        for (int k = 0; k < towers; ++k) r = m * r + (1.f - m) * batch[(long)k * stride + i];
        run[i] = r;
    }
}
extern "C" int mcn_bn_running_chain_strided(float* running, const float* batch, int32_t towers, int64_t n, int64_t tower_stride, float momentum,
                                            void* stream) {
    if (!running || !batch || towers < 0 || n < 0 || tower_stride < n) MCN_FAIL(MCN_E_BADARG, "bn_running_chain: bad argument");
    if (n == 0 || towers == 0) return MCN_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bn_chain_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, running, batch, towers, (long)n, (long)tower_stride,
                       momentum);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_bn_running_chain(float* running, const float* batch, int32_t towers, int64_t n, float momentum, void* stream) {
    return mcn_bn_running_chain_strided(running, batch, towers, n, n, momentum, stream);
}
