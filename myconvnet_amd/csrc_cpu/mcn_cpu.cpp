// mcn_cpu.cpp — libmcn_cpu.so: the C-ABI of include/mcn.h as plain C++ / OpenMP loops ("same symbols in
// libmcn_cpu.so"; stands in for the reference's num_gpus = 0 path, convnet.py:69-95).
//
// What it is for: the graph / executor / launch-list host code (myconvnet_amd/executor.py, optimizers.py) can execute whole training steps in
// the GPU-less build container (`pytest -m "not gpu"`), so a host-side regression shows without a GPU box.  What it is NOT: a fallback.  The
// binding loads it only when MCN_LIB_PATH names it explicitly (myconvnet_amd/_ffi.py); the product path has no CPU route.
//
// Semantics follow include/mcn.h entry by entry (same rounding points: storage type after every stored tensor, fp32 parameters, per-use
// cast of the weights); the fusions that exist for bandwidth only (conv-epilogue BN statistics, dgrad + masked residual, dgrad + BN sums,
// pixel-pair stem, packed operands) announce themselves as "not eligible" through their query entry points (rows = 0, ok = 0, bytes = 0), so
// the executor lowers the plain calls; their compute entry points return MCN_E_UNSUPPORTED.  Loops are naive on purpose.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mcn.h"

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define UNSUPPORTED(name) return fail(MCN_E_UNSUPPORTED, name ": not in libmcn_cpu.so (its query entry point answers \"not eligible\")")
extern "C" const char* mcn_last_error(void) { return g_err; }
extern "C" int mcn_version(void) { return MCN_VERSION; }
extern "C" const char* mcn_build_id(void) { return "cpu"; }

// ---- storage types -------------------------------------------------------------------------------------------------------------------
struct F32 {
    typedef float T;
    static constexpr int VEC = 4;                                  // elements per 16-byte chunk (ReLU byte masks)
    static float ld(const T* p) { return *p; }
    static void st(T* p, float v) { *p = v; }
};
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
struct BF16 {
    typedef uint16_t T;
    static constexpr int VEC = 8;
    static float ld(const T* p) { return bits2f((uint32_t)*p << 16); }
    static void st(T* p, float v) {                                // round to nearest even, NaN stays NaN
        uint32_t u = f2bits(v);
        if ((u & 0x7fffffffu) > 0x7f800000u) { *p = (T)((u >> 16) | 0x40); return; }
        u += 0x7fffu + ((u >> 16) & 1u);
        *p = (T)(u >> 16);
    }
};
struct F16 {
    typedef uint16_t T;
    static constexpr int VEC = 8;
    static float ld(const T* p) {
        const uint32_t h = *p, s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
        if (e == 0) {
            if (m == 0) return bits2f(s);
            float f = (float)m * (1.0f / 16777216.0f);              // subnormal: m * 2^-24
            return (h & 0x8000u) ? -f : f;
        }
        if (e == 31) return bits2f(s | 0x7f800000u | (m << 13));
        return bits2f(s | ((e + 112) << 23) | (m << 13));
    }
    static void st(T* p, float v) {                                // round to nearest even, overflow -> inf (as tf.cast)
        const uint32_t u = f2bits(v), s = (u >> 16) & 0x8000u;
        const uint32_t a = u & 0x7fffffffu;
        if (a > 0x7f800000u) { *p = (T)(s | 0x7e00); return; }
        if (a >= 0x47800000u) { *p = (T)(s | 0x7c00); return; }    // >= 65536 (values that round up to it are caught below)
        if (a < 0x33000001u) { *p = (T)s; return; }                 // < 2^-25 (+ tie): zero
        int e = (int)(a >> 23) - 127;
        uint32_t m = (a & 0x7fffffu) | 0x800000u;
        int shift = e < -14 ? (13 + (-14 - e)) : 13;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        uint32_t h = e < -14 ? r : (((uint32_t)(e + 15) << 10) + (r - 0x400));      // (r may carry into the exponent: the addition handles it)
        if (h >= 0x7c00) h = 0x7c00;
        *p = (T)(s | h);
    }
};
template <typename F>
static int by_dtype(mcn_dtype dt, const char* who, F&& f) {
    if (dt == MCN_F32) return f(F32{});
    if (dt == MCN_BF16) return f(BF16{});
    if (dt == MCN_F16) return f(F16{});
    return fail(MCN_E_UNSUPPORTED, "%s: dtype %d unsupported", who, (int)dt);
}
template <typename S>
static inline float rnd(float v) { typename S::T t; S::st(&t, v); return S::ld(&t); }

// ---- convolution -----------------------------------------------------------------------------------------------------------------------
struct Geo { int N, H, W, Cin, Cout, KH, KW, SH, SW, DH, DW, pT, pB, pL, pR, xcs, OH, OW; };
static int geo_from(const mcn_conv_geom* g, Geo* o) {
    if (!g) return fail(MCN_E_BADARG, "conv: null geometry");
    o->N = g->N; o->H = g->H; o->W = g->W; o->Cin = g->Cin; o->Cout = g->Cout; o->KH = g->KH; o->KW = g->KW; o->SH = g->SH; o->SW = g->SW;
    o->DH = g->DH > 0 ? g->DH : 1; o->DW = g->DW > 0 ? g->DW : 1; o->pT = g->padT; o->pB = g->padB; o->pL = g->padL; o->pR = g->padR;
    o->xcs = g->x_cs > 0 ? g->x_cs : g->Cin;
    if (o->N < 0 || o->H <= 0 || o->W <= 0 || o->Cin <= 0 || o->Cout <= 0 || o->KH <= 0 || o->KW <= 0 || o->SH <= 0 || o->SW <= 0 || o->xcs < o->Cin)
        return fail(MCN_E_BADARG, "conv: bad geometry");
    o->OH = (o->H + o->pT + o->pB - (o->KH - 1) * o->DH - 1) / o->SH + 1;
    o->OW = (o->W + o->pL + o->pR - (o->KW - 1) * o->DW - 1) / o->SW + 1;
    if (o->OH <= 0 || o->OW <= 0) return fail(MCN_E_BADARG, "conv: empty output");
    return MCN_OK;
}
extern "C" int mcn_conv2d_tile_candidates(mcn_conv_op) { return 0; }
extern "C" size_t mcn_conv2d_workspace_bytes(mcn_conv_op, const mcn_conv_geom*, mcn_dtype) { return 0; }
extern "C" int32_t mcn_conv2d_kslices(mcn_conv_op, const mcn_conv_geom*, mcn_dtype) { return 1; }
extern "C" size_t mcn_conv2d_packed_bytes(mcn_conv_op, const mcn_conv_geom*, mcn_dtype) { return 0; }
extern "C" size_t mcn_conv2d_pack_table_bytes(const mcn_pack_job*, int32_t) { return 0; }
extern "C" int mcn_conv2d_pack_table_build(const mcn_pack_job*, int32_t, mcn_dtype, void*, size_t, int32_t* ndesc) { if (ndesc) *ndesc = 0; return MCN_OK; }
extern "C" int mcn_conv2d_pack_run(const void*, int32_t, mcn_dtype, void*) { return MCN_OK; }
extern "C" int mcn_conv2d_pair_geom(const mcn_conv_geom*, mcn_dtype, mcn_conv_geom*) { return 0; }
extern "C" int mcn_conv2d_pair_weights(const float*, float*, const mcn_conv_geom*, mcn_dtype, void*) { UNSUPPORTED("conv2d_pair_weights"); }
extern "C" int mcn_conv2d_pair_wgrad_fold(const float*, float*, const mcn_conv_geom*, mcn_dtype, void*) { UNSUPPORTED("conv2d_pair_wgrad_fold"); }
extern "C" int mcn_conv2d_kernel_name(mcn_conv_op, const mcn_conv_geom*, mcn_dtype, char* buf, size_t n) { if (buf && n) snprintf(buf, n, "cpu_conv_loops"); return 1; }
extern "C" int mcn_conv2d_launch_list(mcn_conv_op, const mcn_conv_geom* g, mcn_dtype, char* buf, size_t n) {
    if (buf && n) snprintf(buf, n, "cpu_conv_loops:%d\n", g ? g->KH * g->KW : 1);
    return 1;
}
extern "C" int32_t mcn_conv2d_dgrad_addmasked_ok(const mcn_conv_geom*, mcn_dtype) { return 0; }
extern "C" int mcn_conv2d_dgrad_addmasked(const void*, const float*, const void*, void*, const void*, const uint8_t*, const mcn_conv_geom*, mcn_dtype, mcn_layout, void*, size_t,
                                          void*) { UNSUPPORTED("conv2d_dgrad_addmasked"); }
extern "C" int32_t mcn_conv2d_bnstats_rows(const mcn_conv_geom*, mcn_dtype, int32_t* rpp) { if (rpp) *rpp = 0; return 0; }
extern "C" int mcn_conv2d_fwd_bnstats(const void*, const float*, const void*, const float*, void*, float*, const mcn_conv_geom*, mcn_dtype, mcn_layout, void*, size_t, void*) {
    UNSUPPORTED("conv2d_fwd_bnstats");
}
extern "C" int32_t mcn_conv2d_dgrad_bnred_rows(const mcn_conv_geom*, mcn_dtype) { return 0; }
extern "C" int mcn_conv2d_dgrad_bnred(const void*, const float*, const void*, void*, const void*, const uint8_t*, float*, const mcn_conv_geom*, mcn_dtype, mcn_layout, void*, size_t,
                                      void*) { UNSUPPORTED("conv2d_dgrad_bnred"); }
extern "C" int mcn_conv2d_dgrad_addmasked_bnred(const void*, const float*, const void*, void*, const void*, const uint8_t*, const void*, const uint8_t*, float*, const mcn_conv_geom*,
                                                mcn_dtype, mcn_layout, void*, size_t, void*) { UNSUPPORTED("conv2d_dgrad_addmasked_bnred"); }

// per-use cast of the fp32 master (convnet.py:1421-1422): the weights as the storage type sees them, kept in fp32
template <typename S>
static std::vector<float> cast_weights(const float* w, size_t n) {
    std::vector<float> q(n);
    for (size_t i = 0; i < n; ++i) q[i] = rnd<S>(w[i]);
    return q;
}
extern "C" int mcn_conv2d_fwd(const void* x, const float* w, const void*, const float* bias, void* y, const mcn_conv_geom* gg, mcn_dtype dtype, mcn_layout layout, void*,
                              size_t, void*) {
    Geo g;
    if (int rc = geo_from(gg, &g)) return rc;
    if (layout != MCN_NHWC) return fail(MCN_E_UNSUPPORTED, "conv2d_fwd: only NHWC activations");
    if (!x || !w || !y) return fail(MCN_E_BADARG, "conv2d_fwd: null pointer");
    return by_dtype(dtype, "conv2d_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const std::vector<float> wq = cast_weights<S>(w, (size_t)g.KH * g.KW * g.Cin * g.Cout);
        const T* xp = (const T*)x;
        T* yp = (T*)y;
        const long rows = (long)g.N * g.OH;
#pragma omp parallel for schedule(static)
        for (long r = 0; r < rows; ++r) {
            const int n = (int)(r / g.OH), oy = (int)(r % g.OH);
            std::vector<float> acc(g.Cout);
            for (int ox = 0; ox < g.OW; ++ox) {
                for (int k = 0; k < g.Cout; ++k) acc[k] = bias ? bias[k] : 0.f;
                for (int kr = 0; kr < g.KH; ++kr) {
                    const int iy = oy * g.SH + kr * g.DH - g.pT;
                    if (iy < 0 || iy >= g.H) continue;
                    for (int ks = 0; ks < g.KW; ++ks) {
                        const int ix = ox * g.SW + ks * g.DW - g.pL;
                        if (ix < 0 || ix >= g.W) continue;
                        const T* xr = xp + (((long)n * g.H + iy) * g.W + ix) * g.xcs;
                        const float* wr = wq.data() + (size_t)(kr * g.KW + ks) * g.Cin * g.Cout;
                        for (int c = 0; c < g.Cin; ++c) {
                            const float xv = S::ld(xr + c);
                            const float* wc = wr + (size_t)c * g.Cout;
                            for (int k = 0; k < g.Cout; ++k) acc[k] += xv * wc[k];
                        }
                    }
                }
                T* yo = yp + (((long)n * g.OH + oy) * g.OW + ox) * g.Cout;
                for (int k = 0; k < g.Cout; ++k) S::st(yo + k, acc[k]);
            }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_conv2d_dgrad(const void* dy, const float* w, const void*, void* dx, const mcn_conv_geom* gg, int accumulate, mcn_dtype dtype, mcn_layout layout, void*, size_t,
                                void*) {
    Geo g;
    if (int rc = geo_from(gg, &g)) return rc;
    if (layout != MCN_NHWC) return fail(MCN_E_UNSUPPORTED, "conv2d_dgrad: only NHWC activations");
    if (!dy || !w || !dx) return fail(MCN_E_BADARG, "conv2d_dgrad: null pointer");
    return by_dtype(dtype, "conv2d_dgrad", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const std::vector<float> wq = cast_weights<S>(w, (size_t)g.KH * g.KW * g.Cin * g.Cout);
        const T* dyp = (const T*)dy;
        T* dxp = (T*)dx;
        const long rows = (long)g.N * g.H;
#pragma omp parallel for schedule(static)
        for (long r = 0; r < rows; ++r) {
            const int n = (int)(r / g.H), iy = (int)(r % g.H);
            std::vector<float> acc(g.Cin);
            for (int ix = 0; ix < g.W; ++ix) {
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int kr = 0; kr < g.KH; ++kr) {
                    const int ty = iy + g.pT - kr * g.DH;
                    if (ty < 0 || ty % g.SH) continue;
                    const int oy = ty / g.SH;
                    if (oy >= g.OH) continue;
                    for (int ks = 0; ks < g.KW; ++ks) {
                        const int tx = ix + g.pL - ks * g.DW;
                        if (tx < 0 || tx % g.SW) continue;
                        const int ox = tx / g.SW;
                        if (ox >= g.OW) continue;
                        const T* dr = dyp + (((long)n * g.OH + oy) * g.OW + ox) * g.Cout;
                        const float* wr = wq.data() + (size_t)(kr * g.KW + ks) * g.Cin * g.Cout;
                        for (int c = 0; c < g.Cin; ++c) {
                            const float* wc = wr + (size_t)c * g.Cout;
                            float a = 0.f;
                            for (int k = 0; k < g.Cout; ++k) a += S::ld(dr + k) * wc[k];
                            acc[c] += a;
                        }
                    }
                }
                T* xo = dxp + (((long)n * g.H + iy) * g.W + ix) * g.Cin;
                for (int c = 0; c < g.Cin; ++c) S::st(xo + c, accumulate ? S::ld(xo + c) + acc[c] : acc[c]);
            }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_conv2d_wgrad(const void* x, const void* dy, float* dw, float* dbias, const mcn_conv_geom* gg, float grad_scale, mcn_dtype dtype, mcn_layout layout, void*, size_t,
                                void*) {
    Geo g;
    if (int rc = geo_from(gg, &g)) return rc;
    if (layout != MCN_NHWC) return fail(MCN_E_UNSUPPORTED, "conv2d_wgrad: only NHWC activations");
    if (!x || !dy || !dw) return fail(MCN_E_BADARG, "conv2d_wgrad: null pointer");
    return by_dtype(dtype, "conv2d_wgrad", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const T* xp = (const T*)x;
        const T* dyp = (const T*)dy;
        const long taps = (long)g.KH * g.KW;
#pragma omp parallel for schedule(dynamic)
        for (long tc = 0; tc < taps * g.Cin; ++tc) {
            const int t = (int)(tc / g.Cin), c = (int)(tc % g.Cin), kr = t / g.KW, ks = t % g.KW;
            std::vector<double> acc(g.Cout, 0.0);
            for (int n = 0; n < g.N; ++n)
                for (int oy = 0; oy < g.OH; ++oy) {
                    const int iy = oy * g.SH + kr * g.DH - g.pT;
                    if (iy < 0 || iy >= g.H) continue;
                    for (int ox = 0; ox < g.OW; ++ox) {
                        const int ix = ox * g.SW + ks * g.DW - g.pL;
                        if (ix < 0 || ix >= g.W) continue;
                        const double xv = S::ld(xp + (((long)n * g.H + iy) * g.W + ix) * g.xcs + c);
                        const T* dr = dyp + (((long)n * g.OH + oy) * g.OW + ox) * g.Cout;
                        for (int k = 0; k < g.Cout; ++k) acc[k] += xv * S::ld(dr + k);
                    }
                }
            float* o = dw + ((size_t)t * g.Cin + c) * g.Cout;
            for (int k = 0; k < g.Cout; ++k) o[k] = (float)(acc[k] * grad_scale);
        }
        if (dbias) {
            const long M = (long)g.N * g.OH * g.OW;
#pragma omp parallel for schedule(static)
            for (int k = 0; k < g.Cout; ++k) {
                double a = 0.0;
                for (long m = 0; m < M; ++m) a += S::ld(dyp + m * g.Cout + k);
                dbias[k] = (float)(a * grad_scale);
            }
        }
        return (int)MCN_OK;
    });
}

// ---- batch normalisation ------------------------------------------------------------------------------------------------------------
extern "C" size_t mcn_bn_workspace_bytes(int64_t, int32_t) { return 0; }
extern "C" size_t mcn_bn_relu_mask_bytes(int64_t M, int32_t C, mcn_dtype dtype) {
    const int vec = dtype == MCN_F32 ? 4 : 8;
    if (M <= 0 || C <= 0 || C % vec) return 0;
    return (size_t)M * (C / vec);
}
static inline float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }
static inline float swish_grad(float z) { const float sg = sigmoidf(z); return sg * (1.f + z * (1.f - sg)); }
extern "C" int mcn_bn_fwd_train(const void* x, const float* gamma, const float* beta, const void* skip, void* y, uint8_t* relu_mask, float* save_mean, float* save_invstd,
                                float* batch_mean, float* batch_var, float* running_mean, float* running_var, float momentum, int64_t M, int32_t C, float eps, mcn_act act,
                                mcn_dtype dtype, void*, size_t, void*) {
    if (!x || !y || !save_mean || !save_invstd || M <= 0 || C <= 0) return fail(MCN_E_BADARG, "bn_fwd_train: bad argument");
    return by_dtype(dtype, "bn_fwd_train", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const T* xp = (const T*)x;
        const T* sp = (const T*)skip;
        T* yp = (T*)y;
        std::vector<float> sc(C), sh(C);
#pragma omp parallel for schedule(static)
        for (int c = 0; c < C; ++c) {
            double a = 0.0, b = 0.0;
            const double piv = S::ld(xp + c);
            for (int64_t m = 0; m < M; ++m) {
                const double d = (double)S::ld(xp + m * C + c) - piv;
                a += d;
                b += d * d;
            }
            const double dm = a / (double)M, mean = piv + dm;
            double var = b / (double)M - dm * dm;
            if (var < 0.0) var = 0.0;
            const double invstd = 1.0 / std::sqrt(var + (double)eps);
            const float fmean = (float)mean, finv = (float)invstd;
            save_mean[c] = fmean;
            save_invstd[c] = finv;
            const double ub = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
            if (batch_mean) batch_mean[c] = fmean;
            if (batch_var) batch_var[c] = (float)ub;
            if (running_mean) running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * fmean;
            if (running_var) running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)ub;
            sc[c] = (gamma ? gamma[c] : 1.f) * finv;
            sh[c] = (beta ? beta[c] : 0.f) - fmean * sc[c];
        }
        const bool mask = relu_mask && act == MCN_ACT_RELU && C % S::VEC == 0;
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; ++m) {
            for (int c0 = 0; c0 < C; c0 += S::VEC) {
                unsigned bits = 0;
                for (int i = 0; i < S::VEC && c0 + i < C; ++i) {
                    const int c = c0 + i;
                    float o = fmaf(S::ld(xp + m * C + c), sc[c], sh[c]);
                    if (sp) o += S::ld(sp + m * C + c);
                    if (act == MCN_ACT_RELU) { bits |= (o > 0.f ? 1u : 0u) << i; o = o > 0.f ? o : 0.f; }
                    if (act == MCN_ACT_SWISH) o = o * sigmoidf(o);
                    S::st(yp + m * C + c, o);
                }
                if (mask) relu_mask[m * (C / S::VEC) + c0 / S::VEC] = (uint8_t)bits;
            }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_bn_fwd_train_fused(const void*, const float*, int32_t, int32_t, const float*, const float*, const void*, void*, uint8_t*, float*, float*, float*, float*, float*,
                                      float*, float, int64_t, int32_t, float, mcn_act, mcn_dtype, void*, size_t, void*) { UNSUPPORTED("bn_fwd_train_fused"); }
extern "C" int mcn_bn_fwd_train_fused_stats(const float*, int32_t, int32_t, const float*, const float*, float*, float*, float*, float*, float*, float*, float, int64_t, int32_t, float,
                                            float*, void*, size_t, void*) { UNSUPPORTED("bn_fwd_train_fused_stats"); }
extern "C" int mcn_bn_fwd_train_fused_affskip(const void*, const float*, int32_t, int32_t, const float*, const float*, const void*, const float*, void*, uint8_t*, float*, float*,
                                              float*, float*, float*, float*, float, int64_t, int32_t, float, mcn_dtype, void*, size_t, void*) {
    UNSUPPORTED("bn_fwd_train_fused_affskip");
}
extern "C" int mcn_bn_fwd_train_fused_maxpool(const void*, const float*, int32_t, int32_t, const float*, const float*, void*, int8_t*, float*, float*, float*, float*, float*, float*,
                                              float, int32_t, int32_t, int32_t, int32_t, float, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, mcn_dtype,
                                              void*, size_t, void*) { UNSUPPORTED("bn_fwd_train_fused_maxpool"); }
extern "C" int mcn_bn_fwd_infer(const void* x, const float* gamma, const float* beta, const float* mean, const float* var, const void* skip, void* y, int64_t M, int32_t C, float eps,
                                mcn_act act, mcn_dtype dtype, void*) {
    if (!x || !y || !mean || !var || M < 0 || C <= 0) return fail(MCN_E_BADARG, "bn_fwd_infer: bad argument");
    return by_dtype(dtype, "bn_fwd_infer", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const T* xp = (const T*)x;
        const T* sp = (const T*)skip;
        T* yp = (T*)y;
        std::vector<float> sc(C), sh(C);
        for (int c = 0; c < C; ++c) {
            sc[c] = (gamma ? gamma[c] : 1.f) / sqrtf(var[c] + eps);
            sh[c] = (beta ? beta[c] : 0.f) - mean[c] * sc[c];
        }
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; ++m)
            for (int c = 0; c < C; ++c) {
                float o = fmaf(S::ld(xp + m * C + c), sc[c], sh[c]);
                if (sp) o += S::ld(sp + m * C + c);
                if (act == MCN_ACT_RELU) o = o > 0.f ? o : 0.f;
                if (act == MCN_ACT_SWISH) o = o * sigmoidf(o);
                S::st(yp + m * C + c, o);
            }
        return (int)MCN_OK;
    });
}
// the output gradient as the BN sees it: dy masked / scaled by the activation (mcn.h, mcn_bn_bwd)
template <typename S>
static inline float act_grad(mcn_act act, const typename S::T* dy, const typename S::T* x, const typename S::T* y, const uint8_t* mask, int64_t m, int C, int c, float sc, float sh) {
    float g = S::ld(dy + m * C + c);
    if (act == MCN_ACT_RELU) {
        bool on;
        if (mask) on = (mask[m * (C / S::VEC) + c / S::VEC] >> (c % S::VEC)) & 1;
        else if (y) on = S::ld(y + m * C + c) > 0.f;
        else on = fmaf(S::ld(x + m * C + c), sc, sh) > 0.f;
        g = on ? g : 0.f;
    } else if (act == MCN_ACT_SWISH) {
        g *= swish_grad(fmaf(S::ld(x + m * C + c), sc, sh));
    }
    return g;
}
template <typename S>
static int bn_bwd_impl(const void* dy, const void* x, const void* y, const uint8_t* relu_mask, const float* gamma, const float* beta, const float* mean, const float* invstd, void* dx,
                       void* dskip, float* dgamma, float* dbeta, float gs, int64_t M, int32_t C, mcn_act act, bool frozen) {
    typedef typename S::T T;
    const T* dyp = (const T*)dy;
    const T* xp = (const T*)x;
    const T* yp = (const T*)y;
    T* dxp = (T*)dx;
    T* dsp = (T*)dskip;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float g0 = gamma ? gamma[c] : 1.f, sc = g0 * invstd[c], sh = (beta ? beta[c] : 0.f) - mean[c] * sc;
        double a = 0.0, b = 0.0;
        for (int64_t m = 0; m < M; ++m) {
            const float g = act_grad<S>(act, dyp, xp, yp, relu_mask, m, C, c, sc, sh);
            a += g;
            b += (double)g * ((S::ld(xp + m * C + c) - mean[c]) * invstd[c]);
        }
        if (dbeta) dbeta[c] = (float)a * gs;
        if (dgamma) dgamma[c] = (float)b * gs;
        const float ca = g0 * invstd[c], cb = frozen ? 0.f : (float)(a / (double)M), cc = frozen ? 0.f : (float)(b / (double)M);
        for (int64_t m = 0; m < M; ++m) {
            const float g = act_grad<S>(act, dyp, xp, yp, relu_mask, m, C, c, sc, sh);
            const float xh = (S::ld(xp + m * C + c) - mean[c]) * invstd[c];
            S::st(dxp + m * C + c, ca * (g - cb - xh * cc));
            if (dsp) S::st(dsp + m * C + c, g);
        }
    }
    return MCN_OK;
}
extern "C" int mcn_bn_bwd(const void* dy, const void* x, const void* y, const uint8_t* relu_mask, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_invstd, void* dx, void* dskip, float* dgamma, float* dbeta, float grad_scale, int64_t M, int32_t C, mcn_act act, mcn_dtype dtype, void*,
                          size_t, void*) {
    if (!dy || !x || !dx || !save_mean || !save_invstd || M <= 0 || C <= 0) return fail(MCN_E_BADARG, "bn_bwd: bad argument");
    if (act == MCN_ACT_SWISH && dskip) return fail(MCN_E_BADARG, "bn_bwd: swish has no fused residual");
    return by_dtype(dtype, "bn_bwd", [&](auto s) {
        typedef decltype(s) S;
        const uint8_t* mk = (relu_mask && C % S::VEC == 0) ? relu_mask : nullptr;
        return bn_bwd_impl<S>(dy, x, y, mk, gamma, beta, save_mean, save_invstd, dx, dskip, dgamma, dbeta, grad_scale, M, C, act, false);
    });
}
extern "C" int mcn_bn_bwd_frozen(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* mean, const float* var, float eps, void* dx,
                                 void* dskip, float* dgamma, float* dbeta, float grad_scale, int64_t M, int32_t C, mcn_act act, mcn_dtype dtype, void*, size_t, void*) {
    if (!dy || !x || !dx || !mean || !var || M <= 0 || C <= 0) return fail(MCN_E_BADARG, "bn_bwd_frozen: bad argument");
    if (act == MCN_ACT_RELU && !y) return fail(MCN_E_BADARG, "bn_bwd_frozen: ReLU needs the stored output");
    std::vector<float> inv(C);
    for (int c = 0; c < C; ++c) inv[c] = 1.f / sqrtf(var[c] + eps);
    return by_dtype(dtype, "bn_bwd_frozen", [&](auto s) {
        typedef decltype(s) S;
        return bn_bwd_impl<S>(dy, x, act == MCN_ACT_RELU ? y : nullptr, nullptr, gamma, beta, mean, inv.data(), dx, dskip, dgamma, dbeta, grad_scale, M, C, act, true);
    });
}
extern "C" int mcn_bn_bwd_from_partials(const void*, const void*, const uint8_t*, const float*, const float*, const float*, const float*, const float*, int32_t, void*, float*, float*,
                                        float, int64_t, int32_t, mcn_dtype, void*, size_t, void*) { UNSUPPORTED("bn_bwd_from_partials"); }
extern "C" int mcn_bn_bwd_maxpool(const void*, const int8_t*, const void*, const float*, const float*, const float*, const float*, void*, float*, float*, float, int32_t, int32_t,
                                  int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, mcn_dtype, void*, size_t, void*) {
    UNSUPPORTED("bn_bwd_maxpool");
}

// ---- squeeze-excite (models/efficientnet.py:152-197) -----------------------------------------------------------------------------------
extern "C" int mcn_channel_scale_fwd(const void* x, const void* m, void* y, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void*) {
    if (!x || !m || !y || N < 0 || HW < 0 || C <= 0) return fail(MCN_E_BADARG, "channel_scale_fwd: bad argument");
    return by_dtype(dtype, "channel_scale_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < (int64_t)N * HW; ++r) {
            const int64_t n = r / HW;
            for (int c = 0; c < C; ++c) S::st((T*)y + r * C + c, S::ld((const T*)x + r * C + c) * S::ld((const T*)m + n * C + c));
        }
        return (int)MCN_OK;
    });
}
template <typename S>
static void chscale_dm(const void* dy, const void* x, void* dm, int32_t N, int64_t HW, int32_t C) {
    typedef typename S::T T;
#pragma omp parallel for schedule(static)
    for (int64_t nc = 0; nc < (int64_t)N * C; ++nc) {
        const int64_t n = nc / C, c = nc % C;
        double a = 0.0;
        for (int64_t p = 0; p < HW; ++p) a += (double)S::ld((const T*)dy + (n * HW + p) * C + c) * S::ld((const T*)x + (n * HW + p) * C + c);
        S::st((T*)dm + n * C + c, (float)a);
    }
}
extern "C" int mcn_channel_scale_bwd(const void* dy, const void* x, const void* m, void* dx, void* dm, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void*) {
    if (!dy || !x || !m || !dx || !dm || N < 0 || HW < 0 || C <= 0) return fail(MCN_E_BADARG, "channel_scale_bwd: bad argument");
    return by_dtype(dtype, "channel_scale_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < (int64_t)N * HW; ++r) {
            const int64_t n = r / HW;
            for (int c = 0; c < C; ++c) S::st((T*)dx + r * C + c, S::ld((const T*)dy + r * C + c) * S::ld((const T*)m + n * C + c));
        }
        chscale_dm<S>(dy, x, dm, N, HW, C);
        return (int)MCN_OK;
    });
}
extern "C" int mcn_channel_scale_bwd_dm(const void* dy, const void* x, void* dm, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void*) {
    if (!dy || !x || !dm || N < 0 || HW < 0 || C <= 0) return fail(MCN_E_BADARG, "channel_scale_bwd_dm: bad argument");
    return by_dtype(dtype, "channel_scale_bwd_dm", [&](auto s) {
        chscale_dm<decltype(s)>(dy, x, dm, N, HW, C);
        return (int)MCN_OK;
    });
}
// BN + swish backward whose output gradient is composed from the squeeze-excite block: round(round(dy * m) + dgap / HW) (mcn.h)
extern "C" int mcn_bn_bwd_se(const void* dy, const void* se_mask, const void* dgap, const void* x, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void*, size_t,
                             void*) {
    if (!dy || !se_mask || !dgap || !x || !dx || !save_mean || !save_invstd || N <= 0 || HW <= 0 || C <= 0) return fail(MCN_E_BADARG, "bn_bwd_se: bad argument");
    return by_dtype(dtype, "bn_bwd_se", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        std::vector<T> g((size_t)N * HW * C);
        const float inv = 1.f / (float)HW;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < (int64_t)N * HW; ++r) {
            const int64_t n = r / HW;
            for (int c = 0; c < C; ++c) {
                const float a = rnd<S>(S::ld((const T*)dy + r * C + c) * S::ld((const T*)se_mask + n * C + c));
                S::st(g.data() + r * C + c, fmaf(S::ld((const T*)dgap + n * C + c), inv, a));
            }
        }
        return bn_bwd_impl<S>(g.data(), x, nullptr, nullptr, gamma, beta, save_mean, save_invstd, dx, nullptr, dgamma, dbeta, grad_scale, (int64_t)N * HW, C, MCN_ACT_SWISH, false);
    });
}

// round 4: the pair with the BN-backward sums taken from per-image sums (mcn.h).  Here: dm from the rebuilt x_se, the sums left at zero, and
// mcn_bn_bwd_se_sums = mcn_bn_bwd_se (the exact composition: what the GPU pair approximates to fp32 summation accuracy)
extern "C" size_t mcn_se_bwd_sums_floats(int32_t N, int64_t, int32_t C, mcn_dtype) { return N > 0 && C > 0 ? (size_t)N * 5 * C : 0; }
extern "C" int mcn_channel_scale_bwd_dm_bnsums(const void* dy, const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dm,
                                               float* sums, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void*) {
    if (!dy || !x || !dm || !sums || !save_mean || !save_invstd || N < 0 || HW <= 0 || C <= 0) return fail(MCN_E_BADARG, "channel_scale_bwd_dm_bnsums: bad argument");
    return by_dtype(dtype, "channel_scale_bwd_dm_bnsums", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        std::vector<T> xse((size_t)N * HW * C);
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < (int64_t)N * HW; ++r)
            for (int c = 0; c < C; ++c) {
                const float sc = (gamma ? gamma[c] : 1.f) * save_invstd[c], sh = (beta ? beta[c] : 0.f) - save_mean[c] * sc;
                const float z = fmaf(S::ld((const T*)x + r * C + c), sc, sh);
                S::st(xse.data() + r * C + c, z * sigmoidf(z));
            }
        chscale_dm<S>(dy, xse.data(), dm, N, HW, C);
        std::fill(sums, sums + (size_t)N * 5 * C, 0.f);
        return (int)MCN_OK;
    });
}
extern "C" int mcn_bn_bwd_se_sums(const void* dy, const void* se_mask, const void* dgap, const void* x, const float* gamma, const float* beta, const float* save_mean,
                                  const float* save_invstd, const float* sums, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int64_t HW, int32_t C,
                                  mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!sums) return fail(MCN_E_BADARG, "bn_bwd_se_sums: bad argument");
    return mcn_bn_bwd_se(dy, se_mask, dgap, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, grad_scale, N, HW, C, dtype, ws, ws_bytes, stream);
}

// ---- depthwise convolution, multiplier 1 (convnet.py:1645) ------------------------------------------------------------------------------
extern "C" size_t mcn_dwconv2d_workspace_bytes(const mcn_conv_geom*, mcn_dtype) { return 0; }
extern "C" int mcn_dwconv2d_fwd(const void* x, const float* w, void* y, const mcn_conv_geom* gg, mcn_dtype dtype, void*) {
    Geo g;
    if (int rc = geo_from(gg, &g)) return rc;
    if (!x || !w || !y || g.Cin != g.Cout) return fail(MCN_E_BADARG, "dwconv2d_fwd: bad argument");
    return by_dtype(dtype, "dwconv2d_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const int C = g.Cin;
        const std::vector<float> wq = cast_weights<S>(w, (size_t)g.KH * g.KW * C);
#pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)g.N * g.OH; ++r) {
            const int n = (int)(r / g.OH), oy = (int)(r % g.OH);
            for (int ox = 0; ox < g.OW; ++ox)
                for (int c = 0; c < C; ++c) {
                    float a = 0.f;
                    for (int kr = 0; kr < g.KH; ++kr) {
                        const int iy = oy * g.SH + kr * g.DH - g.pT;
                        if (iy < 0 || iy >= g.H) continue;
                        for (int ks = 0; ks < g.KW; ++ks) {
                            const int ix = ox * g.SW + ks * g.DW - g.pL;
                            if (ix < 0 || ix >= g.W) continue;
                            a += S::ld((const T*)x + (((long)n * g.H + iy) * g.W + ix) * g.xcs + c) * wq[(size_t)(kr * g.KW + ks) * C + c];
                        }
                    }
                    S::st((T*)y + (((long)n * g.OH + oy) * g.OW + ox) * C + c, a);
                }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_dwconv2d_dgrad(const void* dy, const float* w, void* dx, const mcn_conv_geom* gg, int32_t accumulate, mcn_dtype dtype, void*) {
    Geo g;
    if (int rc = geo_from(gg, &g)) return rc;
    if (!dy || !w || !dx || g.Cin != g.Cout) return fail(MCN_E_BADARG, "dwconv2d_dgrad: bad argument");
    return by_dtype(dtype, "dwconv2d_dgrad", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const int C = g.Cin;
        const std::vector<float> wq = cast_weights<S>(w, (size_t)g.KH * g.KW * C);
#pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)g.N * g.H; ++r) {
            const int n = (int)(r / g.H), iy = (int)(r % g.H);
            for (int ix = 0; ix < g.W; ++ix)
                for (int c = 0; c < C; ++c) {
                    float a = 0.f;
                    for (int kr = 0; kr < g.KH; ++kr) {
                        const int ty = iy + g.pT - kr * g.DH;
                        if (ty < 0 || ty % g.SH || ty / g.SH >= g.OH) continue;
                        for (int ks = 0; ks < g.KW; ++ks) {
                            const int tx = ix + g.pL - ks * g.DW;
                            if (tx < 0 || tx % g.SW || tx / g.SW >= g.OW) continue;
                            a += S::ld((const T*)dy + (((long)n * g.OH + ty / g.SH) * g.OW + tx / g.SW) * C + c) * wq[(size_t)(kr * g.KW + ks) * C + c];
                        }
                    }
                    T* o = (T*)dx + (((long)n * g.H + iy) * g.W + ix) * C + c;
                    S::st(o, accumulate ? S::ld(o) + a : a);
                }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_dwconv2d_wgrad(const void* x, const void* dy, float* dw, const mcn_conv_geom* gg, float grad_scale, mcn_dtype dtype, void*, size_t, void*) {
    Geo g;
    if (int rc = geo_from(gg, &g)) return rc;
    if (!x || !dy || !dw || g.Cin != g.Cout) return fail(MCN_E_BADARG, "dwconv2d_wgrad: bad argument");
    return by_dtype(dtype, "dwconv2d_wgrad", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const int C = g.Cin;
#pragma omp parallel for schedule(static)
        for (long tc = 0; tc < (long)g.KH * g.KW * C; ++tc) {
            const int t = (int)(tc / C), c = (int)(tc % C), kr = t / g.KW, ks = t % g.KW;
            double a = 0.0;
            for (int n = 0; n < g.N; ++n)
                for (int oy = 0; oy < g.OH; ++oy) {
                    const int iy = oy * g.SH + kr * g.DH - g.pT;
                    if (iy < 0 || iy >= g.H) continue;
                    for (int ox = 0; ox < g.OW; ++ox) {
                        const int ix = ox * g.SW + ks * g.DW - g.pL;
                        if (ix < 0 || ix >= g.W) continue;
                        a += (double)S::ld((const T*)x + (((long)n * g.H + iy) * g.W + ix) * g.xcs + c) * S::ld((const T*)dy + (((long)n * g.OH + oy) * g.OW + ox) * C + c);
                    }
                }
            dw[tc] = (float)(a * grad_scale);
        }
        return (int)MCN_OK;
    });
}

// ---- element-wise ------------------------------------------------------------------------------------------------------------------------
template <typename S, typename F>
static int map_n(int64_t n, F&& f) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) f(i);
    return MCN_OK;
}
extern "C" int mcn_channel_affine(const void* x, const float* scale, const float* shift, void* y, int64_t M, int32_t C, mcn_dtype dtype, void*) {
    if (!x || !y || !scale || !shift || M < 0 || C <= 0) return fail(MCN_E_BADARG, "channel_affine: bad argument");
    return by_dtype(dtype, "channel_affine", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(M * C, [&](int64_t i) { const int c = (int)(i % C); S::st((T*)y + i, fmaf(S::ld((const T*)x + i), scale[c], shift[c])); });
    });
}
// ---- depthwise channel multiplier / bias (convnet.py:1634-1650, 1678-1694; mcn.h) -------------------------------------------------------
extern "C" int mcn_channel_repeat_fwd(const void* x, void* y, int64_t M, int32_t C, int32_t mult, mcn_dtype dtype, void*) {
    if (!x || !y || M <= 0 || C <= 0 || mult < 1) return fail(MCN_E_BADARG, "channel_repeat_fwd: bad argument");
    return by_dtype(dtype, "channel_repeat_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const int64_t CO = (int64_t)C * mult;
        return map_n<S>(M * CO, [&](int64_t i) { const int64_t m = i / CO, j = i % CO; ((T*)y)[i] = ((const T*)x)[m * C + j / mult]; });
    });
}
extern "C" int mcn_channel_repeat_bwd(const void* dy, void* dx, int64_t M, int32_t C, int32_t mult, mcn_dtype dtype, void*) {
    if (!dy || !dx || M <= 0 || C <= 0 || mult < 1) return fail(MCN_E_BADARG, "channel_repeat_bwd: bad argument");
    return by_dtype(dtype, "channel_repeat_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(M * C, [&](int64_t i) {
            float a = 0.f;
            for (int q = 0; q < mult; ++q) a += S::ld((const T*)dy + i * mult + q);
            S::st((T*)dx + i, a);
        });
    });
}
extern "C" size_t mcn_bias_grad_workspace_bytes(int64_t M, int32_t C) { return (M > 0 && C > 0) ? 256 : 0; }
extern "C" int mcn_bias_grad(const void* dy, float* dbias, int64_t M, int32_t C, float grad_scale, mcn_dtype dtype, void* ws, size_t ws_bytes, void*) {
    if (!dy || !dbias || M <= 0 || C <= 0) return fail(MCN_E_BADARG, "bias_grad: bad argument");
    if (!ws || ws_bytes < mcn_bias_grad_workspace_bytes(M, C)) return fail(MCN_E_WORKSPACE, "bias_grad: workspace too small");
    return by_dtype(dtype, "bias_grad", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (int c = 0; c < C; ++c) {
            double a = 0.0;
            for (int64_t m = 0; m < M; ++m) a += (double)S::ld((const T*)dy + m * C + c);
            dbias[c] = (float)(a * grad_scale);
        }
        return (int)MCN_OK;
    });
}

extern "C" int mcn_relu_fwd(const void* x, void* y, int64_t n, mcn_dtype dtype, void*) {
    return by_dtype(dtype, "relu_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(n, [&](int64_t i) { const float v = S::ld((const T*)x + i); S::st((T*)y + i, v > 0.f ? v : 0.f); });
    });
}
extern "C" int mcn_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, mcn_dtype dtype, void*) {
    return by_dtype(dtype, "relu_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(n, [&](int64_t i) { S::st((T*)dx + i, S::ld((const T*)y + i) > 0.f ? S::ld((const T*)dy + i) : 0.f); });
    });
}
extern "C" int mcn_add_relu_fwd(const void* a, const void* b, void* y, int64_t n, mcn_act act, mcn_dtype dtype, void*) {
    return by_dtype(dtype, "add_relu_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(n, [&](int64_t i) {
            float v = S::ld((const T*)a + i) + S::ld((const T*)b + i);
            if (act == MCN_ACT_RELU) v = v > 0.f ? v : 0.f;
            S::st((T*)y + i, v);
        });
    });
}
extern "C" int mcn_add_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, mcn_act act, mcn_dtype dtype, void*) {
    return by_dtype(dtype, "add_relu_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(n, [&](int64_t i) { S::st((T*)dx + i, (act != MCN_ACT_RELU || S::ld((const T*)y + i) > 0.f) ? S::ld((const T*)dy + i) : 0.f); });
    });
}
extern "C" int mcn_act_fwd_p(const void* x, void* y, int64_t n, mcn_act act, float param, mcn_dtype dtype, void*) {
    if (act < MCN_ACT_RELU || act > MCN_ACT_TANH) return fail(MCN_E_UNSUPPORTED, "act_fwd: activation unsupported");
    return by_dtype(dtype, "act_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(n, [&](int64_t i) {
            float v = S::ld((const T*)x + i);
            if (act == MCN_ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (act == MCN_ACT_SWISH) v = v * sigmoidf(v);
            else if (act == MCN_ACT_SIGMOID) v = sigmoidf(v);
            else if (act == MCN_ACT_RELU6) v = std::min(std::max(v, 0.f), 6.f);
            else if (act == MCN_ACT_LRELU) v = v > 0.f ? v : param * v;
            else if (act == MCN_ACT_TANH) v = std::tanh(v);
            S::st((T*)y + i, v);
        });
    });
}
extern "C" int mcn_act_fwd(const void* x, void* y, int64_t n, mcn_act act, mcn_dtype dtype, void* s) { return mcn_act_fwd_p(x, y, n, act, 0.2f, dtype, s); }
extern "C" int mcn_act_bwd_p(const void* dy, const void* x, const void* y, void* dx, int64_t n, mcn_act act, float param, mcn_dtype dtype, void*) {
    if (act < MCN_ACT_RELU || act > MCN_ACT_TANH) return fail(MCN_E_UNSUPPORTED, "act_bwd: activation unsupported");
    const bool via_x = act == MCN_ACT_SWISH || act == MCN_ACT_LRELU;
    if ((via_x && !x) || (!via_x && !y)) return fail(MCN_E_BADARG, "act_bwd: missing x / y");
    return by_dtype(dtype, "act_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(n, [&](int64_t i) {
            float g = S::ld((const T*)dy + i);
            if (act == MCN_ACT_RELU) g = S::ld((const T*)y + i) > 0.f ? g : 0.f;
            else if (act == MCN_ACT_SWISH) g *= swish_grad(S::ld((const T*)x + i));
            else if (act == MCN_ACT_SIGMOID) { const float o = S::ld((const T*)y + i); g *= o * (1.f - o); }
            else if (act == MCN_ACT_RELU6) { const float o = S::ld((const T*)y + i); g = (o > 0.f && o < 6.f) ? g : 0.f; }
            else if (act == MCN_ACT_LRELU) g = S::ld((const T*)x + i) > 0.f ? g : param * g;
            else if (act == MCN_ACT_TANH) { const float o = S::ld((const T*)y + i); g *= 1.f - o * o; }
            S::st((T*)dx + i, g);
        });
    });
}
extern "C" int mcn_act_bwd(const void* dy, const void* x, const void* y, void* dx, int64_t n, mcn_act act, mcn_dtype dtype, void* s) {
    return mcn_act_bwd_p(dy, x, y, dx, n, act, 0.2f, dtype, s);
}
extern "C" int mcn_accumulate(void* a, const void* b, int64_t n, mcn_dtype dtype, void*) {
    return by_dtype(dtype, "accumulate", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>(n, [&](int64_t i) { S::st((T*)a + i, S::ld((T*)a + i) + S::ld((const T*)b + i)); });
    });
}
extern "C" int mcn_cast(const void* src, mcn_dtype sdt, void* dst, mcn_dtype ddt, int64_t n, void*) {
    if ((!src || !dst) && n > 0) return fail(MCN_E_BADARG, "cast: null pointer");
    return by_dtype(sdt, "cast", [&](auto a) {
        typedef decltype(a) A;
        return by_dtype(ddt, "cast", [&](auto b) {
            typedef decltype(b) B;
            return map_n<A>(n, [&](int64_t i) { B::st((typename B::T*)dst + i, A::ld((const typename A::T*)src + i)); });
        });
    });
}
extern "C" int mcn_input_prep(const float* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t out_cs, float image_mean, float scale_factor, mcn_layout src_layout,
                              mcn_dtype dtype, void*) {
    if (!x || !y || N < 0 || H <= 0 || W <= 0 || C <= 0 || out_cs < C) return fail(MCN_E_BADARG, "input_prep: bad argument");
    return by_dtype(dtype, "input_prep", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>((int64_t)N * H * W, [&](int64_t p) {
            const int64_t n = p / ((int64_t)H * W), hw = p % ((int64_t)H * W);
            for (int c = 0; c < out_cs; ++c) {
                float v = 0.f;
                if (c < C) v = ((src_layout == MCN_NCHW ? x[(n * C + c) * (int64_t)H * W + hw] : x[p * C + c]) - image_mean) * scale_factor;
                S::st((T*)y + p * out_cs + c, v);
            }
        });
    });
}
extern "C" int mcn_one_hot(const float* labels, float* onehot, int32_t B, int32_t C, void*) {
    if (!labels || !onehot || B < 0 || C <= 0) return fail(MCN_E_BADARG, "one_hot: bad argument");
    for (int b = 0; b < B; ++b) {
        const float l = labels[b];
        for (int c = 0; c < C; ++c) onehot[(size_t)b * C + c] = (l == l && l >= 0.f && l < (float)C && (int)l == c) ? 1.f : 0.f;
    }
    return MCN_OK;
}
extern "C" int mcn_one_hot_seg(const float* labels, float* onehot, int64_t P, int32_t C, void*) {
    if (!labels || !onehot || P < 0 || C <= 0) return fail(MCN_E_BADARG, "one_hot_seg: bad argument");
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; ++p) {
        float l = labels[p];
        if (l != l) l = 0.f;
        const float cls = nearbyintf(l - 1.f);                       // round half to even (tf.round)
        for (int c = 0; c < C; ++c) onehot[p * C + c] = (cls == (float)c) ? 1.f : 0.f;
    }
    return MCN_OK;
}
extern "C" int mcn_copy_channels(const void* src, int32_t ss, int32_t so, void* dst, int32_t ds, int32_t doff, int64_t M, int32_t C, mcn_dtype dtype, void*) {
    if (!src || !dst || M < 0 || C < 0) return fail(MCN_E_BADARG, "copy_channels: bad argument");
    const size_t es = dtype == MCN_F32 ? 4 : 2;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) memcpy((char*)dst + ((size_t)m * ds + doff) * es, (const char*)src + ((size_t)m * ss + so) * es, (size_t)C * es);
    return MCN_OK;
}

// ---- pooling -----------------------------------------------------------------------------------------------------------------------------
template <typename S, bool AFF>
static int maxpool_fwd_impl(const void* x, const float* scale, const float* shift, void* y, int8_t* arg, int N, int H, int W, int C, int KH, int KW, int SH, int SW, int pT, int pL, int OH,
                            int OW) {
    typedef typename S::T T;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)N * OH; ++r) {
        const int n = (int)(r / OH), oy = (int)(r % OH);
        for (int ox = 0; ox < OW; ++ox)
            for (int c = 0; c < C; ++c) {
                float best = -INFINITY;
                int bi = 0;
                for (int kr = 0; kr < KH; ++kr) {
                    const int iy = oy * SH + kr - pT;
                    if (iy < 0 || iy >= H) continue;
                    for (int ks = 0; ks < KW; ++ks) {
                        const int ix = ox * SW + ks - pL;
                        if (ix < 0 || ix >= W) continue;
                        float v = S::ld((const T*)x + (((long)n * H + iy) * W + ix) * C + c);
                        if (AFF) { v = fmaf(v, scale[c], shift[c]); v = rnd<S>(v > 0.f ? v : 0.f); }
                        if (v > best) { best = v; bi = kr * KW + ks; }          // strict: the first maximum wins
                    }
                }
                const long o = (((long)n * OH + oy) * OW + ox) * C + c;
                S::st((T*)y + o, best);
                arg[o] = (int8_t)bi;
            }
    }
    return MCN_OK;
}
extern "C" int mcn_maxpool_fwd(const void* x, void* y, int8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT,
                               int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype, void*) {
    if (!x || !y || !argmax) return fail(MCN_E_BADARG, "maxpool_fwd: null pointer");
    return by_dtype(dtype, "maxpool_fwd", [&](auto s) { return maxpool_fwd_impl<decltype(s), false>(x, nullptr, nullptr, y, argmax, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW); });
}
extern "C" int mcn_maxpool_fwd_affine_relu(const void* x, const float* scale, const float* shift, void* y, int8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH,
                                           int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype, void*) {
    if (!x || !y || !argmax || !scale || !shift) return fail(MCN_E_BADARG, "maxpool_fwd_affine_relu: null pointer");
    return by_dtype(dtype, "maxpool_fwd_affine_relu", [&](auto s) { return maxpool_fwd_impl<decltype(s), true>(x, scale, shift, y, argmax, N, H, W, C, KH, KW, SH, SW, padT, padL, OH, OW); });
}
extern "C" int mcn_maxpool_bwd(const void* dy, const int8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW, int32_t SH, int32_t SW,
                               int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype, void*) {
    if (!dy || !argmax || !dx) return fail(MCN_E_BADARG, "maxpool_bwd: null pointer");
    return by_dtype(dtype, "maxpool_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)N * H; ++r) {
            const int n = (int)(r / H), iy = (int)(r % H);
            for (int ix = 0; ix < W; ++ix)
                for (int c = 0; c < C; ++c) {
                    float a = 0.f;
                    for (int kr = 0; kr < KH; ++kr) {
                        const int ty = iy + padT - kr;
                        if (ty < 0 || ty % SH || ty / SH >= OH) continue;
                        for (int ks = 0; ks < KW; ++ks) {
                            const int tx = ix + padL - ks;
                            if (tx < 0 || tx % SW || tx / SW >= OW) continue;
                            const long o = (((long)n * OH + ty / SH) * OW + tx / SW) * C + c;
                            if (argmax[o] == kr * KW + ks) a += S::ld((const T*)dy + o);
                        }
                    }
                    S::st((T*)dx + (((long)n * H + iy) * W + ix) * C + c, a);
                }
        }
        return (int)MCN_OK;
    });
}
static inline int valid_count(int o, int s, int k, int pad, int size) {
    int n = 0;
    for (int t = 0; t < k; ++t) { const int i = o * s + t - pad; n += (i >= 0 && i < size); }
    return n;
}
extern "C" int mcn_avgpool_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL,
                               int32_t OH, int32_t OW, mcn_dtype dtype, void*) {
    if (!x || !y) return fail(MCN_E_BADARG, "avgpool_fwd: null pointer");
    return by_dtype(dtype, "avgpool_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)N * OH; ++r) {
            const int n = (int)(r / OH), oy = (int)(r % OH);
            for (int ox = 0; ox < OW; ++ox) {
                const int cnt = valid_count(oy, SH, KH, padT, H) * valid_count(ox, SW, KW, padL, W);
                for (int c = 0; c < C; ++c) {
                    float a = 0.f;
                    for (int kr = 0; kr < KH; ++kr) {
                        const int iy = oy * SH + kr - padT;
                        if (iy < 0 || iy >= H) continue;
                        for (int ks = 0; ks < KW; ++ks) {
                            const int ix = ox * SW + ks - padL;
                            if (ix < 0 || ix >= W) continue;
                            a += S::ld((const T*)x + (((long)n * H + iy) * W + ix) * C + c);
                        }
                    }
                    S::st((T*)y + (((long)n * OH + oy) * OW + ox) * C + c, a / (float)(cnt > 0 ? cnt : 1));
                }
            }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL,
                               int32_t OH, int32_t OW, mcn_dtype dtype, void*) {
    if (!dy || !dx) return fail(MCN_E_BADARG, "avgpool_bwd: null pointer");
    return by_dtype(dtype, "avgpool_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)N * H; ++r) {
            const int n = (int)(r / H), iy = (int)(r % H);
            for (int ix = 0; ix < W; ++ix)
                for (int c = 0; c < C; ++c) {
                    float a = 0.f;
                    for (int kr = 0; kr < KH; ++kr) {
                        const int ty = iy + padT - kr;
                        if (ty < 0 || ty % SH || ty / SH >= OH) continue;
                        for (int ks = 0; ks < KW; ++ks) {
                            const int tx = ix + padL - ks;
                            if (tx < 0 || tx % SW || tx / SW >= OW) continue;
                            const int oy = ty / SH, ox = tx / SW;
                            const int cnt = valid_count(oy, SH, KH, padT, H) * valid_count(ox, SW, KW, padL, W);
                            a += S::ld((const T*)dy + (((long)n * OH + oy) * OW + ox) * C + c) / (float)(cnt > 0 ? cnt : 1);
                        }
                    }
                    S::st((T*)dx + (((long)n * H + iy) * W + ix) * C + c, a);
                }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_global_avgpool_fwd(const void* x, void* y, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void*) {
    if (!x || !y || N < 0 || HW <= 0 || C <= 0) return fail(MCN_E_BADARG, "global_avgpool_fwd: bad argument");
    return by_dtype(dtype, "global_avgpool_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        return map_n<S>((int64_t)N * C, [&](int64_t i) {
            const int64_t n = i / C, c = i % C;
            float a = 0.f;
            for (int p = 0; p < HW; ++p) a += S::ld((const T*)x + (n * HW + p) * C + c);
            S::st((T*)y + i, a / (float)HW);
        });
    });
}
template <bool ACC>
static int gap_bwd(const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype) {
    if (!dy || !dx || N < 0 || HW <= 0 || C <= 0) return fail(MCN_E_BADARG, "global_avgpool_bwd: bad argument");
    return by_dtype(dtype, "global_avgpool_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const float inv = 1.f / (float)HW;
        return map_n<S>((int64_t)N * HW * C, [&](int64_t i) {
            const int64_t n = i / ((int64_t)HW * C), c = i % C;
            const float g = S::ld((const T*)dy + n * C + c);
            S::st((T*)dx + i, ACC ? fmaf(g, inv, S::ld((T*)dx + i)) : g * inv);
        });
    });
}
extern "C" int mcn_bn_fwd_train_gap(const void* x, const float* gamma, const float* beta, void* y, void* gap, float* save_mean, float* save_invstd, float* batch_mean,
                                    float* batch_var, float* running_mean, float* running_var, float momentum, int32_t N, int32_t HW, int32_t C, float eps, mcn_act act,
                                    mcn_dtype dtype, void* ws, size_t ws_bytes, void* st) {
    if (!gap || N <= 0 || HW <= 0) return fail(MCN_E_BADARG, "bn_fwd_train_gap: bad argument");
    if ((int)act < 0 || (int)act > 2) return fail(MCN_E_UNSUPPORTED, "bn_fwd_train_gap: activation (none / relu / swish only)");
    std::vector<char> tmp;                                // y == NULL: the means only
    if (!y) { tmp.resize((size_t)N * HW * C * (dtype == MCN_F32 ? 4 : 2)); y = tmp.data(); }
    const int rc = mcn_bn_fwd_train(x, gamma, beta, nullptr, y, nullptr, save_mean, save_invstd, batch_mean, batch_var, running_mean, running_var, momentum, (int64_t)N * HW, C,
                                    eps, act, dtype, ws, ws_bytes, st);
    return rc ? rc : mcn_global_avgpool_fwd(y, gap, N, HW, C, dtype, st);
}
// y = round(round(act(bn(x))) * m[n,c]) from the BN's input (mcn.h)
extern "C" int mcn_bn_act_scale_fwd(const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, const void* m, void* y, int32_t N, int64_t HW,
                                    int32_t C, mcn_act act, mcn_dtype dtype, void*) {
    if (!x || !m || !y || !save_mean || !save_invstd || N < 0 || HW <= 0 || C <= 0) return fail(MCN_E_BADARG, "bn_act_scale_fwd: bad argument");
    return by_dtype(dtype, "bn_act_scale_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < (int64_t)N * HW; ++r) {
            const int64_t n = r / HW;
            for (int c = 0; c < C; ++c) {
                const float sc = (gamma ? gamma[c] : 1.f) * save_invstd[c], sh = (beta ? beta[c] : 0.f) - save_mean[c] * sc;
                float o = fmaf(S::ld((const T*)x + r * C + c), sc, sh);
                if (act == MCN_ACT_RELU) o = o > 0.f ? o : 0.f;
                else if (act == MCN_ACT_SWISH) o = o * sigmoidf(o);
                S::st((T*)y + r * C + c, rnd<S>(o) * S::ld((const T*)m + n * C + c));
            }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_global_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void*) { return gap_bwd<false>(dy, dx, N, HW, C, dtype); }
extern "C" int mcn_global_avgpool_bwd_acc(const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void*) { return gap_bwd<true>(dy, dx, N, HW, C, dtype); }

// ---- fully connected (convnet.py:1743) -----------------------------------------------------------------------------------------------------
extern "C" size_t mcn_fc_workspace_bytes(int32_t, int32_t, int32_t, mcn_dtype) { return 0; }
extern "C" int mcn_fc_fwd(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t In, int32_t Out, mcn_dtype dtype, void*, size_t, void*) {
    if (!x || !w || !y || B < 0 || In <= 0 || Out <= 0) return fail(MCN_E_BADARG, "fc_fwd: bad argument");
    return by_dtype(dtype, "fc_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        const std::vector<float> wq = cast_weights<S>(w, (size_t)In * Out);
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; ++b)
            for (int o = 0; o < Out; ++o) {
                double a = bias ? bias[o] : 0.0;
                for (int i = 0; i < In; ++i) a += (double)S::ld((const T*)x + (size_t)b * In + i) * wq[(size_t)i * Out + o];
                S::st((T*)y + (size_t)b * Out + o, (float)a);
            }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_fc_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias, float grad_scale, int32_t B, int32_t In, int32_t Out, mcn_dtype dtype,
                          void*, size_t, void*) {
    if (!dy || !x || !w || B < 0 || In <= 0 || Out <= 0) return fail(MCN_E_BADARG, "fc_bwd: bad argument");
    return by_dtype(dtype, "fc_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
        if (dx) {
            const std::vector<float> wq = cast_weights<S>(w, (size_t)In * Out);
#pragma omp parallel for schedule(static)
            for (int b = 0; b < B; ++b)
                for (int i = 0; i < In; ++i) {
                    double a = 0.0;
                    for (int o = 0; o < Out; ++o) a += (double)S::ld((const T*)dy + (size_t)b * Out + o) * wq[(size_t)i * Out + o];
                    S::st((T*)dx + (size_t)b * In + i, (float)a);
                }
        }
        if (dw) {
#pragma omp parallel for schedule(static)
            for (int i = 0; i < In; ++i)
                for (int o = 0; o < Out; ++o) {
                    double a = 0.0;
                    for (int b = 0; b < B; ++b) a += (double)S::ld((const T*)x + (size_t)b * In + i) * S::ld((const T*)dy + (size_t)b * Out + o);
                    dw[(size_t)i * Out + o] = (float)(a * grad_scale);
                }
        }
        if (dbias)
            for (int o = 0; o < Out; ++o) {
                double a = 0.0;
                for (int b = 0; b < B; ++b) a += S::ld((const T*)dy + (size_t)b * Out + o);
                dbias[o] = (float)(a * grad_scale);
            }
        return (int)MCN_OK;
    });
}

// ---- loss (convnet.py:528-601) ---------------------------------------------------------------------------------------------------------------
static int xent_rows(const float* logits, const float* labels, const float* class_w, float* pred, float* ce, float* coef, float* dlogits, float* loss, int64_t B, int32_t C, float ls,
                     float loss_scale, const float* avg = nullptr, float fgamma = 0.f, float salpha = 0.f) {
    if (!logits || !labels || !ce || !coef || B <= 0 || C <= 0) return fail(MCN_E_BADARG, "softmax_xent: bad argument");
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* z = logits + b * C;
        const float* yv = labels + b * C;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = std::max(mx, z[c]);
        float se = 0.f, sy = 0.f, bw = 0.f;
        for (int c = 0; c < C; ++c) {
            se += expf(z[c] - mx);
            sy += yv[c];
            bw += yv[c] * (class_w ? class_w[c] : 1.f);
        }
        const float lse = logf(se);
        const float valid = (sy > 1.f - 1e-5f && sy < 1.f + 1e-5f) ? 1.f : 0.f;
        const float cf = bw * valid;
        const float* av = (avg && ls > 0.f) ? avg + b * C : nullptr;
        float lab_sum = ls > 0.f ? sy * (1.f - ls) + ls : sy;
        if (av) {
            lab_sum = 0.f;
            for (int c = 0; c < C; ++c) lab_sum += yv[c] * (1.f - ls) + ls * av[c];
        }
        const float gscale = cf * loss_scale / (float)B;
        float cel = 0.f, pt = 0.f;
        for (int c = 0; c < C; ++c) {
            const float lsm = z[c] - mx - lse;
            const float lab = ls > 0.f ? yv[c] * (1.f - ls) + ls * (av ? av[c] : 1.f / (float)C) : yv[c];
            cel -= lab * lsm;
            pt += yv[c] * expf(lsm);
        }
        // focal factors (convnet.py:581-592): F = (1 - p_t)^gamma differentiated through the softmax, S under stop_gradient
        float F = 1.f, dF = 0.f, S = 1.f;
        if (fgamma > 0.f) {
            const float om = std::max(1.f - pt, 0.f);
            F = std::pow(om, fgamma);
            dF = om > 0.f ? -fgamma * std::pow(om, fgamma - 1.f) : 0.f;
        }
        if (salpha > 0.f) S = (1.f - 1.f / (1.f + expf(-salpha * (pt - 0.5f)))) / (1.f - 1.f / (1.f + expf(0.5f * salpha)));
        for (int c = 0; c < C; ++c) {
            const float lsm = z[c] - mx - lse, p = expf(lsm);
            const float lab = ls > 0.f ? yv[c] * (1.f - ls) + ls * (av ? av[c] : 1.f / (float)C) : yv[c];
            if (pred) pred[b * C + c] = p;
            if (dlogits) dlogits[b * C + c] = (F * S * (p * lab_sum - lab) + cel * dF * S * (yv[c] * p - pt * p)) * gscale;
        }
        cel *= F * S;
        ce[b] = cel;
        coef[b] = cf;
    }
    if (loss) {
        double a = 0.0;
        for (int64_t b = 0; b < B; ++b) a += (double)ce[b] * coef[b];
        loss[0] = (float)(a / (double)B);
    }
    return MCN_OK;
}
extern "C" int mcn_softmax_xent_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce, float* coef, float* dlogits, float* loss, int32_t B,
                                        int32_t C, float label_smoothing, float loss_scale, void*) {
    return xent_rows(logits, labels, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale);
}
extern "C" int mcn_softmax_xent_rows_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce, float* coef, float* dlogits, float* loss,
                                             int64_t B, int32_t C, float label_smoothing, float loss_scale, void*, size_t, void*) {
    return xent_rows(logits, labels, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale);
}
extern "C" int mcn_softmax_xent_rows_soft_fwd_bwd(const float* logits, const float* labels, const float* avg_labels, const float* class_w, float* pred, float* ce, float* coef,
                                                  float* dlogits, float* loss, int64_t B, int32_t C, float label_smoothing, float loss_scale, void*, size_t, void*) {
    return xent_rows(logits, labels, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale, avg_labels);
}
extern "C" int mcn_softmax_xent_focal_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce, float* coef, float* dlogits, float* loss,
                                              int32_t B, int32_t C, float label_smoothing, float loss_scale, float fg, float sa, void*) {
    if (fg < 0.f || sa < 0.f) return fail(MCN_E_BADARG, "softmax_xent: negative focal factor");
    return xent_rows(logits, labels, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale, nullptr, fg, sa);
}
extern "C" int mcn_softmax_xent_rows_focal_fwd_bwd(const float* logits, const float* labels, const float* avg_labels, const float* class_w, float* pred, float* ce, float* coef,
                                                   float* dlogits, float* loss, int64_t B, int32_t C, float label_smoothing, float loss_scale, float fg, float sa, void*, size_t,
                                                   void*) {
    if (fg < 0.f || sa < 0.f) return fail(MCN_E_BADARG, "softmax_xent_rows: negative focal factor");
    return xent_rows(logits, labels, class_w, pred, ce, coef, dlogits, loss, B, C, label_smoothing, loss_scale, avg_labels, fg, sa);
}
extern "C" int mcn_l1_loss(const float* w, int64_t n, float factor, float* out, void*, size_t, void*) {
    if (!w || !out || n < 0) return fail(MCN_E_BADARG, "l1_loss: bad argument");
    double a = 0.0;
    for (int64_t i = 0; i < n; ++i) a += std::fabs((double)w[i]);
    out[0] += (float)((double)factor * a);
    return MCN_OK;
}
extern "C" int mcn_l1_grad_h(float* g, const float* w, int64_t n, float l1, const float* hyper, void*) {
    if (!g || !w || !hyper || n < 0) return fail(MCN_E_BADARG, "l1_grad: bad argument");
    const float f = l1 / hyper[3];
    for (int64_t i = 0; i < n; ++i) g[i] += w[i] > 0.f ? f : (w[i] < 0.f ? -f : 0.f);
    return MCN_OK;
}

// ---- segmentation path: bilinear resize (tf.image.resize_bilinear, convnet.py:2396) ------------------------------------------------------------
static inline void resize_src(int o, int in, int out, bool align, int& i0, int& i1, float& f) {
    const float scale = (align && out > 1) ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out;
    const float pos = (float)o * scale;
    i0 = (int)floorf(pos);
    if (i0 > in - 1) i0 = in - 1;
    i1 = std::min(i0 + 1, in - 1);
    f = pos - (float)i0;
}
extern "C" int mcn_resize_bilinear_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t align, mcn_dtype dtype, void*) {
    if (!x || !y || N < 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return fail(MCN_E_BADARG, "resize_bilinear_fwd: bad argument");
    return by_dtype(dtype, "resize_bilinear_fwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (long r = 0; r < (long)N * OH; ++r) {
            const int n = (int)(r / OH), oy = (int)(r % OH);
            int y0, y1; float fy;
            resize_src(oy, H, OH, align != 0, y0, y1, fy);
            for (int ox = 0; ox < OW; ++ox) {
                int x0, x1; float fx;
                resize_src(ox, W, OW, align != 0, x0, x1, fx);
                for (int c = 0; c < C; ++c) {
                    auto at = [&](int yy, int xx) { return S::ld((const T*)x + (((long)n * H + yy) * W + xx) * C + c); };
                    const float top = at(y0, x0) + (at(y0, x1) - at(y0, x0)) * fx, bot = at(y1, x0) + (at(y1, x1) - at(y1, x0)) * fx;
                    S::st((T*)y + (((long)n * OH + oy) * OW + ox) * C + c, top + (bot - top) * fy);
                }
            }
        }
        return (int)MCN_OK;
    });
}
extern "C" int mcn_resize_bilinear_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t align, mcn_dtype dtype, void*) {
    if (!dy || !dx || N < 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return fail(MCN_E_BADARG, "resize_bilinear_bwd: bad argument");
    return by_dtype(dtype, "resize_bilinear_bwd", [&](auto s) {
        typedef decltype(s) S;
        typedef typename S::T T;
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            std::vector<float> acc((size_t)H * W * C, 0.f);
            for (int oy = 0; oy < OH; ++oy) {
                int y0, y1; float fy;
                resize_src(oy, H, OH, align != 0, y0, y1, fy);
                for (int ox = 0; ox < OW; ++ox) {
                    int x0, x1; float fx;
                    resize_src(ox, W, OW, align != 0, x0, x1, fx);
                    for (int c = 0; c < C; ++c) {
                        const float g = S::ld((const T*)dy + (((long)n * OH + oy) * OW + ox) * C + c);
                        acc[((size_t)y0 * W + x0) * C + c] += g * (1.f - fy) * (1.f - fx);
                        acc[((size_t)y0 * W + x1) * C + c] += g * (1.f - fy) * fx;
                        acc[((size_t)y1 * W + x0) * C + c] += g * fy * (1.f - fx);
                        acc[((size_t)y1 * W + x1) * C + c] += g * fy * fx;
                    }
                }
            }
            for (size_t i = 0; i < acc.size(); ++i) S::st((T*)dx + (size_t)n * H * W * C + i, acc[i]);
        }
        return (int)MCN_OK;
    });
}

// ---- regulariser and optimizer (all fp32) ------------------------------------------------------------------------------------------------------
extern "C" int mcn_l2_loss(const float* w, int64_t n, float factor, float* out, void*, size_t, void*) {
    if ((!w && n > 0) || !out || n < 0) return fail(MCN_E_BADARG, "l2_loss: bad argument");
    double a = 0.0;
#pragma omp parallel for reduction(+ : a) schedule(static)
    for (int64_t i = 0; i < n; ++i) a += (double)w[i] * w[i];
    out[0] += (float)(0.5 * factor * a);
    return MCN_OK;
}
static int sgd(float* w, const float* g, float* a, float* ema, int64_t n, float lr, float mom, float l2, float wd, float d, float gs) {
    if (!w || !g || !a || n < 0) return fail(MCN_E_BADARG, "sgd_nesterov_fused: bad argument");
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float wv = w[i];
        if (ema) ema[i] = d * ema[i] + (1.f - d) * wv;
        const float gt = gs * g[i] + l2 * wv;
        const float av = mom * a[i] + gt;
        float nw = wv - lr * gt - lr * mom * av;
        if (wd > 0.f) nw -= wd * nw;
        w[i] = nw;
        a[i] = av;
    }
    return MCN_OK;
}
extern "C" int mcn_sgd_nesterov_fused(float* w, const float* g, float* accum, float* ema, int64_t n, float lr, float momentum, float l2, float wd, float ema_decay, float grad_scale,
                                      void*) { return sgd(w, g, accum, ema, n, lr, momentum, l2, wd, ema_decay, grad_scale); }
extern "C" int mcn_sgd_nesterov_fused_h(float* w, const float* g, float* accum, float* ema, int64_t n, const float* hyper, float momentum, float l2, int32_t use_wd, void*) {
    if (!hyper) return fail(MCN_E_BADARG, "sgd_nesterov_fused_h: null hyper-parameter buffer");
    return sgd(w, g, accum, ema, n, hyper[0], momentum, l2, use_wd ? hyper[1] : 0.f, hyper[2], hyper[3]);
}
static int decay(float* w, int64_t n, float wd, int32_t mode, float delta) {
    if ((!w && n > 0) || n < 0 || mode < 0 || mode > 2) return fail(MCN_E_BADARG, "decoupled_decay: bad argument");
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float v = w[i];
        float dd;
        if (mode == 0) dd = v;
        else if (mode == 1) dd = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
        else { const float r = v / delta; dd = v / sqrtf(1.f + r * r); }
        w[i] = v - wd * dd;
    }
    return MCN_OK;
}
extern "C" int mcn_decoupled_decay(float* w, int64_t n, float wd, int32_t mode, float delta, void*) { return decay(w, n, wd, mode, delta); }
extern "C" int mcn_decoupled_decay_h(float* w, int64_t n, const float* hyper, int32_t mode, float delta, void*) {
    if (!hyper) return fail(MCN_E_BADARG, "decoupled_decay_h: null hyper-parameter buffer");
    return decay(w, n, hyper[1], mode, delta);
}
extern "C" int mcn_clip_by_global_norm(float* g, const float* w, int64_t n, int64_t n_l2, float l2, float threshold, float* norm_out, void*, size_t, void*) {
    if (!g || n < 0 || n_l2 < 0 || n_l2 > n || !(threshold > 0.f) || (n_l2 > 0 && l2 != 0.f && !w)) return fail(MCN_E_BADARG, "clip_by_global_norm: bad argument");
    double a = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (i < n_l2 && l2 != 0.f) g[i] = fmaf(l2, w[i], g[i]);
        a += (double)g[i] * g[i];
    }
    const float norm = (float)std::sqrt(a), sc = threshold / std::max(norm, threshold);
    for (int64_t i = 0; i < n; ++i) g[i] *= sc;
    if (norm_out) norm_out[0] = norm;
    return MCN_OK;
}
extern "C" int mcn_clip_by_global_norm_runs(float* g, const float* w, const int64_t* runs, int32_t nruns, float l2, float threshold, float* norm_out, void*, size_t,
                                            void*) {
    if (!g || !runs || nruns < 0 || !(threshold > 0.f)) return fail(MCN_E_BADARG, "clip_by_global_norm_runs: bad argument");
    double a = 0.0;
    for (int r = 0; r < nruns; ++r) {
        const int64_t s = runs[3 * r], e = runs[3 * r + 1], le = runs[3 * r + 2];
        if (s < 0 || e < s || le < s || le > e || (le > s && l2 != 0.f && !w)) return fail(MCN_E_BADARG, "clip_by_global_norm_runs: bad run");
        for (int64_t i = s; i < e; ++i) {
            if (i < le && l2 != 0.f) g[i] = fmaf(l2, w[i], g[i]);
            a += (double)g[i] * g[i];
        }
    }
    const float norm = (float)std::sqrt(a), sc = threshold / std::max(norm, threshold);
    for (int r = 0; r < nruns; ++r)
        for (int64_t i = runs[3 * r]; i < runs[3 * r + 1]; ++i) g[i] *= sc;
    if (norm_out) norm_out[0] = norm;
    return MCN_OK;
}
extern "C" int mcn_ema_update(float* shadow, const float* v, int64_t n, float decay_, void*) {
    if (!shadow || !v || n < 0) return fail(MCN_E_BADARG, "ema_update: bad argument");
    for (int64_t i = 0; i < n; ++i) shadow[i] = decay_ * shadow[i] + (1.f - decay_) * v[i];
    return MCN_OK;
}
extern "C" int mcn_ema_update_h(float* shadow, const float* v, int64_t n, const float* hyper, void*) {
    if (!hyper) return fail(MCN_E_BADARG, "ema_update_h: null hyper-parameter buffer");
    return mcn_ema_update(shadow, v, n, hyper[2], nullptr);
}
extern "C" int mcn_bn_running_chain_strided(float* running, const float* batch, int32_t towers, int64_t n, int64_t tower_stride, float momentum, void*) {
    if (!running || !batch || towers < 0 || n < 0 || tower_stride < n) return fail(MCN_E_BADARG, "bn_running_chain: bad argument");
    for (int64_t i = 0; i < n; ++i) {
        float r = running[i];
        for (int k = 0; k < towers; ++k) r = momentum * r + (1.f - momentum) * batch[(size_t)k * tower_stride + i];
        running[i] = r;
    }
    return MCN_OK;
}
extern "C" int mcn_bn_running_chain(float* running, const float* batch, int32_t towers, int64_t n, float momentum, void* s) {
    return mcn_bn_running_chain_strided(running, batch, towers, n, n, momentum, s);
}
