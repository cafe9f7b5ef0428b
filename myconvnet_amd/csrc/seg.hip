// seg.hip — the extra ops of the DeepLabv3+ / SegNet path: bilinear resize with align_corners
// (tf.image.resize_bilinear, reference convnet.py:2378-2406, called from models/deeplabv3plus.py:64,74), channel concat
// (tf.concat, models/deeplabv3plus.py:101,110) and SegNet's label encoding (segmentation/segnet.py:31-50).  HBM-bound.
#include "common.h"

struct ResizeParams {
    int N, H, W, C, OH, OW;
    float sy, sx;          // source step per output index
    int align;
};

// TF ResizeBilinear coordinates: align_corners: src = o * (in-1)/(out-1); else half-pixel centres clamped at 0
__device__ __forceinline__ void rs_coord(int o, float scale, int align, int in, int& lo, int& hi, float& f) {
    float src = align ? (float)o * scale : fmaxf(((float)o + 0.5f) * scale - 0.5f, 0.f);
    int l = (int)floorf(src);
    if (l > in - 1) l = in - 1;
    lo = l;
    hi = l + 1 < in ? l + 1 : in - 1;
    f = src - (float)l;
}

template <typename T, int VEC>
__device__ __forceinline__ void ldvec(const T* p, float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        v[0] = to_f32(p[0]);
    } else {
        const Chunk<T> c = load_chunk<T>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = c.get(i);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void stvec(T* p, const float (&v)[VEC]) {
    if constexpr (VEC == 1) {
        p[0] = from_f32<T>(v[0]);
    } else {
        Chunk<T> c;
#pragma unroll
        for (int i = 0; i < VEC; ++i) c.set(i, v[i]);
        store_chunk<T>(p, c);
    }
}

// CE = 16-byte chunk of channels, or 1 when C is not a chunk multiple (the class-logit maps: C = num_classes)
template <typename T, int CE>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const ResizeParams p) {
    const int cch = p.C / CE;
    const long total = (long)p.N * p.OH * p.OW * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cch);
        long q = i / cch;
        const int ox = (int)(q % p.OW);
        q /= p.OW;
        const int oy = (int)(q % p.OH);
        const long n = q / p.OH;
        int ylo, yhi, xlo, xhi;
        float fy, fx;
        rs_coord(oy, p.sy, p.align, p.H, ylo, yhi, fy);
        rs_coord(ox, p.sx, p.align, p.W, xlo, xhi, fx);
        const T* base = x + n * p.H * p.W * p.C + (long)ch * CE;
        float a[CE], b[CE], c[CE], d[CE], o[CE];
        ldvec<T, CE>(base + ((long)ylo * p.W + xlo) * p.C, a);
        ldvec<T, CE>(base + ((long)ylo * p.W + xhi) * p.C, b);
        ldvec<T, CE>(base + ((long)yhi * p.W + xlo) * p.C, c);
        ldvec<T, CE>(base + ((long)yhi * p.W + xhi) * p.C, d);
#pragma unroll
        for (int k = 0; k < CE; ++k) {
            const float top = a[k] * (1.f - fx) + b[k] * fx;
            const float bot = c[k] * (1.f - fx) + d[k] * fx;
            o[k] = top * (1.f - fy) + bot * fy;
        }
        stvec<T, CE>(y + i * CE, o);
    }
}

// gather form of the gradient (deterministic): an input pixel collects dy from every output whose lower or upper
// neighbour it is; candidates come from the inverse of the coordinate map, membership is re-checked with the forward's
// own arithmetic so that rounding can not drop or duplicate a term.
template <typename T, int CE>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, const ResizeParams p, float isy, float isx) {
    const unsigned cch = (unsigned)(p.C / CE);
    const unsigned total = (unsigned)((long)p.N * p.H * p.W * cch);          // (host: < 2^32; 64-bit divisions cost more than the taps)
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        unsigned q = i / cch;
        const int ch = (int)(i - q * cch);
        const unsigned q2 = q / (unsigned)p.W;
        const int ix = (int)(q - q2 * (unsigned)p.W);
        const long n = q2 / (unsigned)p.H;
        const int iy = (int)(q2 - (unsigned)n * (unsigned)p.H);
        // outputs with src in (i-1, i+1); one candidate of slack on both sides
        int oy0 = (int)floorf(((float)iy - 1.f) * isy) - 1, oy1 = (int)ceilf(((float)iy + 1.f) * isy) + 1;
        int ox0 = (int)floorf(((float)ix - 1.f) * isx) - 1, ox1 = (int)ceilf(((float)ix + 1.f) * isx) + 1;
        if (!p.align) { oy0 -= 1; oy1 += 1; ox0 -= 1; ox1 += 1; }
        if (oy0 < 0) oy0 = 0;
        if (ox0 < 0) ox0 = 0;
        if (oy1 > p.OH - 1) oy1 = p.OH - 1;
        if (ox1 > p.OW - 1) ox1 = p.OW - 1;
        float acc[CE];
#pragma unroll
        for (int k = 0; k < CE; ++k) acc[k] = 0.f;
        const T* base = dy + n * p.OH * p.OW * p.C + (long)ch * CE;
        for (int oy = oy0; oy <= oy1; ++oy) {
            int lo, hi;
            float f;
            rs_coord(oy, p.sy, p.align, p.H, lo, hi, f);
            const float wy = (lo == iy ? 1.f - f : 0.f) + (hi == iy ? f : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                rs_coord(ox, p.sx, p.align, p.W, lo, hi, f);
                const float wx = (lo == ix ? 1.f - f : 0.f) + (hi == ix ? f : 0.f);
                if (wx == 0.f) continue;
                float g[CE];
                ldvec<T, CE>(base + ((long)oy * p.OW + ox) * p.C, g);
                const float w = wy * wx;
#pragma unroll
                for (int k = 0; k < CE; ++k) acc[k] = fmaf(g[k], w, acc[k]);
            }
        }
        stvec<T, CE>(dx + (long)i * CE, acc);
    }
}

// Channel counts that are no chunk multiple (class-logit maps, C = num_classes <= 32): one thread per PIXEL with the
// channels in an inner loop — the coordinate arithmetic runs once per pixel instead of once per element, with 32-bit indices
// (forward: 0.41 -> 0.17 ms on the 513x513x19 logits; the same form of the gather backward has too few threads: 1.6 ms, not kept)
#define MCN_RESIZE_PIX_MAXC 32
template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_pix_kernel(const T* __restrict__ x, T* __restrict__ y, const ResizeParams p) {
    const unsigned total = (unsigned)((long)p.N * p.OH * p.OW);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned q = i / (unsigned)p.OW;
        const int ox = (int)(i - q * (unsigned)p.OW);
        const unsigned n = q / (unsigned)p.OH;
        const int oy = (int)(q - n * (unsigned)p.OH);
        int ylo, yhi, xlo, xhi;
        float fy, fx;
        rs_coord(oy, p.sy, p.align, p.H, ylo, yhi, fy);
        rs_coord(ox, p.sx, p.align, p.W, xlo, xhi, fx);
        const T* base = x + (long)n * p.H * p.W * p.C;
        const T* a = base + ((long)ylo * p.W + xlo) * p.C;
        const T* b = base + ((long)ylo * p.W + xhi) * p.C;
        const T* c = base + ((long)yhi * p.W + xlo) * p.C;
        const T* d = base + ((long)yhi * p.W + xhi) * p.C;
        T* o = y + (long)i * p.C;
        for (int k = 0; k < p.C; ++k) {
            const float top = to_f32(a[k]) * (1.f - fx) + to_f32(b[k]) * fx;
            const float bot = to_f32(c[k]) * (1.f - fx) + to_f32(d[k]) * fx;
            o[k] = from_f32<T>(top * (1.f - fy) + bot * fy);
        }
    }
}
static int resize_setup(const void* a, const void* b, int N, int H, int W, int C, int OH, int OW, int align, mcn_dtype dt, ResizeParams* p, const char* nm) {
    if (!a || !b || N < 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) MCN_FAIL(MCN_E_BADARG, "%s: bad argument", nm);
    if (!mcn_dtype_ok(dt)) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: dtype %d unsupported", nm, (int)dt);
    p->N = N; p->H = H; p->W = W; p->C = C; p->OH = OH; p->OW = OW; p->align = align ? 1 : 0;
    if (align) {
        p->sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
        p->sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    } else {
        p->sy = (float)H / (float)OH;
        p->sx = (float)W / (float)OW;
    }
    return MCN_OK;
}
static inline unsigned seg_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)(b < 1 ? 1 : b);
}
extern "C" int mcn_resize_bilinear_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                                       int32_t align_corners, mcn_dtype dtype, void* stream) {
    ResizeParams p;
    if (int rc = resize_setup(x, y, N, H, W, C, OH, OW, align_corners, dtype, &p, "resize_bilinear_fwd")) return rc;
    const int ce = C % (dtype == MCN_F32 ? 4 : 8) ? 1 : (dtype == MCN_F32 ? 4 : 8);
    const long total = (long)N * OH * OW * (C / ce);
    if (total == 0) return MCN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (ce == 1 && C <= MCN_RESIZE_PIX_MAXC && (long)N * OH * OW < 0x7fffffffL) {
        const long npix = (long)N * OH * OW;
        if (dtype == MCN_F32) hipLaunchKernelGGL((resize_fwd_pix_kernel<float>), dim3(seg_blocks(npix)), dim3(256), 0, st, (const float*)x, (float*)y, p);
        else if (dtype == MCN_F16) hipLaunchKernelGGL((resize_fwd_pix_kernel<f16_t>), dim3(seg_blocks(npix)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, p);
        else hipLaunchKernelGGL((resize_fwd_pix_kernel<bf16_t>), dim3(seg_blocks(npix)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, p);
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    if (dtype == MCN_F32) {
        if (ce == 4) hipLaunchKernelGGL((resize_fwd_kernel<float, 4>), dim3(seg_blocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, p);
        else hipLaunchKernelGGL((resize_fwd_kernel<float, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const float*)x, (float*)y, p);
    } else if (dtype == MCN_F16) {
        if (ce == 8) hipLaunchKernelGGL((resize_fwd_kernel<f16_t, 8>), dim3(seg_blocks(total)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, p);
        else hipLaunchKernelGGL((resize_fwd_kernel<f16_t, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, p);
    } else {
        if (ce == 8) hipLaunchKernelGGL((resize_fwd_kernel<bf16_t, 8>), dim3(seg_blocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, p);
        else if (dtype == MCN_F16) hipLaunchKernelGGL((resize_fwd_kernel<f16_t, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, p);
        else hipLaunchKernelGGL((resize_fwd_kernel<bf16_t, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, p);
    }
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_resize_bilinear_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                                       int32_t align_corners, mcn_dtype dtype, void* stream) {
    ResizeParams p;
    if (int rc = resize_setup(dy, dx, N, H, W, C, OH, OW, align_corners, dtype, &p, "resize_bilinear_bwd")) return rc;
    const int ce = C % (dtype == MCN_F32 ? 4 : 8) ? 1 : (dtype == MCN_F32 ? 4 : 8);
    const long total = (long)N * H * W * (C / ce);
    if (total == 0) return MCN_OK;
    if (total >= 0xffffffffL) MCN_FAIL(MCN_E_UNSUPPORTED, "resize_bilinear_bwd: more than 2^32 elements");
    hipStream_t st = (hipStream_t)stream;
    // inverse steps; a zero forward step (single output row / column) means every output maps to input 0
    const float isy = p.sy > 0.f ? 1.f / p.sy : (float)OH, isx = p.sx > 0.f ? 1.f / p.sx : (float)OW;
    if (dtype == MCN_F32) {
        if (ce == 4) hipLaunchKernelGGL((resize_bwd_kernel<float, 4>), dim3(seg_blocks(total)), dim3(256), 0, st, (const float*)dy, (float*)dx, p, isy, isx);
        else hipLaunchKernelGGL((resize_bwd_kernel<float, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const float*)dy, (float*)dx, p, isy, isx);
    } else if (dtype == MCN_F16) {
        if (ce == 8) hipLaunchKernelGGL((resize_bwd_kernel<f16_t, 8>), dim3(seg_blocks(total)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, p, isy, isx);
        else hipLaunchKernelGGL((resize_bwd_kernel<f16_t, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, p, isy, isx);
    } else {
        if (ce == 8) hipLaunchKernelGGL((resize_bwd_kernel<bf16_t, 8>), dim3(seg_blocks(total)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, p, isy, isx);
        else if (dtype == MCN_F16) hipLaunchKernelGGL((resize_bwd_kernel<f16_t, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const f16_t*)dy, (f16_t*)dx, p, isy, isx);
        else hipLaunchKernelGGL((resize_bwd_kernel<bf16_t, 1>), dim3(seg_blocks(total)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, p, isy, isx);
    }
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// dst[m][dst_off + c] = src[m][src_off + c], c < C: tf.concat along channels (one call per input) and its gradient (slices)
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const T* __restrict__ src, int ss, int so, T* __restrict__ dst, int ds, int dof, long M, int C) {
    constexpr int CE = VecTraits<T>::CE;
    const int cch = C / CE;
    const long total = M * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cch;
        const int c = (int)(i - m * cch) * CE;
        store_chunk<T>(dst + m * ds + dof + c, load_chunk<T>(src + m * ss + so + c));
    }
}
extern "C" int mcn_copy_channels(const void* src, int32_t src_stride, int32_t src_offset, void* dst, int32_t dst_stride, int32_t dst_offset, int64_t M,
                                 int32_t C, mcn_dtype dtype, void* stream) {
    if (!src || !dst || M < 0 || C <= 0 || src_offset < 0 || dst_offset < 0 || src_offset + C > src_stride || dst_offset + C > dst_stride)
        MCN_FAIL(MCN_E_BADARG, "copy_channels: bad argument");
    if (!mcn_dtype_ok(dtype)) MCN_FAIL(MCN_E_UNSUPPORTED, "copy_channels: dtype %d unsupported", (int)dtype);
    const int ce = dtype == MCN_F32 ? 4 : 8;
    if (C % ce || src_stride % ce || dst_stride % ce || src_offset % ce || dst_offset % ce)
        MCN_FAIL(MCN_E_UNSUPPORTED, "copy_channels: channel counts / offsets must be multiples of the 16-byte chunk");
    if (M == 0) return MCN_OK;
    const long total = (long)M * (C / ce);
    if (dtype == MCN_F32) hipLaunchKernelGGL((copy_channels_kernel<float>), dim3(seg_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const float*)src, src_stride, src_offset, (float*)dst, dst_stride, dst_offset, (long)M, C);
    else if (dtype == MCN_F16) hipLaunchKernelGGL((copy_channels_kernel<f16_t>), dim3(seg_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)src, src_stride, src_offset, (f16_t*)dst, dst_stride, dst_offset, (long)M, C);
    else hipLaunchKernelGGL((copy_channels_kernel<bf16_t>), dim3(seg_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, src_stride, src_offset, (bf16_t*)dst, dst_stride, dst_offset, (long)M, C);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// SegNet labels: NaN -> 0; class = round(label - 1) (half to even, tf.math.round); class -1 (label 0) or >= C -> all-zero row
__global__ __launch_bounds__(256) void one_hot_seg_kernel(const float* __restrict__ labels, float* __restrict__ onehot, long P, int C, int vec) {
    // a thread writes four consecutive elements as one 16-byte store (4-byte stores ran at half the write bandwidth); the elements
    // may straddle pixels, so the (pixel, class) pair is stepped rather than divided again
    const long total = P * C;
    const long nvec = vec ? total / 4 : 0;                        // (vec: the output is 16-byte aligned)
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        const long i = v * 4;
        long q = i / C;
        int c = (int)(i - q * C);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float l = labels[q];
            if (l != l) l = 0.f;
            o[k] = (int)rintf(l - 1.f) == c ? 1.f : 0.f;
            if (++c == C) { c = 0; ++q; }
        }
        *reinterpret_cast<f32x4*>(onehot + i) = o;
    }
    for (long i = nvec * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long q = i / C;
        const int c = (int)(i - q * C);
        float l = labels[q];
        if (l != l) l = 0.f;
        onehot[i] = (int)rintf(l - 1.f) == c ? 1.f : 0.f;
    }
}
extern "C" int mcn_one_hot_seg(const float* labels, float* onehot, int64_t P, int32_t C, void* stream) {
    if (!labels || !onehot || P < 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "one_hot_seg: bad argument");
    if (P == 0) return MCN_OK;
    hipLaunchKernelGGL(one_hot_seg_kernel, dim3(seg_blocks(((long)P * C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, labels, onehot, (long)P, C,
                       (((uintptr_t)onehot) & 15) == 0 ? 1 : 0);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
