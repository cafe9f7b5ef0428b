// dwconv.hip — depthwise convolution (channel multiplier 1) and the squeeze-excite channel scale: the HBM-bound
// kernels of the EfficientNet MBConv unit (reference models/efficientnet.py:126-197, convnet.py:1634-1650).
//
// Layout NHWC.  A thread owns one channel chunk (16 bytes forward / dgrad, 4 channels in wgrad) of one pixel; the
// TX threads of a row cover TX consecutive chunks, so every tap read / write is a contiguous run of TX*16 bytes.
// The K*K taps of one output re-read their neighbours' inputs from L1/L2 (every input byte leaves HBM once).
// Filters are the fp32 masters [KH][KW][C]; in bf16 mode they are rounded per use (convnet.py:1421) on the fly.
// (the depthwise kernels re-read rows of x across the filter taps / output rows through L2: plain loads — with streaming loads the
// EfficientNet-B0 step measured 30.1 -> 30.7 ms)
#define MCN_NT_HINT 0
#include "common.h"
#include <string.h>
#include <type_traits>

struct DwParams {
    int N, H, W, C, OH, OW;
    int KH, KW, SH, SW, DH, DW, padT, padL;
    int TX, TY;          // threads per row of chunks / rows per block
    long npix;           // pixels of the tensor the thread grid walks (output: fwd, wgrad; input: dgrad)
    int accumulate;
};

template <typename T>
__device__ __forceinline__ float round_w(float w) {
    return to_f32(from_f32<T>(w));       // the weight as the conv kernels see it (rounded to the storage type)
}

// largest divisor of n that is <= cap (>= 1)
static int best_tx(int n, int cap) {
    int best = 1;
    for (int d = 1; d <= cap && d <= n; ++d)
        if (n % d == 0) best = d;
    if (best < 8 && n > cap) best = cap < 32 ? cap : 32;      // awkward (prime-ish) chunk counts: accept idle lanes
    return best;
}

// ---- forward -------------------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, const DwParams p) {
    constexpr int CE = VecTraits<T>::CE;
    extern __shared__ float sw[];                                // [taps][TX*CE]
    const int KH = K ? K : p.KH, KW = K ? K : p.KW;
    constexpr int UNR = K ? K : 1;                   // K = 0: run-time filter size, nothing to unroll
    const int taps = KH * KW;
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int cch = p.C / CE;
    const int chunk0 = blockIdx.x * p.TX;
    const int wcols = p.TX * CE;
    for (int i = threadIdx.x; i < taps * wcols; i += 256) {
        const int t = i / wcols, c = chunk0 * CE + (i - t * wcols);
        sw[i] = c < p.C ? round_w<T>(w[(long)t * p.C + c]) : 0.f;
    }
    __syncthreads();
    const int chunk = chunk0 + tx;
    if (ty >= p.TY || chunk >= cch) return;
    const float* wl = sw + tx * CE;
    for (unsigned q = blockIdx.y * (unsigned)p.TY + ty; q < (unsigned)p.npix; q += gridDim.y * (unsigned)p.TY) {   // (32-bit: 64-bit divisions cost more than the taps)
        const unsigned t1 = q / (unsigned)p.OW;
        const int ox = (int)(q - t1 * (unsigned)p.OW);
        const long n = t1 / (unsigned)p.OH;
        const int oy = (int)(t1 - (unsigned)n * (unsigned)p.OH);
        float acc[CE];
#pragma unroll
        for (int i = 0; i < CE; ++i) acc[i] = 0.f;
        const int iy0 = oy * p.SH - p.padT, ix0 = ox * p.SW - p.padL;
#pragma unroll UNR
        for (int ky = 0; ky < KH; ++ky) {
            const int iy = iy0 + ky * p.DH;
            if ((unsigned)iy >= (unsigned)p.H) continue;
#pragma unroll UNR
            for (int kx = 0; kx < KW; ++kx) {
                const int ix = ix0 + kx * p.DW;
                if ((unsigned)ix >= (unsigned)p.W) continue;
                const Chunk<T> c = load_chunk<T>(x + ((n * p.H + iy) * p.W + ix) * p.C + (long)chunk * CE);
                const float* wt = wl + (ky * KW + kx) * wcols;
#pragma unroll
                for (int i = 0; i < CE; ++i) acc[i] = fmaf(c.get(i), wt[i], acc[i]);
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int i = 0; i < CE; ++i) o.set(i, acc[i]);
        store_chunk<T>(y + (long)q * p.C + (long)chunk * CE, o);
    }
}

// ---- forward, row-strip form ------------------------------------------------------------------------------------------
// A thread produces S consecutive outputs of one row for its chunk: each of the K input rows is loaded once as a segment
// of (S-1)*STRIDE + K chunks and reused by all K*S (tap, output) pairs — 2-3x fewer L1/L2 reads than a thread per output.
// FLIP: use the filter rotated by 180 degrees (the stride-1 dgrad is this kernel over dy with pads K-1-pad).
template <typename T, int K, int STRIDE, int S, bool FLIP>
__global__ __launch_bounds__(256) void dw_strip_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, const DwParams p) {
    constexpr int CE = VecTraits<T>::CE;
    constexpr int L = (S - 1) * STRIDE + K;
    extern __shared__ float sw[];                                // [taps][TX*CE]
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int cch = p.C / CE;
    const int chunk0 = blockIdx.x * p.TX;
    const int wcols = p.TX * CE;
    for (int i = threadIdx.x; i < K * K * wcols; i += 256) {
        const int t = i / wcols, c = chunk0 * CE + (i - t * wcols);
        const int ts = FLIP ? (K * K - 1 - t) : t;
        sw[i] = c < p.C ? round_w<T>(w[(long)ts * p.C + c]) : 0.f;
    }
    __syncthreads();
    const int chunk = chunk0 + tx;
    if (ty >= p.TY || chunk >= cch) return;
    const float* wl = sw + tx * CE;
    const int nsx = (p.OW + S - 1) / S;
    const long nstrips = (long)p.N * p.OH * nsx;
    for (unsigned q = blockIdx.y * (unsigned)p.TY + ty; q < (unsigned)nstrips; q += gridDim.y * (unsigned)p.TY) {
        const unsigned t1 = q / (unsigned)nsx;
        const int sx = (int)(q - t1 * (unsigned)nsx);
        const long n = t1 / (unsigned)p.OH;
        const int oy = (int)(t1 - (unsigned)n * (unsigned)p.OH);
        const int ox0 = sx * S;
        float acc[S][CE];
#pragma unroll
        for (int j = 0; j < S; ++j)
#pragma unroll
            for (int i = 0; i < CE; ++i) acc[j][i] = 0.f;
        const int ix0 = ox0 * STRIDE - p.padL;
        if constexpr (K == 3) {
            // 3x3: branch-free rows — every segment load goes to a clamped (valid) address and is zeroed afterwards when it lies in the
            // padding, so the loads of row ky+1 can be issued ahead of the FMAs of row ky (with a `continue` per row each row was
            // its own load -> wait -> compute round trip and the kernel ran at 1/3 of its VALU bound).
            Chunk<T> raw[2][L];
            auto load_row = [&](int ky, Chunk<T>* dst) {
                const int iy = oy * STRIDE - p.padT + ky;
                const bool yok = (unsigned)iy < (unsigned)p.H;
                const T* row = x + ((n * p.H + (yok ? iy : 0)) * p.W) * p.C + (long)chunk * CE;
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    const int ix = ix0 + j;
                    const bool ok = yok && (unsigned)ix < (unsigned)p.W;
                    Chunk<T> c = load_chunk<T>(row + (long)(ok ? ix : 0) * p.C);
                    if (!ok) c = Chunk<T>{};
                    dst[j] = c;
                }
            };
            load_row(0, raw[0]);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                if (ky + 1 < K) load_row(ky + 1, raw[(ky + 1) & 1]);
                float seg[L][CE];
#pragma unroll
                for (int j = 0; j < L; ++j)
#pragma unroll
                    for (int i = 0; i < CE; ++i) seg[j][i] = raw[ky & 1][j].get(i);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* wt = wl + (ky * K + kx) * wcols;
                    float wv[CE];
#pragma unroll
                    for (int i = 0; i < CE; ++i) wv[i] = wt[i];
#pragma unroll
                    for (int j = 0; j < S; ++j)
#pragma unroll
                        for (int i = 0; i < CE; ++i) acc[j][i] = fmaf(seg[j * STRIDE + kx][i], wv[i], acc[j][i]);
                }
            }
        } else {
            // (5x5: the same restructuring needs 160+ VGPRs of raw chunks in flight and spills; rows stay one round trip each)
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = oy * STRIDE - p.padT + ky;
                if ((unsigned)iy >= (unsigned)p.H) continue;
                const T* row = x + ((n * p.H + iy) * p.W) * p.C + (long)chunk * CE;
                float seg[L][CE];
#pragma unroll
                for (int j = 0; j < L; ++j) {                     // clamped address + select: the L loads of the row issue back to back
                    const int ix = ix0 + j;
                    const bool ok = (unsigned)ix < (unsigned)p.W;
                    Chunk<T> c = load_chunk<T>(row + (long)(ok ? ix : 0) * p.C);
                    if (!ok) c = Chunk<T>{};
#pragma unroll
                    for (int i = 0; i < CE; ++i) seg[j][i] = c.get(i);
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* wt = wl + (ky * K + kx) * wcols;
                    float wv[CE];
#pragma unroll
                    for (int i = 0; i < CE; ++i) wv[i] = wt[i];
#pragma unroll
                    for (int j = 0; j < S; ++j)
#pragma unroll
                        for (int i = 0; i < CE; ++i) acc[j][i] = fmaf(seg[j * STRIDE + kx][i], wv[i], acc[j][i]);
                }
            }
        }
        T* orow = y + ((n * p.OH + oy) * p.OW + ox0) * p.C + (long)chunk * CE;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (ox0 + j >= p.OW) break;
            Chunk<T> o;
            if (p.accumulate) {
                const Chunk<T> old = load_chunk<T>(orow + (long)j * p.C);
#pragma unroll
                for (int i = 0; i < CE; ++i) o.set(i, acc[j][i] + old.get(i));
            } else {
#pragma unroll
                for (int i = 0; i < CE; ++i) o.set(i, acc[j][i]);
            }
            store_chunk<T>(orow + (long)j * p.C, o);
        }
    }
}

// ---- forward / stride-1 dgrad, LDS-band form ---------------------------------------------------------------------------
// The strip kernel above pays one L1/L2 round trip per filter row of every strip (5x5: five dependent round trips, 1.2-1.4 TB/s
// on the 14x14 / 7x7 layers of EfficientNet-B0 against an HBM floor four times higher).  Here a workgroup owns a tile of TH
// output rows x TNS strips x TC channel chunks of one image and first stages the input band the tile reads — BH x BW pixels
// of TC chunks, halo included — in LDS with LDS-DMA (`buffer_load ... lds`: all passes issued back to back, no staging
// registers, out-of-range offsets arrive as zeros = the padding, so the tap loops carry no bounds tests), then every thread
// computes strips of S outputs from LDS: the K x L segment reads of a strip are independent ds_read_b128s.  The FMAs run
// on packed fp32 pairs (v_pk_fma_f32).  LDS image: slot ((row * BW + col) * TCP + chunk) * 16 bytes, TCP = TC rounded up to
// odd (the lanes of different strips of a wave then start in different banks); DMA lanes of the pad slots fetch zeros.
struct DwBandParams {
    int H, W, C, OH, OW;           // input and output grid of this pass (dgrad: the input is dy)
    int padT, padL;
    int TC, TCP;                   // channel chunks per workgroup, LDS pitch of a pixel in chunks
    int TH, TNS, NS;               // tile: output rows x strips; strips per output row
    int BH, BW;                    // band: input rows x columns
    int tiles_y, tiles_x;          // tiles per image
    int npass;                     // DMA passes of 256 slots
    int accumulate;
    int img_bytes;                 // bytes of one input image (buffer descriptor range)
};
#define DW_BAND_MAXPASS 16
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
template <int N, typename F>
__device__ __forceinline__ void dw_static_for(F&& f) {
    if constexpr (N > 0) {
        dw_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
template <typename T, int K, int STRIDE, int S, bool FLIP>
__global__ __launch_bounds__(256) void dw_band_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, const DwBandParams p) {
    constexpr int CE = VecTraits<T>::CE, CP = CE / 2;
    constexpr int L = (S - 1) * STRIDE + K;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // [npass * 4096 bytes of band][K*K*TC*CE floats of filter]
    const int tid = threadIdx.x, wave = tid >> 6;
    const int cch = p.C / CE;
    const int chunk0 = blockIdx.x * p.TC;
    const int tpi = p.tiles_y * p.tiles_x;
    const int n = blockIdx.y / tpi, tr = blockIdx.y - n * tpi;
    const int tyi = tr / p.tiles_x, txi = tr - tyi * p.tiles_x;
    const int oy0 = tyi * p.TH, sx0 = txi * p.TNS;
    const int iy0 = oy0 * STRIDE - p.padT, ix0 = sx0 * S * STRIDE - p.padL;

    // ---- stage the band: slot e = pass * 256 + tid <-> (row, col, chunk) ----
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (size_t)n * p.H * p.W * p.C), 0, p.img_bytes, 0x00020000);
        __attribute__((address_space(3))) char* const wbase = (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024);
        int tcx = tid % p.TCP, q = tid / p.TCP;
        int col = q % p.BW, row = q / p.BW;
        const int d_t = 256 % p.TCP, d_q = 256 / p.TCP, d_c = d_q % p.BW, d_r = d_q / p.BW;
        const int pixb = p.C * (int)sizeof(T);
        dw_static_for<DW_BAND_MAXPASS>([&](auto pc) {
            constexpr int ps = decltype(pc)::value;
            if (ps < p.npass) {                                             // uniform
                const int iy = iy0 + row, ix = ix0 + col;
                const unsigned bad = (unsigned)iy | (unsigned)(p.H - 1 - iy) | (unsigned)ix | (unsigned)(p.W - 1 - ix) | (unsigned)(p.TC - 1 - tcx) |
                                     (unsigned)(cch - 1 - chunk0 - tcx) | (unsigned)(p.BH - 1 - row);      // sign bits: padding, pad slot, past C, past the band
                const unsigned off = ((unsigned)((iy * p.W + ix) * pixb + (chunk0 + tcx) * 16)) | ((bad >> 31) << 31);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wbase + ps * 4096), 16, (int)off, 0, 0, 0);
                tcx += d_t;
                int cq = d_c;
                if (tcx >= p.TCP) { tcx -= p.TCP; cq++; }
                col += cq;
                int cr = d_r;
                if (col >= p.BW) { col -= p.BW; cr++; }
                row += cr;
                __builtin_amdgcn_sched_barrier(0);            // (one pass at a time: hoisting the index chains of all passes costs 100+ VGPRs)
            }
        });
    }
    // ---- the filter of the block's channels, rounded as the storage type sees it ----
    float* const sw = reinterpret_cast<float*>(smem + p.npass * 4096);
    const int wcols = p.TC * CE;
    for (int i = tid; i < K * K * wcols; i += 256) {
        const int t = i / wcols, c = chunk0 * CE + (i - t * wcols);
        const int ts = FLIP ? (K * K - 1 - t) : t;
        sw[i] = c < p.C ? round_w<T>(w[(long)ts * p.C + c]) : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int tx = tid % p.TC, sid0 = tid / p.TC, nthr = 256 / p.TC;
    const int chunk = chunk0 + tx;
    if (sid0 >= nthr || chunk >= cch) return;
    const float* wl = sw + tx * CE;
    const int nst = p.TH * p.TNS;
    for (int st = sid0; st < nst; st += nthr) {
        const int r = st / p.TNS, sxl = st - r * p.TNS;
        const int oy = oy0 + r, ox0 = (sx0 + sxl) * S;
        if (oy >= p.OH || ox0 >= p.OW) continue;
        dw_f32x2 acc[S][CP];
#pragma unroll
        for (int j = 0; j < S; ++j)
#pragma unroll
            for (int i = 0; i < CP; ++i) acc[j][i] = dw_f32x2{0.f, 0.f};
        const char* band = smem + ((size_t)((r * STRIDE) * p.BW + sxl * S * STRIDE) * p.TCP + tx) * 16;
        // (one filter row per trip: fully unrolled, the scheduler hoists all K x L segment reads and their fp32 images in front of
        // the first FMA — 256 VGPRs + 246 AGPRs of spill space for 5x5, one wave per SIMD; the other waves of the CU cover the LDS latency)
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            dw_f32x2 seg[L][CP];
#pragma unroll
            for (int j = 0; j < L; ++j) {
                Chunk<T> c;
                *reinterpret_cast<i32x4*>(&c.v) = *reinterpret_cast<const i32x4*>(band + (size_t)((ky * p.BW + j) * p.TCP) * 16);
#pragma unroll
                for (int i = 0; i < CP; ++i) seg[j][i] = dw_f32x2{c.get(2 * i), c.get(2 * i + 1)};
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float* wt = wl + (ky * K + kx) * wcols;
                dw_f32x2 wv[CP];
#pragma unroll
                for (int i = 0; i < CP; ++i) wv[i] = dw_f32x2{wt[2 * i], wt[2 * i + 1]};
#pragma unroll
                for (int j = 0; j < S; ++j)
#pragma unroll
                    for (int i = 0; i < CP; ++i) acc[j][i] = __builtin_elementwise_fma(seg[j * STRIDE + kx][i], wv[i], acc[j][i]);
            }
        }
        T* orow = y + (((size_t)n * p.OH + oy) * p.OW + ox0) * p.C + (size_t)chunk * CE;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (ox0 + j >= p.OW) break;
            Chunk<T> o;
            if (p.accumulate) {
                const Chunk<T> old = load_chunk<T>(orow + (size_t)j * p.C);
#pragma unroll
                for (int i = 0; i < CP; ++i) { o.set(2 * i, acc[j][i][0] + old.get(2 * i)); o.set(2 * i + 1, acc[j][i][1] + old.get(2 * i + 1)); }
            } else {
#pragma unroll
                for (int i = 0; i < CP; ++i) { o.set(2 * i, acc[j][i][0]); o.set(2 * i + 1, acc[j][i][1]); }
            }
            store_chunk<T>(orow + (size_t)j * p.C, o);
        }
    }
}

// ---- dgrad: dx[n,iy,ix,c] = sum_{ky,kx} dy[n,(iy+pt-ky*d)/s,(ix+pl-kx*d)/s,c] * w[ky,kx,c] (where divisible, in range) ----
template <typename T, int K>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, const DwParams p) {
    constexpr int CE = VecTraits<T>::CE;
    extern __shared__ float sw[];
    const int KH = K ? K : p.KH, KW = K ? K : p.KW;
    constexpr int UNR = K ? K : 1;                   // K = 0: run-time filter size, nothing to unroll
    const int taps = KH * KW;
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int cch = p.C / CE;
    const int chunk0 = blockIdx.x * p.TX;
    const int wcols = p.TX * CE;
    for (int i = threadIdx.x; i < taps * wcols; i += 256) {
        const int t = i / wcols, c = chunk0 * CE + (i - t * wcols);
        sw[i] = c < p.C ? round_w<T>(w[(long)t * p.C + c]) : 0.f;
    }
    __syncthreads();
    const int chunk = chunk0 + tx;
    if (ty >= p.TY || chunk >= cch) return;
    const float* wl = sw + tx * CE;
    for (unsigned q = blockIdx.y * (unsigned)p.TY + ty; q < (unsigned)p.npix; q += gridDim.y * (unsigned)p.TY) {   // (32-bit: 64-bit divisions cost more than the taps)
        const unsigned t1 = q / (unsigned)p.W;
        const int ix = (int)(q - t1 * (unsigned)p.W);
        const long n = t1 / (unsigned)p.H;
        const int iy = (int)(t1 - (unsigned)n * (unsigned)p.H);
        float acc[CE];
#pragma unroll
        for (int i = 0; i < CE; ++i) acc[i] = 0.f;
#pragma unroll UNR
        for (int ky = 0; ky < KH; ++ky) {
            const int ny = iy + p.padT - ky * p.DH;
            if (ny < 0 || ny % p.SH) continue;
            const int oy = ny / p.SH;
            if (oy >= p.OH) continue;
#pragma unroll UNR
            for (int kx = 0; kx < KW; ++kx) {
                const int nx = ix + p.padL - kx * p.DW;
                if (nx < 0 || nx % p.SW) continue;
                const int ox = nx / p.SW;
                if (ox >= p.OW) continue;
                const Chunk<T> c = load_chunk<T>(dy + ((n * p.OH + oy) * p.OW + ox) * p.C + (long)chunk * CE);
                const float* wt = wl + (ky * KW + kx) * wcols;
#pragma unroll
                for (int i = 0; i < CE; ++i) acc[i] = fmaf(c.get(i), wt[i], acc[i]);
            }
        }
        T* dst = dx + (long)q * p.C + (long)chunk * CE;
        Chunk<T> o;
        if (p.accumulate) {
            const Chunk<T> old = load_chunk<T>(dst);
#pragma unroll
            for (int i = 0; i < CE; ++i) acc[i] += old.get(i);
        }
#pragma unroll
        for (int i = 0; i < CE; ++i) o.set(i, acc[i]);
        store_chunk<T>(dst, o);
    }
}

// ---- wgrad: dw[ky,kx,c] = sum_{n,oy,ox} x[n,oy*s+ky*d-pt,ox*s+kx*d-pl,c] * dy[n,oy,ox,c] ---------------------------
// A thread owns 4 channels and all K*K taps (registers) over its share of the output pixels; the block folds its TY
// pixel rows through LDS and writes one partial [taps][C] slab; a second kernel sums the slabs in a fixed order.
// ---- dgrad, stride 2, row-strip form -----------------------------------------------------------------------------------
// dx[iy][ix] takes the taps (ky, kx) with (iy + padT - ky) and (ix + padL - kx) even.  A thread produces S consecutive dx
// pixels of one row for its chunk: for every filter row of the right parity it loads ONE dy row segment (S/2 + (K-1)/2 + 1
// chunks) and every (pixel, kx) pair of the right parity — known at compile time, PL = padL — reads it from registers.  The
// per-pixel kernel above tests K*K taps and issues a load per hit for every pixel (2-3x the time of the forward pass).
template <typename T, int K, int PL, int S>
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, const DwParams p) {
    constexpr int CE = VecTraits<T>::CE;
    constexpr int OFF = (K - 1) / 2;
    constexpr int LS = OFF + (S - 1 + PL) / 2 + 1;
    extern __shared__ float sw[];                                // [taps][TX*CE]
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int cch = p.C / CE;
    const int chunk0 = blockIdx.x * p.TX;
    const int wcols = p.TX * CE;
    for (int i = threadIdx.x; i < K * K * wcols; i += 256) {
        const int t = i / wcols, c = chunk0 * CE + (i - t * wcols);
        sw[i] = c < p.C ? round_w<T>(w[(long)t * p.C + c]) : 0.f;
    }
    __syncthreads();
    const int chunk = chunk0 + tx;
    if (ty >= p.TY || chunk >= cch) return;
    const float* wl = sw + tx * CE;
    const int nsx = (p.W + S - 1) / S;
    const unsigned nstrips = (unsigned)((long)p.N * p.H * nsx);
    for (unsigned q = blockIdx.y * (unsigned)p.TY + ty; q < nstrips; q += gridDim.y * (unsigned)p.TY) {
        const unsigned t1 = q / (unsigned)nsx;
        const int sx = (int)(q - t1 * (unsigned)nsx);
        const long n = t1 / (unsigned)p.H;
        const int iy = (int)(t1 - (unsigned)n * (unsigned)p.H);
        const int ix0 = sx * S;                                  // multiple of S (even)
        float acc[S][CE];
#pragma unroll
        for (int j = 0; j < S; ++j)
#pragma unroll
            for (int i = 0; i < CE; ++i) acc[j][i] = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ny = iy + p.padT - ky;
            if (ny < 0 || (ny & 1)) continue;
            const int oy = ny >> 1;
            if (oy >= p.OH) continue;
            const T* drow = dy + ((n * p.OH + oy) * p.OW) * p.C + (long)chunk * CE;
            float seg[LS][CE];
#pragma unroll
            for (int i2 = 0; i2 < LS; ++i2) {
                const int ox = (ix0 >> 1) + i2 - OFF;
                const bool ok = (unsigned)ox < (unsigned)p.OW;
                Chunk<T> c = load_chunk<T>(drow + (long)(ok ? ox : 0) * p.C);
                if (!ok) c = Chunk<T>{};
#pragma unroll
                for (int i = 0; i < CE; ++i) seg[i2][i] = c.get(i);
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float* wt = wl + (ky * K + kx) * wcols;
                float wv[CE];
#pragma unroll
                for (int i = 0; i < CE; ++i) wv[i] = wt[i];
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    if (((j + PL - kx) & 1) == 0) {               // compile time
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int r = (j + PL - kx + 2 * OFF) / 2;   // = (j + PL - kx) / 2 + OFF, numerator even and >= 0
#pragma unroll
                        for (int i = 0; i < CE; ++i) acc[j][i] = fmaf(seg[r][i], wv[i], acc[j][i]);
                    }
                }
            }
        }
        T* orow = dx + ((n * p.H + iy) * p.W + ix0) * p.C + (long)chunk * CE;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (ix0 + j >= p.W) break;
            Chunk<T> o;
            if (p.accumulate) {
                const Chunk<T> old = load_chunk<T>(orow + (long)j * p.C);
#pragma unroll
                for (int i = 0; i < CE; ++i) o.set(i, acc[j][i] + old.get(i));
            } else {
#pragma unroll
                for (int i = 0; i < CE; ++i) o.set(i, acc[j][i]);
            }
            store_chunk<T>(orow + (long)j * p.C, o);
        }
    }
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, float* v);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float* v) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* v) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <>
__device__ __forceinline__ void load4<f16_t>(const f16_t* p, float* v) {
    const f16x4 t = *reinterpret_cast<const f16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

template <typename T, int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, const DwParams p) {
    extern __shared__ float red[];                               // [TY][TX*4]
    constexpr int TAPS = K * K;
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int c4 = p.C / 4;
    const int grp = blockIdx.x * p.TX + tx;
    const bool active = ty < p.TY && grp < c4;
    float acc[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    if (active) {
        for (unsigned q = blockIdx.y * (unsigned)p.TY + ty; q < (unsigned)p.npix; q += gridDim.y * (unsigned)p.TY) {   // (32-bit: 64-bit divisions cost more than the taps)
            const unsigned t1 = q / (unsigned)p.OW;
            const int ox = (int)(q - t1 * (unsigned)p.OW);
            const long n = t1 / (unsigned)p.OH;
            const int oy = (int)(t1 - (unsigned)n * (unsigned)p.OH);
            float g[4];
            load4<T>(dy + (long)q * p.C + (long)grp * 4, g);
            const int iy0 = oy * p.SH - p.padT, ix0 = ox * p.SW - p.padL;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = iy0 + ky * p.DH;
                if ((unsigned)iy >= (unsigned)p.H) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int ix = ix0 + kx * p.DW;
                    if ((unsigned)ix >= (unsigned)p.W) continue;
                    float v[4];
                    load4<T>(x + ((n * p.H + iy) * p.W + ix) * p.C + (long)grp * 4, v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[ky * K + kx][i] = fmaf(v[i], g[i], acc[ky * K + kx][i]);
                }
            }
        }
    }
    const int cols = p.TX * 4;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (ty < p.TY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[ty * cols + tx * 4 + i] = acc[t][i];
        }
        __syncthreads();
        if (ty == 0 && grp < c4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
                for (int k = 0; k < p.TY; ++k) s += red[k * cols + tx * 4 + i];
                part[((long)blockIdx.y * TAPS + t) * p.C + (long)grp * 4 + i] = s;
            }
        }
        __syncthreads();
    }
}

// row-strip form of the above: S outputs of one row per iteration, each input row segment loaded once for all K*S pairs
template <typename T, int K, int STRIDE, int S>
__global__ __launch_bounds__(256) void dw_wgrad_strip_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, const DwParams p) {
    extern __shared__ float red[];                               // [TY][TX*4]
    constexpr int TAPS = K * K;
    constexpr int L = (S - 1) * STRIDE + K;
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int c4 = p.C / 4;
    const int grp = blockIdx.x * p.TX + tx;
    const bool active = ty < p.TY && grp < c4;
    float acc[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    if (active) {
        const int nsx = (p.OW + S - 1) / S;
        const long nstrips = (long)p.N * p.OH * nsx;
        for (unsigned q = blockIdx.y * (unsigned)p.TY + ty; q < (unsigned)nstrips; q += gridDim.y * (unsigned)p.TY) {
            const unsigned t1 = q / (unsigned)nsx;
            const int sx = (int)(q - t1 * (unsigned)nsx);
            const long n = t1 / (unsigned)p.OH;
            const int oy = (int)(t1 - (unsigned)n * (unsigned)p.OH);
            const int ox0 = sx * S;
            float g[S][4];
            const T* drow = dy + ((n * p.OH + oy) * p.OW + ox0) * p.C + (long)grp * 4;
#pragma unroll
            for (int j = 0; j < S; ++j) {                          // clamped address + select (no branch per load)
                const bool ok = ox0 + j < p.OW;
                load4<T>(drow + (long)(ok ? j : 0) * p.C, g[j]);
                if (!ok) g[j][0] = g[j][1] = g[j][2] = g[j][3] = 0.f;
            }
            const int ix0 = ox0 * STRIDE - p.padL;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = oy * STRIDE - p.padT + ky;
                if ((unsigned)iy >= (unsigned)p.H) continue;
                const T* row = x + ((n * p.H + iy) * p.W) * p.C + (long)grp * 4;
                float seg[L][4];
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    const int ix = ix0 + j;
                    const bool ok = (unsigned)ix < (unsigned)p.W;
                    load4<T>(row + (long)(ok ? ix : 0) * p.C, seg[j]);
                    if (!ok) seg[j][0] = seg[j][1] = seg[j][2] = seg[j][3] = 0.f;
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < S; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[ky * K + kx][i] = fmaf(seg[j * STRIDE + kx][i], g[j][i], acc[ky * K + kx][i]);
            }
        }
    }
    const int cols = p.TX * 4;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (ty < p.TY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[ty * cols + tx * 4 + i] = acc[t][i];
        }
        __syncthreads();
        if (ty == 0 && grp < c4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
                for (int k = 0; k < p.TY; ++k) s += red[k * cols + tx * 4 + i];
                part[((long)blockIdx.y * TAPS + t) * p.C + (long)grp * 4 + i] = s;
            }
        }
        __syncthreads();
    }
}

// row-split form: a thread owns one 16-byte chunk of channels and ONE filter row ky; per strip of S outputs it loads the S dy
// chunks and the one input row segment it needs (16-byte loads, S + L per K*S*CE FMAs) and keeps K*CE accumulators, so the
// 5x5 case needs 40 registers of accumulators instead of 100 and half as many, twice as wide loads per FMA as the 4-channel
// strip kernel above.  The K threads that share a strip sit next to each other in the block (their dy loads hit L1).
// Measured (B = 512, EfficientNet-B0 layers): bf16 5x5 -18...-49 % per layer; 3x3 (24 accumulators: too little work per
// load) and fp32 (the strip kernel's loads are already 16 bytes) are slower with it and keep the strip kernel.
template <typename T, int K, int STRIDE, int S>
__global__ __launch_bounds__(256) void dw_wgrad_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, const DwParams p) {
    constexpr int CE = VecTraits<T>::CE;
    constexpr int TAPS = K * K;
    constexpr int L = (S - 1) * STRIDE + K;
    extern __shared__ float red[];                               // [TY][TX*CE]
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int cch = p.C / CE;
    const int chunk = blockIdx.x * p.TX + tx;
    const int tys = p.TY / K;                                    // strips walked side by side in a block
    const int ky = ty % K, sl = ty / K;
    const bool active = sl < tys && chunk < cch;
    float acc[K][CE];
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int i = 0; i < CE; ++i) acc[kx][i] = 0.f;
    if (active) {
        const int nsx = (p.OW + S - 1) / S;
        const unsigned nstrips = (unsigned)((long)p.N * p.OH * nsx);
        for (unsigned q = blockIdx.y * (unsigned)tys + sl; q < nstrips; q += gridDim.y * (unsigned)tys) {
            const unsigned t1 = q / (unsigned)nsx;
            const int sx = (int)(q - t1 * (unsigned)nsx);
            const long n = t1 / (unsigned)p.OH;
            const int oy = (int)(t1 - (unsigned)n * (unsigned)p.OH);
            const int iy = oy * STRIDE - p.padT + ky;
            if ((unsigned)iy >= (unsigned)p.H) continue;
            const int ox0 = sx * S, ix0 = ox0 * STRIDE - p.padL;
            const T* drow = dy + ((n * p.OH + oy) * p.OW + ox0) * p.C + (long)chunk * CE;
            const T* row = x + ((n * p.H + iy) * p.W) * p.C + (long)chunk * CE;
            Chunk<T> gr[S], sr[L];
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const bool ok = ox0 + j < p.OW;
                gr[j] = load_chunk<T>(drow + (long)(ok ? j : 0) * p.C);
                if (!ok) gr[j] = Chunk<T>{};
            }
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const int ix = ix0 + j;
                const bool ok = (unsigned)ix < (unsigned)p.W;
                sr[j] = load_chunk<T>(row + (long)(ok ? ix : 0) * p.C);
                if (!ok) sr[j] = Chunk<T>{};
            }
#pragma unroll
            for (int j = 0; j < S; ++j) {
                float g[CE];
#pragma unroll
                for (int i = 0; i < CE; ++i) g[i] = gr[j].get(i);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int i = 0; i < CE; ++i) acc[kx][i] = fmaf(sr[j * STRIDE + kx].get(i), g[i], acc[kx][i]);
            }
        }
    }
    const int cols = p.TX * CE;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
#pragma unroll
        for (int i = 0; i < CE; ++i) red[ty * cols + tx * CE + i] = active ? acc[kx][i] : 0.f;
        __syncthreads();
        if (ty < K && chunk < cch) {                              // ty = filter row; fold the block's strip lanes
#pragma unroll
            for (int i = 0; i < CE; ++i) {
                float sum = 0.f;
                for (int l = 0; l < tys; ++l) sum += red[(l * K + ty) * cols + tx * CE + i];
                part[((long)blockIdx.y * TAPS + ty * K + kx) * p.C + (long)chunk * CE + i] = sum;
            }
        }
        __syncthreads();
    }
}

// ---- wgrad, LDS-band form ------------------------------------------------------------------------------------------------
// The band kernel's staging (dw_band_kernel) for the weight gradient: a workgroup walks tiles t = blockIdx.y, + gridDim.y, ...
// of the (image, row block, strip block) grid; per tile it stages the x band AND the dy tile in LDS by LDS-DMA (out-of-range
// dy pixels arrive as zeros and add nothing, so the loops carry no bounds tests), then thread (chunk, filter row ky, strip
// lane) adds S dy pixels x K taps of row ky into its K x CE accumulators, which live in registers across all tiles.  One
// partial slab [K*K][C] per blockIdx.y at the end (folded over the strip lanes through LDS), summed by dw_wgrad_reduce_kernel.
struct DwWBandParams {
    DwBandParams b;                // x band geometry (H, W = x; OH, OW = dy)
    int DW_;                       // dy tile width in pixels (TNS * S)
    int npass_y;                   // DMA passes of the dy tile
    int ntiles;                    // N * tiles_y * tiles_x
    int dy_img_bytes;
};
template <typename T, int K, int STRIDE, int S>
__global__ __launch_bounds__(256) void dw_wgrad_band_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, const DwWBandParams pw) {
    constexpr int CE = VecTraits<T>::CE, CP = CE / 2;
    constexpr int TAPS = K * K;
    constexpr int L = (S - 1) * STRIDE + K;
    const DwBandParams& p = pw.b;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // [x band: npass * 4096][dy tile: npass_y * 4096]; reused by the final fold
    const int tid = threadIdx.x, wave = tid >> 6;
    const int cch = p.C / CE;
    const int chunk0 = blockIdx.x * p.TC;
    const int tpi = p.tiles_y * p.tiles_x;
    const int tx = tid % p.TC, rest = tid / p.TC;
    const int ky = rest % K, sid0 = rest / K, nthr = 256 / (p.TC * K);
    const int chunk = chunk0 + tx;
    const bool active = sid0 < nthr && chunk < cch;
    dw_f32x2 acc[K][CP];
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int i = 0; i < CP; ++i) acc[kx][i] = dw_f32x2{0.f, 0.f};
    __attribute__((address_space(3))) char* const wbase = (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024);
    const int pixb = p.C * (int)sizeof(T);
    const char* const dyl = smem + p.npass * 4096;
    for (int t = blockIdx.y; t < pw.ntiles; t += gridDim.y) {
        const int n = t / tpi, tr = t - n * tpi;
        const int tyi = tr / p.tiles_x, txi = tr - tyi * p.tiles_x;
        const int oy0 = tyi * p.TH, ox00 = txi * p.TNS * S;
        const int iy0 = oy0 * STRIDE - p.padT, ix0 = ox00 * STRIDE - p.padL;
        __syncthreads();                                                    // the previous tile's readers are done
        int tidv = tid;                                                     // opaque per trip: otherwise the index chains of all DMA passes are
        asm volatile("" : "+v"(tidv));                                      // hoisted out of the tile loop as loop invariants (190+ VGPRs)
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (size_t)n * p.H * p.W * p.C), 0, p.img_bytes, 0x00020000);
            int tcx = tidv % p.TCP, q = tidv / p.TCP;
            int col = q % p.BW, row = q / p.BW;
            const int d_t = 256 % p.TCP, d_q = 256 / p.TCP, d_c = d_q % p.BW, d_r = d_q / p.BW;
            dw_static_for<DW_BAND_MAXPASS>([&](auto pc) {
                constexpr int ps = decltype(pc)::value;
                if (ps < p.npass) {
                    const int iy = iy0 + row, ix = ix0 + col;
                    const unsigned bad = (unsigned)iy | (unsigned)(p.H - 1 - iy) | (unsigned)ix | (unsigned)(p.W - 1 - ix) | (unsigned)(p.TC - 1 - tcx) |
                                         (unsigned)(cch - 1 - chunk0 - tcx) | (unsigned)(p.BH - 1 - row);
                    const unsigned off = ((unsigned)((iy * p.W + ix) * pixb + (chunk0 + tcx) * 16)) | ((bad >> 31) << 31);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wbase + ps * 4096), 16, (int)off, 0, 0, 0);
                    tcx += d_t;
                    int cq = d_c;
                    if (tcx >= p.TCP) { tcx -= p.TCP; cq++; }
                    col += cq;
                    int cr = d_r;
                    if (col >= p.BW) { col -= p.BW; cr++; }
                    row += cr;
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy + (size_t)n * p.OH * p.OW * p.C), 0, pw.dy_img_bytes, 0x00020000);
            __attribute__((address_space(3))) char* const ybase = wbase + p.npass * 4096;
            int tcx = tidv % p.TCP, q = tidv / p.TCP;
            int col = q % pw.DW_, row = q / pw.DW_;
            const int d_t = 256 % p.TCP, d_q = 256 / p.TCP, d_c = d_q % pw.DW_, d_r = d_q / pw.DW_;
            dw_static_for<DW_BAND_MAXPASS>([&](auto pc) {
                constexpr int ps = decltype(pc)::value;
                if (ps < pw.npass_y) {
                    const int oy = oy0 + row, ox = ox00 + col;
                    const unsigned bad = (unsigned)(p.OH - 1 - oy) | (unsigned)(p.OW - 1 - ox) | (unsigned)(p.TC - 1 - tcx) | (unsigned)(cch - 1 - chunk0 - tcx) |
                                         (unsigned)(p.TH - 1 - row);
                    const unsigned off = ((unsigned)((oy * p.OW + ox) * pixb + (chunk0 + tcx) * 16)) | ((bad >> 31) << 31);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ybase + ps * 4096), 16, (int)off, 0, 0, 0);
                    tcx += d_t;
                    int cq = d_c;
                    if (tcx >= p.TCP) { tcx -= p.TCP; cq++; }
                    col += cq;
                    int cr = d_r;
                    if (col >= pw.DW_) { col -= pw.DW_; cr++; }
                    row += cr;
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (active) {
            const int nst = p.TH * p.TNS;
#pragma unroll 1
            for (int st = sid0; st < nst; st += nthr) {
                const int r = st / p.TNS, sxl = st - r * p.TNS;
                const char* xb = smem + ((size_t)((r * STRIDE + ky) * p.BW + sxl * S * STRIDE) * p.TCP + tx) * 16;
                const char* gb = dyl + ((size_t)(r * pw.DW_ + sxl * S) * p.TCP + tx) * 16;
                // all L + S segment reads go out first (raw 16-byte chunks); the dy pixels are widened once, every x chunk when its turn
                // comes: chunk jp feeds the (kx, j) pairs with j * STRIDE + kx == jp.  (Widening the whole segment up front as the forward
                // kernel does needs 250+ VGPRs here: K x CE accumulators and S dy pixels stay live.)
                Chunk<T> graw[S], raw[L];
#pragma unroll
                for (int j = 0; j < S; ++j) *reinterpret_cast<i32x4*>(&graw[j].v) = *reinterpret_cast<const i32x4*>(gb + (size_t)(j * p.TCP) * 16);
#pragma unroll
                for (int j = 0; j < L; ++j) *reinterpret_cast<i32x4*>(&raw[j].v) = *reinterpret_cast<const i32x4*>(xb + (size_t)(j * p.TCP) * 16);
                dw_f32x2 g[S][CP];
#pragma unroll
                for (int j = 0; j < S; ++j)
#pragma unroll
                    for (int i = 0; i < CP; ++i) g[j][i] = dw_f32x2{graw[j].get(2 * i), graw[j].get(2 * i + 1)};
#pragma unroll
                for (int jp = 0; jp < L; ++jp) {
                    dw_f32x2 v[CP];
#pragma unroll
                    for (int i = 0; i < CP; ++i) v[i] = dw_f32x2{raw[jp].get(2 * i), raw[jp].get(2 * i + 1)};
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int d = jp - kx;                                // (compile time after unrolling)
                        if (d >= 0 && d % STRIDE == 0 && d / STRIDE < S) {
#pragma unroll
                            for (int i = 0; i < CP; ++i) acc[kx][i] = __builtin_elementwise_fma(v[i], g[d / STRIDE][i], acc[kx][i]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // fold the strip lanes: red[(sid * K + ky)][tx * CE + i], one filter column kx per round
    __syncthreads();
    float* const red = reinterpret_cast<float*>(smem);
    const int cols = p.TC * CE;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        if (sid0 < nthr) {
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                red[(sid0 * K + ky) * cols + tx * CE + 2 * i] = active ? acc[kx][i][0] : 0.f;
                red[(sid0 * K + ky) * cols + tx * CE + 2 * i + 1] = active ? acc[kx][i][1] : 0.f;
            }
        }
        __syncthreads();
        if (sid0 == 0 && chunk < cch) {
#pragma unroll
            for (int i = 0; i < CE; ++i) {
                float sum = 0.f;
                for (int l = 0; l < nthr; ++l) sum += red[(l * K + ky) * cols + tx * CE + i];
                part[((long)blockIdx.y * TAPS + ky * K + kx) * p.C + (long)chunk * CE + i] = sum;
            }
        }
        __syncthreads();
    }
}

// generic filter sizes: one tap per blockIdx.z (K*K passes over the data; not on the EfficientNet path)
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_tap_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, const DwParams p) {
    extern __shared__ float red[];
    const int tap = blockIdx.z, ky = tap / p.KW, kx = tap - ky * p.KW;
    const int taps = p.KH * p.KW;
    const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
    const int c4 = p.C / 4;
    const int grp = blockIdx.x * p.TX + tx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ty < p.TY && grp < c4) {
        for (unsigned q = blockIdx.y * (unsigned)p.TY + ty; q < (unsigned)p.npix; q += gridDim.y * (unsigned)p.TY) {   // (32-bit: 64-bit divisions cost more than the taps)
            const unsigned t1 = q / (unsigned)p.OW;
            const int ox = (int)(q - t1 * (unsigned)p.OW);
            const long n = t1 / (unsigned)p.OH;
            const int oy = (int)(t1 - (unsigned)n * (unsigned)p.OH);
            const int iy = oy * p.SH - p.padT + ky * p.DH, ix = ox * p.SW - p.padL + kx * p.DW;
            if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
            float g[4], v[4];
            load4<T>(dy + (long)q * p.C + (long)grp * 4, g);
            load4<T>(x + ((n * p.H + iy) * p.W + ix) * p.C + (long)grp * 4, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(v[i], g[i], acc[i]);
        }
    }
    const int cols = p.TX * 4;
    if (ty < p.TY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[ty * cols + tx * 4 + i] = acc[i];
    }
    __syncthreads();
    if (ty == 0 && grp < c4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
            for (int k = 0; k < p.TY; ++k) s += red[k * cols + tx * 4 + i];
            part[((long)blockIdx.y * taps + tap) * p.C + (long)grp * 4 + i] = s;
        }
    }
}

// slab fold: E elements x 256/E slab lanes per block (a thread per element walking all slabs alone took 270 us on average for
// the 600-2000 slabs x 300-29000 elements of EfficientNet-B0: more than the wgrad kernel itself); fixed summation order, double.
template <int E>
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long n, int slabs, float scale) {
    constexpr int L = 256 / E;
    __shared__ double red[256];
    const int e = threadIdx.x % E, l = threadIdx.x / E;
    const long i = (long)blockIdx.x * E + e;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        int b = l;
        for (; b + 3 * L < slabs; b += 4 * L) {
            const float v0 = part[(long)b * n + i], v1 = part[(long)(b + L) * n + i], v2 = part[(long)(b + 2 * L) * n + i], v3 = part[(long)(b + 3 * L) * n + i];
            s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
        }
        for (; b < slabs; b += L) s0 += (double)part[(long)b * n + i];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (l == 0 && i < n) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < L; ++k) s += red[k * E + e];
        dw[i] = (float)s * scale;
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------
static int dw_check(const mcn_conv_geom* g, mcn_dtype dt, const char* what) {
    if (!g) MCN_FAIL(MCN_E_BADARG, "%s: null geometry", what);
    if (g->N < 0 || g->H <= 0 || g->W <= 0 || g->Cin <= 0 || g->KH <= 0 || g->KW <= 0 || g->SH <= 0 || g->SW <= 0 || g->DH <= 0 || g->DW <= 0)
        MCN_FAIL(MCN_E_BADARG, "%s: bad geometry", what);
    if (g->Cout != g->Cin) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: channel multiplier %d/%d != 1 is not built", what, g->Cout, g->Cin);
    if (!mcn_dtype_ok(dt)) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: dtype %d unsupported", what, (int)dt);
    if (g->Cin % (dt == MCN_F32 ? 4 : 8)) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: C=%d must be a multiple of the 16-byte chunk", what, g->Cin);
    if (g->x_cs && g->x_cs != g->Cin) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: strided input channels are not built", what);
    if (g->KH * g->KW > 121) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: filter too large", what);
    if ((long)g->N * g->H * g->W >= 0x40000000L) MCN_FAIL(MCN_E_UNSUPPORTED, "%s: more than 2^30 pixels (32-bit pixel indices)", what);
    return MCN_OK;
}
static inline int dw_out(int in, int k, int s, int d, int pa, int pb) { return (in + pa + pb - (k - 1) * d - 1) / s + 1; }

static DwParams dw_params(const mcn_conv_geom* g, int ce, bool walk_input, unsigned* gx, unsigned* gy) {
    DwParams p;
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->Cin;
    p.OH = dw_out(g->H, g->KH, g->SH, g->DH, g->padT, g->padB);
    p.OW = dw_out(g->W, g->KW, g->SW, g->DW, g->padL, g->padR);
    p.KH = g->KH; p.KW = g->KW; p.SH = g->SH; p.SW = g->SW; p.DH = g->DH; p.DW = g->DW; p.padT = g->padT; p.padL = g->padL;
    const int cch = p.C / ce;
    p.TX = best_tx(cch, 32);
    p.TY = 256 / p.TX;
    p.npix = walk_input ? (long)p.N * p.H * p.W : (long)p.N * p.OH * p.OW;
    p.accumulate = 0;
    *gx = (unsigned)((cch + p.TX - 1) / p.TX);
    long rows = (p.npix + p.TY - 1) / p.TY;
    long want = 4096 / *gx;
    if (want < 1) want = 1;
    *gy = (unsigned)(rows < want ? (rows < 1 ? 1 : rows) : want);
    return p;
}

// the row-strip kernels cover the EfficientNet cases: 3x3 / 5x5, stride 1 or 2 (both axes), no dilation
static bool dw_strip_ok(const DwParams& p) {
    return p.KH == p.KW && (p.KH == 3 || p.KH == 5) && p.SH == p.SW && (p.SH == 1 || p.SH == 2) && p.DH == 1 && p.DW == 1;
}

// Tile of the band kernel for one pass (in: H x W, out: OH x OW, square K, stride s): TC chunks x TH rows x TNS strips with
// the band within the LDS budget; scored by outputs per staged input pixel (halo overhead) x thread utilisation.
// MCN_DW_BAND=0 switches the band kernels off.
static bool dw_band_enabled() {
    static const int v = [] { const char* e = getenv("MCN_DW_BAND"); return e ? atoi(e) : 1; }();
    return v != 0;
}
#define DW_BAND_S 4
static long dw_band_budget() {
    static const long v = [] { const char* e = getenv("MCN_DW_BAND_KB"); return (long)(e ? atoi(e) : 40) * 1024; }();
    return v;
}
#define DW_BAND_LDS_BUDGET dw_band_budget()
// kthreads > 1 (wgrad): K threads share a strip (one filter row each) and the dy tile is staged too (its passes in *npass_y)
static bool dw_band_plan(int H, int W, int C, int OH, int OW, int K, int stride, int padT, int padL, int ce, size_t es, DwBandParams* out,
                         int kthreads = 1, int* npass_y = nullptr) {
    if (!dw_band_enabled() || (size_t)H * W * C * es >= 0x7fffffffull) return false;
    // stride 2 stages four input pixels per output: on the large maps (56x56 / 112x112 inputs) the strip kernel, whose loads hit
    // L1 / L2 directly, is 25-40 % faster (measured); the small maps gain 10-20 % from the band
    static const int s2max = [] { const char* e = getenv("MCN_DW_BAND_S2MAX"); return e ? atoi(e) : 14; }();
    if (stride == 2 && OH > s2max && kthreads == 1) return false;
    const int cch = C / ce;
    int TC = cch < 8 ? cch : 8;
    for (int d = 8; d >= 5; --d)
        if (cch % d == 0) { TC = d; break; }
    const int TCP = TC | 1;
    const int NS = (OW + DW_BAND_S - 1) / DW_BAND_S, nthr = 256 / (TC * kthreads);
    if (nthr < 1) return false;
    double best = -1;
    DwBandParams b;
    memset(&b, 0, sizeof(b));
    int last_tns = -1;
    for (int parts = 1; parts <= NS; ++parts) {
        const int TNS = (NS + parts - 1) / parts;
        if (TNS == last_tns) continue;
        last_tns = TNS;
        for (int TH = 1; TH <= OH && TH <= 32; ++TH) {
            const int BH = (TH - 1) * stride + K, BW = (TNS * DW_BAND_S - 1) * stride + K;
            const long slots = (long)BH * BW * TCP;
            const int npass = (int)((slots + 255) / 256);
            const int npy = npass_y ? (int)(((long)TH * TNS * DW_BAND_S * TCP + 255) / 256) : 0;
            if (npass > DW_BAND_MAXPASS || npy > DW_BAND_MAXPASS || (long)(npass + npy) * 4096 > DW_BAND_LDS_BUDGET) break;
            const int strips = TH * TNS;
            const double util = (double)strips / (((strips + nthr - 1) / nthr) * nthr);
            const int ty = (OH + TH - 1) / TH, txn = (NS + TNS - 1) / TNS;
            const double cover = ((double)OH / (ty * TH)) * ((double)NS / (txn * TNS));       // partly empty last tiles
            const double score = (double)(TH * TNS * DW_BAND_S) * stride * stride / ((double)BH * BW) * util * cover;
            if (score > best) {
                best = score;
                b.TH = TH; b.TNS = TNS; b.BH = BH; b.BW = BW; b.npass = npass;
                b.tiles_y = ty; b.tiles_x = txn;
                if (npass_y) *npass_y = npy;
            }
        }
    }
    if (best < 0) return false;
    b.H = H; b.W = W; b.C = C; b.OH = OH; b.OW = OW; b.padT = padT; b.padL = padL;
    b.TC = TC; b.TCP = TCP; b.NS = NS; b.accumulate = 0; b.img_bytes = (int)((size_t)H * W * C * es);
    *out = b;
    return true;
}
template <typename T, bool FLIP>
static int dw_band_launch(const void* x, const float* w, void* y, int N, int K, int stride, const DwBandParams& b, hipStream_t st) {
    const int cch = b.C / VecTraits<T>::CE;
    const dim3 grid((unsigned)((cch + b.TC - 1) / b.TC), (unsigned)((long)N * b.tiles_y * b.tiles_x)), block(256);
    const size_t lds = (size_t)b.npass * 4096 + (size_t)K * K * b.TC * VecTraits<T>::CE * sizeof(float);
#define DW_BAND(KK, SS) hipLaunchKernelGGL((dw_band_kernel<T, KK, SS, DW_BAND_S, FLIP>), grid, block, lds, st, (const T*)x, w, (T*)y, b)
    if (K == 3 && stride == 1) DW_BAND(3, 1);
    else if (K == 3) DW_BAND(3, 2);
    else if (stride == 1) DW_BAND(5, 1);
    else DW_BAND(5, 2);
#undef DW_BAND
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

template <typename T>
static int dw_fwd_t(const void* x, const float* w, void* y, const mcn_conv_geom* g, hipStream_t st) {
    unsigned gx, gy;
    const DwParams p = dw_params(g, VecTraits<T>::CE, false, &gx, &gy);
    if (p.npix == 0) return MCN_OK;
    const size_t lds = (size_t)p.KH * p.KW * p.TX * VecTraits<T>::CE * sizeof(float);
    const dim3 grid(gx, gy), block(256);
    DwBandParams b;
    if (dw_strip_ok(p) && dw_band_plan(p.H, p.W, p.C, p.OH, p.OW, p.KH, p.SH, p.padT, p.padL, VecTraits<T>::CE, sizeof(T), &b))
        return dw_band_launch<T, false>(x, w, y, p.N, p.KH, p.SH, b, st);
    if (dw_strip_ok(p)) {
        constexpr int S = 4;
#define DW_STRIP(KK, SS) hipLaunchKernelGGL((dw_strip_kernel<T, KK, SS, S, false>), grid, block, lds, st, (const T*)x, w, (T*)y, p)
        if (p.KH == 3 && p.SH == 1) DW_STRIP(3, 1);
        else if (p.KH == 3) DW_STRIP(3, 2);
        else if (p.SH == 1) DW_STRIP(5, 1);
        else DW_STRIP(5, 2);
#undef DW_STRIP
    } else if (p.KH == 3 && p.KW == 3) hipLaunchKernelGGL((dw_fwd_kernel<T, 3>), grid, block, lds, st, (const T*)x, w, (T*)y, p);
    else if (p.KH == 5 && p.KW == 5) hipLaunchKernelGGL((dw_fwd_kernel<T, 5>), grid, block, lds, st, (const T*)x, w, (T*)y, p);
    else hipLaunchKernelGGL((dw_fwd_kernel<T, 0>), grid, block, lds, st, (const T*)x, w, (T*)y, p);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_dwconv2d_fwd(const void* x, const float* w, void* y, const mcn_conv_geom* g, mcn_dtype dtype, void* stream) {
    if (int rc = dw_check(g, dtype, "dwconv2d_fwd")) return rc;
    if (!x || !w || !y) MCN_FAIL(MCN_E_BADARG, "dwconv2d_fwd: null pointer");
    if (dtype == MCN_F16) return dw_fwd_t<f16_t>(x, w, y, g, (hipStream_t)stream);
    return dtype == MCN_F32 ? dw_fwd_t<float>(x, w, y, g, (hipStream_t)stream) : dw_fwd_t<bf16_t>(x, w, y, g, (hipStream_t)stream);
}

template <typename T>
static int dw_dgrad_t(const void* dy, const float* w, void* dx, const mcn_conv_geom* g, int accumulate, hipStream_t st) {
    unsigned gx, gy;
    DwParams p = dw_params(g, VecTraits<T>::CE, true, &gx, &gy);
    p.accumulate = accumulate;
    if (p.npix == 0) return MCN_OK;
    const size_t lds = (size_t)p.KH * p.KW * p.TX * VecTraits<T>::CE * sizeof(float);
    const dim3 grid(gx, gy), block(256);
    if (dw_strip_ok(p) && p.SH == 1) {
        // stride 1: dx = dy (*) rot180(w) with pads K-1-pad — the forward strip kernel over dy
        DwParams f = p;
        f.H = p.OH; f.W = p.OW; f.OH = p.H; f.OW = p.W;
        f.padT = p.KH - 1 - p.padT; f.padL = p.KW - 1 - p.padL;
        DwBandParams b;
        if (dw_band_plan(f.H, f.W, f.C, f.OH, f.OW, p.KH, 1, f.padT, f.padL, VecTraits<T>::CE, sizeof(T), &b)) {
            b.accumulate = accumulate;
            return dw_band_launch<T, true>(dy, w, dx, p.N, p.KH, 1, b, st);
        }
        if (p.KH == 3) hipLaunchKernelGGL((dw_strip_kernel<T, 3, 1, 4, true>), grid, block, lds, st, (const T*)dy, w, (T*)dx, f);
        else hipLaunchKernelGGL((dw_strip_kernel<T, 5, 1, 4, true>), grid, block, lds, st, (const T*)dy, w, (T*)dx, f);
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    if (dw_strip_ok(p) && p.SH == 2 && p.padL >= 0 && p.padL <= 2 && p.padL < p.KW) {
        // stride 2: strips of dx pixels, one dy row segment per contributing filter row (grid: strips of the INPUT grid)
        const long nstrips = (long)p.N * p.H * ((p.W + 3) / 4);
        long rows = (nstrips + p.TY - 1) / p.TY, want = 4096 / gx;
        if (want < 1) want = 1;
        const dim3 sgrid(gx, (unsigned)(rows < want ? (rows < 1 ? 1 : rows) : want));
#define DW_DS2(KK, PLV) hipLaunchKernelGGL((dw_dgrad_s2_kernel<T, KK, PLV, 4>), sgrid, block, lds, st, (const T*)dy, w, (T*)dx, p)
        if (p.KH == 3) { if (p.padL == 0) DW_DS2(3, 0); else if (p.padL == 1) DW_DS2(3, 1); else DW_DS2(3, 2); }
        else { if (p.padL == 0) DW_DS2(5, 0); else if (p.padL == 1) DW_DS2(5, 1); else DW_DS2(5, 2); }
#undef DW_DS2
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    if (p.KH == 3 && p.KW == 3) hipLaunchKernelGGL((dw_dgrad_kernel<T, 3>), grid, block, lds, st, (const T*)dy, w, (T*)dx, p);
    else if (p.KH == 5 && p.KW == 5) hipLaunchKernelGGL((dw_dgrad_kernel<T, 5>), grid, block, lds, st, (const T*)dy, w, (T*)dx, p);
    else hipLaunchKernelGGL((dw_dgrad_kernel<T, 0>), grid, block, lds, st, (const T*)dy, w, (T*)dx, p);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_dwconv2d_dgrad(const void* dy, const float* w, void* dx, const mcn_conv_geom* g, int32_t accumulate, mcn_dtype dtype,
                                  void* stream) {
    if (int rc = dw_check(g, dtype, "dwconv2d_dgrad")) return rc;
    if (!dy || !w || !dx) MCN_FAIL(MCN_E_BADARG, "dwconv2d_dgrad: null pointer");
    if (dtype == MCN_F16) return dw_dgrad_t<f16_t>(dy, w, dx, g, accumulate, (hipStream_t)stream);
    return dtype == MCN_F32 ? dw_dgrad_t<float>(dy, w, dx, g, accumulate, (hipStream_t)stream)
                            : dw_dgrad_t<bf16_t>(dy, w, dx, g, accumulate, (hipStream_t)stream);
}

// MCN_DW_WGRAD_ROWS: 1 (default) = the row-split kernel for the bf16 5x5 case, 0 = the 4-channel strip kernel everywhere
static bool dw_wgrad_rows() {
    static const int v = [] { const char* e = getenv("MCN_DW_WGRAD_ROWS"); return e ? atoi(e) : 1; }();
    return v != 0;
}
static DwParams dw_wgrad_params(const mcn_conv_geom* g, mcn_dtype dtype, unsigned* gx, unsigned* gy) {
    unsigned a, b;
    DwParams p = dw_params(g, 4, false, &a, &b);
    const int ce = dtype == MCN_F32 ? 4 : 8;
    if (dw_wgrad_rows() && dw_strip_ok(p) && p.C % ce == 0 && dtype != MCN_F32 && p.KH == 5) {    // (measured: 3x3 and fp32 lose with it)
        // row-split kernel: TX chunks x TY rows of threads, TY / K strips side by side
        const int cch = p.C / ce;
        p.TX = best_tx(cch, 32);
        p.TY = 256 / p.TX;
        *gx = (unsigned)((cch + p.TX - 1) / p.TX);
        const long tys = p.TY / p.KH;
        const long nstrips = (long)p.N * p.OH * ((p.OW + 3) / 4);
        long rows = (nstrips + tys - 1) / tys;
        long want = 2048 / *gx;
        if (want < 1) want = 1;
        *gy = (unsigned)(rows < want ? (rows < 1 ? 1 : rows) : want);
        return p;
    }
    const int c4 = p.C / 4;
    p.TX = best_tx(c4, 64);
    p.TY = 256 / p.TX;
    *gx = (unsigned)((c4 + p.TX - 1) / p.TX);
    long rows = (p.npix + p.TY - 1) / p.TY;
    long want = 1024 / *gx;
    if (want < 1) want = 1;
    *gy = (unsigned)(rows < want ? (rows < 1 ? 1 : rows) : want);
    return p;
}
// band form of the wgrad (dw_wgrad_band_kernel): plan + grid.  MCN_DW_WBAND=0 switches it off.
static bool dw_wgrad_band_plan(const mcn_conv_geom* g, mcn_dtype dtype, DwWBandParams* pw, unsigned* gx, unsigned* gy) {
    static const int on = [] { const char* e = getenv("MCN_DW_WBAND"); return e ? atoi(e) : 1; }();
    unsigned a, b2;
    const DwParams p = dw_params(g, 4, false, &a, &b2);
    const int ce = dtype == MCN_F32 ? 4 : 8;
    const size_t es = mcn_dtype_size(dtype);
    // stride 1 only: measured per layer (EfficientNet-B0, bf16, B = 512) 5x5 215 -> 149, 163 -> 113, 294 -> 212 us, 3x3 82 -> 75, 266 -> 233
    // (56x56 144ch 298 -> 318); stride 2 stages four x pixels per dy pixel and loses 2-38 % against the strip / row-split kernels
    if (on < 2 && p.SH != 1) return false;
    if (!on || !dw_strip_ok(p) || p.C % ce || p.npix == 0 || (size_t)p.OH * p.OW * p.C * es >= 0x7fffffffull) return false;
    DwWBandParams w;
    memset(&w, 0, sizeof(w));
    int npy = 0;
    if (!dw_band_plan(p.H, p.W, p.C, p.OH, p.OW, p.KH, p.SH, p.padT, p.padL, ce, es, &w.b, p.KH, &npy)) return false;
    w.DW_ = w.b.TNS * DW_BAND_S;
    w.npass_y = npy;
    w.ntiles = p.N * w.b.tiles_y * w.b.tiles_x;
    w.dy_img_bytes = (int)((size_t)p.OH * p.OW * p.C * es);
    const int cch = p.C / ce;
    *gx = (unsigned)((cch + w.b.TC - 1) / w.b.TC);
    long want = 1024 / *gx;                                            // ~4 workgroups per CU in all
    if (want < 1) want = 1;
    *gy = (unsigned)(w.ntiles < want ? w.ntiles : want);
    *pw = w;
    return true;
}
extern "C" size_t mcn_dwconv2d_workspace_bytes(const mcn_conv_geom* g, mcn_dtype dtype) {
    if (!g || g->Cin <= 0 || g->Cin % 4 || g->KH <= 0 || g->KW <= 0 || g->N < 0) return 0;
    unsigned gx, gy;
    dw_wgrad_params(g, dtype, &gx, &gy);
    DwWBandParams pw;
    unsigned bx, by = 0;
    if (mcn_dtype_ok(dtype) && g->H > 0 && g->W > 0 && g->SH > 0 && g->SW > 0 && g->DH > 0 && g->DW > 0 && dw_wgrad_band_plan(g, dtype, &pw, &bx, &by) && by > gy) gy = by;
    return align_up((size_t)gy * g->KH * g->KW * g->Cin * sizeof(float), 256);
}
template <typename T>
static int dw_wgrad_t(const void* x, const void* dy, float* dw, const mcn_conv_geom* g, float scale, void* ws, hipStream_t st) {
    unsigned gx, gy;
    const DwParams p = dw_wgrad_params(g, DtypeOf<T>::value, &gx, &gy);
    const long n = (long)p.KH * p.KW * p.C;
    if (p.npix == 0) {
        if (hipMemsetAsync(dw, 0, (size_t)n * sizeof(float), st) != hipSuccess) MCN_FAIL(MCN_E_LAUNCH, "dwconv2d_wgrad: memset failed");
        return MCN_OK;
    }
    float* part = (float*)ws;
    const size_t lds = (size_t)p.TY * p.TX * 4 * sizeof(float);
    const dim3 block(256);
    DwWBandParams pw;
    unsigned bx, by;
    if (dw_wgrad_band_plan(g, DtypeOf<T>::value, &pw, &bx, &by)) {
        const size_t bl = (size_t)(pw.b.npass + pw.npass_y) * 4096;
#define DW_WBAND(KK, SS) hipLaunchKernelGGL((dw_wgrad_band_kernel<T, KK, SS, DW_BAND_S>), dim3(bx, by), block, bl, st, (const T*)x, (const T*)dy, part, pw)
        if (p.KH == 3 && p.SH == 1) DW_WBAND(3, 1);
        else if (p.KH == 3) DW_WBAND(3, 2);
        else if (p.SH == 1) DW_WBAND(5, 1);
        else DW_WBAND(5, 2);
#undef DW_WBAND
        gy = by;
    } else if (dw_wgrad_rows() && dw_strip_ok(p) && p.C % VecTraits<T>::CE == 0 && sizeof(T) == 2 && p.KH == 5) {
        const size_t rl = (size_t)p.TY * p.TX * VecTraits<T>::CE * sizeof(float);
#define DW_WROWS(KK, SS) hipLaunchKernelGGL((dw_wgrad_rows_kernel<T, KK, SS, 4>), dim3(gx, gy), block, rl, st, (const T*)x, (const T*)dy, part, p)
        if (p.SH == 1) DW_WROWS(5, 1);
        else DW_WROWS(5, 2);
#undef DW_WROWS
    } else if (dw_strip_ok(p)) {
        constexpr int S = 4;
#define DW_WSTRIP(KK, SS) hipLaunchKernelGGL((dw_wgrad_strip_kernel<T, KK, SS, S>), dim3(gx, gy), block, lds, st, (const T*)x, (const T*)dy, part, p)
        if (p.KH == 3 && p.SH == 1) DW_WSTRIP(3, 1);
        else if (p.KH == 3) DW_WSTRIP(3, 2);
        else if (p.SH == 1) DW_WSTRIP(5, 1);
        else DW_WSTRIP(5, 2);
#undef DW_WSTRIP
    } else if (p.KH == 3 && p.KW == 3) hipLaunchKernelGGL((dw_wgrad_kernel<T, 3>), dim3(gx, gy), block, lds, st, (const T*)x, (const T*)dy, part, p);
    else if (p.KH == 5 && p.KW == 5) hipLaunchKernelGGL((dw_wgrad_kernel<T, 5>), dim3(gx, gy), block, lds, st, (const T*)x, (const T*)dy, part, p);
    else hipLaunchKernelGGL((dw_wgrad_tap_kernel<T>), dim3(gx, gy, p.KH * p.KW), block, lds, st, (const T*)x, (const T*)dy, part, p);
    MCN_CHECK_LAUNCH();
    if (gy >= 64 && n <= 8192) hipLaunchKernelGGL(dw_wgrad_reduce_kernel<8>, dim3((unsigned)((n + 7) / 8)), block, 0, st, (const float*)part, dw, n, (int)gy, scale);
    else if (gy >= 16) hipLaunchKernelGGL(dw_wgrad_reduce_kernel<32>, dim3((unsigned)((n + 31) / 32)), block, 0, st, (const float*)part, dw, n, (int)gy, scale);
    else hipLaunchKernelGGL(dw_wgrad_reduce_kernel<256>, dim3((unsigned)((n + 255) / 256)), block, 0, st, (const float*)part, dw, n, (int)gy, scale);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_dwconv2d_wgrad(const void* x, const void* dy, float* dw, const mcn_conv_geom* g, float grad_scale, mcn_dtype dtype, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (int rc = dw_check(g, dtype, "dwconv2d_wgrad")) return rc;
    if (!x || !dy || !dw) MCN_FAIL(MCN_E_BADARG, "dwconv2d_wgrad: null pointer");
    if (!ws || ws_bytes < mcn_dwconv2d_workspace_bytes(g, dtype)) MCN_FAIL(MCN_E_WORKSPACE, "dwconv2d_wgrad: workspace too small");
    if (dtype == MCN_F16) return dw_wgrad_t<f16_t>(x, dy, dw, g, grad_scale, ws, (hipStream_t)stream);
    return dtype == MCN_F32 ? dw_wgrad_t<float>(x, dy, dw, g, grad_scale, ws, (hipStream_t)stream)
                            : dw_wgrad_t<bf16_t>(x, dy, dw, g, grad_scale, ws, (hipStream_t)stream);
}

// ---- squeeze-excite channel scale: y[n,h,w,c] = x[n,h,w,c] * m[n,c] (models/efficientnet.py:161) ---------------------
template <typename T>
__global__ __launch_bounds__(256) void chscale_fwd_kernel(const T* __restrict__ x, const T* __restrict__ m, T* __restrict__ y, long total_, long HW, int cch) {
    constexpr int CE = VecTraits<T>::CE;
    const unsigned total = (unsigned)total_;                       // (host: < 2^32 chunks — two 64-bit divisions per 16 bytes were most of this kernel)
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned pix = i / (unsigned)cch;
        const int ch = (int)(i - pix * (unsigned)cch);
        const long n = pix / (unsigned)HW;
        const Chunk<T> a = load_chunk<T>(x + (long)i * CE), b = load_chunk<T>(m + (n * cch + ch) * CE);
        Chunk<T> o;
#pragma unroll
        for (int k = 0; k < CE; ++k) o.set(k, a.get(k) * b.get(k));
        store_chunk<T>(y + (long)i * CE, o);
    }
}
extern "C" int mcn_channel_scale_fwd(const void* x, const void* m, void* y, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void* stream) {
    if (!x || !m || !y || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "channel_scale_fwd: bad argument");
    if (!mcn_dtype_ok(dtype)) MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_fwd: dtype %d unsupported", (int)dtype);
    const int ce = dtype == MCN_F32 ? 4 : 8;
    if (C % ce) MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_fwd: C=%d must be a multiple of the 16-byte chunk", C);
    const long total = (long)N * HW * (C / ce);
    if (total == 0) return MCN_OK;
    if (total >= 0xffffffffl) MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_fwd: more than 2^32 chunks (32-bit index arithmetic)");
    long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (dtype == MCN_F32) hipLaunchKernelGGL((chscale_fwd_kernel<float>), dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)m, (float*)y, total, (long)HW, C / ce);
    else if (dtype == MCN_F16) hipLaunchKernelGGL((chscale_fwd_kernel<f16_t>), dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, (const f16_t*)m, (f16_t*)y, total, (long)HW, C / ce);
    else hipLaunchKernelGGL((chscale_fwd_kernel<bf16_t>), dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)m, (bf16_t*)y, total, (long)HW, C / ce);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// dx = dy * m ; dm[n,c] = sum_hw dy * x  (one block per image x group of TX chunks)
template <typename T>
__global__ __launch_bounds__(256) void chscale_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ m, T* __restrict__ dx,
                                                          T* __restrict__ dm, long HW, int C, int TX, int TY) {
    constexpr int CE = VecTraits<T>::CE;
    extern __shared__ float red[];                               // [TY][TX*CE]
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int cch = C / CE;
    const int chunk = blockIdx.x * TX + tx;
    const long n = blockIdx.y;
    const bool active = ty < TY && chunk < cch;
    float acc[CE];
#pragma unroll
    for (int i = 0; i < CE; ++i) acc[i] = 0.f;
    if (active) {
        const Chunk<T> mm = load_chunk<T>(m + (n * cch + chunk) * CE);
        for (long r = ty; r < HW; r += TY) {
            const long off = ((n * HW + r) * cch + chunk) * CE;
            const Chunk<T> g = load_chunk<T>(dy + off), v = load_chunk<T>(x + off);
            Chunk<T> o;
#pragma unroll
            for (int i = 0; i < CE; ++i) {
                o.set(i, g.get(i) * mm.get(i));
                acc[i] = fmaf(g.get(i), v.get(i), acc[i]);
            }
            store_chunk<T>(dx + off, o);
        }
    }
    const int cols = TX * CE;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < CE; ++i) red[ty * cols + tx * CE + i] = acc[i];
    }
    __syncthreads();
    if (ty == 0 && chunk < cch) {
        Chunk<T> o;
#pragma unroll
        for (int i = 0; i < CE; ++i) {
            float s = 0.f;
            for (int k = 0; k < TY; ++k) s += red[k * cols + tx * CE + i];
            o.set(i, s);
        }
        store_chunk<T>(dm + (n * cch + chunk) * CE, o);
    }
}
// dm only (the dx half of chscale_bwd_kernel is composed inside the BN backward: mcn_bn_bwd_se)
template <typename T>
__global__ __launch_bounds__(256) void chscale_bwd_dm_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dm, long HW, int C, int TX, int TY) {
    constexpr int CE = VecTraits<T>::CE;
    extern __shared__ float red[];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int cch = C / CE;
    const int chunk = blockIdx.x * TX + tx;
    const long n = blockIdx.y;
    const bool active = ty < TY && chunk < cch;
    float acc[CE];
#pragma unroll
    for (int i = 0; i < CE; ++i) acc[i] = 0.f;
    if (active) {
        for (long r = ty; r < HW; r += TY) {
            const long off = ((n * HW + r) * cch + chunk) * CE;
            const Chunk<T> g = load_chunk<T>(dy + off), v = load_chunk<T>(x + off);
#pragma unroll
            for (int i = 0; i < CE; ++i) acc[i] = fmaf(g.get(i), v.get(i), acc[i]);
        }
    }
    const int cols = TX * CE;
    if (ty < TY) {
#pragma unroll
        for (int i = 0; i < CE; ++i) red[ty * cols + tx * CE + i] = acc[i];
    }
    __syncthreads();
    if (ty == 0 && chunk < cch) {
        Chunk<T> o;
#pragma unroll
        for (int i = 0; i < CE; ++i) {
            float s = 0.f;
            for (int k = 0; k < TY; ++k) s += red[k * cols + tx * CE + i];
            o.set(i, s);
        }
        store_chunk<T>(dm + (n * cch + chunk) * CE, o);
    }
}
extern "C" int mcn_channel_scale_bwd_dm(const void* dy, const void* x, void* dm, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void* stream) {
    if (!dy || !x || !dm || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "channel_scale_bwd_dm: bad argument");
    if (!mcn_dtype_ok(dtype)) MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_bwd_dm: dtype %d unsupported", (int)dtype);
    const int ce = dtype == MCN_F32 ? 4 : 8;
    if (C % ce) MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_bwd_dm: C=%d must be a multiple of the 16-byte chunk", C);
    if (N == 0) return MCN_OK;
    const int cch = C / ce;
    const int TX = best_tx(cch, 32), TY = 256 / TX;
    const dim3 grid((unsigned)((cch + TX - 1) / TX), (unsigned)N), block(256);
    const size_t lds = (size_t)TY * TX * ce * sizeof(float);
    if (dtype == MCN_F32) hipLaunchKernelGGL((chscale_bwd_dm_kernel<float>), grid, block, lds, (hipStream_t)stream, (const float*)dy, (const float*)x, (float*)dm, (long)HW, C, TX, TY);
    else if (dtype == MCN_F16) hipLaunchKernelGGL((chscale_bwd_dm_kernel<f16_t>), grid, block, lds, (hipStream_t)stream, (const f16_t*)dy, (const f16_t*)x, (f16_t*)dm, (long)HW, C, TX, TY);
    else hipLaunchKernelGGL((chscale_bwd_dm_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dm, (long)HW, C, TX, TY);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_channel_scale_bwd(const void* dy, const void* x, const void* m, void* dx, void* dm, int32_t N, int64_t HW, int32_t C,
                                     mcn_dtype dtype, void* stream) {
    if (!dy || !x || !m || !dx || !dm || N < 0 || HW <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "channel_scale_bwd: bad argument");
    if (!mcn_dtype_ok(dtype)) MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_bwd: dtype %d unsupported", (int)dtype);
    const int ce = dtype == MCN_F32 ? 4 : 8;
    if (C % ce) MCN_FAIL(MCN_E_UNSUPPORTED, "channel_scale_bwd: C=%d must be a multiple of the 16-byte chunk", C);
    if (N == 0) return MCN_OK;
    const int cch = C / ce;
    const int TX = best_tx(cch, 32), TY = 256 / TX;
    const dim3 grid((unsigned)((cch + TX - 1) / TX), (unsigned)N), block(256);
    const size_t lds = (size_t)TY * TX * ce * sizeof(float);
    if (dtype == MCN_F32) hipLaunchKernelGGL((chscale_bwd_kernel<float>), grid, block, lds, (hipStream_t)stream, (const float*)dy, (const float*)x, (const float*)m, (float*)dx, (float*)dm, (long)HW, C, TX, TY);
    else if (dtype == MCN_F16) hipLaunchKernelGGL((chscale_bwd_kernel<f16_t>), grid, block, lds, (hipStream_t)stream, (const f16_t*)dy, (const f16_t*)x, (const f16_t*)m, (f16_t*)dx, (f16_t*)dm, (long)HW, C, TX, TY);
    else hipLaunchKernelGGL((chscale_bwd_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)m, (bf16_t*)dx, (bf16_t*)dm, (long)HW, C, TX, TY);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
