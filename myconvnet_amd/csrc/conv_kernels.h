// conv_kernels.h — device kernels for conv2d fwd / dgrad / wgrad on gfx950.
//
// Two MFMA kernels cover every convolution of the path:
//
//  * conv_gemm_nt : OUT[m][n] (+)= sum_k A[m][k] * Bt[n][k]
//      m = output pixel of a (sub-)grid, k = (tap, channel), A gathered on the fly from an NHWC
//      tensor through a per-tap (dy,dx) table with hardware bounds-checked buffer loads (padding
//      = out-of-range => 0), Bt = weights pre-packed K-contiguous.  Used for the forward pass
//      (A = x, taps = filter taps) and for dgrad (A = dy, taps = the filter taps that reach one
//      stride-parity class of dx; stride-2 dgrad is 4 exact sub-convolutions, no wasted MACs).
//  * conv_gemm_tn : DW[(tap,c)][n] = sum_m X[pix(m,tap)][c] * DY[m][n]   (split over m)
//      both operands are pixel-major in memory, i.e. the reduction index is the slow one; bf16
//      fragments come from LDS through ds_read_b64_tr_b16, fp32 through conflict-free b32 reads.
//
// Both kernels stage their tiles with LDS-DMA loads (buffer_load_dwordx4 ... lds) into XOR-swizzled, double-buffered LDS
// images (the DMA of K-step ks+1 flies under the MFMAs of step ks; one barrier per K-step; the swizzle sits on the SOURCE
// address because a wave-instruction writes LDS in lane order).  bf16 / fp16 use v_mfma_f32_16x16x32_{bf16,f16},
// fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32; no TF32 on gfx950).  Operand order is
// (weights, activations) so that each lane ends with 4 consecutive output channels of one pixel
// => 8/16-byte epilogue stores.
#pragma once
#include <type_traits>
#include "common.h"

#define MCN_MAX_TAPS 64
#define MCN_OOB 0x80000000u
// Timing probes inside the kernels (MCN_NT_EPI_FLAGS / MCN_TN_DBG: skip a kind of instruction, wrong results) are compiled in only with
// -DMCN_KERNEL_PROBES=1: as run-time branches they split the K loop into basic blocks the scheduler does not interleave across
// (measured on conv_gemm_nt_wpp: 49 us without, 60 us with four such branches in the loop, profiles/round4_nt_wpp_probe_bf16.txt).
#ifndef MCN_KERNEL_PROBES
#define MCN_KERNEL_PROBES 0
#endif
static constexpr bool MCN_PROBES = MCN_KERNEL_PROBES != 0;

struct GemmNTParams {
    const void* in;
    const void* wt;
    void* out;
    const float* bias;
    float* stats;           // optional [2*ntm][3][Nn] per-wave-row (sum(y-p), sum((y-p)^2), p) of the stored output y, p = the
                            // wave row's first pixel (BN statistics; the shift keeps E[y^2]-E[y]^2 cancellation out of fp32)
    int M, OH, OW;          // GEMM rows = N*OH*OW of the (sub-)grid
    int m_begin, m_end;     // row range this launch covers (a conv may be split into a big-tile body and a small-tile tail)
    int IH, IW, Cs;         // gathered tensor: spatial dims, channel stride (elements)
    int cpt;                // 16-byte chunks per tap
    int ntaps;
    int nchunks;            // ntaps*cpt
    int sy, sx;             // gathered-tensor step per grid step
    int Nn;                 // GEMM N
    int OWf, OHf, ldo;      // full output grid (pixels) and channel stride
    int osy, osx, oy0, ox0; // scatter of the sub-grid into the full output grid
    // stream-K tail (see launch_nt): sk_mode 1 = workgroups >= sk_body compute one of sk_slices K-slices of a tail tile each
    // and write their accumulators to `partial`; conv_nt_sk_reduce sums them and runs the epilogue
    float* partial;
    int sk_mode, sk_slices, sk_body;
    int accumulate;         // 0: store; 1: out += old value; 2: out += add_src * [bit of add_mask] (the masked gradient of a
                            // residual block's output, see mcn_conv2d_dgrad_addmasked)
    const void* add_src;
    const unsigned char* add_mask;
    unsigned in_bytes, wt_bytes;
    int win_dmin, win_rows; // conv_gemm_nt_win: smallest tap offset in pixels (dy*IW + dx) and the rows of the input window BM + (dmax - dmin)
    unsigned out_bytes;     // bytes of the whole output tensor (set by launch_nt: buffer range of the epilogue's stores / loads)
    int epi_flags;          // experiments (MCN_NT_EPI_FLAGS): 1 = no lane pairing (8-byte bf16 accesses), 2 = byte-wise mask loads
    // NT_EPI_BNRED (dgrad whose output IS the gradient of a BN + ReLU output): the BN's backward sums ride in this epilogue.
    // red_x = the BN's input (layout of `out`), red_mask = the forward's ReLU byte mask (one byte per 16-byte chunk), red_part =
    // partial rows [rows][2][Nn] = sum dy', sum dy' * x over the pixel rows of a wave row, dy' = the stored gradient where the mask bit is set
    const void* red_x;
    const unsigned char* red_mask;
    float* red_part;
    int red_row0;           // first partial row of this launch (a strided dgrad launches once per stride-parity class)
    int tap[MCN_MAX_TAPS];  // (dy & 0xffff) | (dx << 16) per filter tap: 32-bit so that a wave-uniform tap index is a scalar load
                            // (byte tables are fetched with vector loads whose waits drain the LDS-DMA queue)
};

// Cache policy of the LDS-DMA loads of the activation operands (build-time experiments): the aux immediate of buffer_load ... lds,
// 2 = nt (streaming).  MCN_DMA_AUX_A: A operand of the NT kernels (activations; the packed filter stays cached); MCN_DMA_AUX_X: x and dy
// of the wgrad.
#ifndef MCN_DMA_AUX_A
#define MCN_DMA_AUX_A 0
#endif
#ifndef MCN_DMA_AUX_X
#define MCN_DMA_AUX_X 0
#endif
struct GemmTNParams {
    const void* x;
    const void* dy;
    float* slab;            // [splits][rows][Nn]
    int M, OH, OW;
    int IH, IW, Cs;
    int Cp;                 // channels per tap (padded to chunk)
    int ntaps;
    int rows;               // ntaps*Cp
    int sy, sx;
    int Nn, ldy;
    int nsteps, steps_per_split;
    unsigned x_bytes, dy_bytes;
    int dbg;                // DEBUG probes (MCN_TN_DBG): 1 = X staged for the first K-step only, 2 = DY likewise (wrong results, timing only); 32 = early DMA issue (correct results)
    int grp;                // > 0: XCD-aware order — groups of grp tiles of one split next to each other on one XCD
    signed char tdy[MCN_MAX_TAPS];
    signed char tdx[MCN_MAX_TAPS];
};

__device__ __forceinline__ i32x4 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
}
// the epilogue's read of a tensor in the output's layout (residual / accumulate source, BN input): read once (MCN_EPI_AUX: 2 = nt)
#ifndef MCN_EPI_AUX
#define MCN_EPI_AUX 0
#endif
__device__ __forceinline__ i32x4 buf_load16_epi(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, MCN_EPI_AUX);
}

// ------------------------------------------------------------------------------------------------
// MFMA policies
// ------------------------------------------------------------------------------------------------
template <typename T>
struct MmaNT;

// bf16: 16x16x32, K-step = 64 elements = 2 slabs; fragment = one 16-byte chunk
template <>
struct MmaNT<bf16_t> {
    static constexpr int MT = 16;       // tile edge
    static constexpr int SLABS = 2;     // per 128-byte K-step
    static constexpr int CPS = 4;       // chunks per slab
    typedef f32x4 Acc;
    typedef bf16x8 Frag;
    static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
    static __device__ __forceinline__ int frag_chunk(int lane) { return lane >> 4; }
    static __device__ __forceinline__ void mma(Acc& acc, const Frag& w, const Frag& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
    }
};
// fp16: the bf16 policy with the f16 MFMA (same cycles, same operand / accumulator lane maps)
template <>
struct MmaNT<f16_t> {
    static constexpr int MT = 16;
    static constexpr int SLABS = 2;
    static constexpr int CPS = 4;
    typedef f32x4 Acc;
    typedef f16x8 Frag;
    static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
    static __device__ __forceinline__ int frag_chunk(int lane) { return lane >> 4; }
    static __device__ __forceinline__ void mma(Acc& acc, const Frag& w, const Frag& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, acc, 0, 0, 0);
    }
};
// fp32: 32x32x2; a 16-byte fragment (4 consecutive k) feeds 4 MFMAs (k pairs {e, 4+e})
template <>
struct MmaNT<float> {
    static constexpr int MT = 32;
    static constexpr int SLABS = 4;
    static constexpr int CPS = 2;
    typedef f32x16 Acc;
    typedef f32x4 Frag;
    static __device__ __forceinline__ int frag_row(int lane) { return lane & 31; }
    static __device__ __forceinline__ int frag_chunk(int lane) { return lane >> 5; }
    static __device__ __forceinline__ void mma(Acc& acc, const Frag& w, const Frag& x) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], x[e], acc, 0, 0, 0);
    }
};

// LDS image of an NT tile: rows of 128 bytes, 16-byte chunk c of row r at r*128 + ((c ^ ((r>>1)&7))<<4)
__device__ __forceinline__ int nt_lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// Halving butterfly over lanes (see the statistics epilogue of conv_gemm_nt): C values left, partner offset O.  All array
// indices are compile-time constants (a run-time count makes the compiler index the register arrays with select chains).
// The exchanges are DPP moves inside the 16-lane rows and v_permlane16_swap across them — no LDS traffic (the ds_bpermute
// that __shfl_xor compiles to cost 46 LDS round trips per tile in four dependent levels, queued behind the other workgroup's
// fragment reads).  A step only has to pair every lane with one whose bit O differs and that kept the same values so far:
// offset 8 = rotate the row by 8, offset 4 = mirror the half row (partner lane ^ 7), offsets 2 / 1 = quad permutes.
template <int O>
__device__ __forceinline__ float lane_partner(float x) {
    static_assert(O == 8 || O == 4 || O == 2 || O == 1, "DPP partner inside a 16-lane row");
    constexpr int ctrl = O == 8 ? 0x128 /* row_ror:8 */ : (O == 4 ? 0x141 /* row_half_mirror */ : (O == 2 ? 0x4E /* quad_perm:[2,3,0,1] */ : 0xB1 /* [1,0,3,2] */));
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false));
}
// lane 0 of every 16-lane row to the whole row
__device__ __forceinline__ float row_first(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x150 /* row_newbcast:0 */, 0xf, 0xf, false));
}
template <int C, int O>
struct LaneFold {
    static __device__ __forceinline__ void run(float* a, float* b, int lane, int& base, bool& writer) {
        const bool hi = (lane & O) != 0;
        if constexpr (C > 1) {
            constexpr int H = C / 2;
#pragma unroll
            for (int t = 0; t < H; ++t) {
                if constexpr (O == 16) {
                    // rows 1 / 3 of the first operand trade places with rows 0 / 2 of the second: every lane then holds its own kept
                    // value in one result and the partner's copy of it in the other — no selects
                    // (as inline asm: hipcc 7.2 compiled `r = permlane16_swap(x, y); r[0] + r[1]` to `v_add_f32 x, x, x`)
                    float x0 = a[t], x1 = a[t + H], y0 = b[t], y1 = b[t + H];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
                    a[t] = x0 + x1;
                    b[t] = y0 + y1;
                } else {
                    const float sa = hi ? a[t] : a[t + H], sb = hi ? b[t] : b[t + H];
                    const float ka = hi ? a[t + H] : a[t], kb = hi ? b[t + H] : b[t];
                    a[t] = ka + lane_partner<O>(sa);
                    b[t] = kb + lane_partner<O>(sb);
                }
            }
            base += hi ? H : 0;
            LaneFold<H, O / 2>::run(a, b, lane, base, writer);
        } else {
            if constexpr (O == 16) {
                a[0] += __shfl_xor(a[0], O);
                b[0] += __shfl_xor(b[0], O);
            } else {
                a[0] += lane_partner<O>(a[0]);
                b[0] += lane_partner<O>(b[0]);
            }
            writer = writer && !hi;
            LaneFold<1, O / 2>::run(a, b, lane, base, writer);
        }
    }
};
template <int C>
struct LaneFold<C, 0> {
    static __device__ __forceinline__ void run(float*, float*, int, int&, bool&) {}
};

// ------------------------------------------------------------------------------------------------
// conv_gemm_nt
// ------------------------------------------------------------------------------------------------
// MODE 0 (NT_LINEAR) : A is a plain [M][Cs] matrix (1x1 / stride 1 / no padding).
// MODE 1 (NT_UNIFORM): every K-step lies inside ONE filter tap (chunks per tap % 8 == 0, true for every conv of the
//                      path except the stem): the tap and its (dy,dx) are wave-uniform scalars, the per-row work is
//                      two adds, two unsigned compares and a select — branch-free, so the compiler can schedule it
//                      between the MFMAs of the previous K-step instead of in front of them.
// MODE 2 (NT_GENERIC): a K-step may straddle taps (stem: 1-2 chunks per tap); per-thread tap decode from an LDS table.
enum { NT_LINEAR = 0, NT_UNIFORM = 1, NT_GENERIC = 2 };

// NW waves per workgroup: 4 = 2x2 waves, 8 = 4x2 waves (BM = 256: the B tile is shared by four wave rows — 25 % fewer
// operand bytes per FLOP out of L2 than 128x128 at the same waves, registers and LDS per CU as two 4-wave workgroups).
// Staging: LDS-DMA loads (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass; out-of-range offsets
// still return 0 = the conv padding).  A wave-instruction writes 1 KiB of LDS linearly (8 rows x 128 B in lane order), so
// the XOR swizzle of the LDS image is applied on the SOURCE side: the lane at slot s of row r fetches chunk s ^ key(r).
// Two LDS buffers: the DMA of step ks+1 is issued after the barrier of step ks and flies under that step's MFMAs
// (vmcnt(0) + one barrier per K-step).  Measured against the earlier register-staged pipeline (two register sets, prefetch
// distance 2): bf16 5-25 % faster per layer, fp32 equal (MFMA-bound).  STATS: epilogue with the BN-statistics partials.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Epilogue of conv_gemm_nt / conv_nt_sk_reduce: bias, store (or accumulate), optional BN-statistics partials.
// EPI: 0 = store, 1 = store + BN-statistics partials, 2 = accumulate (p.accumulate 1 / 2) — compile-time so that the plain
// and statistics instantiations do not carry the registers of the accumulate path's batched loads.
//
// Branch-free by construction (round 2; PMC showed the 1x1 layers of the 2-byte types VALU-issue bound in this code: ~870
// vector instructions per wave and tile, most of them exec-mask branches around single stores / loads, 64-bit address
// arithmetic and per-element conversions): every store and every epilogue load is a raw BUFFER access whose 32-bit byte
// offset carries "row past the end" / "column past Nn" in bit 31 (out of range => the store is dropped, the load returns
// 0), the bias is folded into the accumulators under ONE uniform branch, the statistics run on packed fp32 pairs
// (v_pk_add_f32 / v_pk_fma_f32), and the pixel of a row is its GEMM row unless the launch scatters a sub-grid (strided dgrad).
enum { NT_EPI_STORE = 0, NT_EPI_STATS = 1, NT_EPI_ACC = 2, NT_EPI_STATSC = 3, NT_EPI_BNRED = 4, NT_EPI_ACCRED = 5 };
// NT_EPI_ACCRED (round 4) = NT_EPI_ACC (p.accumulate == 2: the masked gradient of the NEXT residual unit's output is added) AND NT_EPI_BNRED
// in one epilogue: the launch writes the COMPLETE gradient of a residual unit's output y_b = relu(bn(x_b) + skip_b), so the backward sums of
// that unit's own output BN (sum dy', sum dy' * x_b with dy' = the stored gradient where y_b > 0) ride along and its reduction pass over
// (dy, x_b) disappears — two tensors in the output's layout are loaded: add_src / add_mask and red_x / red_mask.
// NT_EPI_STATSC (conv_gemm_nt_pers only): the per-lane statistics sums are CARRIED across the tiles a persistent workgroup walks
// (all its tiles share the channel block) and folded / written once per workgroup as a "counted" partial row
// [s1][s2][pivot][count] — one lane fold and one row per workgroup and wave row instead of one per tile.
template <typename T, int TN>
struct NtStatsCarry {
    static constexpr int NG = (int)(sizeof(typename MmaNT<T>::Acc) / 16);
    float s1[TN][NG][4], s2[TN][NG][4], piv[TN][NG][4];
    float count;            // pixel rows summed so far (wave-uniform)
    bool have;              // pivot taken
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
// offsets that carry "out of range" in bit 31 are added with saturation (v_add_u32 ... clamp): bad row + bad column must not
// wrap around to a valid address
__device__ __forceinline__ unsigned sat_add(unsigned a, unsigned b) { return __builtin_elementwise_add_sat(a, b); }

// rounded storage value(s) of an accumulator quadruple: T x 4 and the same values back in fp32 (what the statistics see)
template <typename T>
struct Quad;
template <>
struct Quad<float> {
    typedef f32x4 Bits;                                     // 16 bytes
    static __device__ __forceinline__ Bits pack(const float (&v)[4]) { return f32x4{v[0], v[1], v[2], v[3]}; }
    static __device__ __forceinline__ void unpack(const Bits& b, float (&r)[4]) { r[0] = b[0]; r[1] = b[1]; r[2] = b[2]; r[3] = b[3]; }
};
template <typename T>
struct Quad {                                               // bf16 / fp16: 8 bytes
    typedef u32x2 Bits;
    typedef T TV4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ Bits pack(const float (&v)[4]) {
        const TV4 o = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
        return __builtin_bit_cast(u32x2, o);
    }
    static __device__ __forceinline__ void unpack(const Bits& b, float (&r)[4]) {
        const TV4 o = __builtin_bit_cast(TV4, b);
        r[0] = to_f32(o[0]); r[1] = to_f32(o[1]); r[2] = to_f32(o[2]); r[3] = to_f32(o[3]);
    }
};

template <>
struct Quad<bf16_t> {                                       // bf16 -> fp32 is a 16-bit shift: two integer ops per packed pair
    typedef u32x2 Bits;
    typedef bf16_t TV4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ Bits pack(const float (&v)[4]) {
        const TV4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        return __builtin_bit_cast(u32x2, o);
    }
    static __device__ __forceinline__ void unpack(const Bits& b, float (&r)[4]) {
        r[0] = __builtin_bit_cast(float, b[0] << 16); r[1] = __builtin_bit_cast(float, b[0] & 0xffff0000u);
        r[2] = __builtin_bit_cast(float, b[1] << 16); r[3] = __builtin_bit_cast(float, b[1] & 0xffff0000u);
    }
};

template <typename T, int BM, int BN, bool TAPS, int NW, int EPI>
__device__ __forceinline__ void nt_epilogue(const GemmNTParams& p,
                                            typename MmaNT<T>::Acc (&acc)[BN / 2 / MmaNT<T>::MT][BM / (NW / 2) / MmaNT<T>::MT],
                                            const int m0, const int n0, const int lane, const int wm, const int wn,
                                            NtStatsCarry<T, BN / 2 / MmaNT<T>::MT>* carry = nullptr) {
    typedef MmaNT<T> MM;
    typedef Quad<T> Q;
    constexpr bool CARRY = EPI == NT_EPI_STATSC;
    constexpr bool BOTH = EPI == NT_EPI_ACCRED;                  // masked residual add AND BN-backward sums: two loaded tensors
    constexpr bool STATS = EPI == NT_EPI_STATS || CARRY, ACC = EPI == NT_EPI_ACC || BOTH;
    constexpr bool BNRED = EPI == NT_EPI_BNRED || BOTH;
    constexpr bool LD = ACC || BNRED;                          // the epilogue loads a tensor in the output's layout (+ a byte mask)
    constexpr int WROWS = NW / 2;
    constexpr int WTM = BM / WROWS, WTN = BN / 2;
    constexpr int TM = WTM / MM::MT, TN = WTN / MM::MT;
    constexpr int NG = (int)(sizeof(typename MM::Acc) / 16);   // groups of 4 channels per accumulator
    constexpr int ES = (int)sizeof(T);
    // 2-byte types / 16x16 accumulators: a lane holds 4 channels = 8 bytes of a pixel, and 8-byte stores run at 2/3 of the rate
    // of 16-byte ones (4.5 vs 6.4-6.8 TB/s measured for these row shapes).  Lanes l and l+16 hold neighbouring channel groups
    // of the same pixel: one v_permlane16_swap per dword trades row block 2q+1 of the even 16-lane rows for row block 2q of the
    // odd ones, after which every lane owns 8 consecutive channels = one 16-byte access (loads of the accumulate path alike).
    constexpr bool PAIR = ES == 2;
    static_assert(!PAIR || (MM::MT == 16 && TM % 2 == 0), "lane pairing needs 16x16 accumulators and an even row-block count");
    if (MCN_PROBES && (p.epi_flags & 32)) return;                              // DEBUG probe (MCN_NT_EPI_FLAGS=32): no epilogue at all — timing only
    const int fr = MM::frag_row(lane);
    // first of the 4 consecutive output channels that accumulator group (j, g) holds in this lane
    auto col = [&](int j, int g) -> int { return n0 + wn * WTN + j * MM::MT + (MM::MT == 16 ? 4 * (lane >> 4) : 8 * g + 4 * (lane >> 5)); };

    // (the bias is not added here: the accumulators START at it — nt_init_acc — so that a biased conv costs no epilogue code)

    // byte offset of each row block's pixel in the output tensor (< 2^31, mfma_path_ok), bit 31 = past the end of the GEMM rows
    const bool scat = p.osy != 1 || p.osx != 1 || p.OH != p.OHf || p.OW != p.OWf;      // sub-grid of a strided dgrad (uniform)
    unsigned rowoff[TM];
    const unsigned rowbytes = (unsigned)(p.ldo * ES);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mraw = m0 + wm * WTM + i * MM::MT + fr;
        const unsigned bad = ((unsigned)(p.m_end - 1 - mraw) >> 31) << 31;
        unsigned pix = bad ? 0u : (unsigned)mraw;
        if (scat) {
            const int hw = p.OH * p.OW;
            const int img = (int)pix / hw, rem = (int)pix - img * hw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            pix = (unsigned)((img * p.OHf + (oy * p.osy + p.oy0)) * p.OWf + (ox * p.osx + p.ox0));
        }
        rowoff[i] = pix * rowbytes | bad;
    }
    // column byte offsets (bit 31 = past Nn; Nn is a chunk multiple, so a 4-channel group is in or out as a whole)
    unsigned coloff[TN][NG];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int n = col(j, g);
            coloff[j][g] = (unsigned)(n * ES) | (((unsigned)(p.Nn - 1 - n) >> 31) << 31);
        }
    const bool oddrow = PAIR && ((lane >> 4) & 1);
    const unsigned oddmask = oddrow ? ~0u : 0u;
    const unsigned pairshift = oddrow ? 4u * ES : 0u;         // an odd 16-lane row stores / loads the chunk that starts 4 channels lower
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);

    // ---- accumulate modes: ALL loads of the epilogue are issued back to back in front of the first store (with LDS-DMA in the
    // kernel hipcc waits vmcnt(0) at the first use of every ordinary load, which also drains the stores in front of it) ----
    typename Q::Bits prev[LD ? TM : 1][LD ? TN : 1][LD ? NG : 1];
    unsigned char mbits[LD ? TM : 1][LD ? TN : 1][LD ? NG : 1];
    typename Q::Bits prev2[BOTH ? TM : 1][BOTH ? TN : 1][BOTH ? NG : 1];   // NT_EPI_ACCRED: the BN's input x and its ReLU bits
    unsigned char mbits2[BOTH ? TM : 1][BOTH ? TN : 1][BOTH ? NG : 1];
    if constexpr (LD) {
        constexpr int CEL = 16 / ES;                                    // elements per mask byte
        constexpr int MB = WTN / CEL;                                   // mask bytes per pixel for this wave's WTN channels: 4, 8 or 16
        static_assert(MB % 4 == 0, "mask bytes per wave row");
        // one tensor in the output's layout + its byte mask -> (pv, mbv)
        auto load_src = [&](const void* const lsrc, const unsigned char* const lmask, const bool masked, auto& pv, auto& mbv) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(lsrc), 0, (int)p.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(lmask), 0, masked ? (int)(p.out_bytes >> 4) : 0, 0x00020000);
        // one wide mask load per row block (instead of a byte per accumulator) when the wave's channel range is whole
        const bool wide = masked && p.Nn % WTN == 0 && (p.ldo / CEL) % MB == 0 && ((size_t)lmask & 15) == 0 && !(MCN_PROBES && (p.epi_flags & 2));
        unsigned mw[TM][MB / 4];
        const unsigned wcol = (unsigned)((n0 + wn * WTN) * ES) | (((unsigned)(p.Nn - 1 - (n0 + wn * WTN)) >> 31) << 31);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int k = 0; k < MB / 4; ++k) mw[i][k] = 0;
            if (wide) {                                                    // uniform
                const unsigned bo = sat_add(rowoff[i], wcol);                       // byte offset of the wave's first channel of this row (bit 31: invalid)
                const unsigned mo = ((bo & 0x7fffffffu) >> 4) | (bo & 0x80000000u);
                if constexpr (MB == 4) mw[i][0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsM, (int)mo, 0, 0);
                else if constexpr (MB == 8) { const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsM, (int)mo, 0, 0); mw[i][0] = t[0]; mw[i][1] = t[1]; }
                else { const i32x4 t = buf_load16(rsM, mo); mw[i][0] = t[0]; mw[i][1] = t[1]; mw[i][2] = t[2]; mw[i][3] = t[3]; }
            }
        }
        if constexpr (PAIR) {
            // 16-byte loads in the paired layout of the store loop, then the same swap hands every lane its own channels
#pragma unroll
            for (int q = 0; q < TM / 2; ++q) {
                const unsigned srow = rowoff[2 * q] ^ ((rowoff[2 * q] ^ rowoff[2 * q + 1]) & oddmask);   // (a select here compiles to a stack array + indexed load)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const i32x4 w = buf_load16_epi(rsS, sat_add(srow, coloff[j][0] - pairshift));
                    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)w[0], (unsigned)w[2], false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)w[1], (unsigned)w[3], false, false);
                    pv[2 * q][j][0] = u32x2{lo[0], hi[0]};
                    pv[2 * q + 1][j][0] = u32x2{lo[1], hi[1]};
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < NG; ++g) pv[i][j][g] = __builtin_bit_cast(typename Q::Bits, buf_load16_epi(rsS, sat_add(rowoff[i], coloff[j][g])));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (!masked) {
                        mbv[i][j][g] = 0xff;
                    } else if (wide) {
                        const int nl = (MM::MT == 16) ? (4 * (lane >> 4)) : (8 * g + 4 * (lane >> 5));
                        const int idx = (j * MM::MT + nl) / CEL;           // byte of this accumulator within the wave row's MB bytes
                        unsigned dw = mw[i][0];
#pragma unroll
                        for (int k = 1; k < MB / 4; ++k) dw = (idx >> 2) == k ? mw[i][k] : dw;
                        mbv[i][j][g] = (unsigned char)(dw >> ((idx & 3) * 8));
                    } else {
                        const unsigned bo = sat_add(rowoff[i], coloff[j][g]);
                        mbv[i][j][g] = (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsM, (int)(((bo & 0x7fffffffu) >> 4) | (bo & 0x80000000u)), 0, 0);
                    }
                }
        };
        if constexpr (BOTH) {
            load_src(p.add_src, p.add_mask, true, prev, mbits);
            load_src(p.red_x, p.red_mask, true, prev2, mbits2);
        } else {
            const bool masked = BNRED || p.accumulate == 2;
            load_src(BNRED ? p.red_x : (masked ? p.add_src : (const void*)p.out), BNRED ? p.red_mask : p.add_mask, masked, prev, mbits);
        }
    }

    // ---- BN statistics of the STORED (rounded) values: per-lane sums of (y - pivot) and of its square over the wave's row
    // blocks, pivot = pixel row 0 of the wave row (keeps the E[y^2] - E[y]^2 cancellation out of fp32) ----
    float ls1[TN][NG][4], ls2[TN][NG][4], lpiv[TN][NG][4];                  // (unused and eliminated unless EPI == NT_EPI_STATS)
    typedef float (&StatRef)[TN][NG][4];
    StatRef s1 = [&]() -> StatRef { if constexpr (CARRY) return carry->s1; else return ls1; }();
    StatRef s2 = [&]() -> StatRef { if constexpr (CARRY) return carry->s2; else return ls2; }();
    StatRef piv = [&]() -> StatRef { if constexpr (CARRY) return carry->piv; else return lpiv; }();
    if constexpr (STATS) {
        bool init = true;
        if constexpr (CARRY) init = !carry->have;                            // (uniform) the first tile of the workgroup sets the pivot
        if (init) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float v[4], r[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][0][g * 4 + e];
                    Q::unpack(Q::pack(v), r);                                 // the values as stored
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        piv[j][g][e] = MM::MT == 16 ? row_first(r[e]) : __shfl(r[e], lane & ~(MM::MT - 1));        // pixel row 0 of the wave row
                        s1[j][g][e] = s2[j][g][e] = 0.f;
                    }
                }
        }
        if constexpr (CARRY) {
            carry->have = true;
            const int left = p.m_end - (m0 + wm * WTM);                       // valid pixel rows of this wave row in this tile
            carry->count += (float)(left < 0 ? 0 : (left > WTM ? WTM : left));
        }
    }
    if constexpr (BNRED) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) s1[j][g][e] = s2[j][g][e] = 0.f;
    }
    const bool full = m0 + BM <= p.m_end;                      // wave-uniform: no row of this tile is past the end
    // the 4 output values of accumulator (i, j, g) as stored (residual added, rounded) + their statistics
    auto value = [&](int i, int j, int g, bool rowvalid) -> typename Q::Bits {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[j][i][g * 4 + e];
        if constexpr (ACC) {
            float pr[4];
            Q::unpack(prev[i][j][g], pr);
            // accumulate 1: every bit of mbits is set; 2: the residual's ReLU mask, one bit per element (8 per byte for 2-byte types)
            const unsigned mb = ES == 4 ? (unsigned)mbits[i][j][g] : ((unsigned)mbits[i][j][g] >> (col(j, g) & 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (mb >> e) & 1u ? pr[e] : 0.f;
        }
        const typename Q::Bits o = Q::pack(v);
        if constexpr (BNRED) {
            // BN-backward sums of the gradient as stored: dy' = dy where the forward's ReLU bit is set; rows past the end load x = 0 and
            // mask = 0 (out-of-range buffer offsets), so they add nothing
            float r[4], xv[4];
            Q::unpack(o, r);
            unsigned char mraw;
            if constexpr (BOTH) { Q::unpack(prev2[i][j][g], xv); mraw = mbits2[i][j][g]; }
            else { Q::unpack(prev[i][j][g], xv); mraw = mbits[i][j][g]; }
            const unsigned mb = ES == 4 ? (unsigned)mraw : ((unsigned)mraw >> (col(j, g) & 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 d = f32x2{(mb >> (2 * h)) & 1u ? r[2 * h] : 0.f, (mb >> (2 * h + 1)) & 1u ? r[2 * h + 1] : 0.f};
                const f32x2 a = f32x2{s1[j][g][2 * h], s1[j][g][2 * h + 1]} + d;
                const f32x2 b = __builtin_elementwise_fma(d, f32x2{xv[2 * h], xv[2 * h + 1]}, f32x2{s2[j][g][2 * h], s2[j][g][2 * h + 1]});
                s1[j][g][2 * h] = a[0]; s1[j][g][2 * h + 1] = a[1];
                s2[j][g][2 * h] = b[0]; s2[j][g][2 * h + 1] = b[1];
            }
        }
        if constexpr (STATS) {
            float r[4];
            Q::unpack(o, r);
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                     // packed pairs: v_pk_add_f32 / v_pk_fma_f32
                f32x2 d = f32x2{r[2 * h], r[2 * h + 1]} - f32x2{piv[j][g][2 * h], piv[j][g][2 * h + 1]};
                if (!rowvalid) d = f32x2{0.f, 0.f};
                const f32x2 a = f32x2{s1[j][g][2 * h], s1[j][g][2 * h + 1]} + d;
                const f32x2 b = __builtin_elementwise_fma(d, d, f32x2{s2[j][g][2 * h], s2[j][g][2 * h + 1]});
                s1[j][g][2 * h] = a[0]; s1[j][g][2 * h + 1] = a[1];
                s2[j][g][2 * h] = b[0]; s2[j][g][2 * h + 1] = b[1];
            }
        }
        return o;
    };
    auto store_all = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;                         // FULL: every row is valid (no per-row statistics select)
        if constexpr (PAIR) {
#pragma unroll
            for (int q = 0; q < TM / 2; ++q) {
                const unsigned srow = rowoff[2 * q] ^ ((rowoff[2 * q] ^ rowoff[2 * q + 1]) & oddmask);   // (a select here compiles to a stack array + indexed load)
                const bool v0 = FULL || !(rowoff[2 * q] >> 31), v1 = FULL || !(rowoff[2 * q + 1] >> 31);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const u32x2 a = value(2 * q, j, 0, v0);
                    const u32x2 b = value(2 * q + 1, j, 0, v1);
                    const auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    // even rows: (own, partner) of row block 2q; odd rows: (partner, own) of 2q+1 — lower channels first
                    const i32x4 w = {(int)lo[0], (int)hi[0], (int)lo[1], (int)hi[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(w, rsO, (int)sat_add(srow, coloff[j][0] - pairshift), 0, 0);
                    // (without branches the scheduler would hoist every accumulator read and conversion of the tile in front of
                    // the first store: +60-110 VGPRs, an occupancy step; one store group at a time keeps the old footprint)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bool rv = FULL || !(rowoff[i] >> 31);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                    {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, value(i, j, g, rv)), rsO, (int)sat_add(rowoff[i], coloff[j][g]), 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        }
    };
    if (!STATS || full) store_all(std::true_type{});
    else store_all(std::false_type{});

    if constexpr (BNRED) {
        constexpr int V = TN * NG * 4;
        float a[V], b[V];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[(j * NG + g) * 4 + e] = s1[j][g][e];
                    b[(j * NG + g) * 4 + e] = s2[j][g][e];
                }
        int base = 0;
        bool writer = true;
        LaneFold<V, MM::MT / 2>::run(a, b, lane, base, writer);
        const int prow = p.red_row0 + ((m0 - p.m_begin) / BM) * WROWS + wm;
        const int e = base & 3, gj = base >> 2, g = gj % NG, j = gj / NG;
        const int nl = (MM::MT == 16) ? (4 * (lane >> 4)) : (8 * g + 4 * (lane >> 5));
        const int n = n0 + wn * WTN + j * MM::MT + nl + e;
        if (writer && n < p.Nn) {
            p.red_part[((long)prow * 2 + 0) * p.Nn + n] = a[0];
            p.red_part[((long)prow * 2 + 1) * p.Nn + n] = b[0];
        }
    }
    if constexpr (STATS && !CARRY) {
        // Fold over the MT lanes that hold different pixel rows of the same channels with a halving butterfly: at each
        // step a lane keeps half of its values and receives the partner's copies of that half (V/2 + V/4 + ... shuffles
        // instead of V per step); when one value is left the remaining steps are plain all-reduce steps.  At the end lane
        // (lane & (MT-1)) owns the total of flat value index `base` (j, g, e order).
        constexpr int V = TN * NG * 4;
        float a[V], b[V];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[(j * NG + g) * 4 + e] = s1[j][g][e];
                    b[(j * NG + g) * 4 + e] = s2[j][g][e];
                }
        int base = 0;
        bool writer = true;
        LaneFold<V, MM::MT / 2>::run(a, b, lane, base, writer);
        const int prow = (m0 / BM) * WROWS + wm;                     // absolute: body and stream-K tail launches share the buffer
        const int e = base & 3, gj = base >> 2, g = gj % NG, j = gj / NG;
        const int nl = (MM::MT == 16) ? (4 * (lane >> 4)) : (8 * g + 4 * (lane >> 5));
        const int n = n0 + wn * WTN + j * MM::MT + nl + e;
        if (writer && n < p.Nn) {
            p.stats[((long)prow * 3 + 0) * p.Nn + n] = a[0];
            p.stats[((long)prow * 3 + 1) * p.Nn + n] = b[0];
        }
        if (fr == 0) {                                                          // the pivots: 4 consecutive channels per (j, g)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    const int n2 = col(jj, gg);
                    if (n2 < p.Nn)
                        *reinterpret_cast<f32x4*>(p.stats + ((long)prow * 3 + 2) * p.Nn + n2) = f32x4{piv[jj][gg][0], piv[jj][gg][1], piv[jj][gg][2], piv[jj][gg][3]};
                }
        }
    }
}

// end of a persistent workgroup with carried statistics: fold the lanes once, merge the wave rows through LDS and write ONE counted
// partial row per workgroup: sum(y - p), sum((y - p)^2), p, pixel rows summed (layout: "COMPACT counted rows" below); prow = workgroup.
// Wave row w > 0 summed around its own pivot p_w: with d = p_w - p_0, sum(y - p_0) = s1_w + n_w d and
// sum((y - p_0)^2) = s2_w + 2 d s1_w + n_w d^2 (d is of the order of the standard deviation: no cancellation added).
template <typename T, int BM, int BN, int NW>
__device__ __forceinline__ void nt_stats_flush(const GemmNTParams& p, NtStatsCarry<T, BN / 2 / MmaNT<T>::MT>& st, const int n0, const int lane,
                                               const int wm, const int wn, const int prow, char* smem) {
    typedef MmaNT<T> MM;
    constexpr int WROWS = NW / 2, WTN = BN / 2, TN = WTN / MM::MT;
    constexpr int NG = (int)(sizeof(typename MM::Acc) / 16);
    constexpr int V = TN * NG * 4;
    const int fr = MM::frag_row(lane);
    float a[V], b[V];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[(j * NG + g) * 4 + e] = st.s1[j][g][e];
                b[(j * NG + g) * 4 + e] = st.s2[j][g][e];
            }
    int base = 0;
    bool writer = true;
    LaneFold<V, MM::MT / 2>::run(a, b, lane, base, writer);
    // the pivot of the channel whose totals this lane ended up with (flat index `base` in (j, g, e) order)
    float pv = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) pv = ((j * NG + g) * 4 + e) == base ? st.piv[j][g][e] : pv;
    float* const sh = reinterpret_cast<float*>(smem);                  // [WROWS - 1][2][64][4] floats: s1, s2, pivot, count
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has left the K loop's LDS tiles
    if (wm > 0) {
        float* d = sh + ((((wm - 1) * 2 + wn) * 64 + lane) << 2);
        d[0] = a[0]; d[1] = b[0]; d[2] = pv; d[3] = st.have ? st.count : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wm != 0) return;
    float count = st.count;
#pragma unroll
    for (int w = 1; w < WROWS; ++w) {
        const float* d = sh + ((((w - 1) * 2 + wn) * 64 + lane) << 2);
        const float n1 = d[3];
        if (n1 > 0.f) {                                                // (uniform per wave: a wave row without pixel rows has no pivot)
            const float dp = d[2] - pv;
            a[0] += d[0] + n1 * dp;
            b[0] += d[1] + 2.f * dp * d[0] + n1 * dp * dp;
            count += n1;
        }
    }
    // COMPACT counted rows (round 3): the row is keyed by the workgroup's channel block — stats[((nb * RPB + j) * 4 + plane) * BW + cl], nb = n0 / BN,
    // BW = min(BN, Nn) channels per block (a layer narrower than the tile is one block of Nn channels: the buffer holds rows x 4 x Nn floats),
    // j = this workgroup's rank among the workgroups of its block (all of them walk tiles of one channel block: nt_stats_counted), RPB rows
    // per block.  With grid < total tiles the workgroups b = xcd + 8 idx start at tile base(xcd) + idx and the host guarantees
    // (grid / 8) % ntn == 0: per XCD, idx / ntn counts the ones that share a block.  Otherwise every workgroup has exactly one tile L and
    // j is its M tile.  (Before: rows of Nn channels with zero counts for the other blocks, up to 3.9 MB of zeros per launch and
    // 512-1024 rows scanned per channel by the finalize.)
    const int ntn = (p.Nn + BN - 1) / BN;
    const int total = ((p.m_end - p.m_begin + BM - 1) / BM) * ntn, grid = (int)gridDim.x;
    int jrow, RPB;
    if (grid < total) {
        const int per = grid / 8 / ntn;
        jrow = (prow & 7) * per + (prow >> 3) / ntn;
        RPB = grid / ntn;
    } else {
        jrow = xcd_remap(prow, total) / ntn;
        RPB = total / ntn;
    }
    const int BW = p.Nn < BN ? p.Nn : BN;
    float* const row = p.stats + ((size_t)(n0 / BN) * RPB + jrow) * 4 * BW;
    const int e = base & 3, gj = base >> 2, g = gj % NG, j = gj / NG;
    const int nl = (MM::MT == 16) ? (4 * (lane >> 4)) : (8 * g + 4 * (lane >> 5));
    const int cl = wn * WTN + j * MM::MT + nl + e;                    // channel within the block
    if (writer && n0 + cl < p.Nn) {
        row[0 * BW + cl] = a[0];
        row[1 * BW + cl] = b[0];
    }
    if (fr == 0) {
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                const int c2 = wn * WTN + jj * MM::MT + (MM::MT == 16 ? 4 * (lane >> 4) : 8 * gg + 4 * (lane >> 5));
                if (n0 + c2 < p.Nn) {
                    *reinterpret_cast<f32x4*>(row + 2 * BW + c2) = f32x4{st.piv[jj][gg][0], st.piv[jj][gg][1], st.piv[jj][gg][2], st.piv[jj][gg][3]};
                    *reinterpret_cast<f32x4*>(row + 3 * BW + c2) = f32x4{count, count, count, count};
                }
            }
    }
}

// Accumulators start at the bias of their output channels (or at zero): the lane holds 4 consecutive channels per (j, g) for
// every row block i, so a biased convolution (VGG, the fc layer) costs one uniform branch here and nothing in the epilogue.
template <typename T, int TN, int TM>
__device__ __forceinline__ void nt_init_acc(typename MmaNT<T>::Acc (&acc)[TN][TM], const float* bias, int ncol0, int Nn, int lane) {
    typedef MmaNT<T> MM;
    constexpr int NG = (int)(sizeof(typename MM::Acc) / 16);
    if (bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int n = ncol0 + j * MM::MT + (MM::MT == 16 ? 4 * (lane >> 4) : 8 * g + 4 * (lane >> 5));
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + min(n, Nn - 4));       // (clamped: columns past Nn are never stored)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][i][g * 4 + e] = bv[e];
            }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < (int)(sizeof(typename MM::Acc) / 4); ++e) acc[j][i][e] = 0.f;
    }
}

// minimum waves per SIMD the register allocator must leave room for: the 8-wave 128x128 tile exists to put two workgroups
// (4 waves per SIMD) on a CU; everything else takes what it gets
template <typename T, int BM, int BN, int MODE, int NW = 4, int EPI = NT_EPI_STORE>
// (2-byte 128 x 128 tiles: two workgroups per CU by LDS — keep the registers there too; the fp16 statistics epilogue had drifted to 304)
__global__ __launch_bounds__(NW * 64, (NW == 8 && BM == 128 ? 4 : (sizeof(T) == 2 && BM == 128 && BN == 128 ? 2 : 1))) void conv_gemm_nt(const GemmNTParams p) {
    typedef MmaNT<T> MM;
    constexpr int CE = VecTraits<T>::CE;
    constexpr bool TAPS = MODE != NT_LINEAR;
    constexpr int NT = NW * 64;                        // threads
    constexpr int RPP = NT / 8;                        // tile rows covered by one staging pass (8 chunks per row)
    constexpr int WROWS = NW / 2;                      // wave rows (x 2 wave columns)
    constexpr int AR = BM / RPP, BR = BN / RPP;        // staged chunks per thread
    constexpr int WTM = BM / WROWS, WTN = BN / 2;      // wave tile
    constexpr int TM = WTM / MM::MT, TN = WTN / MM::MT;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ signed char s_tdy[MCN_MAX_TAPS], s_tdx[MCN_MAX_TAPS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.Nn + BN - 1) / BN;
    const int ntm = (p.m_end - p.m_begin + BM - 1) / BM;
    // Stream-K tail (sk_mode 1): the first sk_body workgroups compute whole tiles, the rest compute one K-slice each of the
    // tiles behind them (sk_rel = tile * sk_slices + slice) and park their accumulators for conv_nt_sk_reduce.
    const int sk_rel = (int)blockIdx.x - p.sk_body;
    const bool sk_slice = p.sk_mode == 1 && sk_rel >= 0;
    const int L = sk_slice ? p.sk_body + sk_rel / p.sk_slices : xcd_remap(blockIdx.x, p.sk_mode == 1 ? p.sk_body : ntm * ntn);
    const int m0 = p.m_begin + (L / ntn) * BM, n0 = (L % ntn) * BN;

    if (MODE == NT_GENERIC) {
        if (tid < MCN_MAX_TAPS) {
            s_tdy[tid] = (signed char)(short)(p.tap[tid] & 0xffff);
            s_tdx[tid] = (signed char)(p.tap[tid] >> 16);
        }
        __syncthreads();
    }

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, (int)p.wt_bytes, 0x00020000);

    const int crow = tid >> 3;
    const int cid = (tid & 7) ^ ((crow >> 1) & 7);     // K chunk this thread fetches: the swizzle sits on the source side
    // per-thread A rows: byte offset of the row's pixel at tap offset (0,0) + its grid coordinates for the bounds test
    unsigned a_off[AR];
    int a_y[AR], a_x[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + crow + RPP * i;
        const bool mv = m < p.m_end;
        if (TAPS) {
            const int mm = mv ? m : 0;
            const int hw = p.OH * p.OW;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            a_y[i] = mv ? oy * p.sy : -0x40000000;       // invalid rows fail the unsigned bounds test for every tap
            a_x[i] = ox * p.sx;
            a_off[i] = (unsigned)((img * p.IH + oy * p.sy) * p.IW + ox * p.sx) * (unsigned)p.Cs * (unsigned)sizeof(T) + (unsigned)cid * 16u;
        } else {
            a_y[i] = 0;
            a_x[i] = 0;
            a_off[i] = mv ? (unsigned)m * (unsigned)p.Cs * (unsigned)sizeof(T) + (unsigned)cid * 16u : MCN_OOB;
        }
    }
    unsigned b_off[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + crow + RPP * i;
        b_off[i] = n < p.Nn ? (unsigned)n * (unsigned)p.nchunks * 16u + (unsigned)cid * 16u : MCN_OOB;
    }
    const int pix_bytes = p.Cs * (int)sizeof(T);
    const int kpt = p.cpt >> 3;                          // K-steps per tap (NT_UNIFORM)

    // LDS-DMA destination of this wave for staging pass i: wave-uniform base (M0), the hardware adds lane * 16
    // (wave base + compile-time constant: distinct constants let the compiler see that the DMAs of different buffers do
    // not overlap — with an opaque address it orders every new DMA behind all outstanding ones with vmcnt waits)
    __attribute__((address_space(3))) char* const wbase =
        (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024);
    auto dma = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, auto ldsc) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wbase + decltype(ldsc)::value), 16, (int)off, 0, 0, 0);
    };
    auto dma_act = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, auto ldsc) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wbase + decltype(ldsc)::value), 16, (int)off, 0, 0, MCN_DMA_AUX_A);
    };
    // setc: LDS buffer index
    auto issue = [&](int ks, auto setc) {
        constexpr int S = decltype(setc)::value;
        auto ldA = [&](auto ic, unsigned off) {
            dma_act(rsA, off, std::integral_constant<int, S * TILE_BYTES + decltype(ic)::value * RPP * 128>{});
        };
        auto ldB = [&](auto ic, unsigned off) {
            dma(rsB, off, std::integral_constant<int, S * TILE_BYTES + BM * 128 + decltype(ic)::value * RPP * 128>{});
        };
        if (MODE == NT_UNIFORM) {
            const int tap = ks / kpt;                    // wave-uniform (scalar ALU)
            const int cb = ks - tap * kpt;
            const int tw = p.tap[tap];
            const int dy = (short)(tw & 0xffff), dx = tw >> 16;
            const unsigned toff = (unsigned)((dy * p.IW + dx) * pix_bytes + cb * 128);
            if (!(MCN_PROBES && (p.epi_flags & 8) && tap != 0))      // DEBUG probe (MCN_NT_EPI_FLAGS=8): A staged for the first tap only — wrong results, timing only
            static_for<AR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const bool ok = (unsigned)(a_y[i] + dy) < (unsigned)p.IH && (unsigned)(a_x[i] + dx) < (unsigned)p.IW;
                ldA(ic, ok ? a_off[i] + toff : MCN_OOB);
            });
        } else if (MODE == NT_LINEAR) {
            // K tail as arithmetic (bit 31 = out of range), not as a select: hipcc turned `kv ? off : OOB` into two DMA
            // instructions under complementary exec masks behind a branch — 12 DMAs and 11 branches per K-step instead of 8 and 0
            const unsigned oob = ((unsigned)(p.nchunks - 1 - (ks * 8 + cid)) >> 31) << 31;
            if (!(MCN_PROBES && (p.epi_flags & 8) && ks != 0))
            static_for<AR>([&](auto ic) { ldA(ic, (a_off[decltype(ic)::value] + (unsigned)ks * 128u) | oob); });
        } else {
            const int j = ks * 8 + cid;
            const bool kv = j < p.nchunks;
            const int tap = kv ? j / p.cpt : 0;
            const int cc = j - tap * p.cpt;
            const int dy = s_tdy[tap], dx = s_tdx[tap];
            const unsigned toff = (unsigned)((dy * p.IW + dx) * pix_bytes + (cc - cid) * 16);
            const unsigned kbad = (unsigned)(p.nchunks - 1 - j);              // sign bit: past the end of K
            static_for<AR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int yy = a_y[i] + dy, xx = a_x[i] + dx;
                const unsigned bad = kbad | (unsigned)yy | (unsigned)(p.IH - 1 - yy) | (unsigned)xx | (unsigned)(p.IW - 1 - xx);
                ldA(ic, (a_off[i] + toff) | ((bad >> 31) << 31));                // (arithmetic, not a select: see NT_LINEAR)
            });
        }
        const unsigned oobb = MODE == NT_UNIFORM ? 0u : ((unsigned)(p.nchunks - 1 - (ks * 8 + cid)) >> 31) << 31;
        if (!(MCN_PROBES && (p.epi_flags & 16) && ks % 9 != 0))      // DEBUG probe (MCN_NT_EPI_FLAGS=16): B staged every 9th K-step only
        static_for<BR>([&](auto ic) { ldB(ic, (b_off[decltype(ic)::value] + (unsigned)ks * 128u) | oobb); });   // OOB + small stays OOB
    };
    typename MM::Acc acc[TN][TM];
    nt_init_acc<T, TN, TM>(acc, sk_slice ? nullptr : p.bias, n0 + wn * WTN, p.Nn, lane);     // (K-slices of a stream-K tail: the reduce pass adds the bias)

    const int fr = MM::frag_row(lane), fc = MM::frag_chunk(lane);
    const int fsw = (fr >> 1) & 7;   // tile rows start at multiples of 16 => swizzle key depends on the lane only
    const int a_rd = (wm * WTM + fr) * 128;
    const int b_rd = BM * 128 + (wn * WTN + fr) * 128;

    // Fragment registers are double-buffered and the K-loop is skewed by one slab: the ds_reads of slab s+1 are in
    // flight under the MFMAs of slab s, and the first slab of the NEXT K-step is read right after the barrier, under
    // the MFMAs of the current K-step's last slab.  (With one register set the reads can only issue after the last
    // MFMA that sources them, exposing the LDS latency once per slab: -27 % on the fp32 loop, worse on bf16.)
    static_assert(MM::SLABS % 2 == 0, "slab skew needs an even slab count");
    typename MM::Frag xa[2][TM], wb[2][TN];
    auto load_frags = [&](int set, int sl, const char* base) {
        const int coff = (((sl * MM::CPS + fc) ^ fsw) << 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[set][i] = *reinterpret_cast<const typename MM::Frag*>(base + a_rd + i * MM::MT * 128 + coff);
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[set][j] = *reinterpret_cast<const typename MM::Frag*>(base + b_rd + j * MM::MT * 128 + coff);
    };
    auto mma_set = [&](int set) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) MM::mma(acc[j][i], wb[set][j], xa[set][i]);
    };

    const int nk_all = (p.nchunks + 7) >> 3;
    // K-step range of this workgroup: everything, or one slice of a stream-K tail tile (sk_mode 1), or nothing (sk_mode 2)
    int ks0 = 0, nk = nk_all;
    if (sk_slice) {
        const int per = (nk_all + p.sk_slices - 1) / p.sk_slices, sl = sk_rel % p.sk_slices;
        ks0 = sl * per;
        nk = min(nk_all, ks0 + per);
    }
    constexpr int ACCSZ = (int)(sizeof(typename MM::Acc) / 4);
    constexpr int NREG = TN * TM * ACCSZ;                       // accumulator registers per thread
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    if (ks0 < nk) {
        // LDS-DMA pipeline: buffer `cur` holds step ks (its DMA was issued one step ago), the DMA of step ks+1 into the
        // other buffer is issued right after the barrier that retires that buffer's readers and flies under this
        // step's MFMAs.  vmcnt(0) + barrier per step: own DMA landed, then everybody's.
        auto compute = [&](const char* base) {
            load_frags(0, 0, base);
#pragma unroll
            for (int s = 0; s < MM::SLABS; ++s) {
                if (s + 1 < MM::SLABS) load_frags((s + 1) & 1, s + 1, base);
                mma_set(s & 1);
            }
        };
        // (A three-buffer variant that keeps the DMA of step ks+2 in flight across the barrier — counted vmcnt, raw
        // s_barrier — was measured and is not faster at these tile sizes: with two workgroups per CU the other
        // workgroup already covers the wait.  It needs every other load out of the loop: see GemmNTParams::tap.)
        // a K loop of exactly two steps fits the two buffers: both DMAs go out at once (one exposed round trip instead of two)
        // (fp32 only: in the bf16 instantiations the third inlined copy of the staging code costs an occupancy step, -9 %)
        const bool both = sizeof(T) == 4 && nk - ks0 == 2;
        auto gstep = [&](int ks, auto cur, auto nxt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ks + 1 < nk && !both) issue(ks + 1, nxt);
            compute(smem + decltype(cur)::value * TILE_BYTES);
        };
        issue(ks0, S0{});
        if (both) issue(ks0 + 1, S1{});
        for (int ks = ks0; ks < nk; ks += 2) {
            gstep(ks, S0{}, S1{});
            if (ks + 1 < nk) gstep(ks + 1, S1{}, S0{});
        }
    }
    if (sk_slice) {
        // stream-K slice: park the accumulators ([tile][slice][register][thread]: coalesced) — the reduce pass runs the epilogue
        float* dst = p.partial + ((size_t)sk_rel * NREG) * NT + tid;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < ACCSZ; ++e) dst[(size_t)((j * TM + i) * ACCSZ + e) * NT] = acc[j][i][e];
        return;
    }
    nt_epilogue<T, BM, BN, TAPS, NW, EPI>(p, acc, m0, n0, lane, wm, wn);
}

// Stream-K reduce: tail tile blockIdx.x = the sum of its K-slices in a fixed order (deterministic), then the ordinary epilogue.
template <typename T, int BM, int BN, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void conv_nt_sk_reduce(const GemmNTParams p) {
    typedef MmaNT<T> MM;
    constexpr int NT = NW * 64;
    constexpr int WROWS = NW / 2;
    constexpr int TM = BM / WROWS / MM::MT, TN = BN / 2 / MM::MT;
    constexpr int ACCSZ = (int)(sizeof(typename MM::Acc) / 4);
    constexpr int NREG = TN * TM * ACCSZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = (p.Nn + BN - 1) / BN;
    const int L = p.sk_body + (int)blockIdx.x;
    const int m0 = p.m_begin + (L / ntn) * BM, n0 = (L % ntn) * BN;
    typename MM::Acc acc[TN][TM];
    nt_init_acc<T, TN, TM>(acc, p.bias, n0 + (wave & 1) * (BN / 2), p.Nn, lane);
    for (int sl = 0; sl < p.sk_slices; ++sl) {
        const float* src = p.partial + (((size_t)blockIdx.x * p.sk_slices + sl) * NREG) * NT + tid;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < ACCSZ; ++e) acc[j][i][e] += src[(size_t)((j * TM + i) * ACCSZ + e) * NT];
    }
    nt_epilogue<T, BM, BN, true, NW, EPI>(p, acc, m0, n0, lane, wave >> 1, wave & 1);
}

// ------------------------------------------------------------------------------------------------
// conv_gemm_nt_pers: the NT_LINEAR conv_gemm_nt (1x1 / stride 1: A is a plain [M][Cs] matrix) as a PERSISTENT kernel.
//
// Probes on the 1x1 layers of ResNet-50 (round 2, bf16, B = 256: `MCN_NT_EPI_FLAGS=32` = no epilogue, `=24` = operands staged
// for the first K-step only) showed that their time is the SUM of the K loop and the epilogue, not the larger of the two:
// 56x56 64 -> 256 forward 157 us = 47 (loads + MFMA) + 110 (stores + statistics), 28x28 128 -> 512 95 = 30 + 65, 14x14 1024 -> 256
// dgrad + residual 73 = 24 + 49 — a workgroup loads, multiplies, stores and retires, and the next one on that slot starts with an
// address prologue and a full DMA round trip before its first MFMA, while HBM idles through every K loop and the MFMAs through
// every epilogue.  Here a workgroup walks tiles v = blockIdx.x, + gridDim.x, ... (grid = resident slots; xcd_remap keeps an
// XCD on a contiguous tile range) and issues the first K-step of tile i+1 into the free LDS buffer right after the barrier
// of the LAST K-step of tile i: that DMA flies under the last MFMAs and the whole epilogue of tile i.
//   * vmcnt counts loads, stores and LDS-DMAs together, in issue order: the first wait of tile i+1 is `vmcnt(NSTORE)`,
//     NSTORE = the store instructions the epilogue always issues — the prefetched DMAs are older than those and have landed,
//     the stores themselves stay in flight.  (Statistics partials are extra, conditional stores: they only make the wait more
//     conservative.)  Raw s_barrier: a __syncthreads() fence would drain the counter.
//   * the two LDS buffers alternate over the flattened (tile, K-step) sequence; buffer indices are compile-time constants
//     (DMA aliasing, see conv_gemm_nt), so the tile body exists for both starting parities and a K loop with an odd number
//     of steps alternates between them.
// Not for: a bias (its load would wait vmcnt(0) behind the previous tile's stores), stream-K launches.
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int NW, int EPI>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : (BM == 128 && BN == 128 ? 2 : (BM == 128 ? 3 : 4)))) void conv_gemm_nt_pers(const GemmNTParams p) {
    typedef MmaNT<T> MM;
    constexpr int NT = NW * 64, RPP = NT / 8, WROWS = NW / 2;
    constexpr int AR = BM / RPP, BR = BN / RPP;
    constexpr int WTM = BM / WROWS, WTN = BN / 2;
    constexpr int TM = WTM / MM::MT, TN = WTN / MM::MT;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    constexpr int NG = (int)(sizeof(typename MM::Acc) / 16);
    constexpr int NSTORE = sizeof(T) == 2 ? (TM / 2) * TN : TM * TN * NG;      // nt_epilogue's unconditional stores per wave
    static_assert(NSTORE >= 1 && NSTORE < 48, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.Nn + BN - 1) / BN;
    const int total = ((p.m_end - p.m_begin + BM - 1) / BM) * ntn;
    const int nk = (p.nchunks + 7) >> 3;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, (int)p.wt_bytes, 0x00020000);
    const int crow = tid >> 3;
    const int cid = (tid & 7) ^ ((crow >> 1) & 7);
    const unsigned rowb = (unsigned)p.Cs * (unsigned)sizeof(T);

    int v = blockIdx.x, m0 = 0, n0 = 0;                // (grid <= total)
    unsigned a_off[AR], b_off[BR];
    auto set_tile = [&](int vv) {
        const int L = xcd_remap(vv, total);
        m0 = p.m_begin + (L / ntn) * BM;
        n0 = (L % ntn) * BN;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = m0 + crow + RPP * i;
            a_off[i] = ((unsigned)m * rowb + (unsigned)cid * 16u) | (((unsigned)(p.m_end - 1 - m) >> 31) << 31);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + crow + RPP * i;
            b_off[i] = ((unsigned)n * (unsigned)p.nchunks * 16u + (unsigned)cid * 16u) | (((unsigned)(p.Nn - 1 - n) >> 31) << 31);
        }
    };
    __attribute__((address_space(3))) char* const wbase =
        (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024);
    auto issue = [&](int ks, auto setc) {
        constexpr int S = decltype(setc)::value;
        const unsigned oob = ((unsigned)(p.nchunks - 1 - (ks * 8 + cid)) >> 31) << 31;      // K tail (bit 31 = out of range: zero fill)
        static_for<AR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(wbase + (S * TILE_BYTES + i * RPP * 128)), 16,
                                                     (int)((a_off[i] + (unsigned)ks * 128u) | oob), 0, 0, MCN_DMA_AUX_A);
        });
        static_for<BR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(wbase + (S * TILE_BYTES + BM * 128 + i * RPP * 128)), 16,
                                                     (int)((b_off[i] + (unsigned)ks * 128u) | oob), 0, 0, 0);
        });
    };

    const int fr = MM::frag_row(lane), fc = MM::frag_chunk(lane);
    const int fsw = (fr >> 1) & 7;
    const int a_rd = (wm * WTM + fr) * 128;
    const int b_rd = BM * 128 + (wn * WTN + fr) * 128;
    static_assert(MM::SLABS % 2 == 0, "slab skew needs an even slab count");

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    bool first = true;
    NtStatsCarry<T, TN> carry;                          // (NT_EPI_STATSC; otherwise unused)
    carry.count = 0.f;
    carry.have = false;
    // one tile whose first K-step sits (or is on its way) in buffer P; returns true after the workgroup's last tile
    auto body = [&](auto parc) -> bool {
        constexpr int P = decltype(parc)::value;
        typedef std::integral_constant<int, P> C0;
        typedef std::integral_constant<int, P ^ 1> C1;
        typename MM::Acc acc[TN][TM];
        nt_init_acc<T, TN, TM>(acc, p.bias, n0 + wn * WTN, p.Nn, lane);       // (a bias costs a vmcnt(0) wait behind the previous tile's stores: the host avoids it)
        typename MM::Frag xa[2][TM], wb[2][TN];
        auto load_frags = [&](int set, int sl, const char* base) {
            const int coff = (((sl * MM::CPS + fc) ^ fsw) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[set][i] = *reinterpret_cast<const typename MM::Frag*>(base + a_rd + i * MM::MT * 128 + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) wb[set][j] = *reinterpret_cast<const typename MM::Frag*>(base + b_rd + j * MM::MT * 128 + coff);
        };
        auto compute = [&](const char* base) {
            load_frags(0, 0, base);
#pragma unroll
            for (int s = 0; s < MM::SLABS; ++s) {
                if (s + 1 < MM::SLABS) load_frags((s + 1) & 1, s + 1, base);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) MM::mma(acc[j][i], wb[s & 1][j], xa[s & 1][i]);
            }
        };
        const int em0 = __builtin_amdgcn_readfirstlane(m0), en0 = __builtin_amdgcn_readfirstlane(n0);   // this tile (set_tile moves m0 / n0 on to the next one inside the K loop)
        const int vnext = v + (int)gridDim.x;
        const bool more = vnext < total;
        auto gstep = [&](int ks, auto cur, auto nxt) {
            if (ks == 0 && !first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");      // the previous tile's stores stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (ks + 1 < nk) issue(ks + 1, nxt);
            else if (more) {
                set_tile(vnext);
                issue(0, nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
            compute(smem + decltype(cur)::value * TILE_BYTES);
        };
        for (int ks = 0; ks < nk; ks += 2) {
            gstep(ks, C0{}, C1{});
            if (ks + 1 < nk) gstep(ks + 1, C1{}, C0{});
        }
        __builtin_amdgcn_sched_barrier(0);              // the prefetch stays in front of the epilogue's stores
        nt_epilogue<T, BM, BN, false, NW, EPI>(p, acc, em0, en0, lane, wm, wn, &carry);
        __builtin_amdgcn_sched_barrier(0);
        first = false;
        v = vnext;
        if (more) set_tile(v);                          // (recomputed: the operand offsets are not kept alive across the epilogue)
        return !more;
    };
    set_tile(v);
    issue(0, S0{});
    const int fn0 = n0;                                 // (NT_EPI_STATSC: every tile of this workgroup has this channel block — the host checks)
    for (;;) {
        if (body(S0{})) break;
        if (nk & 1) {
            if (body(S1{})) break;
        }
    }
    if constexpr (EPI == NT_EPI_STATSC) nt_stats_flush<T, BM, BN, NW>(p, carry, fn0, lane, wm, wn, (int)blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------------
// conv_gemm_nt_win: conv_gemm_nt for stride-1 tap convolutions whose output grid is the input grid (3x3 SAME forward, its dgrad,
// dilated 3x3) with the INPUT WINDOW of the tile kept in LDS across the taps.
//
// GEMM row m is output pixel m of the flattened (image, y, x) grid and tap t reads input pixel m + doff_t, doff_t = dy_t*IW +
// dx_t: all taps of a BM-row tile read the contiguous pixel range [m0 + dmin, m0 + BM + dmax) — the window, P = BM + dmax - dmin
// rows.  conv_gemm_nt stages a fresh BM-row A tile per K-step (the nine taps re-read the same rows with their halo through L2:
// PMC 2.3x the algorithmic bytes, and the staging probes put 8 % (fp32) / 20 % (bf16) of the kernel on exactly that); here the
// K loop runs channel chunk OUTER, tap INNER: the window of one 128-byte channel chunk is staged once (P rows instead of
// 9 x BM) and the nine K-steps read their A fragments from it at row offset doff_t - dmin.  A tap that leaves the image for a
// given output pixel must contribute zero although its window row holds a real neighbour (previous image row / image): the
// lane's fragment address is redirected to a row of zeros — rows >= P of the window allocation are staged out of range, i.e.
// zero-filled by every window load.  One window buffer, two B buffers: at a chunk boundary the next window is issued after the
// barrier that retires the last tap's reads (one exposed DMA latency per ntaps K-steps, covered by the other workgroups of
// the CU).  Everything else (tile -> workgroup map, stream-K slices, epilogue) is conv_gemm_nt's.
// ------------------------------------------------------------------------------------------------
#define MCN_WIN_MAXPASS 16
template <typename T, int BM, int BN, int NW, int EPI>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 2 && NW == 4 && BM == 128 && BN == 128 ? 2 : 1)) void conv_gemm_nt_win(const GemmNTParams p) {
    typedef MmaNT<T> MM;
    constexpr int NT = NW * 64, RPP = NT / 8, WROWS = NW / 2;
    constexpr int BR = BN / RPP;
    constexpr int WTM = BM / WROWS, WTN = BN / 2;
    constexpr int TM = WTM / MM::MT, TN = WTN / MM::MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.Nn + BN - 1) / BN;
    const int ntm = (p.m_end - p.m_begin + BM - 1) / BM;
    const int sk_rel = (int)blockIdx.x - p.sk_body;
    const bool sk_slice = p.sk_mode == 1 && sk_rel >= 0;
    const int L = sk_slice ? p.sk_body + sk_rel / p.sk_slices : xcd_remap(blockIdx.x, p.sk_mode == 1 ? p.sk_body : ntm * ntn);
    const int m0 = p.m_begin + (L / ntn) * BM, n0 = (L % ntn) * BN;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, (int)p.wt_bytes, 0x00020000);

    // LDS: [window: wrows8 rows x 128 B (rows >= P are zero)] [B buffer 0] [B buffer 1]
    const int P = p.win_rows;
    const int wrows8 = (P + 8) & ~7;                   // at least one zero row behind the window
    const int WB = wrows8 * 128;
    const int ZOFF = P * 128;
    const int crow = tid >> 3;
    // Swizzle key of a WINDOW row: fragment reads start at arbitrary row offsets (tap shifts).  For the 32-row fp32 fragments
    // (row >> 1) & 7 is conflict-free at every offset; for the 16-row x 4-chunk fragments of the 2-byte types it is 2-way
    // conflicted unless the offset is a multiple of 4, and row & 7 is conflict-free at every offset (enumerated over the
    // ds_read_b128 lane groups).  The B tile keeps the tile key (its rows start at multiples of 16).
    auto wkey = [](int r) -> int { return sizeof(T) == 4 ? ((r >> 1) & 7) : (r & 7); };
    const int cid = (tid & 7) ^ ((crow >> 1) & 7);     // B tile: chunk this thread fetches (source-side swizzle; RPP % 16 == 0: the key is pass-invariant)
    const int cidw = (tid & 7) ^ wkey(crow);           // window rows (RPP % 8 == 0)
    const int pix_bytes = p.Cs * (int)sizeof(T);
    const int ntaps = p.ntaps, kpt = p.cpt >> 3;
    unsigned b_off[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + crow + RPP * i;
        b_off[i] = n < p.Nn ? (unsigned)n * (unsigned)p.nchunks * 16u + (unsigned)cid * 16u : MCN_OOB;
    }
    __attribute__((address_space(3))) char* const wbaseA =
        (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024);
    __attribute__((address_space(3))) char* const wbaseB =
        (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024 + WB);
    const int wrow0 = __builtin_amdgcn_readfirstlane(wave * 8);          // first window row this wave stages in pass 0
    // window of channel chunk cb: pass ps stages rows [ps*RPP, ps*RPP + RPP); waves whose rows lie behind the allocation skip
    auto issue_window = [&](int cb) {
        static_for<MCN_WIN_MAXPASS>([&](auto pc) {
            constexpr int ps = decltype(pc)::value;
            if (ps * RPP + wrow0 < wrows8) {                              // wave-uniform
                const int j = ps * RPP + crow;
                const int q = m0 + p.win_dmin + j;                        // input pixel of window row j
                const unsigned bad = (unsigned)q | (unsigned)(p.M - 1 - q) | (unsigned)(P - 1 - j);      // sign bits: outside the tensor / behind the window
                const unsigned off = ((unsigned)q * (unsigned)pix_bytes + (unsigned)(cb * 128 + cidw * 16)) | ((bad >> 31) << 31);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(wbaseA + ps * RPP * 128), 16, (int)off, 0, 0, MCN_DMA_AUX_A);
            }
        });
    };
    auto issue_b = [&](int cb, int t, auto setc) {
        constexpr int S = decltype(setc)::value;
        const unsigned koff = (unsigned)(t * kpt + cb) * 128u;
        static_for<BR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(wbaseB + (S * BN * 128 + i * RPP * 128)), 16,
                                                     (int)(b_off[i] + koff), 0, 0, 0);                          // (OOB + small stays OOB)
        });
    };

    // per fragment row block: window row of the output pixel itself and the taps that stay inside the image (bit t)
    const int fr = MM::frag_row(lane), fc = MM::frag_chunk(lane);
    int rloc[TM];
    unsigned vm[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        rloc[i] = wm * WTM + i * MM::MT + fr - p.win_dmin;
        const int m = m0 + wm * WTM + i * MM::MT + fr;
        unsigned bits = 0;
        if (m < p.m_end) {
            const int hw = p.OH * p.OW;
            const int rem = m - (m / hw) * hw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            for (int t = 0; t < ntaps; ++t) {
                const int tw = p.tap[t];
                const int dy = (short)(tw & 0xffff), dx = tw >> 16;
                bits |= (((unsigned)(oy + dy) < (unsigned)p.IH && (unsigned)(ox + dx) < (unsigned)p.IW) ? 1u : 0u) << t;
            }
        }
        vm[i] = bits;
    }

    typename MM::Acc acc[TN][TM];
    nt_init_acc<T, TN, TM>(acc, sk_slice ? nullptr : p.bias, n0 + wn * WTN, p.Nn, lane);

    const int fsw = (fr >> 1) & 7;
    const int b_rd = (wn * WTN + fr) * 128;
    static_assert(MM::SLABS % 2 == 0, "slab skew needs an even slab count");
    typename MM::Frag xa[2][TM], wb[2][TN];
    int abase[TM], akey[TM];                             // per K-step: byte base and swizzle key of each row block's window row (-1: zero row)
    auto load_frags = [&](int set, int sl, const char* bbase) {
        const int ch = sl * MM::CPS + fc;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int a = akey[i] < 0 ? ZOFF + (ch << 4) : abase[i] + ((ch ^ akey[i]) << 4);
            xa[set][i] = *reinterpret_cast<const typename MM::Frag*>(smem + a);
        }
        const int coff = ((ch ^ fsw) << 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[set][j] = *reinterpret_cast<const typename MM::Frag*>(bbase + b_rd + j * MM::MT * 128 + coff);
    };
    auto mma_set = [&](int set) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) MM::mma(acc[j][i], wb[set][j], xa[set][i]);
    };
    auto compute = [&](int t, const char* bbase) {
        const int tw = p.tap[t];
        const int sh = (short)(tw & 0xffff) * p.IW + (tw >> 16);          // (scalar)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int wr = rloc[i] + sh;
            abase[i] = wr << 7;
            akey[i] = ((vm[i] >> t) & 1u) ? wkey(wr) : -1;
        }
        load_frags(0, 0, bbase);
#pragma unroll
        for (int s = 0; s < MM::SLABS; ++s) {
            if (s + 1 < MM::SLABS) load_frags((s + 1) & 1, s + 1, bbase);
            mma_set(s & 1);
        }
    };

    const int nk_all = ntaps * kpt;
    int ks0 = 0, nk = nk_all;
    if (sk_slice) {
        const int per = (nk_all + p.sk_slices - 1) / p.sk_slices, sl = sk_rel % p.sk_slices;
        ks0 = sl * per;
        nk = min(nk_all, ks0 + per);
    }
    constexpr int ACCSZ = (int)(sizeof(typename MM::Acc) / 4);
    constexpr int NREG = TN * TM * ACCSZ;
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    if (ks0 < nk) {
        int cb = ks0 / ntaps, t = ks0 - cb * ntaps;      // (a stream-K slice may start in the middle of a channel chunk)
        bool need_win = true;
        auto step = [&](int s, auto cur, auto nxt) {
            if (need_win) {
                if (s > ks0) __syncthreads();            // every wave has finished reading the previous chunk's window
                issue_window(cb);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int t2 = t + 1, cb2 = cb;
            if (t2 == ntaps) { t2 = 0; cb2 = cb + 1; }
            if (s + 1 < nk) issue_b(cb2, t2, nxt);
            compute(t, smem + WB + decltype(cur)::value * (BN * 128));
            need_win = t2 == 0;
            t = t2;
            cb = cb2;
        };
        issue_b(cb, t, S0{});
        for (int s = ks0; s < nk; s += 2) {
            step(s, S0{}, S1{});
            if (s + 1 < nk) step(s + 1, S1{}, S0{});
        }
    }
    if (sk_slice) {
        float* dst = p.partial + ((size_t)sk_rel * NREG) * NT + tid;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < ACCSZ; ++e) dst[(size_t)((j * TM + i) * ACCSZ + e) * NT] = acc[j][i][e];
        return;
    }
    nt_epilogue<T, BM, BN, true, NW, EPI>(p, acc, m0, n0, lane, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// conv_gemm_nt_wpp (round 4, 2-byte types): 3x3 / stride-1 / same-grid convolutions (forward and dgrad) as 256 x BN tiles on 8 waves with
//   * the tile's INPUT WINDOW (256 + 2 W + 2 pixel rows of one 64-channel chunk) kept in LDS across the nine taps and DOUBLE-BUFFERED: the window
//     of chunk cb + 1 arrives — one 64-row pass per K-step, six passes — under the nine K-steps of chunk cb;
//   * the filter tiles (BN rows x 128 B per (tap, chunk) K-step) on a ring of three LDS slots, the tile of step s + 2 issued during step s;
//   * two wave groups ONE PHASE APART (waves 4-7 share their SIMDs with waves 0-3): while one group multiplies (32 MFMAs at BN = 128) the other reads
//     its fragments and issues its DMAs; every phase ends at a workgroup barrier (the late group runs one extra barrier up front, the early
//     group one at the end);
//   * COUNTED vmcnt throughout: the chunk loop is unrolled over (window parity, tap), every K-step issues a fixed number of LDS-DMAs — a window pass
//     in steps 0-5 of a chunk (past the last chunk: out of range, i.e. zeros into a buffer nobody reads again), the filter tile of step s + 2 in every
//     step (past the end likewise) — so that "all DMAs up to the filter tile of step s have landed" is `s_waitcnt vmcnt(constant(tap))`.
// Why this shape: the two-buffer kernels and the plain ring / ping-pong forms of this round spend their K-step in DMA ISSUE (an LDS-DMA among
// ds_reads costs the issuing wave 100-185 cycles; 6-8 of them per 32 MFMAs): here a wave issues 2-3 per 32 MFMAs (profiles/round4_*probe*).
// LDS: 2 x 384 window rows + 3 x BN filter rows, 128 B each = 144 KB at BN = 128: one workgroup per CU.
// ------------------------------------------------------------------------------------------------
#define WPP_WROWS 384           /* window rows per buffer: six staging passes of 64 rows (P = 256 + span <= 376 + at least one zero row) */
template <typename T, int BN, int EPI>
__global__ __launch_bounds__(512) void conv_gemm_nt_wpp(const GemmNTParams p) {
    typedef MmaNT<T> MM;
    static_assert(sizeof(T) == 2 && MM::SLABS == 2 && MM::MT == 16 && MM::CPS == 4, "2-byte types: 16x16x32 MFMA, two 64-byte slabs per 128-byte K-step");
    constexpr int BM = 256, NW = 8, RPP = 64;
    constexpr int WTM = 64, WTN = BN / 2;
    constexpr int TM = WTM / MM::MT, TN = WTN / MM::MT;
    constexpr int NB = BN / RPP;                          // filter-tile DMAs per wave and K-step
    constexpr int WB = WPP_WROWS * 128;                   // bytes of one window buffer
    constexpr int SB = BN * 128;                          // bytes of one filter ring slot
    constexpr int WOFF = 3 * SB;                          // LDS: [filter ring (ds_read immediates reach every slot)] [window 0] [window 1]
    constexpr int NTAPS = 9, WPASS = WPP_WROWS / RPP;     // 6 window passes per chunk
    static_assert(2 * SB + (TN - 1) * MM::MT * 128 + 64 < 65536 && WB < 65536, "fragment reads address ring slots / window buffers through the 16-bit ds_read offset");
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(3))) typename MM::Frag lds_frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const lds = (lds_char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.Nn + BN - 1) / BN;
    const int ntm = (p.m_end - p.m_begin + BM - 1) / BM;
    const int L = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = p.m_begin + (L / ntn) * BM, n0 = (L % ntn) * BN;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt), 0, (int)p.wt_bytes, 0x00020000);

    const int P = p.win_rows;                             // BM + span rows hold pixels; rows P .. 383 are staged out of range (zeros: at least the 8 rows behind P rounded up to 8)
    const int crow = tid >> 3;
    const int cid = (tid & 7) ^ ((crow >> 1) & 7);        // filter tile: source-side swizzle of the tile image (rows start at multiples of 16)
    const int cidw = (tid & 7) ^ (crow & 7);              // window rows are read at arbitrary row offsets: key = row & 7 (64 % 8 == 0: pass-invariant)
    const int kpt = p.cpt >> 3;                           // 64-channel chunks
    // everything a K-step needs is ADDRESS ARITHMETIC HOISTED OUT OF THE LOOP (measured: with ~85 VALU instructions per wave and K-step for the
    // fragment addresses the load phase alone took 620 cycles against the 512 of the other group's 32 MFMAs):
    //   * DMA sources: one byte offset per window pass / filter row block; a K-step adds a scalar (saturating: bit 31 = out of range = zeros);
    //   * fragment reads: the LDS address of (tap, row block) for slab 0 (slab 1 = ^ 64; window buffer and ring slot ride in the instruction's offset).
    unsigned b_off[NB], w_off[WPASS];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + crow + RPP * i;
        b_off[i] = n < p.Nn ? (unsigned)n * (unsigned)p.nchunks * 16u + (unsigned)cid * 16u : MCN_OOB;
    }
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
        const int j = ps * RPP + crow;
        const int q = m0 + p.win_dmin + j;                // input pixel of window row j
        const unsigned bad = (unsigned)q | (unsigned)(p.M - 1 - q) | (unsigned)(P - 1 - j);     // sign bits
        w_off[ps] = (bad >> 31) ? MCN_OOB : (unsigned)q * (unsigned)(p.Cs * (int)sizeof(T)) + (unsigned)cidw * 16u;
    }
    lds_char* const wbase = lds + __builtin_amdgcn_readfirstlane(wave * 1024);
    // chunk offset of a DMA source (scalar): past the last chunk -> out of range, which keeps the DMA count of every K-step fixed
    auto chunk_term = [&](int cb, int k128) -> unsigned { return cb < kpt ? (unsigned)k128 * 128u : MCN_OOB; };
    // window pass `ps` (64 rows) of channel chunk cb into window buffer WSEL
    auto issue_wpass = [&](int cb, auto wselc, auto psc) {
        constexpr int WSEL = decltype(wselc)::value, ps = decltype(psc)::value;
        const unsigned off = __builtin_elementwise_add_sat(w_off[ps], chunk_term(cb, cb));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(wbase + (WOFF + WSEL * WB + ps * RPP * 128)), 16, (int)off, 0, 0, MCN_DMA_AUX_A);
    };
    // filter tile of K-step (cb, t) into ring slot S
    auto issue_b = [&](int cb, int t, auto setc) {
        constexpr int S = decltype(setc)::value;
        const unsigned koff = chunk_term(cb, t * kpt + cb);
        static_for<NB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(wbase + (S * SB + i * RPP * 128)), 16,
                                                     (int)__builtin_elementwise_add_sat(b_off[i], koff), 0, 0, 0);
        });
    };

    const int fr = MM::frag_row(lane), fc = MM::frag_chunk(lane);
    unsigned wa[NTAPS][TM];                               // slab 0 of (tap, row block) in window buffer 0; taps outside the image read the zero row P
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * MM::MT + fr;
        const int hw = p.OH * p.OW;
        const int rem = m - (m / hw) * hw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int tw = p.tap[t];
            const int dy = (short)(tw & 0xffff), dx = tw >> 16;
            const bool ok = m < p.m_end && (unsigned)(oy + dy) < (unsigned)p.IH && (unsigned)(ox + dx) < (unsigned)p.IW;
            const int wr = wm * WTM + i * MM::MT + fr - p.win_dmin + dy * p.IW + dx;          // window row of this tap's input pixel
            // a tap outside the image reads zeros — from the zero row with the SAME bank image as the row it replaces (rows P8 + (wr & 7), P8 = P rounded up to 8:
            // same row parity, same swizzle key): with one shared zero row the redirected lanes collide with their neighbours' banks and a fragment read with
            // any border pixel in its 16-lane group takes 8 LDS cycles instead of 4 (SQ_LDS_BANK_CONFLICT 26 % of SQ_LDS_IDX_ACTIVE on the 7x7 layer -> 0; same-box A/B: no change in time,
            // profiles/round4_pmc_layers_bf16.txt)
            const int zr = ((P + 7) & ~7) + (wr & 7);
            wa[t][i] = (unsigned)WOFF + (unsigned)((ok ? wr : zr) << 7) + (unsigned)((fc ^ (wr & 7)) << 4);
        }
    }
    const unsigned ba = (unsigned)((wn * WTN + fr) * 128 + ((fc ^ ((fr >> 1) & 7)) << 4));      // slab 0 of the wave's filter rows in ring slot 0

    typename MM::Acc acc[TN][TM];
    nt_init_acc<T, TN, TM>(acc, p.bias, n0 + wn * WTN, p.Nn, lane);

    typename MM::Frag xa[2][TM], wb[2][TN];
    // all fragments of K-step (window buffer WSEL, tap TAP, ring slot S): 2 slabs x (TM + TN) ds_read_b128, one v_xor per slab-1 read
    auto load_step = [&](auto tapc, auto wselc, auto setc) {
        constexpr int TAP = decltype(tapc)::value, WSEL = decltype(wselc)::value, S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            xa[0][i] = *(lds_frag*)(lds + wa[TAP][i] + WSEL * WB);
            xa[1][i] = *(lds_frag*)(lds + (wa[TAP][i] ^ 64u) + WSEL * WB);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            wb[0][j] = *(lds_frag*)(lds + ba + (S * SB + j * MM::MT * 128));
            wb[1][j] = *(lds_frag*)(lds + (ba ^ 64u) + (S * SB + j * MM::MT * 128));
        }
    };
    auto mma_set = [&](int set) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) MM::mma(acc[j][i], wb[set][j], xa[set][i]);
    };

    const bool late = __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
    auto bar = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // K-step (chunk cb, tap TAP) on window buffer WSEL.  DMAs of this step: [window pass TAP of chunk cb + 1 (TAP < 6)], then the filter tile of step
    // s + 2.  "Everything up to the filter tile of step k has landed" leaves the DMAs of step k - 1 in flight: NB + (1 if step k - 1 staged a window pass).
    auto pstep = [&](int cb, auto wselc, auto tapc) {
        constexpr int WSEL = decltype(wselc)::value, TAP = decltype(tapc)::value;
        constexpr int PREV = (TAP + NTAPS - 1) % NTAPS;                   // tap of the step in front of this one
        constexpr int N_EARLY = NB + (PREV < WPASS ? 1 : 0);              // early group, in front of its load phase: waits for step s
        constexpr int N_LATE = NB + (TAP < WPASS ? 1 : 0);                // late group, behind its load phase: waits for step s + 1
        if (!late) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_EARLY) : "memory");
        bar();                                             // ---- load phase
        load_step(tapc, wselc, std::integral_constant<int, TAP % 3>{});
        if constexpr (TAP < WPASS) issue_wpass(cb + 1, std::integral_constant<int, WSEL ^ 1>{}, std::integral_constant<int, TAP>{});
        {
            // step s + 2 = (cb, TAP + 2) or (cb + 1, TAP + 2 - 9)
            constexpr int T2 = (TAP + 2) % NTAPS;
            issue_b(cb + (TAP + 2 >= NTAPS ? 1 : 0), T2, std::integral_constant<int, (TAP + 2) % 3>{});
        }
        if (late) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_LATE) : "memory");
        bar();                                             // ---- compute phase
        __builtin_amdgcn_s_setprio(1);
        mma_set(0);
        mma_set(1);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: the whole window of chunk 0, the filter tiles of steps 0 and 1
    static_for<WPASS>([&](auto pc) { issue_wpass(0, std::integral_constant<int, 0>{}, pc); });
    issue_b(0, 0, std::integral_constant<int, 0>{});
    issue_b(0, 1, std::integral_constant<int, 1>{});
    if (late) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");        // window 0 and the tile of step 0 (the tile of step 1 stays in flight)
        bar();
    }
    for (int cb = 0; cb < kpt; cb += 2) {
        static_for<NTAPS>([&](auto tc) { pstep(cb, std::integral_constant<int, 0>{}, tc); });
        if (cb + 1 < kpt) static_for<NTAPS>([&](auto tc) { pstep(cb + 1, std::integral_constant<int, 1>{}, tc); });
    }
    if (!late) bar();                                      // pairs with the late group's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the out-of-range DMAs behind the last step)
    nt_epilogue<T, BM, BN, true, NW, EPI>(p, acc, m0, n0, lane, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// conv_gemm_tn (wgrad): tile = 128 rows of (tap,c) x BN columns (cout), reduction over pixels
// ------------------------------------------------------------------------------------------------
template <typename T>
struct TNCfg;
template <>
struct TNCfg<bf16_t> {
    static constexpr int KP = 64;       // pixels per K-step
    static constexpr int MT = 16;
};
template <>
struct TNCfg<f16_t> {
    static constexpr int KP = 64;
    static constexpr int MT = 16;
};
template <>
struct TNCfg<float> {
    static constexpr int KP = 32;
    static constexpr int MT = 32;
};
// transposing LDS read + MFMA of the 2-byte storage types
__device__ __forceinline__ bf16x4 lds_tr16(const char* p, bf16_t) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
typedef __fp16 fp16v4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));      // the builtin's own vector type (same bits as f16x4)
__device__ __forceinline__ f16x4 lds_tr16(const char* p, f16_t) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16v4_t*)p));
}
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// bf16 pixel-major tile: row stride RS bytes, 32-byte granule g of pixel row p stored at granule g ^ key(p)
__device__ __forceinline__ int tn_key(int p) { return (p & 3) | (((p >> 3) & 1) << 2); }

// NW waves per workgroup: 4 = 2x2 waves; 8 = 4x2 waves on the same tile (wave tile BR/4 x BN/2): the 128x128 tile needs 64 KB of LDS,
// i.e. two workgroups per CU — with 4 waves each a SIMD holds 2 waves and its MFMA pipe idles whenever both sit at the
// barrier / DMA wait (fp32: 69 % busy by SQ_VALU_MFMA_BUSY_CYCLES); 8 waves put 4 on every SIMD at the same tile traffic.
// STAGES = 3: a ring of three LDS buffers — the DMA of K-step k+2 is issued at K-step k, two steps of MFMAs cover its latency and the
// wait in front of a step is `vmcnt(DMAs of one stage)`, not vmcnt(0).  (conv_gemm_tn3: 96 KB for the fp32 128 x 128 tile, one
// 8-wave workgroup per CU.)
template <typename T, int BR, int BN, bool LINEAR, int NW, int STAGES>
__device__ __forceinline__ void tn_body(const GemmTNParams& p) {
    typedef TNCfg<T> CF;
    constexpr int CE = VecTraits<T>::CE;
    constexpr int KP = CF::KP;
    constexpr int XRS = BR * (int)sizeof(T);           // X tile row stride (bytes): 256 / 512
    constexpr int DRS = BN * (int)sizeof(T);
    constexpr int XCPR = BR / CE, DCPR = BN / CE;      // chunks per pixel row
    constexpr int NT = NW * 64;                        // threads
    constexpr int XPR = NT / XCPR, DPR = NT / DCPR;    // pixel rows per staging pass
    constexpr int XN = KP / XPR, DN = KP / DPR;        // chunks per thread
    constexpr int XBYTES = KP * XRS, DBYTES = KP * DRS;
    constexpr int TILE_BYTES = XBYTES + DBYTES;
    constexpr int WTR = BR / (NW / 2), WTN = BN / 2;
    constexpr int TR = WTR / CF::MT, TNn = WTN / CF::MT;
    constexpr int DGM = (DRS / 32) - 1;                // granule masks of the D / X tiles
    constexpr int XGM = (XRS / 32) - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntn = (p.Nn + BN - 1) / BN;
    // the tiles of one split (all taps / channel blocks over the same pixel range) read the same dy rows and overlapping x rows: keep
    // groups of p.grp of them on one XCD, next to each other in its dispatch order, so that its L2 serves all but the first of a group
    // (order: group of tiles, split, tile in the group; every group but the last has exactly p.grp tiles)
    int tile = blockIdx.x, split = blockIdx.y;
    if (p.grp > 0) {
        const int nt = gridDim.x, ns = gridDim.y;
        const int L = xcd_remap(blockIdx.y * nt + blockIdx.x, nt * ns);
        const int k = L / (ns * p.grp);
        const int rem = L - k * ns * p.grp;
        const int sz = min(p.grp, nt - k * p.grp);
        split = rem / sz;
        tile = k * p.grp + rem - split * sz;
    }
    const int r0 = (tile / ntn) * BR, n0 = (tile % ntn) * BN;
    const int ks0 = split * p.steps_per_split;
    const int ks1 = min(p.nsteps, ks0 + p.steps_per_split);

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

    // ---- X staging (LDS-DMA): the thread at linear LDS slot (pixel row xpr, slot tid % XCPR) fetches the chunk that the
    // swizzled image keeps there (bf16: 32-byte granule g of row p sits at granule g ^ key(p); XPR is a multiple of 16,
    // so the key is the same for all of a thread's rows); fixed column (tap, c) per thread, XN pixel rows ----
    const int xpr = tid / XCPR, xsl = tid % XCPR;
    const int xcc = sizeof(T) == 2 ? ((((xsl >> 1) ^ (tn_key(xpr) & XGM)) << 1) | (xsl & 1)) : xsl;
    const int gcol = r0 + xcc * CE;
    const bool xcol_ok = gcol < p.rows;
    int xt = 0, xc = gcol;
    if (!LINEAR && xcol_ok) {
        xt = gcol / p.Cp;
        xc = gcol - xt * p.Cp;
    }
    const int tdy = p.tdy[xt], tdx = p.tdx[xt];
    int px[XN], py[XN], pimg[XN];
    int dKx = 0, dKy = 0, dKi = 0;
    if (!LINEAR) {
        const int hw = p.OH * p.OW;
        dKi = KP / hw;
        const int rem = KP - dKi * hw;
        dKy = rem / p.OW;
        dKx = rem - dKy * p.OW;
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int m = ks0 * KP + xpr + XPR * i;
            const int img = m / hw, r = m - img * hw;
            pimg[i] = img;
            py[i] = r / p.OW;
            px[i] = r - py[i] * p.OW;
        }
    }
    // ---- D staging ---------------------------------------------------------------------------
    const int dpr = tid / DCPR, dsl = tid % DCPR;
    const int dcc = sizeof(T) == 2 ? ((((dsl >> 1) ^ (tn_key(dpr) & DGM)) << 1) | (dsl & 1)) : dsl;
    const int dn = n0 + dcc * CE;
    const bool dcol_ok = dn < p.Nn;

    const unsigned xoob = xcol_ok ? 0u : MCN_OOB, doob = dcol_ok ? 0u : MCN_OOB;       // loop-invariant column masks
    static_assert(XPR * XRS == NT * 16 && DPR * DRS == NT * 16, "one staging pass of the workgroup = 16 bytes per thread");
    static_assert(XN >= 1 && DN >= 1 && WTR % CF::MT == 0, "tile too small for this many waves");
    __attribute__((address_space(3))) char* const wbase =
        (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024);
    auto issue = [&](int ks, auto bufc) {
        constexpr int B = decltype(bufc)::value;
        if (!(MCN_PROBES && (p.dbg & 1) && ks != ks0))
        static_for<XN>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = ks * KP + xpr + XPR * i;
            // out-of-range as arithmetic on bit 31 (sign bits of the range tests), not as `if (ok) off = ...`: hipcc turns that
            // select into two DMA instructions under complementary exec masks behind a branch (8-12 branches per K-step)
            unsigned off;
            if (LINEAR) {
                off = (((unsigned)m * (unsigned)p.Cs + (unsigned)xc) * (unsigned)sizeof(T)) | xoob | (((unsigned)(p.M - 1 - m) >> 31) << 31);
            } else {
                const int iy = py[i] * p.sy + tdy, ix = px[i] * p.sx + tdx;
                const unsigned bad = (unsigned)(p.M - 1 - m) | (unsigned)iy | (unsigned)(p.IH - 1 - iy) | (unsigned)ix | (unsigned)(p.IW - 1 - ix);
                off = (((unsigned)((pimg[i] * p.IH + iy) * p.IW + ix) * (unsigned)p.Cs + (unsigned)xc) * (unsigned)sizeof(T)) | xoob | ((bad >> 31) << 31);
                // advance this row's pixel coordinates by KP for the next step
                px[i] += dKx;
                if (px[i] >= p.OW) { px[i] -= p.OW; py[i] += 1; }
                py[i] += dKy;
                if (py[i] >= p.OH) { py[i] -= p.OH; pimg[i] += 1; }
                pimg[i] += dKi;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(wbase + (B * TILE_BYTES + i * (NT * 16))), 16, (int)off, 0, 0, MCN_DMA_AUX_X);
        });
        if (!(MCN_PROBES && (p.dbg & 2) && ks != ks0))
        static_for<DN>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = ks * KP + dpr + DPR * i;
            const unsigned off = (((unsigned)m * (unsigned)p.ldy + (unsigned)dn) * (unsigned)sizeof(T)) | doob | (((unsigned)(p.M - 1 - m) >> 31) << 31);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(wbase + (B * TILE_BYTES + XBYTES + i * (NT * 16))), 16, (int)off, 0, 0, MCN_DMA_AUX_X);
        });
    };
    constexpr int ACCN = CF::MT == 16 ? 4 : 16;
    float acc[TR][TNn][ACCN];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TNn; ++j)
#pragma unroll
            for (int e = 0; e < ACCN; ++e) acc[i][j][e] = 0.f;

    typedef std::integral_constant<int, 0> B0;
    typedef std::integral_constant<int, 1> B1;
    // one K-step (KP pixels) from LDS buffer `buf`; the DMA of the next step goes to the other buffer right after the
    // barrier that retires its readers and flies under this step's MFMAs
    auto kstep = [&](int ks, auto cur, auto nxt) {
        if constexpr (STAGES >= 3) {
            // ring of STAGES buffers: `nxt` is the buffer of step ks + STAGES - 1 (read last in step ks - 1: every wave is past that once it
            // reaches this barrier); the DMAs of steps ks + 1 .. ks + STAGES - 2, issued earlier, may stay in flight (loads retire in issue order)
            constexpr int PER = XN + DN;
            const int ahead = min(STAGES - 2, ks1 - 1 - ks);                       // later steps whose DMAs are already issued (uniform)
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            static_assert(STAGES <= 4 && 2 * PER < 64, "vmcnt immediates of the ring");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (ks + STAGES - 1 < ks1) issue(ks + STAGES - 1, nxt);
        } else if (MCN_PROBES && (p.dbg & 32)) {
            // early issue: the next step's DMAs do not wait for THIS step's data — barrier (everybody has finished reading `nxt`), issue,
            // then wait for this step's DMAs only (the ones just issued stay in flight) and a second barrier to see the other threads' data
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (ks + 1 < ks1) {
                issue(ks + 1, nxt);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XN + DN) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ks + 1 < ks1) issue(ks + 1, nxt);
        }
        const char* xs = smem + decltype(cur)::value * TILE_BYTES;
        const char* ds = xs + XBYTES;
        if constexpr (sizeof(T) == 2) {
            // bf16: 16x16x32; operand rows come from transposing reads of the pixel-major tiles
            const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
#pragma unroll
            for (int s = 0; s < KP / 32; ++s) {
                typename VecTraits<T>::V4 xh[2][TR], dh[2][TNn];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pix = 32 * s + 8 * g + 4 * h + q;
                    const int key = tn_key(pix);
#pragma unroll
                    for (int i = 0; i < TR; ++i) {
                        const int gran = wr * (WTR / 16) + i;
                        xh[h][i] = lds_tr16(xs + pix * XRS + ((gran ^ (key & XGM)) << 5) + 8 * pp, T{});
                    }
#pragma unroll
                    for (int j = 0; j < TNn; ++j) {
                        const int gran = wc * (WTN / 16) + j;
                        dh[h][j] = lds_tr16(ds + pix * DRS + ((gran ^ (key & DGM)) << 5) + 8 * pp, T{});
                    }
                }
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TNn; ++j) {
                        f32x4 a = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        const typename VecTraits<T>::V8 xa = __builtin_shufflevector(xh[0][i], xh[1][i], 0, 1, 2, 3, 4, 5, 6, 7);
                        const typename VecTraits<T>::V8 db = __builtin_shufflevector(dh[0][j], dh[1][j], 0, 1, 2, 3, 4, 5, 6, 7);
                        a = mfma16(xa, db, a);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][e] = a[e];
                    }
            }
        } else {
            // fp32: 32x32x2; lane (i = lane&31, k = lane>>5) reads one float per operand
            const int li = lane & 31, kh = lane >> 5;
#pragma unroll 4
            for (int pp = 0; pp < KP / 2; ++pp) {
                const int pix = 2 * pp + kh;
                float xa[TR], db[TNn];
#pragma unroll
                for (int i = 0; i < TR; ++i) xa[i] = *reinterpret_cast<const float*>(xs + pix * XRS + (wr * WTR + i * 32 + li) * 4);
#pragma unroll
                for (int j = 0; j < TNn; ++j) db[j] = *reinterpret_cast<const float*>(ds + pix * DRS + (wc * WTN + j * 32 + li) * 4);
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TNn; ++j) {
                        f32x16 a;
#pragma unroll
                        for (int e = 0; e < 16; ++e) a[e] = acc[i][j][e];
                        a = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i], db[j], a, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = a[e];
                    }
            }
        }
    };
    if constexpr (STAGES == 3) {
        typedef std::integral_constant<int, 2> B2;
        if (ks0 < ks1) issue(ks0, B0{});
        if (ks0 + 1 < ks1) issue(ks0 + 1, B1{});
        for (int ks = ks0; ks < ks1; ks += 3) {
            kstep(ks, B0{}, B2{});
            if (ks + 1 < ks1) kstep(ks + 1, B1{}, B0{});
            if (ks + 2 < ks1) kstep(ks + 2, B2{}, B1{});
        }
    } else if constexpr (STAGES == 4) {
        typedef std::integral_constant<int, 2> B2;
        typedef std::integral_constant<int, 3> B3;
        if (ks0 < ks1) issue(ks0, B0{});
        if (ks0 + 1 < ks1) issue(ks0 + 1, B1{});
        if (ks0 + 2 < ks1) issue(ks0 + 2, B2{});
        for (int ks = ks0; ks < ks1; ks += 4) {
            kstep(ks, B0{}, B3{});
            if (ks + 1 < ks1) kstep(ks + 1, B1{}, B0{});
            if (ks + 2 < ks1) kstep(ks + 2, B2{}, B1{});
            if (ks + 3 < ks1) kstep(ks + 3, B3{}, B2{});
        }
    } else {
        if (ks0 < ks1) issue(ks0, B0{});
        for (int ks = ks0; ks < ks1; ks += 2) {
            kstep(ks, B0{}, B1{});
            if (ks + 1 < ks1) kstep(ks + 1, B1{}, B0{});
        }
    }

    // ---- store the partial tile to this split's slab (row-major [rows][Nn]) --------------------
    float* slab = p.slab + (size_t)split * p.rows * p.Nn;
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TNn; ++j)
#pragma unroll
            for (int e = 0; e < ACCN; ++e) {
                int row, col;
                if (CF::MT == 16) {
                    row = (lane >> 4) * 4 + e;
                    col = lane & 15;
                } else {
                    row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    col = lane & 31;
                }
                const int r = r0 + wr * WTR + i * CF::MT + row;
                const int n = n0 + wc * WTN + j * CF::MT + col;
                if (r < p.rows && n < p.Nn) slab[(size_t)r * p.Nn + n] = acc[i][j][e];
            }
}

template <typename T, int BR, int BN, bool LINEAR, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv_gemm_tn(const GemmTNParams p) {
    tn_body<T, BR, BN, LINEAR, NW, 2>(p);
}
template <typename T, int BR, int BN, bool LINEAR, int NW>
__global__ __launch_bounds__(NW * 64) void conv_gemm_tn3(const GemmTNParams p) {
    tn_body<T, BR, BN, LINEAR, NW, 3>(p);
}
// four-stage ring (round 3): the DMAs of K-steps k+1 .. k+3 are in flight while step k multiplies.  For the 1x1 wgrads of the 2-byte types:
// all tiles of a split walk the same pixel rows in lockstep, so every K-step's operands come from HBM (the XCD's L2 only merges the
// tiles' requests) and a two-buffer loop runs one K-step per HBM round trip.
template <typename T, int BR, int BN, bool LINEAR, int NW>
__global__ __launch_bounds__(NW * 64) void conv_gemm_tn4(const GemmTNParams p) {
    tn_body<T, BR, BN, LINEAR, NW, 4>(p);
}

// dw[t][c][n] = scale * sum_split slab[split][t*Cp + c][n]   (c < Cin).  16 outputs x 16 split lanes per block (the stem: 9408
// outputs, up to 1024 splits — a thread per output walking them alone took 170 us); lanes fold through LDS in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int ntaps, int Cp,
                                                           int Cin, int Nn, float scale) {
    __shared__ float red[256];
    const long total = (long)ntaps * Cin * Nn;
    const long rowsNn = (long)ntaps * Cp * Nn;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const long idx = (long)blockIdx.x * 16 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
        const int n = (int)(idx % Nn);
        const long tc = idx / Nn;
        const int c = (int)(tc % Cin), t = (int)(tc / Cin);
        const float* src = slab + ((long)t * Cp + c) * Nn + n;
        int k = ty;
        for (; k + 48 < splits; k += 64) {
            const float v0 = src[(long)k * rowsNn], v1 = src[(long)(k + 16) * rowsNn], v2 = src[(long)(k + 32) * rowsNn], v3 = src[(long)(k + 48) * rowsNn];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; k < splits; k += 16) s0 += src[(long)k * rowsNn];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && idx < total) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) s += red[l * 16 + tx];
        dw[idx] = s * scale;
    }
}
// Cp == Cin (every conv but the stem): the slab rows are the HWIO rows, so the reduction is a plain strided sum of
// `splits` arrays — 16-byte lanes.  SL lanes share one output vector and take the splits s = l, l+SL, ... (small filters
// have few output vectors and up to 512 splits: one thread per output was a serial chain of ~500 dependent-issue loads in
// a handful of workgroups); the SL partial sums are folded through LDS in a fixed order (deterministic).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_linear_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits,
                                                                  long total4, float scale) {
    constexpr int VPB = 256 / SL;                                 // output vectors per block
    __shared__ f32x4 red[SL > 1 ? 256 : 1];
    const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
    f32x4* d4 = reinterpret_cast<f32x4*>(dw);
    const int v = threadIdx.x % VPB, l = threadIdx.x / VPB;
    for (long i0 = (long)blockIdx.x * VPB; i0 < total4; i0 += (long)gridDim.x * VPB) {
        const long i = i0 + v;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if (i < total4) {
            // four slabs per trip, all loads before the first add (a lane walking 12-64 slabs two at a time is a chain of dependent L2 /
            // Infinity Cache round trips: these launches ran 6-9 us for a few hundred KB)
            int k = l;
            for (; k + 3 * SL < splits; k += 4 * SL) {
                const f32x4 v0 = s4[(long)k * total4 + i], v1 = s4[(long)(k + SL) * total4 + i];
                const f32x4 v2 = s4[(long)(k + 2 * SL) * total4 + i], v3 = s4[(long)(k + 3 * SL) * total4 + i];
                a0 += v0; a1 += v1;
                a0 += v2; a1 += v3;
            }
            for (; k < splits; k += SL) a0 += s4[(long)k * total4 + i];
        }
        a0 += a1;
        if (SL == 1) {
            if (i < total4) d4[i] = a0 * scale;
        } else {
            red[l * VPB + v] = a0;
            __syncthreads();
            if (l == 0 && i < total4) {
                f32x4 t = red[v];
#pragma unroll
                for (int j = 1; j < SL; ++j) t += red[j * VPB + v];
                d4[i] = t * scale;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 HWIO master -> K-contiguous operand in the compute dtype
//   mode 0 (fwd)  : out[n][t*Cp + c] = w[r_t][s_t][c][n]      rows = Cout, per-tap width Cp >= Cin
//   mode 1 (dgrad): out[c][t*Cp + k] = w[r_t][s_t][c][k]      rows = Cin,  per-tap width Cp >= Cout
// ------------------------------------------------------------------------------------------------
//   mode 2 / 3    : U = G g G^T of a 3x3 filter for the Winograd kernels (forward / dgrad = rotated + transposed), fp32
__device__ __forceinline__ void wino_u_elements(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, bool transposed, long id);
struct PackParams {
    const float* w;
    void* out;
    int KW, Cin, Cout, rows, Cp, ntaps, mode;
    int job0, job1;      // batch kernel: the slice of this operand's jobs the descriptor covers (job1 == 0: all of them)
    signed char tr[MCN_MAX_TAPS], ts[MCN_MAX_TAPS];
};
template <typename T>
__global__ void pack_weights_kernel(const PackParams p) {
    const long total = (long)p.rows * p.ntaps * p.Cp;
    T* out = reinterpret_cast<T*>(p.out);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int col = (int)(idx % p.Cp);
        const long rt = idx / p.Cp;
        const int t = (int)(rt % p.ntaps), row = (int)(rt / p.ntaps);
        const int r = p.tr[t], s = p.ts[t];
        float v = 0.f;
        if (p.mode == 0) {
            if (col < p.Cin) v = p.w[(((long)r * p.KW + s) * p.Cin + col) * p.Cout + row];
        } else {
            if (col < p.Cout) v = p.w[(((long)r * p.KW + s) * p.Cin + row) * p.Cout + col];
        }
        out[idx] = from_f32<T>(v);
    }
}

// all packed operands of a model in ONE launch: blockIdx.y selects the descriptor (device table; big operands are cut into
// several descriptors of at most MCN_PACK_SLICE_* jobs so that no operand is left to 64 workgroups), blockIdx.x strides
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackParams* __restrict__ table) {
    const PackParams& p = table[blockIdx.y];
    const long total = (long)p.rows * p.ntaps * p.Cp;
    T* out = reinterpret_cast<T*>(p.out);
    if (p.mode >= 2) {                                   // Winograd filter transform (wino_kernels.h): one job = 256 elements of U
        for (int job = p.job0 + blockIdx.x; job < p.job1; job += gridDim.x)
            wino_u_elements(p.w, reinterpret_cast<float*>(p.out), p.Cin, p.Cout, p.mode == 3, (long)job * 256 + threadIdx.x);
        return;
    }
    if (p.mode == 0) {
        // forward operand = per-tap TRANSPOSE of the HWIO filter ([cout][tap][cin] from [tap][cin][cout]): 32x32 tiles through
        // LDS so that both the fp32 reads (along cout) and the packed writes (along cin) are coalesced — the element-wise form
        // below reads with a stride of Cout floats and took 260 us per step for ResNet-50's 25 M weights
        __shared__ float tile[32][33];
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
        const int nci = (p.Cp + 31) / 32, nco = (p.rows + 31) / 32;
        const int njobs = p.job1 ? p.job1 : p.ntaps * nci * nco;
        for (int job = p.job0 + blockIdx.x; job < njobs; job += gridDim.x) {
            const int co0 = (job % nco) * 32;
            const int rest = job / nco;
            const int ci0 = (rest % nci) * 32, t = rest / nci;
            const long wbase = ((long)p.tr[t] * p.KW + p.ts[t]) * p.Cin;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = ci0 + ty + 8 * k, co = co0 + tx;
                tile[ty + 8 * k][tx] = (ci < p.Cin && co < p.rows) ? p.w[(wbase + ci) * p.Cout + co] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = co0 + ty + 8 * k, ci = ci0 + tx;
                if (co < p.rows && ci < p.Cp) out[((long)co * p.ntaps + t) * p.Cp + ci] = from_f32<T>(tile[tx][ty + 8 * k]);
            }
            __syncthreads();
        }
        return;
    }
    // dgrad operand ([cin][tap][cout]): every (cin, tap) row is a contiguous run of Cout weights — one job per row, no
    // per-element 64-bit division
    (void)total;
    const int njobs = p.job1 ? p.job1 : p.rows * p.ntaps;
    for (int job = p.job0 + blockIdx.x; job < njobs; job += gridDim.x) {
        const int row = job / p.ntaps, t = job - row * p.ntaps;
        const float* src = p.w + (((long)p.tr[t] * p.KW + p.ts[t]) * p.Cin + row) * p.Cout;
        T* dst = out + (long)job * p.Cp;
        for (int col = threadIdx.x; col < p.Cp; col += 256) dst[col] = from_f32<T>(col < p.Cout ? src[col] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// naive fallbacks (any channel count; also an in-library cross-check of the MFMA path)
// ------------------------------------------------------------------------------------------------
struct NaiveConvParams {
    const void* x;
    const float* w;
    const void* dy;
    void* y;
    void* dx;
    float* dw;
    const float* bias;
    int N, H, W, Cin, Cout, KH, KW, SH, SW, DH, DW, padT, padL, OH, OW, x_cs;
    int accumulate;
    float scale;
};
template <typename T>
__global__ void naive_conv_fwd(const NaiveConvParams p) {
    const long total = (long)p.N * p.OH * p.OW * p.Cout;
    const T* x = reinterpret_cast<const T*>(p.x);
    T* y = reinterpret_cast<T*>(p.y);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % p.Cout);
        long r = idx / p.Cout;
        const int ox = (int)(r % p.OW);
        r /= p.OW;
        const int oy = (int)(r % p.OH), n = (int)(r / p.OH);
        float acc = 0.f;
        for (int kr = 0; kr < p.KH; ++kr) {
            const int iy = oy * p.SH + kr * p.DH - p.padT;
            if (iy < 0 || iy >= p.H) continue;
            for (int ks = 0; ks < p.KW; ++ks) {
                const int ix = ox * p.SW + ks * p.DW - p.padL;
                if (ix < 0 || ix >= p.W) continue;
                const T* xp = x + (((long)n * p.H + iy) * p.W + ix) * p.x_cs;
                const float* wp = p.w + ((long)(kr * p.KW + ks) * p.Cin) * p.Cout + k;
                for (int c = 0; c < p.Cin; ++c) acc = fmaf(to_f32(xp[c]), to_f32(from_f32<T>(wp[(long)c * p.Cout])), acc);
            }
        }
        if (p.bias) acc += p.bias[k];
        y[idx] = from_f32<T>(acc);
    }
}
template <typename T>
__global__ void naive_conv_dgrad(const NaiveConvParams p) {
    const long total = (long)p.N * p.H * p.W * p.Cin;
    const T* dy = reinterpret_cast<const T*>(p.dy);
    T* dx = reinterpret_cast<T*>(p.dx);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % p.Cin);
        long r = idx / p.Cin;
        const int ix = (int)(r % p.W);
        r /= p.W;
        const int iy = (int)(r % p.H), n = (int)(r / p.H);
        float acc = 0.f;
        for (int kr = 0; kr < p.KH; ++kr) {
            const int ty = iy + p.padT - kr * p.DH;
            if (ty < 0 || ty % p.SH) continue;
            const int oy = ty / p.SH;
            if (oy >= p.OH) continue;
            for (int ks = 0; ks < p.KW; ++ks) {
                const int tx = ix + p.padL - ks * p.DW;
                if (tx < 0 || tx % p.SW) continue;
                const int ox = tx / p.SW;
                if (ox >= p.OW) continue;
                const T* dp = dy + (((long)n * p.OH + oy) * p.OW + ox) * p.Cout;
                const float* wp = p.w + ((long)(kr * p.KW + ks) * p.Cin + c) * p.Cout;
                for (int k = 0; k < p.Cout; ++k) acc = fmaf(to_f32(dp[k]), to_f32(from_f32<T>(wp[k])), acc);
            }
        }
        if (p.accumulate) acc += to_f32(dx[idx]);
        dx[idx] = from_f32<T>(acc);
    }
}
// Fallback wgrad (channel counts that are not chunk multiples: the class-logit conv of the segmentation head, the
// squeeze-excite convs).  blockDim = (32 weight elements, 8 lanes); the (image, output row) pairs are cut into gridDim.y
// slices, the 8 lanes of a block stride over their slice and fold through LDS; slice partials go to `part` and are summed
// in a fixed order by naive_wgrad_reduce (deterministic).  With one slice the result is written directly.
template <typename T>
__global__ __launch_bounds__(256) void naive_conv_wgrad(const NaiveConvParams p, float* __restrict__ part) {
    __shared__ float red[8][32];
    const long total = (long)p.KH * p.KW * p.Cin * p.Cout;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* dy = reinterpret_cast<const T*>(p.dy);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long nrows = (long)p.N * p.OH;
    const long rps = (nrows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * rps, r1 = min(nrows, r0 + rps);
    for (long base = (long)blockIdx.x * 32; base < total; base += (long)gridDim.x * 32) {
        const long idx = base + tx;
        float acc = 0.f;
        if (idx < total) {
            const int k = (int)(idx % p.Cout);
            long r = idx / p.Cout;
            const int c = (int)(r % p.Cin);
            r /= p.Cin;
            const int ks = (int)(r % p.KW), kr = (int)(r / p.KW);
            for (long row = r0 + ty; row < r1; row += 8) {
                const int n = (int)(row / p.OH), oy = (int)(row - (long)n * p.OH);
                const int iy = oy * p.SH + kr * p.DH - p.padT;
                if (iy < 0 || iy >= p.H) continue;
                for (int ox = 0; ox < p.OW; ++ox) {
                    const int ix = ox * p.SW + ks * p.DW - p.padL;
                    if (ix < 0 || ix >= p.W) continue;
                    acc = fmaf(to_f32(x[(((long)n * p.H + iy) * p.W + ix) * p.x_cs + c]),
                               to_f32(dy[(((long)n * p.OH + oy) * p.OW + ox) * p.Cout + k]), acc);
                }
            }
        }
        red[ty][tx] = acc;
        __syncthreads();
        if (ty == 0 && idx < total) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += red[j][tx];
            if (gridDim.y == 1) p.dw[idx] = s * p.scale;
            else part[(long)blockIdx.y * total + idx] = s;
        }
        __syncthreads();
    }
}
// 32 elements x 8 slice lanes per block, fixed order
__global__ __launch_bounds__(256) void naive_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, long total, int slices, float scale) {
    __shared__ float red[256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + tx;
    float s = 0.f;
    if (i < total)
        for (int k = ty; k < slices; k += 8) s += part[(long)k * total + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + tx];
        dw[i] = t * scale;
    }
}

// ------------------------------------------------------------------------------------------------
// "Skinny" 1x1 convolutions: few output channels that are not a chunk multiple on top of a chunked input (the class-logit
// conv of the segmentation head: 256 -> 19 at 129x129).  One thread per pixel; the weights sit in a zero-padded fp32 table
// [Cin][CO] in the workspace (rounded per use in bf16 mode, like every weight), read with wave-uniform (scalar) loads, so
// the inner loop is one v_fmac per (channel, output) with an SGPR operand.  VALU bound: 2*M*Cin*CO flops at fp32 rate.
// ------------------------------------------------------------------------------------------------
// wp[c][n] (c < Cin "many", n < CO "few", zero beyond Cout) from w[c][n] (row stride Cout), or — transposed — from
// w[n][c] (row stride Cin): the roles of the two channel counts swap for a conv with FEW INPUT channels (SE expand convs)
template <typename T>
__global__ void skinny_pack_w(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CO, int transposed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * CO) return;
    const int c = i / CO, n = i - c * CO;
    wp[i] = n < Cout ? to_f32(from_f32<T>(transposed ? w[(long)n * Cin + c] : w[(long)c * Cout + n])) : 0.f;
}
template <typename T, int CO>
__global__ __launch_bounds__(256) void skinny_conv_fwd(const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                                       T* __restrict__ y, long M, int Cin, int Cout, int accumulate) {
    constexpr int CE = VecTraits<T>::CE;
    const int nch = Cin / CE;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < M; pix += (long)gridDim.x * blockDim.x) {
        float acc[CO];
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[n] = 0.f;
        const T* xr = x + pix * Cin;
        for (int ch = 0; ch < nch; ++ch) {
            const Chunk<T> c = load_chunk<T>(xr + ch * CE);
            const float* wr = wp + (long)ch * CE * CO;            // wave-uniform address: scalar loads
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                const float xv = c.get(e);
#pragma unroll
                for (int n = 0; n < CO; ++n) acc[n] = fmaf(xv, wr[e * CO + n], acc[n]);
            }
        }
        T* yr = y + pix * Cout;
#pragma unroll
        for (int n = 0; n < CO; ++n)
            if (n < Cout) yr[n] = from_f32<T>(acc[n] + (bias ? bias[n] : 0.f) + (accumulate ? to_f32(yr[n]) : 0.f));
    }
}
// the same for FEW pixels (squeeze-excite bottlenecks: M = batch, Cin up to 1152): one WAVE per pixel, lane = slice of the
// channel chunks (coalesced x row), butterfly fold over the wave.  The thread-per-pixel form above ran M = 512 threads
// through 84-144 chunks each: 60-100 us of pure latency per call.
template <typename T, int CO>
__global__ __launch_bounds__(256) void skinny_conv_fwd_split(const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                                             T* __restrict__ y, long M, int Cin, int Cout, int accumulate) {
    constexpr int CE = VecTraits<T>::CE;
    static_assert(CO % 4 == 0, "weight rows are read as float4");
    const int nch = Cin / CE;
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= M) return;                                          // (whole waves leave: no barrier below)
    float acc[CO];
#pragma unroll
    for (int n = 0; n < CO; ++n) acc[n] = 0.f;
    const T* xr = x + pix * Cin;
    for (int ch = lane; ch < nch; ch += 64) {
        const Chunk<T> c = load_chunk<T>(xr + ch * CE);
        const float4* wr = reinterpret_cast<const float4*>(wp + (long)ch * CE * CO);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const float xv = c.get(e);
#pragma unroll
            for (int n4 = 0; n4 < CO / 4; ++n4) {
                const float4 w4 = wr[e * (CO / 4) + n4];
                acc[n4 * 4] = fmaf(xv, w4.x, acc[n4 * 4]);
                acc[n4 * 4 + 1] = fmaf(xv, w4.y, acc[n4 * 4 + 1]);
                acc[n4 * 4 + 2] = fmaf(xv, w4.z, acc[n4 * 4 + 2]);
                acc[n4 * 4 + 3] = fmaf(xv, w4.w, acc[n4 * 4 + 3]);
            }
        }
    }
    T* yr = y + pix * Cout;
#pragma unroll
    for (int n = 0; n < CO; ++n) {
        float v = acc[n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && n < Cout) yr[n] = from_f32<T>(v + (bias ? bias[n] : 0.f) + (accumulate ? to_f32(yr[n]) : 0.f));
    }
}
// dx[pixel][c] = sum_n dy[pixel][n] * w[c][n]  (accumulate: += the old value)
template <typename T, int CO>
__global__ __launch_bounds__(256) void skinny_conv_dgrad(const T* __restrict__ dy, const float* __restrict__ wp, T* __restrict__ dx, long M, int Cin,
                                                         int Cout, int accumulate) {
    constexpr int CE = VecTraits<T>::CE;
    const int nch = Cin / CE;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < M; pix += (long)gridDim.x * blockDim.x) {
        float g[CO];
        const T* dr = dy + pix * Cout;
#pragma unroll
        for (int n = 0; n < CO; ++n) g[n] = n < Cout ? to_f32(dr[n]) : 0.f;
        T* xr = dx + pix * Cin;
        for (int ch = blockIdx.y; ch < nch; ch += gridDim.y) {   // (gridDim.y > 1 when there are few pixels: SE bottlenecks, M = batch)
            const float* wr = wp + (long)ch * CE * CO;
            Chunk<T> o;
            Chunk<T> old;
            if (accumulate) old = load_chunk<T>(xr + ch * CE);
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                float a = 0.f;
#pragma unroll
                for (int n = 0; n < CO; ++n) a = fmaf(g[n], wr[e * CO + n], a);
                if (accumulate) a += old.get(e);
                o.set(e, a);
            }
            store_chunk<T>(xr + ch * CE, o);
        }
    }
}
// dw[c][n] = sum_pixels x[pixel][c] * dy[pixel][n].  blockIdx.x = group of 8 channel chunks, blockIdx.y = pixel slab.  Lanes
// 0-7 of every 8 hold the 8 chunks of ONE pixel (a 128-byte line of x per 8 lanes: the earlier one-chunk-per-block form
// fetched every line of x 8 times and read dy with 2-byte loads 42 bytes apart, 478 us for the 129x129 class-logit conv);
// the dy rows of 256 pixels are staged through LDS as zero-padded fp32 [pixel][CO] with coalesced loads and read back as
// 16-byte broadcasts.  Per-thread accumulators [CE][CO], folded over the 8 pixel lanes of a wave by shuffles and over the
// block's waves through LDS; slab partials are summed in a fixed order by naive_wgrad_reduce (deterministic).
template <typename T, int CO>
__global__ __launch_bounds__(256) void skinny_conv_wgrad(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, long M, int Cin,
                                                         int Cout, long slab, int transposed) {
    constexpr int CE = VecTraits<T>::CE;
    static_assert(CO % 4 == 0, "CO rows are read as float4");
    __shared__ __attribute__((aligned(16))) float sm[256 * CO];   // dy stage [256][CO]; afterwards the wave partials [4][8][CE*CO]
    const int nch = Cin / CE;
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int ch = blockIdx.x * 8 + cl;
    const bool chok = ch < nch;
    const long p0 = (long)blockIdx.y * slab, p1 = min(M, p0 + slab);
    float acc[CE][CO];
#pragma unroll
    for (int e = 0; e < CE; ++e)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[e][n] = 0.f;
    for (int i = threadIdx.x; i < 256 * CO; i += 256) sm[i] = 0.f;            // the pad columns stay zero
    for (long q0 = p0; q0 < p1; q0 += 256) {
        const int np = (int)min((long)256, p1 - q0);
        __syncthreads();
        const T* dq = dy + q0 * Cout;
        const int nit = (np + 31) >> 5;
        for (int i = threadIdx.x; i < nit * 32 * Cout; i += 256) {            // contiguous span of dy: coalesced
            const int px = i / Cout, n = i - px * Cout;
            sm[px * CO + n] = px < np ? to_f32(dq[i]) : 0.f;
        }
        __syncthreads();
        // x chunks two iterations ahead (2 waves per SIMD at 192 accumulators: nothing else hides the load latency)
        auto ldx = [&](int it) {
            const int q = it * 32 + pl;
            return load_chunk<T>(x + (q0 + ((chok && q < np) ? q : 0)) * Cin + (chok ? ch : 0) * CE);
        };
        Chunk<T> c0 = ldx(0), c1 = ldx(1);
        for (int it = 0; it < nit; ++it) {
            const int px = it * 32 + pl;
            const bool ok = chok && px < np;
            Chunk<T> c = c0;
            c0 = c1;
            c1 = ldx(it + 2);
            if (!ok) c = Chunk<T>{};
            float g[CO];
#pragma unroll
            for (int n4 = 0; n4 < CO / 4; ++n4) {
                const float4 v = *reinterpret_cast<const float4*>(&sm[px * CO + n4 * 4]);
                g[n4 * 4] = v.x; g[n4 * 4 + 1] = v.y; g[n4 * 4 + 2] = v.z; g[n4 * 4 + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                const float xv = c.get(e);
#pragma unroll
                for (int n = 0; n < CO; ++n) acc[e][n] = fmaf(xv, g[n], acc[e][n]);
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < CE; ++e)
#pragma unroll
        for (int n = 0; n < CO; ++n) {
            float v = acc[e][n];
#pragma unroll
            for (int o = 32; o >= 8; o >>= 1) v += __shfl_xor(v, o);
            if (lane < 8) sm[(wave * 8 + lane) * (CE * CO) + e * CO + n] = v;
        }
    __syncthreads();
    const long total = (long)Cin * Cout;
    for (int i = threadIdx.x; i < 8 * CE * CO; i += 256) {
        const int c8 = i / (CE * CO), r = i - c8 * (CE * CO);
        const int e = r / CO, n = r - e * CO;
        const int chn = blockIdx.x * 8 + c8;
        // transposed: the caller swapped the operands (few INPUT channels): element (many = chn*CE+e, few = n) belongs at [few][many]
        if (chn < nch && n < Cout)
            part[(long)blockIdx.y * total + (transposed ? (long)n * Cin + (chn * CE + e) : (long)(chn * CE + e) * Cout + n)] =
                (sm[i] + sm[8 * CE * CO + i]) + (sm[2 * 8 * CE * CO + i] + sm[3 * 8 * CE * CO + i]);
    }
}

// column sums of a [M][C] matrix -> fp32 [C] (bias gradient); two-stage, deterministic.  A block is TX column lanes x
// 256/TX row lanes (TX = C rounded up to a power of two, at most 256), so narrow matrices (21 class logits) still use every
// thread; the row lanes are folded through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ part, long M, int C, int rows_per_block, int TX) {
    __shared__ float red[256];
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = blockIdx.x * TX + tx;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        long r = r0 + ty;
        for (; r + 3 * TY < r1; r += 4 * TY) {
            const T v0 = x[r * C + c], v1 = x[(r + TY) * C + c], v2 = x[(r + 2 * TY) * C + c], v3 = x[(r + 3 * TY) * C + c];
            s0 += to_f32(v0); s1 += to_f32(v1); s2 += to_f32(v2); s3 += to_f32(v3);
        }
        for (; r < r1; r += TY) s0 += to_f32(x[r * C + c]);
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && c < C) {
        float s = 0.f;
        for (int k = 0; k < TY; ++k) s += red[k * TX + tx];
        part[(long)blockIdx.y * C + c] = s;
    }
}
// 32 columns x 8 part lanes per block
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int C, float scale) {
    __shared__ float red[256];
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    const int c = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < C)
        for (int k = ty; k < nparts; k += 8) s += part[(long)k * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + tx];
        out[c] = t * scale;
    }
}
