// Winograd F(2x2, 3x3) convolution for fp32 (3x3, stride 1, dilation 1, SAME): minimal filtering (Lavin & Gray 2016) on the MFMA path.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile / 4x4 input patch, summed over input channels
//
// 16 multiplications per 2x2 outputs instead of 36: the channel sum of each of the 16 "frequencies" f is a GEMM
// M_f[cout][tile] = sum_c U_f[cout][c] V_f[tile][c], all 16 of them accumulated by ONE workgroup for its (64 tiles x 64 couts) block, so
// that the output transform is lane-local: a lane of the 32x32 MFMA accumulator holds the same (tile, cout) position in all 16 accumulators.
// The input transform sits on the staging path (global -> registers -> 32 adds per 4-channel chunk -> LDS), the filter transform is a small
// launch of its own per step (U, 16/9 of the filter bytes).
//
// The block of a workgroup is 64 tiles x 64 couts x 16 frequencies = 256 accumulator registers per lane of a 32 x 32 wave tile: one
// workgroup per CU, eight waves (conv_wino_f2k3_w8 below; the first, 4-wave form — all 16 frequencies in one wave, one wave per SIMD — is
// kept in profiles/probes/wino_w4_kernel.h).
#pragma once
#include <type_traits>
#include "common.h"
#include "conv_kernels.h"

struct WinoParams {
    const float* in;        // NHWC, channel stride Cs
    const float* u;         // transformed filter: [ceil(Nn/64)][Cin/32][r 4][f 4][chunk 8][64 cout][4 c]
    float* out;             // NHWC, channel stride ldo
    const float* bias;      // optional [Nn]
    int H, W, Cs, Cin;      // Cin: multiple of 32
    int TH, TW, ntiles;     // tile grid per image, N*TH*TW
    int Nn, ldo;
    unsigned in_bytes, u_bytes, out_bytes;
    float* stats;           // NT_EPI_STATS: counted rows [4 * tile blocks][4][Nn]
    const float* red_x;     // NT_EPI_BNRED (as in GemmNTParams)
    const unsigned char* red_mask;
    float* red_part;
    int red_row0;
    // K-sliced tail (as conv_gemm_nt's stream-K tail): workgroups >= sk_body compute one of sk_slices slices of the 32-channel super-steps of a
    // tail block each and park their accumulators in `partial` ([slice block][wave 8][register 128][lane 64]); a second launch of the kernel
    // with DBG = WINO_REDUCE sums the slices of each tail block in a fixed order and runs the epilogue
    float* partial;
    int sk_body, sk_slices;
    int dbg;                // DEBUG probes (timing only, wrong results): 1 = no staging after the first K-step, 2 = no MFMAs, 4 = no fragment reads
};

#define WINO_OOB 0x80000000u
#define WINO_REDUCE 64      // DBG bit of conv_wino_f2k3_w8: no K loop, accumulators = the sum of a tail block's parked slices
#define WINO_SLICE 128      // DBG bit: the workgroups of this launch are K-slices of the tail blocks (park the accumulators, no epilogue)

template <int N, typename F, int I = 0>
__device__ __forceinline__ void wino_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wino_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

__device__ __forceinline__ f32x4 wino_ld16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

// U = G g G^T of HWIO weights w[3][3][Cin][Cout] into the kernel's operand layout [nb][Kin/32][r 4][f 4][chunk 8][64 cout][4 c]
// (frequency F = 4 r + f; TRANSPOSED: the dgrad's filter — rotated by 180 degrees, channel roles swapped: Kin = Cout, Kout = Cin)
__device__ __forceinline__ void wino_u_elements(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, bool transposed, long id) {
    const int Kin = transposed ? Cout : Cin, Kout = transposed ? Cin : Cout;
    const int ns = Kin / 32, nb = (Kout + 63) / 64;
    const long total = (long)nb * ns * 8 * 64 * 4;
    if (id >= total) return;
    // 256 consecutive ids = one (64 couts x 4 channels) chunk of U; the HWIO filter is contiguous along Cout: forward reads run along the
    // 64 couts (co fastest), dgrad reads along the 4 channels (its Kin is the filter's Cout)
    const int t = (int)(id & 255);
    const int c4 = transposed ? (t & 3) : (t >> 6), co = transposed ? (t >> 2) : (t & 63);
    const int ch = (int)((id >> 8) & 7);
    const long rest = id >> 11;
    const int sidx = (int)(rest % ns), b = (int)(rest / ns);
    const int ci = sidx * 32 + ch * 4 + c4, n = b * 64 + co;
    float gg[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v = 0.f;
            if (n < Kout) v = transposed ? w[(((2 - r) * 3 + (2 - q)) * Cin + n) * (long)Cout + ci] : w[((r * 3 + q) * Cin + ci) * (long)Cout + n];
            gg[r][q] = v;
        }
    float t3[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        t3[0][q] = gg[0][q];
        t3[1][q] = 0.5f * (gg[0][q] + gg[1][q] + gg[2][q]);
        t3[2][q] = 0.5f * (gg[0][q] - gg[1][q] + gg[2][q]);
        t3[3][q] = gg[2][q];
    }
    float* dst = u + ((long)b * ns + sidx) * (16 * 2048) + (ch * 64 + co) * 4 + c4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dst[(r * 4 + 0) * 2048] = t3[r][0];
        dst[(r * 4 + 1) * 2048] = 0.5f * (t3[r][0] + t3[r][1] + t3[r][2]);
        dst[(r * 4 + 2) * 2048] = 0.5f * (t3[r][0] - t3[r][1] + t3[r][2]);
        dst[(r * 4 + 3) * 2048] = t3[r][2];
    }
}
__global__ __launch_bounds__(256) void wino_filter_transform_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int transposed) {
    wino_u_elements(w, u, Cin, Cout, transposed != 0, (long)blockIdx.x * 256 + threadIdx.x);
}

// LDS stage = the operands of ONE frequency row r (4 frequencies) x 32 input channels:
//   V [f 4][chunk 8][64 tiles][16 B]   chunk stride 1056 B (bank-conflict-free 16-byte writes with lanes = 8 chunks x 8 tiles)
//   U [f 4][chunk 8][64 couts][16 B]   the same chunk stride: an LDS-DMA instruction writes one chunk (1 KB); with the global image's 1024-byte
//                                      stride lanes l and l + 32 of a fragment read hit the same banks (PMC: one conflict cycle per MFMA)
constexpr int WINO_VCH = 1056, WINO_VF = 8 * WINO_VCH, WINO_VBYTES = 4 * WINO_VF, WINO_UBYTES = 32768;      // (UBYTES: bytes of a stage's U in global memory)
constexpr int WINO_STAGE = 2 * WINO_VBYTES;

// ------------------------------------------------------------------------------------------------------------------------------------
// K loop: stages q = 4 s + r (s = 32-channel super-step, r = frequency row).  A stage reads whole 128-byte lines of the input (8 lanes =
// the 8 chunks of one pixel) — with 8-channel stages every line was fetched four times over and L2 -> CU traffic bound the kernel.  The
// stage is software-pipelined three deep inside ONE instruction stream, one staging "slot" behind every MFMA (an instruction that waits
// at issue — a buffer load behind a full address queue — starves the MFMA pipe, which holds no more than the instruction in flight):
//   stage q:  MFMAs on LDS buffer q&1  |  U of stage q+1 by LDS-DMA  |  input transform of stage q+1 (registers loaded during stage q-1) ->
//             ds_write into buffer (q+1)&1  |  global loads of stage q+2 re-issued into the registers the transform has just consumed
//   ONE barrier per stage, late in the stage: every read of this stage's buffer has returned, every write into the other buffer is
//   complete (own DMAs: vmcnt, own ds_writes: lgkmcnt); the next stage's first fragments are read behind it, before the stage ends.
// conv_wino_f2k3_w8: the block (64 tiles x 64 couts x 16 frequencies, the same LDS stages) on EIGHT waves — two per SIMD, each owning
// two COLUMNS of the 4 x 4 frequency grid (f = 2 fh + {0, 1} of every frequency row: 128 accumulators).  One wave per SIMD issues an
// instruction every 5-8 clocks and nothing runs under its waits; two waves fill each other's issue gaps.  A wave's half of the output
// transform: S = A^T M over its two columns, then the partial 2x2 outputs P[a][b] (fh 0: S0 + S1, S1; fh 1: S2, -S2 - S3); wave fh keeps
// output row a = fh and hands the other row to its partner (wave ^ 4) through LDS once per block.  Statistics / BN-backward partial rows:
// one per (tile block, wave column, fh) — 4 per tile block.
template <int DBG, int EPI>
__global__ __launch_bounds__(512, 1) void conv_wino_f2k3_w8(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int ntn = (p.Nn + 63) >> 6, ntm = (p.ntiles + 63) >> 6;
    // three instantiations per epilogue (compile-time: the slice bookkeeping in the plain kernel cost 17 spilled registers and 9 % of the
    // 56 x 56 layer): plain = the whole blocks 0 .. sk_body-1; SLICE = workgroup (tail block, slice) of a second launch; REDUCE = the third
    constexpr bool REDUCE = (DBG & WINO_REDUCE) != 0, SLICE = (DBG & WINO_SLICE) != 0;
    const int sk_rel = REDUCE ? (int)blockIdx.x * p.sk_slices : (int)blockIdx.x;
    const int L = REDUCE ? p.sk_body + (int)blockIdx.x : (SLICE ? p.sk_body + sk_rel / p.sk_slices : xcd_remap(blockIdx.x, p.sk_body));
    const int t0 = (L / ntn) * 64, nb = L % ntn, n0 = nb * 64;
    const int ns_all = p.Cin >> 5;
    // super-step range of this workgroup: everything, or one slice of a tail block
    int s0 = 0, ns = ns_all;
    if constexpr (SLICE) {
        const int per = (ns_all + p.sk_slices - 1) / p.sk_slices, sl = sk_rel % p.sk_slices;
        s0 = sl * per;
        ns = min(ns_all, s0 + per);
    }
    (void)ntm;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)p.u_bytes, 0x00020000);

    // ---- loader role: ONE item per thread (tile = tid / 8, chunk j = tid & 7): 8 lanes read one 128-byte line
    const int lj = tid & 7, ltile = tid >> 3;
    unsigned poff[4][4];
    {
        const int T = t0 + ltile;
        const int tpi = p.TH * p.TW;
        const int img = T / tpi, rem = T - img * tpi;
        const int th = rem / p.TW, tw = rem - th * p.TW;
        const bool tv = T < p.ntiles;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int y = 2 * th - 1 + r, x = 2 * tw - 1 + c;
                const bool ok = tv && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                poff[r][c] = ok ? (unsigned)((((img * p.H + y) * p.W + x) * p.Cs + lj * 4) * 4) : WINO_OOB;
            }
    }
    const unsigned ubase = (unsigned)(((long)nb * ns_all) * (4 * WINO_UBYTES) + lane * 16);
    __attribute__((address_space(3))) char* const wbase = (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * WINO_VCH);

    f32x4 raw[2][4], tt[4];
    auto load_raw = [&](auto rc, int sidx) {
        constexpr int r = decltype(rc)::value;
        constexpr int ia = r == 0 ? 0 : (r == 2 ? 2 : 1), ib = r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
        const unsigned co = (unsigned)(sidx * 128);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            raw[0][c] = wino_ld16(rsX, poff[ia][c] + co);
            raw[1][c] = wino_ld16(rsX, poff[ib][c] + co);
        }
    };
    auto rows = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) tt[c] = r == 1 ? raw[0][c] + raw[1][c] : raw[0][c] - raw[1][c];
    };
    auto cols = [&](int buf) {
        char* const vs = smem + buf * WINO_STAGE + lj * WINO_VCH + ltile * 16;
        *reinterpret_cast<f32x4*>(vs + 0 * WINO_VF) = tt[0] - tt[2];
        *reinterpret_cast<f32x4*>(vs + 1 * WINO_VF) = tt[1] + tt[2];
        *reinterpret_cast<f32x4*>(vs + 2 * WINO_VF) = tt[2] - tt[1];
        *reinterpret_cast<f32x4*>(vs + 3 * WINO_VF) = tt[1] - tt[3];
    };
    auto dma_u = [&](auto bufc, int q) {                                           // U of stage q -> buffer buf: 4 x 1 KB per wave
        constexpr int buf = decltype(bufc)::value;
        const unsigned go = ubase + (unsigned)q * WINO_UBYTES + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(wbase + (buf * WINO_STAGE + WINO_VBYTES + i * WINO_VF)), 16,
                                                     (int)(go + (unsigned)(i * 8192)), 0, 0, 0);
    };

    f32x16 acc[8];                                                                 // [frequency row r][column j of this wave's two]
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    const int fragA = (lane >> 5) * WINO_VCH + (wm * 32 + (lane & 31)) * 16 + WINO_VBYTES + fh * 2 * WINO_VF;
    const int fragB = (lane >> 5) * WINO_VCH + (wn * 32 + (lane & 31)) * 16 + fh * 2 * WINO_VF;
    f32x4 fa[2][2], fb[2][2];                                                      // [set][column j]; group g = chunk pair i
    auto read_frags = [&](const char* st, int g, int set) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fa[set][j] = *reinterpret_cast<const f32x4*>(st + fragA + j * WINO_VF + g * 2 * WINO_VCH);
            fb[set][j] = *reinterpret_cast<const f32x4*>(st + fragB + j * WINO_VF + g * 2 * WINO_VCH);
        }
    };
    // one stage: 32 MFMAs (4 groups = chunk pairs, 2 columns x 4 elements each), each followed by one slot:
    //   0 B^T d of stage q+1 | 1-4 the wave's 4 DMAs of U(q+1) | 5-12 the 8 loads of stage q+2 | 13-16 (.)B + 4 ds_writes |
    //   8g .. 8g+3 fragments of group g+1 | 26 the barrier | 28-31 the next stage's first fragments
    auto stage = [&](auto rc, auto nc, int sidx) {
        constexpr int R = decltype(rc)::value, R1 = (R + 1) & 3, R2 = (R + 2) & 3;
        constexpr bool next = (decltype(nc)::value & 1) && !(DBG & 1), next2 = (decltype(nc)::value & 2) && !(DBG & 1);
        constexpr int ia = R2 == 0 ? 0 : (R2 == 2 ? 2 : 1), ib = R2 == 0 ? 2 : (R2 == 1 ? 2 : (R2 == 2 ? 1 : 3));
        const int q = sidx * 4 + R;
        const char* const st = smem + (R & 1) * WINO_STAGE;
        const char* const sn = smem + ((R + 1) & 1) * WINO_STAGE;
        char* const nv = smem + ((R + 1) & 1) * WINO_STAGE + lj * WINO_VCH + ltile * 16;
        const unsigned co2 = (unsigned)((sidx + (R >= 2 ? 1 : 0)) * 128);
        const unsigned go = ubase + (unsigned)(q + 1) * WINO_UBYTES + (unsigned)wave * 1024u;
        wino_for<32>([&](auto mc) {
            constexpr int m = decltype(mc)::value, g = m >> 3, w = m & 7, e = w >> 1, j = w & 1;
            if constexpr (!(DBG & 2))
                acc[R * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][j][e], fb[g & 1][j][e], acc[R * 2 + j], 0, 0, 0);
            if constexpr (g < 3 && w < 4) {
                constexpr int g1 = g + 1, jj = w >> 1;
                if constexpr ((w & 1) == 0) fa[g1 & 1][jj] = *reinterpret_cast<const f32x4*>(st + fragA + jj * WINO_VF + g1 * 2 * WINO_VCH);
                else fb[g1 & 1][jj] = *reinterpret_cast<const f32x4*>(st + fragB + jj * WINO_VF + g1 * 2 * WINO_VCH);
            }
            if constexpr (next) {
                if constexpr (m == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) tt[c] = R1 == 1 ? raw[0][c] + raw[1][c] : raw[0][c] - raw[1][c];
                } else if constexpr (m >= 1 && m < 5) {
                    constexpr int i = m - 1;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(wbase + (((R + 1) & 1) * WINO_STAGE + WINO_VBYTES + i * WINO_VF)),
                                                             16, (int)(go + (unsigned)(i * 8192)), 0, 0, 0);
                } else if constexpr (m >= 5 && m < 13) {
                    constexpr int l = m - 5, h = l >> 2, c = l & 3;
                    if constexpr (next2) raw[h][c] = wino_ld16(rsX, poff[h ? ib : ia][c] + co2);
                } else if constexpr (m >= 13 && m < 17) {
                    constexpr int f = m - 13;
                    const f32x4 v = f == 0 ? tt[0] - tt[2] : (f == 1 ? tt[1] + tt[2] : (f == 2 ? tt[2] - tt[1] : tt[1] - tt[3]));
                    *reinterpret_cast<f32x4*>(nv + f * WINO_VF) = v;
                } else if constexpr (m == 26) {
                    if constexpr (next2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");      // the DMAs are older than the 8 loads
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else if constexpr (m >= 28) {
                    constexpr int jj = (m - 28) >> 1;
                    if constexpr ((m & 1) == 0) fa[0][jj] = *reinterpret_cast<const f32x4*>(sn + fragA + jj * WINO_VF);
                    else fb[0][jj] = *reinterpret_cast<const f32x4*>(sn + fragB + jj * WINO_VF);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    if constexpr (!REDUCE) {
        if (!SLICE || s0 < ns) {                                   // (a slice past the end of K: nothing to add)
            dma_u(I0{}, s0 * 4);
            load_raw(I0{}, s0);
            rows(I0{});
            cols(0);
            load_raw(I1{}, s0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            read_frags(smem, 0, 0);
            for (int sidx = s0; sidx < ns - 1; ++sidx) {
                stage(I0{}, I3{}, sidx);
                stage(I1{}, I3{}, sidx);
                stage(I2{}, I3{}, sidx);
                stage(I3{}, I3{}, sidx);
            }
            stage(I0{}, I3{}, ns - 1);
            stage(I1{}, I3{}, ns - 1);
            stage(I2{}, I1{}, ns - 1);
            stage(I3{}, I0{}, ns - 1);
        }
        if constexpr (SLICE) {                                     // park the accumulators: the reduce launch runs the epilogue
            float* dst = p.partial + ((size_t)(sk_rel * 8 + wave) * 128) * 64 + lane;
#pragma unroll
            for (int f = 0; f < 8; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(f * 16 + r) * 64] = acc[f][r];
            return;
        }
    } else {
        for (int sl = 0; sl < p.sk_slices; ++sl) {
            const float* src = p.partial + ((size_t)((sk_rel + sl) * 8 + wave) * 128) * 64 + lane;
#pragma unroll
            for (int f = 0; f < 8; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][r] += src[(f * 16 + r) * 64];
        }
    }

    // ---- output transform: this wave's partial outputs, exchange with the partner, then its output row a = fh
    constexpr bool STATS = EPI == NT_EPI_STATS, BNRED = EPI == NT_EPI_BNRED, ACC = EPI == NT_EPI_ACC;
    __syncthreads();                                               // every wave has left the K loop: the stage buffers are free
    float* const xch = reinterpret_cast<float*>(smem);             // [wave 8][value 32][lane 64]
    f32x4 keep[4][2];                                              // [q][b]: own row a = fh
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 P[2][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = q * 4 + e;
            // S[a][j] over the frequency rows of column j: a = 0: M0 + M1 + M2, a = 1: M1 - M2 - M3
            const float s00 = acc[0][r] + acc[2][r] + acc[4][r], s01 = acc[1][r] + acc[3][r] + acc[5][r];
            const float s10 = acc[2][r] - acc[4][r] - acc[6][r], s11 = acc[3][r] - acc[5][r] - acc[7][r];
            // fh 0 holds columns 0, 1: P[a][0] = S0 + S1, P[a][1] = S1; fh 1 holds columns 2, 3: P[a][0] = S2, P[a][1] = -S2 - S3
            P[0][0][e] = fh ? s00 : s00 + s01;
            P[0][1][e] = fh ? -s00 - s01 : s01;
            P[1][0][e] = fh ? s10 : s10 + s11;
            P[1][1][e] = fh ? -s10 - s11 : s11;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            keep[q][b] = fh ? P[1][b] : P[0][b];
            const f32x4 give = fh ? P[0][b] : P[1][b];
#pragma unroll
            for (int e = 0; e < 4; ++e) xch[(wave * 32 + (q * 2 + b) * 4 + e) * 64 + lane] = give[e];
        }
    }
    __syncthreads();
    const int T = t0 + wn * 32 + (lane & 31);
    const int tpi = p.TH * p.TW;
    const int img = T / tpi, rem = T - img * tpi;
    const int th = rem / p.TW, tw = rem - th * p.TW;
    const bool tv = T < p.ntiles;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    unsigned ooff[2];
    float npix = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int y = 2 * th + fh, x = 2 * tw + b;
        const bool ok = tv && y < p.H && x < p.W;
        ooff[b] = ok ? (unsigned)(((img * p.H + y) * p.W + x) * p.ldo * 4) : WINO_OOB;
        if constexpr (STATS) npix += (float)__builtin_popcount((unsigned)__builtin_amdgcn_ballot_w64(ok));
    }
    unsigned coff[4];
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + wm * 32 + 8 * q + 4 * (lane >> 5);
        coff[q] = n < p.Nn ? (unsigned)(n * 4) : WINO_OOB;
        bq[q] = (p.bias && n < p.Nn) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 xin[BNRED || ACC ? 4 : 1][2];
    unsigned char mb[BNRED ? 4 : 1][2];
    if constexpr (ACC) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 2; ++b) xin[q][b] = wino_ld16(rsO, __builtin_elementwise_add_sat(ooff[b], coff[q]));
    }
    if constexpr (BNRED) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.red_x), 0, (int)p.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.red_mask), 0, (int)(p.out_bytes >> 4), 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const unsigned bo = __builtin_elementwise_add_sat(ooff[b], coff[q]);
                xin[q][b] = wino_ld16(rsS, bo);
                mb[q][b] = (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsM, (int)(((bo & 0x7fffffffu) >> 4) | (bo & 0x80000000u)), 0, 0);
            }
    }
    float s1[16], s2[16], piv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 Y[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) Y[b][e] = keep[q][b][e] + xch[((wave ^ 4) * 32 + (q * 2 + b) * 4 + e) * 64 + lane] + bq[q][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (STATS) {
                const int v = __builtin_bit_cast(int, Y[0][e]);
                piv[q * 4 + e] = __builtin_bit_cast(float, lane < 32 ? __builtin_amdgcn_readlane(v, 0) : __builtin_amdgcn_readlane(v, 32));
            }
            if constexpr (STATS || BNRED) s1[q * 4 + e] = s2[q * 4 + e] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if constexpr (STATS) {
                const bool ok = !(ooff[b] >> 31);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = ok ? Y[b][e] - piv[q * 4 + e] : 0.f;
                    s1[q * 4 + e] += d;
                    s2[q * 4 + e] = __builtin_fmaf(d, d, s2[q * 4 + e]);
                }
            }
            if constexpr (BNRED) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = (mb[q][b] >> e) & 1 ? Y[b][e] : 0.f;
                    s1[q * 4 + e] += d;
                    s2[q * 4 + e] = __builtin_fmaf(d, xin[q][b][e], s2[q * 4 + e]);
                }
            }
            if constexpr (ACC) Y[b] += xin[q][b];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, Y[b]), rsO, (int)__builtin_elementwise_add_sat(ooff[b], coff[q]), 0, 0);
        }
    }
    if constexpr (STATS || BNRED) {
        float pv[16];
        if constexpr (STATS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = piv[r];
        }
        int base = 0;
        bool writer = true;
        LaneFold<16, 16>::run(s1, s2, lane, base, writer);
        const int prow = ((t0 >> 5) + wn) * 2 + fh;
        const int n = n0 + wm * 32 + 8 * (base >> 2) + 4 * (lane >> 5) + (base & 3);
        if (writer && n < p.Nn) {
            if constexpr (STATS) {
                p.stats[((long)prow * 4 + 0) * p.Nn + n] = s1[0];
                p.stats[((long)prow * 4 + 1) * p.Nn + n] = s2[0];
            } else {
                p.red_part[((long)(p.red_row0 + prow) * 2 + 0) * p.Nn + n] = s1[0];
                p.red_part[((long)(p.red_row0 + prow) * 2 + 1) * p.Nn + n] = s2[0];
            }
        }
        if constexpr (STATS) {
            if ((lane & 31) == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n2 = n0 + wm * 32 + 8 * q + 4 * (lane >> 5);
                    if (n2 < p.Nn) {
                        *reinterpret_cast<f32x4*>(p.stats + ((long)prow * 4 + 2) * p.Nn + n2) = f32x4{pv[q * 4], pv[q * 4 + 1], pv[q * 4 + 2], pv[q * 4 + 3]};
                        *reinterpret_cast<f32x4*>(p.stats + ((long)prow * 4 + 3) * p.Nn + n2) = f32x4{npix, npix, npix, npix};
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Winograd weight gradient, F(3x3, 2x2):   dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G
// The 16 frequency GEMMs dU_f[c][n] = sum_tiles V_f[tile][c] Z_f[tile][n] run over the TILES (K dimension); one workgroup owns a
// (64 channels x 64 couts) block of all 16 frequencies for one split of the tiles, applies G^T . G lane-locally (16 -> 9 taps, linear: it
// commutes with the sum over the splits) and writes its share of the split's slab in the layout of conv_gemm_tn ([split][tap * Cin + c][n]),
// so the slab reduce of the direct wgrad finishes the job.  Both operands are transformed on the staging path.
//
// LDS stage = one frequency row r (4 frequencies) x 32 tiles:  V [f 4][tile 32][64 c] and Z [f 4][tile 32][64 n] floats, rows of 256 bytes;
// odd tiles store their channels with bit 5 flipped (the two halves of a wave read tiles 2s and 2s+1 of the same 32 columns: without the
// flip both halves hit the same 32 banks).
struct WinoWgradParams {
    const float* x;         // NHWC, channel stride Cs
    const float* dy;        // NHWC, channel stride ldy
    float* slab;            // [splits][9 * Cin][Nn]
    int H, W, Cs, Cin, ldy, Nn;
    int TH, TW, ntiles;
    int tiles_per_split;    // multiple of 32
    int nbc, nbn;           // channel / cout blocks of 64
    unsigned x_bytes, dy_bytes;
};
constexpr int WINO_WG_HALF = 32768;                 // V or Z of one stage
constexpr int WINO_WG_STAGE = 2 * WINO_WG_HALF;
constexpr int WINO_WG_LDS_W8 = 8 * 72 * 64 * 4;   // conv_wino_wgrad_f3k2_w8: the epilogue's exchange area (> the two stages)

// conv_wino_wgrad_f3k2_w8: eight waves (two per SIMD), each owning two columns of the frequency grid — the first, 4-wave form (all 16
// frequencies in one wave: profiles/probes/wino_w4_wgrad_kernel.h) was bound by its own instruction issue, 480 instructions per stage in one stream.  Epilogue: a wave applies G^T over
// the rows and its two columns' share of (.) G, which gives 9 PARTIAL taps per accumulator register; wave fh finishes registers 8 fh ..
// 8 fh + 7 and hands the other eight to its partner (wave ^ 4) through LDS (72 floats per lane, once per block).
__global__ __launch_bounds__(512, 1) void conv_wino_wgrad_f3k2_w8(const WinoWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int nblk = p.nbc * p.nbn;
    const int L = xcd_remap(blockIdx.x, (int)gridDim.x);
    const int split = L / nblk, blk = L - split * nblk;
    const int c0 = (blk / p.nbn) * 64, n0 = (blk % p.nbn) * 64;
    const int tbeg = split * p.tiles_per_split;
    const int ngrp = p.tiles_per_split >> 5;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

    // ---- loader role: one item per thread (tile = tid / 16, chunk j = tid & 15) for x AND dy
    const int lj = tid & 15, ltile = tid >> 4;
    const unsigned cbad = c0 + lj * 4 < p.Cin ? 0u : WINO_OOB, nbad = n0 + lj * 4 < p.Nn ? 0u : WINO_OOB;
    int xbyte, ybyte, ymask;
    auto locate = [&](int grp) {
        const int T = tbeg + grp * 32 + ltile;
        const int tpi = p.TH * p.TW;
        const int img = T / tpi, rem = T - img * tpi;
        const int th = rem / p.TW, tw = rem - th * p.TW;
        const int y0 = 2 * th - 1, x0 = 2 * tw - 1;
        int m = 0;
        if (T < p.ntiles) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((unsigned)(y0 + r) < (unsigned)p.H) m |= 1 << r;
                if ((unsigned)(x0 + r) < (unsigned)p.W) m |= 16 << r;
            }
        }
        const int xbase = (img * p.H + y0) * p.W + x0;
        xbyte = (xbase * p.Cs + c0 + lj * 4) * 4;
        ybyte = ((xbase + p.W + 1) * p.ldy + n0 + lj * 4) * 4;
        ymask = m;
    };
    auto xoff = [&](int r, int c) -> unsigned {
        const unsigned ok = ((unsigned)ymask >> r) & ((unsigned)ymask >> (4 + c)) & 1u;
        return (unsigned)(xbyte + (r * p.W + c) * p.Cs * 4) | ((ok ^ 1u) << 31) | cbad;
    };
    auto yoff = [&](int a, int b) -> unsigned {
        const unsigned ok = ((unsigned)ymask >> (1 + a)) & ((unsigned)ymask >> (5 + b)) & 1u;
        return (unsigned)(ybyte + (a * p.W + b) * p.ldy * 4) | ((ok ^ 1u) << 31) | nbad;
    };

    f32x4 raw[2][4], rdy[2][2], tt[4], zz[2];
    auto load_stage = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int ia = r == 0 ? 0 : (r == 2 ? 2 : 1), ib = r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            raw[0][c] = wino_ld16(rsX, xoff(ia, c));
            raw[1][c] = wino_ld16(rsX, xoff(ib, c));
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (r != 3) rdy[0][b] = wino_ld16(rsY, yoff(0, b));
            if (r != 0) rdy[1][b] = wino_ld16(rsY, yoff(1, b));
        }
    };
    auto rows = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) tt[c] = r == 1 ? raw[0][c] + raw[1][c] : raw[0][c] - raw[1][c];
#pragma unroll
        for (int b = 0; b < 2; ++b) zz[b] = r == 0 ? rdy[0][b] : (r == 1 ? rdy[0][b] + rdy[1][b] : (r == 2 ? rdy[0][b] - rdy[1][b] : -rdy[1][b]));
    };
    const int lslot = ltile * 256 + ((lj ^ ((ltile & 1) * 8)) << 4);
    auto put_v = [&](int buf) {
        char* const vs = smem + buf * WINO_WG_STAGE + lslot;
        *reinterpret_cast<f32x4*>(vs + 0 * 8192) = tt[0] - tt[2];
        *reinterpret_cast<f32x4*>(vs + 1 * 8192) = tt[1] + tt[2];
        *reinterpret_cast<f32x4*>(vs + 2 * 8192) = tt[2] - tt[1];
        *reinterpret_cast<f32x4*>(vs + 3 * 8192) = tt[1] - tt[3];
    };
    auto put_z = [&](int buf) {
        char* const zs = smem + buf * WINO_WG_STAGE + WINO_WG_HALF + lslot;
        *reinterpret_cast<f32x4*>(zs + 0 * 8192) = zz[0];
        *reinterpret_cast<f32x4*>(zs + 1 * 8192) = zz[0] + zz[1];
        *reinterpret_cast<f32x4*>(zs + 2 * 8192) = zz[0] - zz[1];
        *reinterpret_cast<f32x4*>(zs + 3 * 8192) = -zz[1];
    };

    f32x16 acc[8];                                                                 // [frequency row][column j of this wave's two]
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    const int fragA = (lane >> 5) * 256 + (((wm * 32 + (lane & 31)) ^ ((lane >> 5) * 32)) << 2) + fh * 2 * 8192;
    const int fragB = (lane >> 5) * 256 + (((wn * 32 + (lane & 31)) ^ ((lane >> 5) * 32)) << 2) + WINO_WG_HALF + fh * 2 * 8192;
    float fa[2][2][4], fb[2][2][4];                                               // [set][column j][k-step of the group]

    locate(0);
    load_stage(I0{});
    rows(I0{});
    put_v(0); put_z(0);
    load_stage(I1{});
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            fa[0][j][s2] = *reinterpret_cast<const float*>(smem + fragA + j * 8192 + s2 * 512);
            fb[0][j][s2] = *reinterpret_cast<const float*>(smem + fragB + j * 8192 + s2 * 512);
        }

    // one stage = 32 MFMAs (4 groups of 4 k-steps x 2 columns), one slot behind each:
    //   0 transforms' first halves of stage q+1 | 2, 4 second halves + 8 ds_writes | 5 next tile group | 6-13 the 8 x loads of stage q+2 |
    //   14-17 its dy loads | fragments of group g+1: two reads per slot of group g | 26 barrier | 28-31 the next stage's first fragments
    auto stage = [&](auto rc, auto nc, int grp) {
        constexpr int R = decltype(rc)::value, R1 = (R + 1) & 3, R2 = (R + 2) & 3;
        constexpr bool next = decltype(nc)::value & 1, next2 = decltype(nc)::value & 2;
        constexpr int ia = R2 == 0 ? 0 : (R2 == 2 ? 2 : 1), ib = R2 == 0 ? 2 : (R2 == 1 ? 2 : (R2 == 2 ? 1 : 3));
        const char* const st = smem + (R & 1) * WINO_WG_STAGE;
        const char* const sn = smem + ((R + 1) & 1) * WINO_WG_STAGE;
        wino_for<32>([&](auto mc) {
            constexpr int m = decltype(mc)::value, g = m >> 3, w = m & 7, s = w >> 1, j = w & 1;
            acc[R * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][j][s], fb[g & 1][j][s], acc[R * 2 + j], 0, 0, 0);
            if constexpr (g < 3) {
                constexpr int g1 = g + 1;
                fa[g1 & 1][j][s] = *reinterpret_cast<const float*>(st + fragA + j * 8192 + (g1 * 4 + s) * 512);
                fb[g1 & 1][j][s] = *reinterpret_cast<const float*>(st + fragB + j * 8192 + (g1 * 4 + s) * 512);
            }
            if constexpr (next) {
                if constexpr (m == 0) rows(std::integral_constant<int, R1>{});
                else if constexpr (m == 2) put_v((R + 1) & 1);
                else if constexpr (m == 4) put_z((R + 1) & 1);
                else if constexpr (m == 5) {
                    if constexpr (next2 && R2 == 0) locate(grp + 1);
                } else if constexpr (m >= 6 && m < 14) {
                    constexpr int l = m - 6, h = l >> 2, c = l & 3;
                    if constexpr (next2) raw[h][c] = wino_ld16(rsX, xoff(h ? ib : ia, c));
                } else if constexpr (m >= 14 && m < 18) {
                    constexpr int l = m - 14, a = l >> 1, b = l & 1;
                    if constexpr (next2 && !(R2 == 3 && a == 0) && !(R2 == 0 && a == 1)) rdy[a][b] = wino_ld16(rsY, yoff(a, b));
                } else if constexpr (m == 26) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else if constexpr (m >= 28) {
                    constexpr int jj = (m - 28) >> 1, h = (m - 28) & 1;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        fa[0][jj][2 * h + s2] = *reinterpret_cast<const float*>(sn + fragA + jj * 8192 + (2 * h + s2) * 512);
                        fb[0][jj][2 * h + s2] = *reinterpret_cast<const float*>(sn + fragB + jj * 8192 + (2 * h + s2) * 512);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int grp = 0; grp < ngrp - 1; ++grp) {
        stage(I0{}, I3{}, grp);
        stage(I1{}, I3{}, grp);
        stage(I2{}, I3{}, grp);
        stage(I3{}, I3{}, grp);
    }
    stage(I0{}, I3{}, ngrp - 1);
    stage(I1{}, I3{}, ngrp - 1);
    stage(I2{}, I1{}, ngrp - 1);
    stage(I3{}, I0{}, ngrp - 1);

    // ---- G^T over the rows, this wave's columns' share of (.) G: 9 partial taps per register; exchange, finish, slab
    __syncthreads();
    float* const xch = reinterpret_cast<float*>(smem);             // [wave 8][72][lane 64]
    float mine[8][9];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float pt[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float tj[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float x0 = acc[0 + j][r], x1 = acc[2 + j][r], x2 = acc[4 + j][r], x3 = acc[6 + j][r];
                tj[j] = a == 0 ? x0 + 0.5f * (x1 + x2) : (a == 1 ? 0.5f * (x1 - x2) : 0.5f * (x1 + x2) + x3);
            }
            // columns 0, 1 (fh 0): G rows (1, 0, 0), (.5, .5, .5); columns 2, 3 (fh 1): (.5, -.5, .5), (0, 0, 1)
            pt[a][0] = fh ? 0.5f * tj[0] : tj[0] + 0.5f * tj[1];
            pt[a][1] = fh ? -0.5f * tj[0] : 0.5f * tj[1];
            pt[a][2] = fh ? 0.5f * tj[0] + tj[1] : 0.5f * tj[1];
        }
        const bool keep = (r >> 3) == fh;                          // (wave-uniform)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                if (keep) mine[r & 7][a * 3 + s2] = pt[a][s2];
                else xch[(wave * 72 + (r & 7) * 9 + a * 3 + s2) * 64 + lane] = pt[a][s2];
            }
    }
    __syncthreads();
    const int n = n0 + wn * 32 + (lane & 31);
    float* const sl = p.slab + (long)split * 9 * p.Cin * p.Nn + n;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = fh * 8 + rr;
        const int c = c0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (c < p.Cin && n < p.Nn) {
#pragma unroll
            for (int t = 0; t < 9; ++t) sl[((long)t * p.Cin + c) * p.Nn] = mine[rr][t] + xch[((wave ^ 4) * 72 + rr * 9 + t) * 64 + lane];
        }
    }
}
