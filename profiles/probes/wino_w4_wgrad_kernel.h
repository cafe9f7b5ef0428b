// the 4-wave (one wave per SIMD) form of the Winograd wgrad kernel, kept for the record: the product runs conv_wino_wgrad_f3k2_w8.
// Include after wino_kernels.h.
#pragma once
__global__ __launch_bounds__(256, 1) void conv_wino_wgrad_f3k2(const WinoWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nblk = p.nbc * p.nbn;
    const int L = xcd_remap(blockIdx.x, (int)gridDim.x);
    const int split = L / nblk, blk = L - split * nblk;
    const int c0 = (blk / p.nbn) * 64, n0 = (blk % p.nbn) * 64;
    const int tbeg = split * p.tiles_per_split;
    const int ngrp = p.tiles_per_split >> 5, nq = ngrp * 4;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

    // ---- loader role: items (tile = tid/16 + 16 it, chunk j = tid & 15): 16 lanes read the 256 bytes of one pixel's 64 channels
    const int lj = tid & 15, ltile = tid >> 4;
    const bool cok = c0 + lj * 4 < p.Cin, nok = n0 + lj * 4 < p.Nn;
    // per item and tile group: patch origin (pixel index of (2th-1, 2tw-1), may be negative) and validity masks of its 4 rows / 4 columns
    // (bits 0-3 rows, 4-7 columns; for dy: rows 2th, 2th+1 are patch rows 1, 2, likewise the columns)
    int xbyte[2], ybyte[2], ymask[2];
    auto locate = [&](int grp) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int T = tbeg + grp * 32 + ltile + 16 * it;
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
            xbyte[it] = (xbase * p.Cs + c0 + lj * 4) * 4;                    // (pixel strides below are wave-uniform: scalar multiplies)
            ybyte[it] = ((xbase + p.W + 1) * p.ldy + n0 + lj * 4) * 4;
            ymask[it] = m;
        }
    };
    // (arithmetic, not selects: hipcc turns `ok ? off : OOB` in front of a buffer load into exec-mask branches)
    const unsigned cbad = cok ? 0u : WINO_OOB, nbad = nok ? 0u : WINO_OOB;
    auto xoff = [&](int it, int r, int c) -> unsigned {
        const unsigned ok = ((unsigned)ymask[it] >> r) & ((unsigned)ymask[it] >> (4 + c)) & 1u;
        return (unsigned)(xbyte[it] + (r * p.W + c) * p.Cs * 4) | ((ok ^ 1u) << 31) | cbad;
    };
    auto yoff = [&](int it, int a, int b) -> unsigned {                           // output pixel (2th + a, 2tw + b) = patch (1 + a, 1 + b)
        const unsigned ok = ((unsigned)ymask[it] >> (1 + a)) & ((unsigned)ymask[it] >> (5 + b)) & 1u;
        return (unsigned)(ybyte[it] + (a * p.W + b) * p.ldy * 4) | ((ok ^ 1u) << 31) | nbad;
    };

    f32x4 raw[2][2][4], rdy[2][2][2], tt[2][4], zz[2][2];
    // x rows (ia, ib) of frequency row r: d0 - d2, d1 + d2, d2 - d1, d1 - d3; dy rows: dy0, dy0 + dy1, dy0 - dy1, -dy1
    auto load_stage = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int ia = r == 0 ? 0 : (r == 2 ? 2 : 1), ib = r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
#pragma unroll
        for (int it = 0; it < 2; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                raw[it][0][c] = wino_ld16(rsX, xoff(it, ia, c));
                raw[it][1][c] = wino_ld16(rsX, xoff(it, ib, c));
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (r != 3) rdy[it][0][b] = wino_ld16(rsY, yoff(it, 0, b));
                if (r != 0) rdy[it][1][b] = wino_ld16(rsY, yoff(it, 1, b));
            }
        }
    };
    auto rows = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) tt[it][c] = r == 1 ? raw[it][0][c] + raw[it][1][c] : raw[it][0][c] - raw[it][1][c];
#pragma unroll
            for (int b = 0; b < 2; ++b)
                zz[it][b] = r == 0 ? rdy[it][0][b] : (r == 1 ? rdy[it][0][b] + rdy[it][1][b] : (r == 2 ? rdy[it][0][b] - rdy[it][1][b] : -rdy[it][1][b]));
        }
    };
    // LDS byte offset of (tile, chunk j) inside a frequency plane
    auto slot = [&](int it) -> int {
        const int t = ltile + 16 * it;
        return t * 256 + ((lj ^ ((t & 1) * 8)) << 4);
    };
    auto put_v = [&](int buf, int it) {
        char* const vs = smem + buf * WINO_WG_STAGE + slot(it);
        *reinterpret_cast<f32x4*>(vs + 0 * 8192) = tt[it][0] - tt[it][2];
        *reinterpret_cast<f32x4*>(vs + 1 * 8192) = tt[it][1] + tt[it][2];
        *reinterpret_cast<f32x4*>(vs + 2 * 8192) = tt[it][2] - tt[it][1];
        *reinterpret_cast<f32x4*>(vs + 3 * 8192) = tt[it][1] - tt[it][3];
    };
    auto put_z = [&](int buf, int it) {
        char* const zs = smem + buf * WINO_WG_STAGE + WINO_WG_HALF + slot(it);
        *reinterpret_cast<f32x4*>(zs + 0 * 8192) = zz[it][0];
        *reinterpret_cast<f32x4*>(zs + 1 * 8192) = zz[it][0] + zz[it][1];
        *reinterpret_cast<f32x4*>(zs + 2 * 8192) = zz[it][0] - zz[it][1];
        *reinterpret_cast<f32x4*>(zs + 3 * 8192) = -zz[it][1];
    };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    // fragment addresses: A = V[f][2 ks + (lane >> 5)][wm * 32 + (lane & 31)], odd tiles (lanes 32-63) with the column half flipped
    const int fragA = (lane >> 5) * 256 + (((wm * 32 + (lane & 31)) ^ ((lane >> 5) * 32)) << 2);
    const int fragB = (lane >> 5) * 256 + (((wn * 32 + (lane & 31)) ^ ((lane >> 5) * 32)) << 2) + WINO_WG_HALF;
    float fa[2][2][4], fb[2][2][4];                                               // [set][f of the pair][k-step of the group]
    auto read_frags = [&](const char* st, int g, int set) {
        const int f0 = 2 * (g >> 2), k0 = (g & 3) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                fa[set][j][s] = *reinterpret_cast<const float*>(st + fragA + (f0 + j) * 8192 + (k0 + s) * 512);
                fb[set][j][s] = *reinterpret_cast<const float*>(st + fragB + (f0 + j) * 8192 + (k0 + s) * 512);
            }
    };

    locate(0);
    load_stage(I0{});
    rows(I0{});
    put_v(0, 0); put_v(0, 1); put_z(0, 0); put_z(0, 1);
    load_stage(I1{});
    __syncthreads();
    read_frags(smem, 0, 0);

    // one stage: R = frequency row (LDS buffer R & 1), grp = its tile group; NC bit 0: stage q+1 exists, bit 1: stage q+2 exists
    auto stage = [&](auto rc, auto nc, int grp) {
        constexpr int R = decltype(rc)::value, R1 = (R + 1) & 3, R2 = (R + 2) & 3;
        constexpr bool next = decltype(nc)::value & 1, next2 = decltype(nc)::value & 2;
        constexpr int ia = R2 == 0 ? 0 : (R2 == 2 ? 2 : 1), ib = R2 == 0 ? 2 : (R2 == 1 ? 2 : (R2 == 2 ? 1 : 3));
        const char* const st = smem + (R & 1) * WINO_WG_STAGE;
        const char* const sn = smem + ((R + 1) & 1) * WINO_WG_STAGE;
        wino_for<64>([&](auto mc) {
            constexpr int m = decltype(mc)::value, g = m >> 3, w = m & 7, s = w >> 1, j = w & 1;
            acc[R * 4 + 2 * (g >> 2) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][j][s], fb[g & 1][j][s], acc[R * 4 + 2 * (g >> 2) + j], 0, 0, 0);
            if constexpr (g < 7) {                                              // fragments of group g+1: two reads per slot
                constexpr int g1 = g + 1, f0 = 2 * (g1 >> 2), k0 = (g1 & 3) * 4;
                fa[g1 & 1][j][s] = *reinterpret_cast<const float*>(st + fragA + (f0 + j) * 8192 + (k0 + s) * 512);
                fb[g1 & 1][j][s] = *reinterpret_cast<const float*>(st + fragB + (f0 + j) * 8192 + (k0 + s) * 512);
            }
            if constexpr (next) {
                // slot 0: the transforms' first halves of stage q+1 (registers loaded during stage q-1); slots 2-8: second halves + the 16
                // ds_writes (the other buffer's readers retired at the previous stage's barrier); slots 10-33: the loads of stage q+2, one per slot
                if constexpr (m == 0) rows(std::integral_constant<int, R1>{});
                else if constexpr (m == 2) put_v((R + 1) & 1, 0);
                else if constexpr (m == 4) put_v((R + 1) & 1, 1);
                else if constexpr (m == 6) put_z((R + 1) & 1, 0);
                else if constexpr (m == 8) put_z((R + 1) & 1, 1);
                else if constexpr (m == 9) {
                    if constexpr (next2 && R2 == 0) locate(grp + 1);               // stage q+2 opens the next tile group
                } else if constexpr (m >= 10 && m < 26) {
                    constexpr int l = m - 10, it = l >> 3, h = (l >> 2) & 1, c = l & 3;
                    if constexpr (next2) raw[it][h][c] = wino_ld16(rsX, xoff(it, h ? ib : ia, c));
                } else if constexpr (m >= 26 && m < 34) {
                    constexpr int l = m - 26, it = l >> 2, a = (l >> 1) & 1, b = l & 1;
                    if constexpr (next2 && !(R2 == 3 && a == 0) && !(R2 == 0 && a == 1)) rdy[it][a][b] = wino_ld16(rsY, yoff(it, a, b));
                }
                else if constexpr (m == 52) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else if constexpr (m >= 56 && m < 60) {
                    constexpr int jj = (m - 56) >> 1, h = (m - 56) & 1;
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
    (void)nq;

    // ---- G^T . G (lane-local) and the slab: lane = cout n0 + wn*32 + (lane & 31), channels c0 + wm*32 + 8 q + 4 (lane >> 5) + e
    const int n = n0 + wn * 32 + (lane & 31);
    float* const sl = p.slab + (long)split * 9 * p.Cin * p.Nn + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = c0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        float t3[3][4];
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const float x0 = acc[0 + col][r], x1 = acc[4 + col][r], x2 = acc[8 + col][r], x3 = acc[12 + col][r];
            t3[0][col] = x0 + 0.5f * (x1 + x2);
            t3[1][col] = 0.5f * (x1 - x2);
            t3[2][col] = 0.5f * (x1 + x2) + x3;
        }
        if (c < p.Cin && n < p.Nn) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                sl[((long)(a * 3 + 0) * p.Cin + c) * p.Nn] = t3[a][0] + 0.5f * (t3[a][1] + t3[a][2]);
                sl[((long)(a * 3 + 1) * p.Cin + c) * p.Nn] = 0.5f * (t3[a][1] - t3[a][2]);
                sl[((long)(a * 3 + 2) * p.Cin + c) * p.Nn] = 0.5f * (t3[a][1] + t3[a][2]) + t3[a][3];
            }
        }
    }
}
