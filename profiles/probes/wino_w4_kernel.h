// the 4-wave (one wave per SIMD) form of the Winograd forward kernel, kept for the record: the product runs conv_wino_f2k3_w8
// (342 -> 326 us at 14x14, 395 -> 356 us at 56x56).  Include after wino_kernels.h.
#pragma once
// K loop: stages q = 4 s + r (s = 32-channel super-step, r = frequency row).  A stage reads 128-byte lines of the input (8 lanes = the 8
// chunks of one pixel) — with 8-channel stages every line was fetched four times over and the kernel was bound by L2 -> CU traffic.
// One wave per SIMD, so the stage is software-pipelined inside one instruction stream:
//   stage q:  MFMAs on LDS buffer q&1  |  U of stage q+1 by LDS-DMA  |  input transform of stage q+1 (registers loaded during stage q-1) ->
//             ds_write into buffer (q+1)&1  |  global loads of stage q+2 re-issued into the registers the transform has just consumed
// The 64 MFMAs of a stage go out as 8 groups of 8 (two frequencies x one chunk pair); the fragments of group g+1 are read under group g.
// (template order: the epilogue LAST, as in conv_gemm_nt — bench.py and mcn_conv2d_kernel_name rewrite the trailing parameter)
template <int DBG, int EPI>
__global__ __launch_bounds__(256, 1) void conv_wino_f2k3(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.Nn + 63) >> 6, ntm = (p.ntiles + 63) >> 6;
    const int L = xcd_remap(blockIdx.x, ntm * ntn);
    const int t0 = (L / ntn) * 64, nb = L % ntn, n0 = nb * 64;
    const int ns = p.Cin >> 5, nq = ns * 4;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)p.u_bytes, 0x00020000);

    // ---- loader role: items (tile = tid/8 + 32 it, chunk j = tid & 7): 8 lanes read one 128-byte line
    const int lj = tid & 7;
    unsigned poff[2][4][4];                                       // byte offset of patch pixel (row, col) at chunk j; bit 31 = outside
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int T = t0 + (tid >> 3) + 32 * it;
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
                poff[it][r][c] = ok ? (unsigned)((((img * p.H + y) * p.W + x) * p.Cs + lj * 4) * 4) : WINO_OOB;
            }
    }
    const unsigned ubase = (unsigned)(((long)nb * ns) * (4 * WINO_UBYTES) + lane * 16);
    __attribute__((address_space(3))) char* const wbase = (__attribute__((address_space(3))) char*)smem + __builtin_amdgcn_readfirstlane(wave * 1024);

    f32x4 raw[2][2][4], tt[2][4];
    // rows (ia, ib) of frequency row r: B^T d = d0 - d2, d1 + d2, d2 - d1, d1 - d3
    auto load_raw = [&](auto rc, int sidx) {
        constexpr int r = decltype(rc)::value;
        constexpr int ia = r == 0 ? 0 : (r == 2 ? 2 : 1), ib = r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
        const unsigned co = (unsigned)(sidx * 128);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                raw[it][0][c] = wino_ld16(rsX, poff[it][ia][c] + co);                  // (OOB | small stays out of range)
                raw[it][1][c] = wino_ld16(rsX, poff[it][ib][c] + co);
            }
    };
    auto rows = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) tt[it][c] = r == 1 ? raw[it][0][c] + raw[it][1][c] : raw[it][0][c] - raw[it][1][c];
    };
    auto cols = [&](int buf, int it) {                                             // (.) B, into LDS buffer buf
        char* const vs = smem + buf * WINO_STAGE + lj * WINO_VCH + ((tid >> 3) + 32 * it) * 16;
        *reinterpret_cast<f32x4*>(vs + 0 * WINO_VF) = tt[it][0] - tt[it][2];
        *reinterpret_cast<f32x4*>(vs + 1 * WINO_VF) = tt[it][1] + tt[it][2];
        *reinterpret_cast<f32x4*>(vs + 2 * WINO_VF) = tt[it][2] - tt[it][1];
        *reinterpret_cast<f32x4*>(vs + 3 * WINO_VF) = tt[it][1] - tt[it][3];
    };
    auto dma_u = [&](auto bufc, int q) {                                           // U of stage q -> buffer buf: 8 x 1 KB per wave
        constexpr int buf = decltype(bufc)::value;
        const unsigned go = ubase + (unsigned)q * WINO_UBYTES + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(wbase + (buf * WINO_STAGE + WINO_VBYTES + i * 4096)), 16,
                                                     (int)(go + (unsigned)(i * 4096)), 0, 0, 0);
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
    dma_u(I0{}, 0);
    load_raw(I0{}, 0);
    rows(I0{});
    cols(0, 0);
    cols(0, 1);
    load_raw(I1{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fragA = (lane >> 5) * 1024 + (wm * 32 + (lane & 31)) * 16 + WINO_VBYTES;
    const int fragB = (lane >> 5) * WINO_VCH + (wn * 32 + (lane & 31)) * 16;
    f32x4 fa[2][2], fb[2][2];
    auto read_frags = [&](const char* st, int g, int set) {
        const int f0 = 2 * (g >> 2), i = g & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fa[set][j] = *reinterpret_cast<const f32x4*>(st + fragA + (f0 + j) * 8192 + i * 2048);
            fb[set][j] = *reinterpret_cast<const f32x4*>(st + fragB + (f0 + j) * WINO_VF + i * 2 * WINO_VCH);
        }
    };
    // one stage: R = frequency row of this stage (LDS buffer R & 1), sidx = its super-step.  64 MFMAs, each followed by ONE slot of staging
    // work: with a single wave per SIMD an instruction that waits at issue (a buffer load behind a full address queue) starves the MFMA
    // pipe, which holds no more than the instruction in flight — 16 loads issued back to back cost 0.6 us per stage.
    //   slots 0-1    B^T d of stage q+1 from the registers loaded during stage q-1
    //   slots 2-9    the 8 LDS-DMAs of U(q+1)
    //   slots 10-25  the 16 global loads of stage q+2, one per slot
    //   slots 26-33  (.) B and the 8 ds_writes of V(q+1)
    //   slots 8g .. 8g+3: the fragments of group g+1 (two groups ahead of their first use)
    //   slot 52      THE barrier of the stage: every read of this stage's buffer has returned (group 7's fragments were read in slots
    //                48-51), every write into the other buffer is complete (own DMAs: vmcnt, own ds_writes: lgkmcnt)
    //   slots 56-59  the fragments of the next stage's group 0 from the other buffer
    auto stage = [&](auto rc, auto nc, int sidx) {
        constexpr int R = decltype(rc)::value, R1 = (R + 1) & 3, R2 = (R + 2) & 3;
        constexpr bool next = (decltype(nc)::value & 1) && !(DBG & 1), next2 = (decltype(nc)::value & 2) && !(DBG & 1);   // stages q+1 / q+2 exist
        constexpr int ia = R2 == 0 ? 0 : (R2 == 2 ? 2 : 1), ib = R2 == 0 ? 2 : (R2 == 1 ? 2 : (R2 == 2 ? 1 : 3));
        const int q = sidx * 4 + R;
        const char* const st = smem + (R & 1) * WINO_STAGE;
        const char* const sn = smem + ((R + 1) & 1) * WINO_STAGE;
        char* const nv = smem + ((R + 1) & 1) * WINO_STAGE + lj * WINO_VCH + (tid >> 3) * 16;
        const unsigned co2 = (unsigned)((sidx + (R >= 2 ? 1 : 0)) * 128);
        const unsigned go = ubase + (unsigned)(q + 1) * WINO_UBYTES + (unsigned)wave * 1024u;
        wino_for<64>([&](auto mc) {
            constexpr int m = decltype(mc)::value, g = m >> 3, w = m & 7, e = w >> 1, j = w & 1;
            if constexpr (!(DBG & 2))
                acc[R * 4 + 2 * (g >> 2) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][j][e], fb[g & 1][j][e], acc[R * 4 + 2 * (g >> 2) + j], 0, 0, 0);
            if constexpr (g < 7 && w < 4) {                           // fragments of group g+1: A/B of frequency f0 + (w >> 1)
                constexpr int g1 = g + 1, f0 = 2 * (g1 >> 2), i1 = g1 & 3, jj = w >> 1;
                if constexpr ((w & 1) == 0) fa[g1 & 1][jj] = *reinterpret_cast<const f32x4*>(st + fragA + (f0 + jj) * 8192 + i1 * 2048);
                else fb[g1 & 1][jj] = *reinterpret_cast<const f32x4*>(st + fragB + (f0 + jj) * WINO_VF + i1 * 2 * WINO_VCH);
            }
            if constexpr (next) {
                if constexpr (m < 2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) tt[m][c] = R1 == 1 ? raw[m][0][c] + raw[m][1][c] : raw[m][0][c] - raw[m][1][c];
                } else if constexpr (m < 10) {
                    constexpr int i = m - 2;
                    if constexpr (!(DBG & 16))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (__attribute__((address_space(3))) void*)(wbase + (((R + 1) & 1) * WINO_STAGE + WINO_VBYTES + i * 4096)),
                                                                 16, (int)(go + (unsigned)(i * 4096)), 0, 0, 0);
                } else if constexpr (m < 26) {
                    constexpr int l = m - 10, it = l >> 3, h = (l >> 2) & 1, c = l & 3;
                    if constexpr (next2 && !(DBG & 8)) raw[it][h][c] = wino_ld16(rsX, poff[it][h ? ib : ia][c] + co2);
                } else if constexpr (m < 34) {
                    constexpr int l = m - 26, it = l >> 2, f = l & 3;
                    if constexpr (!(DBG & 4)) {
                        const f32x4 v = f == 0 ? tt[it][0] - tt[it][2] : (f == 1 ? tt[it][1] + tt[it][2] : (f == 2 ? tt[it][2] - tt[it][1] : tt[it][1] - tt[it][3]));
                        *reinterpret_cast<f32x4*>(nv + it * 512 + f * WINO_VF) = v;
                    }
                } else if constexpr (m == 52) {
                    if constexpr (next2 && !(DBG & 8)) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");      // the DMAs are older than the 16 loads
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else if constexpr (m >= 56 && m < 60) {
                    constexpr int jj = (m - 56) >> 1;
                    if constexpr ((m & 1) == 0) fa[0][jj] = *reinterpret_cast<const f32x4*>(sn + fragA + jj * 8192);
                    else fb[0][jj] = *reinterpret_cast<const f32x4*>(sn + fragB + jj * WINO_VF);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    read_frags(smem, 0, 0);
    for (int sidx = 0; sidx < ns - 1; ++sidx) {
        stage(I0{}, I3{}, sidx);
        stage(I1{}, I3{}, sidx);
        stage(I2{}, I3{}, sidx);
        stage(I3{}, I3{}, sidx);
    }
    stage(I0{}, I3{}, ns - 1);
    stage(I1{}, I3{}, ns - 1);
    stage(I2{}, I1{}, ns - 1);
    stage(I3{}, I0{}, ns - 1);

    // ---- output transform (lane-local) + store: lane = tile (lane & 31), couts n0 + wm*32 + 8q + 4(lane>>5) + {0..3}
    // EPI (values of conv_gemm_nt's epilogue): NT_EPI_STORE; NT_EPI_STATS = BN-statistics partials of the output as COUNTED rows
    // [row][4][Nn] (sum(y-p), sum((y-p)^2), p, pixels; row = 2 * tile block + wave column; the cout blocks of a tile block fill the
    // channels of its rows between them); NT_EPI_BNRED = the BN-backward sums of the dgrad whose output is the gradient of a BN + ReLU
    // output (rows [red_row0 + row][2][Nn], see GemmNTParams)
    constexpr bool STATS = EPI == NT_EPI_STATS, BNRED = EPI == NT_EPI_BNRED, ACC = EPI == NT_EPI_ACC;     // ACC: out += (a second gradient contribution)
    const int T = t0 + wn * 32 + (lane & 31);
    const int tpi = p.TH * p.TW;
    const int img = T / tpi, rem = T - img * tpi;
    const int th = rem / p.TW, tw = rem - th * p.TW;
    const bool tv = T < p.ntiles;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    unsigned ooff[2][2];
    float npix = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int y = 2 * th + a, x = 2 * tw + b;
            const bool ok = tv && y < p.H && x < p.W;
            ooff[a][b] = ok ? (unsigned)(((img * p.H + y) * p.W + x) * p.ldo * 4) : WINO_OOB;
            if constexpr (STATS) npix += (float)__builtin_popcount((unsigned)__builtin_amdgcn_ballot_w64(ok));      // lanes 0-31: the wave's tiles
        }
    unsigned coff[4];
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + wm * 32 + 8 * q + 4 * (lane >> 5);
        coff[q] = n < p.Nn ? (unsigned)(n * 4) : WINO_OOB;
        bq[q] = (p.bias && n < p.Nn) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // BNRED: every load of the epilogue goes out before the first store
    f32x4 xin[BNRED || ACC ? 4 : 1][2][2];
    unsigned char mb[BNRED ? 4 : 1][2][2];
    if constexpr (ACC) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) xin[q][a][b] = wino_ld16(rsO, __builtin_elementwise_add_sat(ooff[a][b], coff[q]));
    }
    if constexpr (BNRED) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.red_x), 0, (int)p.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.red_mask), 0, (int)(p.out_bytes >> 4), 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const unsigned bo = __builtin_elementwise_add_sat(ooff[a][b], coff[q]);
                    xin[q][a][b] = wino_ld16(rsS, bo);
                    mb[q][a][b] = (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsM, (int)(((bo & 0x7fffffffu) >> 4) | (bo & 0x80000000u)), 0, 0);
                }
    }
    float s1[16], s2[16], piv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 Y[2][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = q * 4 + e;
            float u0[4], u1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u0[c] = acc[0 + c][r] + acc[4 + c][r] + acc[8 + c][r];
                u1[c] = acc[4 + c][r] - acc[8 + c][r] - acc[12 + c][r];
            }
            Y[0][0][e] = u0[0] + u0[1] + u0[2] + bq[q][e];
            Y[0][1][e] = u0[1] - u0[2] - u0[3] + bq[q][e];
            Y[1][0][e] = u1[0] + u1[1] + u1[2] + bq[q][e];
            Y[1][1][e] = u1[1] - u1[2] - u1[3] + bq[q][e];
            if constexpr (STATS) {
                // pivot = output pixel (0, 0) of the wave's first tile (its lane 0 / 32: the two halves hold different channels)
                const int v = __builtin_bit_cast(int, Y[0][0][e]);
                piv[r] = __builtin_bit_cast(float, lane < 32 ? __builtin_amdgcn_readlane(v, 0) : __builtin_amdgcn_readlane(v, 32));
            }
            if constexpr (STATS || BNRED) s1[r] = s2[r] = 0.f;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if constexpr (STATS) {
                    const bool ok = !(ooff[a][b] >> 31);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = ok ? Y[a][b][e] - piv[q * 4 + e] : 0.f;
                        s1[q * 4 + e] += d;
                        s2[q * 4 + e] = __builtin_fmaf(d, d, s2[q * 4 + e]);
                    }
                }
                if constexpr (BNRED) {
                    // rows past the end load x = 0 and mask = 0 (out-of-range offsets): they add nothing
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = (mb[q][a][b] >> e) & 1 ? Y[a][b][e] : 0.f;
                        s1[q * 4 + e] += d;
                        s2[q * 4 + e] = __builtin_fmaf(d, xin[q][a][b][e], s2[q * 4 + e]);
                    }
                }
                if constexpr (ACC) Y[a][b] += xin[q][a][b];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, Y[a][b]), rsO, (int)__builtin_elementwise_add_sat(ooff[a][b], coff[q]), 0, 0);
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
        LaneFold<16, 16>::run(s1, s2, lane, base, writer);                     // lane & 31 ends up with the totals of value `base`
        const int prow = (t0 >> 5) + wn;
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

