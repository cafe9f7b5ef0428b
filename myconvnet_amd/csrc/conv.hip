// conv.hip — host side of mcn_conv2d_{fwd,dgrad,wgrad} and mcn_fc_*: geometry -> tap tables,
// weight packing into caller workspace, kernel selection and launch.  No device allocation, no sync.
#include "conv_kernels.h"
#include "wino_kernels.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ---- error string ------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void mcn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mcn_last_error(void) { return g_err; }
extern "C" int mcn_version(void) { return MCN_VERSION; }

static bool force_naive() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MCN_FORCE_NAIVE");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

// ---- geometry helpers ----------------------------------------------------------------------------
// worst case of the stream-K partials (sk_plan): tail tiles x slices <= resident workgroup slots, largest tile
#define MCN_SK_MAX_BYTES ((size_t)MCN_NUM_CU * 256 * 128 * sizeof(float))

struct Geo {
    int N, H, W, Cin, Cout, KH, KW, SH, SW, DH, DW, pT, pB, pL, pR, xcs, OH, OW, tile;
};
static int geo_from(const mcn_conv_geom* g, Geo* o) {
    if (!g) MCN_FAIL(MCN_E_BADARG, "conv: null geometry");
    o->N = g->N; o->H = g->H; o->W = g->W; o->Cin = g->Cin; o->Cout = g->Cout;
    o->KH = g->KH; o->KW = g->KW; o->SH = g->SH; o->SW = g->SW; o->DH = g->DH; o->DW = g->DW;
    o->pT = g->padT; o->pB = g->padB; o->pL = g->padL; o->pR = g->padR;
    o->xcs = g->x_cs > 0 ? g->x_cs : g->Cin;
    o->tile = g->tile;
    if (o->N < 0 || o->H <= 0 || o->W <= 0 || o->Cin <= 0 || o->Cout <= 0 || o->KH <= 0 || o->KW <= 0 || o->SH <= 0 ||
        o->SW <= 0 || o->DH <= 0 || o->DW <= 0 || o->pT < 0 || o->pB < 0 || o->pL < 0 || o->pR < 0 || o->xcs < o->Cin)
        MCN_FAIL(MCN_E_BADARG, "conv: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d K=%dx%d S=%dx%d D=%dx%d x_cs=%d", o->N, o->H,
                 o->W, o->Cin, o->Cout, o->KH, o->KW, o->SH, o->SW, o->DH, o->DW, o->xcs);
    const int eh = (o->KH - 1) * o->DH + 1, ew = (o->KW - 1) * o->DW + 1;
    if (o->H + o->pT + o->pB < eh || o->W + o->pL + o->pR < ew) MCN_FAIL(MCN_E_BADARG, "conv: filter larger than padded input");
    o->OH = (o->H + o->pT + o->pB - eh) / o->SH + 1;
    o->OW = (o->W + o->pL + o->pR - ew) / o->SW + 1;
    return MCN_OK;
}
static inline int ce_of(mcn_dtype t) { return t == MCN_F32 ? 4 : 8; }
static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

static bool mfma_path_ok(const Geo& g, mcn_dtype dt) {
    if (force_naive()) return false;
    const int ce = ce_of(dt);
    const size_t es = mcn_dtype_size(dt);
    if (g.KH * g.KW > MCN_MAX_TAPS) return false;
    if ((g.KH - 1) * g.DH + g.pT > 127 || (g.KW - 1) * g.DW + g.pL > 127) return false;
    if (g.xcs % ce) return false;                       // 16-byte chunks of x
    if (round_up(g.Cin, ce) > g.xcs) return false;
    if (g.Cout % ce) return false;                      // 16-byte chunks of dy, vector epilogue
    if (g.Cin % 4) {
        if (g.xcs == g.Cin) return false;
    }
    if ((size_t)g.N * g.H * g.W * g.xcs * es >= 0x7fffffffull) return false;
    if ((size_t)g.N * g.OH * g.OW * g.Cout * es >= 0x7fffffffull) return false;
    return true;
}
// dgrad writes dx with channel stride Cin through the vector epilogue
static bool mfma_dgrad_ok(const Geo& g, mcn_dtype dt) { return mfma_path_ok(g, dt) && g.Cin % 4 == 0 && g.xcs == g.Cin; }

// "skinny" 1x1 convs (conv_kernels.h): off the MFMA path because Cout is no chunk multiple, few outputs, chunked dense input
static int skinny_co(const Geo& g) { return round_up(g.Cout, 8); }
static bool skinny_ok(const Geo& g, mcn_dtype dt, int max_co) {
    return g.KH == 1 && g.KW == 1 && g.SH == 1 && g.SW == 1 && g.pT == 0 && g.pB == 0 && g.pL == 0 && g.pR == 0 && g.OH == g.H && g.OW == g.W &&
           g.xcs == g.Cin && g.Cin % ce_of(dt) == 0 && skinny_co(g) <= max_co && mcn_dtype_ok(dt);
}
// ... and the mirror case, few INPUT channels in front of a chunked output (SE expand convs): dgrad and wgrad run the same
// kernels with the roles of the two channel counts swapped
static int skinny_ci(const Geo& g) { return round_up(g.Cin, 8); }
static bool skinny_in_ok(const Geo& g, mcn_dtype dt, int max_co) {
    return g.KH == 1 && g.KW == 1 && g.SH == 1 && g.SW == 1 && g.pT == 0 && g.pB == 0 && g.pL == 0 && g.pR == 0 && g.OH == g.H && g.OW == g.W &&
           g.xcs == g.Cin && g.Cout % ce_of(dt) == 0 && skinny_ci(g) <= max_co && mcn_dtype_ok(dt);
}
static size_t skinny_in_w_bytes(const Geo& g) { return align_up((size_t)g.Cout * skinny_ci(g) * sizeof(float), 256); }
#define MCN_SKINNY_MAX_CO 32        /* fwd / dgrad: accumulators per thread */
#define MCN_SKINNY_MAX_CO_WGRAD 24  /* wgrad: [chunk elements][CO] accumulators per thread */
// few pixels and many channel chunks: one wave per pixel instead of one thread (skinny_conv_fwd_split)
static bool skinny_split(long M, int chunks) { return M <= 8192 && chunks >= 16; }
static size_t skinny_w_bytes(const Geo& g) { return align_up((size_t)g.Cin * skinny_co(g) * sizeof(float), 256); }
static long skinny_wgrad_slab(long M, int chunks) {      // pixels per slab: a multiple of the 32 pixel lanes, at most 256 slabs and ~2048 workgroups
    const long gx = (chunks + 7) / 8;
    long slab = 32;
    while ((M + slab - 1) / slab > 256 || (M + slab - 1) / slab * gx > 2048) slab *= 2;
    return slab;
}

// ---- Winograd F(2x2, 3x3) path (wino_kernels.h): fp32, 3x3 / stride 1 / dilation 1 / pad 1 — 16 multiplications per 2x2 outputs
// instead of 36.  MCN_WINOGRAD=0 (2: forward and dgrad only) or MCN_TILE_NOWINO in mcn_conv_geom.tile keep the direct kernels (the flag must be the same for the
// pack job and the call: the packed operand of an eligible layer is the transformed filter U).
static int wino_level() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MCN_WINOGRAD");
        v = e ? atoi(e) : 1;
    }
    return v;
}
static bool wino_geom(const Geo& g, mcn_dtype dt) {
    return dt == MCN_F32 && wino_level() > 0 && !(g.tile & MCN_TILE_NOWINO) && g.KH == 3 && g.KW == 3 && g.SH == 1 && g.SW == 1 && g.DH == 1 &&
           g.DW == 1 && g.pT == 1 && g.pL == 1 && g.pB == 1 && g.pR == 1 && !force_naive();
}
static bool wino_fwd_ok(const Geo& g, mcn_dtype dt) { return wino_geom(g, dt) && mfma_path_ok(g, dt) && g.Cin % 32 == 0 && g.Cout % 4 == 0; }
static bool wino_dgrad_ok(const Geo& g, mcn_dtype dt) { return wino_geom(g, dt) && mfma_dgrad_ok(g, dt) && g.Cout % 32 == 0 && g.Cin % 4 == 0; }
static size_t wino_u_bytes(int Kin, int Kout) { return (size_t)((Kout + 63) / 64) * (Kin / 32) * 16 * 2048 * sizeof(float); }
static bool wino_wgrad_ok(const Geo& g, mcn_dtype dt) { return wino_geom(g, dt) && mfma_path_ok(g, dt) && g.Cin % 4 == 0 && g.Cout % 4 == 0 && wino_level() == 1; }
// split of the tiles over workgroups: one round of the chip (1 workgroup per CU) over (channel block, cout block, split); whole 32-tile groups
static int wino_wgrad_splits(const Geo& g, int* tiles_per_split) {
    const long ntiles = (long)g.N * ((g.H + 1) / 2) * ((g.W + 1) / 2);
    const int nblk = ((g.Cin + 63) / 64) * ((g.Cout + 63) / 64);
    long groups = (ntiles + 31) / 32;
    if (groups < 1) groups = 1;
    long splits = MCN_NUM_CU / nblk;
    if (splits < 1) splits = 1;
    if (splits > groups) splits = groups;
    const long gps = (groups + splits - 1) / splits;
    splits = (groups + gps - 1) / gps;
    if (tiles_per_split) *tiles_per_split = (int)(gps * 32);
    return (int)splits;
}
static int wino_rows(const Geo& g) { return (int)(4 * (((long)g.N * ((g.H + 1) / 2) * ((g.W + 1) / 2) + 63) / 64)); }       // partial rows of its epilogues: (tile block, wave column, frequency half)

static size_t fwd_pack_bytes(const Geo& g, mcn_dtype dt) {
    if (wino_fwd_ok(g, dt)) return align_up(wino_u_bytes(g.Cin, g.Cout), 256);
    return align_up((size_t)g.Cout * g.KH * g.KW * round_up(g.Cin, ce_of(dt)) * mcn_dtype_size(dt), 256);
}
static size_t dgrad_pack_bytes(const Geo& g, mcn_dtype dt) {
    if (wino_dgrad_ok(g, dt)) return align_up(wino_u_bytes(g.Cout, g.Cin), 256);
    return align_up((size_t)g.Cin * g.KH * g.KW * round_up(g.Cout, ce_of(dt)) * mcn_dtype_size(dt), 256) + 256 * (size_t)g.SH * g.SW;
}
// wgrad tile: fp32 runs 64x64 tiles (32 KB of LDS -> 4 workgroups per CU, same finding as conv_gemm_nt); bf16 keeps
// 128-row tiles unless the operand has no more than 64 rows / columns (1x1 convs on 64 channels)
// (per-layer A/B on MI355X: fp32 1x1/stride-1 wgrads gain 5-55 % from 64x64 tiles, the gathered (3x3 / strided) ones lose
// 8-25 % because the per-K-step pixel bookkeeping is amortised over fewer MFMAs)
static void tn_tile(int rows, int Cout, mcn_dtype dt, bool linear, int forced, int* br, int* bn) {
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    forced &= 0xff;
    if (forced >= 1 && forced <= 4) { *br = cand[forced - 1][0]; *bn = cand[forced - 1][1]; return; }
    static const int env_tile = [] { const char* e = getenv("MCN_TN_TILE"); return e ? atoi(e) : 0; }();      // experiments: gathered (non-linear) wgrads only
    if (env_tile >= 1 && env_tile <= 4 && !linear) { *br = cand[env_tile - 1][0]; *bn = cand[env_tile - 1][1]; return; }
    if (dt == MCN_F32 && linear) { *br = 64; *bn = 64; return; }
    // 256x256 on 8 waves (wave tile 64x128; 128 KB of LDS, one workgroup per CU) for the 2-byte types: half the operand bytes staged
    // per FLOP.  Measured (bf16, B = 256): the 3x3 wgrads with >= 256 output channels gain 7-13 % (14x14 256ch 111 -> 103 us, 7x7 512ch
    // 111 -> 99, 28x28 256ch / 2 129 -> 113, 14x14 512ch / 2 118 -> 106); the 1x1 wgrads (rows = Cin: few tiles, short splits) lose up to
    // 60 %.  In the real step (wgrad on the side stream beside the dgrad / BN-backward chain) the rule "rows >= 4 x Cout" measured
    // 22.86 -> 22.83 and 22.74 -> 22.96 ms: noise, and a 128 KB workgroup keeps the main stream's 64 KB dgrad workgroups off its CU —
    // off by default.  MCN_TN_256: 0 = off (default), 1 = that rule, 2 = wherever it fits.
    static const int big = [] { const char* e = getenv("MCN_TN_256"); return e ? atoi(e) : 0; }();
    if (big && dt != MCN_F32 && rows >= 256 && Cout >= 256 && (big > 1 || (!linear && rows >= 4 * Cout))) { *br = 256; *bn = 256; return; }
    *br = rows <= 64 ? 64 : 128;
    *bn = Cout <= 64 ? 64 : 128;
}
// three-stage LDS ring (conv_gemm_tn3, 8 waves, one workgroup per CU) for the fp32 128 x 128 tile: MCN_TN_RING 0 = off, 1 = gathered
// (3x3 / strided) wgrads, 2 = every fp32 128 x 128 wgrad
static bool tn_ring(size_t es, int BR, int BN, bool linear) {
    static const int v = [] { const char* e = getenv("MCN_TN_RING"); return e ? atoi(e) : 0; }();      // bits: 1 gathered 128x128, 2 linear 128x128, 4 the 64x64 tile (48 KB: three workgroups per CU)
    if (es != 4) return false;
    if (BR == 128 && BN == 128) return (v & (linear ? 2 : 1)) != 0;
    return BR == 64 && BN == 64 && (v & 4) != 0;
}
// ring of 3 / 4 LDS buffers for the 128 x 128 wgrad tile of the 2-byte types (conv_gemm_tn3 / conv_gemm_tn4 on 8 waves, one workgroup per CU):
// MCN_TN_RING2 = stages (3 or 4) for the 1x1 wgrads, 10 + stages for the gathered (3x3 / strided) ones too; 0 = two buffers
static int tn_ring2(size_t es, int BR, int BN, bool linear) {
    static const int v = [] { const char* e = getenv("MCN_TN_RING2"); return e ? atoi(e) : 0; }();
    if (es != 2 || BR != 128 || BN != 128 || v <= 0) return 0;
    const int st = v % 10;
    if (st != 3 && st != 4) return 0;
    return (linear || v >= 10) ? st : 0;
}
static inline bool conv_is_linear(const Geo& g) {
    return g.KH * g.KW == 1 && g.SH == 1 && g.SW == 1 && g.pT == 0 && g.pL == 0 && g.OH == g.H && g.OW == g.W;
}
static int wgrad_splits(const Geo& g, mcn_dtype dt, int* nsteps_out, int* sps_out) {
    const int KP = dt == MCN_F32 ? 32 : 64;
    const long M = (long)g.N * g.OH * g.OW;
    const int nsteps = (int)((M + KP - 1) / KP);
    const int rows = g.KH * g.KW * round_up(g.Cin, ce_of(dt));
    int BR, BN;
    tn_tile(rows, g.Cout, dt, conv_is_linear(g), g.tile, &BR, &BN);
    const int tiles = ((rows + BR - 1) / BR) * ((g.Cout + BN - 1) / BN);
    // workgroups in total: ~4 per CU in fp32; bf16 (2-3 resident per CU, HBM-side bound) prefers fewer, longer splits — same-box
    // A/B of the whole step: 1536 / 1024 / 768 -> fp32 73.1 / 72.5 / 73.2 ms, bf16 25.6 / 25.4 / 25.1 ms; 512 / 256 -> bf16 24.95 / 25.2 ms
    static const int target_env = [] { const char* e = getenv("MCN_TN_TARGET"); return e ? atoi(e) : 0; }();      // experiments
    // (re-measured at the end of round 2, XCD-aware order and streaming BN loads in place: bf16 384 / 512 / 768 / 1024 -> 21.45-21.61 /
    // 21.26-21.54 / 21.19-21.44 / 21.73 ms; fp32 768 ... 1536 within 0.2 %)
    const int target = target_env > 0 ? target_env : (dt == MCN_F32 ? 1024 : 768);
    int splits = (target + tiles - 1) / tiles;
    if (splits > nsteps) splits = nsteps;
    if (splits > 512) splits = 512;
    if (splits < 1) splits = 1;
    static const int model = [] { const char* e = getenv("MCN_TN_SPLITS"); return e ? atoi(e) : 1; }();
    const int ring2 = tn_ring2(mcn_dtype_size(dt), BR, BN, conv_is_linear(g));
    if (model > 1 || (model == 1 && BR == 128 && BN == 128 && (!conv_is_linear(g) || ring2))) {
        // Whole rounds: a CU holds occ = 2 workgroups of the 128 x 128 tile (64 KB of LDS), the chip 512; a launch of W = tiles x splits
        // workgroups runs W / slots full rounds and a tail.  A full round costs occ x (steps + c0) MFMA-bound step times per CU, a tail
        // round ceil(rem / 256) of them (one workgroup alone on a CU does not fill its pipes: at least 1.3); c0 = slab write + read
        // back + prologue, in steps.  Pick the split count with the lowest modelled time.  The fixed target above gave the 3x3 layers
        // with 256+ channels 1044-1152 workgroups = two rounds + a nearly empty third: 14x14 256ch 547 -> 389 us (bf16) / 545 -> 493
        // (fp32), 28x28 256ch / 2 127 -> 85, 7x7 512ch 112 -> 93 (serial launches, B = 256).  The smaller tiles keep the target: their
        // 3-5 workgroups per CU are not priced well by this model (64 x 64 fp32 1x1 layers +8 %, MCN_TN_SPLITS=2 to see it).
        int occ = (160 * 1024) / ((ring2 ? ring2 : (tn_ring(mcn_dtype_size(dt), BR, BN, conv_is_linear(g)) ? 3 : 2)) * KP * (BR + BN) * (int)mcn_dtype_size(dt));
        if (occ > 4) occ = 4;
        const int slots = 256 * (occ < 1 ? 1 : occ);
        const double c0 = 4.0;
        double best = 1e30;
        int best_s = splits;
        const int smax = nsteps < 512 ? nsteps : 512;
        for (int s2 = 1; s2 <= smax; ++s2) {
            const int sps2 = (nsteps + s2 - 1) / s2;
            const int sr = (nsteps + sps2 - 1) / sps2;          // the split count this step count really gives
            if (sr != s2) continue;
            const long W = (long)tiles * s2;
            const long full = W / slots, rem = W % slots;
            double tail = 0.0;
            if (rem) {
                tail = (double)((rem + 255) / 256);
                if (tail < 1.3) tail = 1.3;
            }
            const double cost = ((double)full * occ + tail) * ((double)sps2 + c0);
            if (cost < best * 0.999) { best = cost; best_s = s2; }
        }
        splits = best_s;
    }
    int sps = (nsteps + splits - 1) / splits;
    if (sps < 1) sps = 1;
    splits = (nsteps + sps - 1) / sps;
    if (splits < 1) splits = 1;
    if (nsteps_out) *nsteps_out = nsteps;
    if (sps_out) *sps_out = sps;
    return splits;
}
static size_t colsum_parts(long M) {
    long parts = (M + 15) / 16;
    if (parts > 1024) parts = 1024;
    if (parts < 1) parts = 1;
    return (size_t)parts;
}
// slices of the (image, output row) range for the fallback wgrad: enough blocks to fill the chip, at least 8 rows each
static int naive_wgrad_slices(const Geo& g) {
    const long total = (long)g.KH * g.KW * g.Cin * g.Cout, nrows = (long)g.N * g.OH;
    const long wblocks = (total + 31) / 32;
    long s = 2048 / (wblocks < 1 ? 1 : wblocks);
    if (s > nrows / 8) s = nrows / 8;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
}
static size_t wgrad_ws_bytes(const Geo& g, mcn_dtype dt) {
    size_t b = 0;
    if (!mfma_path_ok(g, dt)) {
        const long M = (long)g.N * g.OH * g.OW;
        int sl = naive_wgrad_slices(g);
        if (skinny_ok(g, dt, MCN_SKINNY_MAX_CO_WGRAD) || skinny_in_ok(g, dt, MCN_SKINNY_MAX_CO_WGRAD)) {
            const long slab = skinny_wgrad_slab(M, (skinny_ok(g, dt, MCN_SKINNY_MAX_CO_WGRAD) ? g.Cin : g.Cout) / ce_of(dt));
            sl = (int)((M + slab - 1) / slab) + 1;
        }
        if (sl > 1) b += align_up((size_t)sl * g.KH * g.KW * g.Cin * g.Cout * 4, 256);
    }
    if (mfma_path_ok(g, dt)) {
        const int splits = wino_wgrad_ok(g, dt) ? wino_wgrad_splits(g, nullptr) : wgrad_splits(g, dt, nullptr, nullptr);
        const size_t rows = (size_t)g.KH * g.KW * round_up(g.Cin, ce_of(dt));
        b += align_up((size_t)splits * rows * g.Cout * 4, 256);
    }
    b += align_up(colsum_parts((long)g.N * g.OH * g.OW) * g.Cout * 4, 256);
    return b;
}

extern "C" int mcn_conv2d_tile_candidates(mcn_conv_op op) { return op == MCN_CONV_WGRAD ? 4 : 6; }      /* (the 256x256 wgrad tile of the 2-byte types is chosen by rule, not a candidate) */    /* NT: 128x128, 128x64, 64x64, 256x128 / 8 waves and 128x128 / 8 waves (2-byte types); TN: 4 shapes */

extern "C" size_t mcn_conv2d_workspace_bytes(mcn_conv_op op, const mcn_conv_geom* gg, mcn_dtype dtype) {
    Geo g;
    if (geo_from(gg, &g) != MCN_OK) return 0;
    if (!mcn_dtype_ok(dtype)) return 0;
    switch (op) {
        /* packed weights (unless the caller keeps them) + room for the stream-K partials */
        case MCN_CONV_FWD: return mfma_path_ok(g, dtype) ? fwd_pack_bytes(g, dtype) + MCN_SK_MAX_BYTES : (skinny_ok(g, dtype, MCN_SKINNY_MAX_CO) ? skinny_w_bytes(g) : 0);
        case MCN_CONV_DGRAD:
            if (mfma_dgrad_ok(g, dtype)) return dgrad_pack_bytes(g, dtype) + MCN_SK_MAX_BYTES;
            if (skinny_ok(g, dtype, MCN_SKINNY_MAX_CO)) return skinny_w_bytes(g);
            return skinny_in_ok(g, dtype, MCN_SKINNY_MAX_CO) ? skinny_in_w_bytes(g) : 0;
        case MCN_CONV_WGRAD: return wgrad_ws_bytes(g, dtype);
    }
    return 0;
}

// ---- launch helpers ---------------------------------------------------------------------------------
// (returns false when the runtime refuses the request: the launch behind it would fail with a less telling error)
template <typename K>
static bool allow_lds(K kernel, int bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

// Tile shape for conv_gemm_nt.  In the MFMA-bound regime a CU's time is (tiles it receives) x (work per tile), so a layer
// with W equal tiles costs ceil(W/256) tile-times: at B=256 the late ResNet stages have only 392..1568 tiles of 128x128
// (1.53 / 3.06 / 6.125 per CU) and lose 13-23 % to the last, nearly empty round, while the K-loop itself runs at 91 % of
// the MFMA rate (in-kernel stamps).  Smaller tiles quantise better and fit 4 workgroups per CU instead of 2 (their
// prologue / epilogue / barrier bubbles overlap better), but re-read more LDS per MFMA.  Per-area cost factors from
// whole-network A/B runs on MI355X: fp32 (64-cycle MFMAs, LDS far from limiting) prefers 64x64 everywhere
// (conv fwd+dgrad 38.4 ms vs 41.5 ms per step with 128x128); bf16 is LDS-bandwidth sensitive and keeps the big tiles.
// (A body/tail split — big tiles for whole rounds, small tiles for the remainder in a second launch — was measured
// and lost 4 %: the kernel boundary costs more than the shorter tail saves.)
struct NtTile { int bm, bn, nw, wpp; };
// candidate 3 (256x128, 8 waves) is bf16 only: fp32 is MFMA-bound and prefers the smallest tile
// candidate 4 (128x128 on 8 waves of 32x64): half the accumulators and epilogue registers per wave -> 4 waves per SIMD instead of
// 2 at the same tile / L2 traffic: for the memory- and epilogue-bound 1x1 layers of the 2-byte types (statistics / residual epilogues)
// candidates 5 / 6 (round 4; tile hint 6 picks the one that fits Cout): conv_gemm_nt_wpp — 3x3 / stride-1 window kernel of the 2-byte types, 256 x 128 or
// 256 x 64 on 8 waves (double-buffered input window, filter ring, two wave groups one phase apart); geometries it does not take run candidate 3 / 1
static const NtTile kNtCand[7] = {{128, 128, 4, 0}, {128, 64, 4, 0}, {64, 64, 4, 0}, {256, 128, 8, 0}, {128, 128, 8, 0}, {256, 128, 8, 1}, {256, 64, 8, 1}};
#define MCN_NT_CANDS 7
static inline double nt_tile_work(int c, size_t es) {
    static const double w[3] = {128.0 * 128, 128.0 * 64, 64.0 * 64};
    static const double f32[3] = {1.08, 1.03, 1.00}, bf16[3] = {1.00, 1.30, 1.70};      // (round 2: the 2-byte kernels are bound by L2 -> LDS
    // staging, not by tile-count quantisation — every layer for which the old 1.08 / 1.35 picked 128x64 over 128x128 ran 10-17 % faster on 128x128)
    return w[c] * (es == 4 ? f32[c] : bf16[c]);
}
template <typename T>
static int pick_nt_tile(int M, int Nn, int hint = 0) {
    hint &= 0xff;
    if (hint >= 1 && hint <= 3) return hint - 1;
    if ((hint == 4 || hint == 5) && sizeof(T) == 2 && Nn > 64) return hint - 1;      // the 8-wave tiles: 2-byte types only; otherwise the heuristic below
    if (hint == 6 && sizeof(T) == 2) return Nn > 64 ? 5 : 6;                          // the window ping-pong kernel's two tile widths
    static const int forced = [] { const char* e = getenv("MCN_NT_TILE"); return e ? atoi(e) : -1; }();
    if ((forced == 4 || forced == 3) && sizeof(T) == 2 && Nn > 64) return forced;
    const NtTile* cand = kNtCand;
    int best = Nn <= 64 ? 1 : 0;
    double best_cost = -1;
    for (int c = 0; c < 3; ++c) {
        if (Nn <= 64 && cand[c].bn > 64) continue;
        if (forced >= 0 && forced < 3 && c != forced && !(Nn <= 64 && forced == 0)) continue;
        const long w = (long)((M + cand[c].bm - 1) / cand[c].bm) * ((Nn + cand[c].bn - 1) / cand[c].bn);
        const double cost = (double)((w + MCN_NUM_CU - 1) / MCN_NUM_CU) * nt_tile_work(c, sizeof(T));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

// Stream-K tail (fp32).  The fp32 conv_gemm_nt is MFMA bound, so a CU's time is the tile work it receives: W tiles over S
// resident workgroup slots cost ceil(W / S) rounds and the last round of the late ResNet stages is mostly empty (14x14 / 7x7
// layers at B = 256: 2.45 / 1.2 rounds).  The whole rounds run as ordinary tiles; the tiles of the last, partial round are
// cut along K into `slices` workgroups each so that they fill the slots once, in the SAME launch (no kernel boundary in
// front of the tail); the slices park their fp32 accumulators in the workspace and a short second launch sums them in a
// fixed order (deterministic, unlike an atomic tail) and runs the epilogue.  Measured per layer on MI355X (B = 256):
// 7x7 3x3 512 -> 512 fwd 540 -> 479 us, 7x7 1x1 fwd / dgrad -8 %, 14x14 3x3 -3 %; layers with many rounds or a short K loop
// lose (stem: 39 rounds, 7 K-steps, +4 %), hence the two guards below.  bf16 is off: its layers are LDS / L2 / HBM bound,
// the tiles of a thin last round run faster than those of a full one, and the split cost 4 % of the step (per-layer +10-38 %).
struct SkPlan { int body, tail, slices; size_t bytes; };
// resident workgroups per CU: 160 KB of LDS / (2 buffers x (BM + BN) x 128 B); registers allow at least as many
static const int kNtSlotsPerCU[MCN_NT_CANDS] = {2, 3, 5, 1, 2, 1, 1};
#define MCN_SK_MAX_ROUNDS 8          /* more whole rounds than this: the tail is too small a share of the layer to pay */
#define MCN_SK_MIN_KSTEPS 4          /* K-steps per slice (below: prologue + partial traffic outweigh the MFMAs) */
static SkPlan sk_plan(int tile, long W, int nk, size_t es) {
    // MCN_NT_STREAMK: 0 = off, 1 = default (fp32 only), 2 = every dtype (experiments)
    static const int enabled = [] { const char* e = getenv("MCN_NT_STREAMK"); return e ? atoi(e) : 1; }();
    SkPlan sp = {(int)W, 0, 1, 0};
    if (!enabled || (es != 4 && enabled < 2)) return sp;
    if (kNtCand[tile].wpp) return sp;                      // (launch_nt hands the window ping-pong tiles to launch_nt_wpp before any split: never K-sliced)
    const long S = (long)kNtSlotsPerCU[tile] * MCN_NUM_CU;
    const long tail = W % S;
    if (W < S || W / S >= MCN_SK_MAX_ROUNDS || tail == 0 || 4 * tail >= 3 * S) return sp;   // last round already >= 75 % full
    int slices = (int)(S / tail);
    if (slices > nk / MCN_SK_MIN_KSTEPS) slices = nk / MCN_SK_MIN_KSTEPS;
    if (slices > 32) slices = 32;
    if (slices < 2) return sp;
    sp.body = (int)(W - tail);
    sp.tail = (int)tail;
    sp.slices = slices;
    sp.bytes = (size_t)tail * slices * kNtCand[tile].bm * kNtCand[tile].bn * sizeof(float);
    return sp;
}

// conv_gemm_nt_win (input window with halo kept in LDS across the taps): stride-1 tap convolutions on the input's own grid whose
// window (BM + span rows of 128 bytes, span = (KH-1)*DH*IW + (KW-1)*DW pixels) plus two B buffers stay within 64 KB and 1.6x the
// LDS of the two-buffer kernel (occupancy: fp32 64x64 keeps 4-5 workgroups per CU up to 56-wide maps).
// Measured per layer (B = 256): fp32 7x7 492 -> 460 us, 14x14 504 -> 468, 28x28 510 -> 482, 56x56 518 -> 504 (the staging probe's
// bound was -8 %).  The 2-byte types do NOT gain (bf16 14x14 68.9 -> 69.6 us, 28x28 76.5 -> 78.9, 56x56 95 -> 104, 7x7 69 -> 63 with a
// window swizzle that is conflict-free at every row offset; the first build, 2-way conflicted at odd offsets, measured the
// same): the one exposed window load per channel chunk and the per-K-step fragment address arithmetic are a larger share
// of their short K-steps (512 MFMA cycles against 1024 in fp32) — fp32 only by default.
// MCN_NT_WINDOW: 0 = off, 1 = fp32 (default), 2 = every dtype.
#define NT_WINDOW 3
static int nt_window_level() {
    static const int v = [] { const char* e = getenv("MCN_NT_WINDOW"); return e ? atoi(e) : 1; }();
    return v;
}
static bool nt_window_enabled(size_t es) { return nt_window_level() >= (es == 4 ? 1 : 2); }
static int nt_window_lds(int span, const NtTile& t) {           // 0 = not eligible
    const int P = t.bm + span, rpp = t.nw * 8;
    const int wrows8 = (P + 8) & ~7;
    if (t.nw != 4 || wrows8 > 16 * rpp) return 0;
    const int lds = wrows8 * 128 + 2 * t.bn * 128;
    if (lds > 64 * 1024 || lds * 10 > 2 * (t.bm + t.bn) * 128 * 16) return 0;
    return lds;
}
// span of the tap offsets in pixels, or -1 when the geometry does not qualify (same-grid, stride 1, 2..32 taps, whole K-steps per tap)
static int nt_window_span(size_t es, int ntaps, int cpt, int sy, int sx, bool same_grid, const int* tap, int IW, int* dmin_out) {
    if (!nt_window_enabled(es) || ntaps < 2 || ntaps > 32 || sy != 1 || sx != 1 || !same_grid || cpt % 8) return -1;
    int dmin = 0x7fffffff, dmax = -0x7fffffff;
    for (int t = 0; t < ntaps; ++t) {
        const int d = (int)(short)(tap[t] & 0xffff) * IW + (tap[t] >> 16);
        if (d < dmin) dmin = d;
        if (d > dmax) dmax = d;
    }
    if (dmin_out) *dmin_out = dmin;
    return dmax - dmin;
}

// the same decision from the conv geometry alone (introspection: kernel names for profiling tables)
static bool nt_window_geom(const Geo& g, size_t es, int cpt, const NtTile& t) {
    if (!nt_window_enabled(es) || g.KH * g.KW < 2 || g.KH * g.KW > 32 || g.SH != 1 || g.SW != 1 || g.OH != g.H || g.OW != g.W || cpt % 8) return false;
    return nt_window_lds((g.KH - 1) * g.DH * g.W + (g.KW - 1) * g.DW, t) != 0;
}

// conv_gemm_nt_wpp (tile hint 6): 2-byte types, 3x3 taps (any dilation that keeps the window in 376 rows), stride 1, output grid = input grid,
// whole 64-channel K-steps per tap.  The one place that decides: every caller turns the geometry's hint into the effective one with nt_hint().
#define MCN_WPP_MAX_SPAN (WPP_WROWS - 8 - 256)
static bool nt_wpp_geom(const Geo& g, mcn_dtype dtype, bool dgrad) {
    if (mcn_dtype_size(dtype) != 2 || g.KH != 3 || g.KW != 3 || g.SH != 1 || g.SW != 1 || g.OH != g.H || g.OW != g.W) return false;
    const int ce = ce_of(dtype), cpt = round_up(dgrad ? g.Cout : g.Cin, ce) / ce;
    return cpt % 8 == 0 && 2 * g.DH * g.W + 2 * g.DW <= MCN_WPP_MAX_SPAN;
}
// Without a hint the kernel takes the layers it was measured faster on (MI355X, B = 256, bf16, us: fwd / fwd + statistics / dgrad / dgrad + masked add)
//   14x14 256 -> 256: 56.1 / 56.4 / 55.3 / 60.2 against 65.4 / 66.8 / 65.0 / 68.3 of the best two-buffer tile;  7x7 512 -> 512: 49.2 / 49.9 / 49.6 / 51.8 against 59.3 / 61.4 / 61.1 / 64.4;
//   28x28 128 -> 128 (two chunks = 18 K-steps per tile: the one-workgroup-per-CU prologue shows) 73.4 against 68.4, 56x56 64 -> 64 (one chunk, BN = 64) 112 against 82:
// i.e. four or more 64-channel chunks per tap and a full 128-column tile.  MCN_NT_WPP: 0 = only on request (tile hint 6), 1 = forward, 2 = forward and dgrad (default).
static int nt_wpp_level() {
    static const int v = [] { const char* e = getenv("MCN_NT_WPP"); return e ? atoi(e) : 2; }();
    return v;
}
static int nt_hint(const Geo& g, mcn_dtype dtype, bool dgrad) {
    const int hint = g.tile & 0xff;
    if (hint == 6 && !nt_wpp_geom(g, dtype, dgrad)) return (g.tile & ~0xff) | 4;      // the plain 256 x 128 / 8-wave tile (Cout <= 64: the heuristic)
    if (hint == 0 && nt_wpp_level() >= (dgrad ? 2 : 1) && (dgrad ? g.Cout : g.Cin) >= 256 && (dgrad ? g.Cin : g.Cout) >= 128 && nt_wpp_geom(g, dtype, dgrad))
        return (g.tile & ~0xff) | 6;
    return g.tile;
}

// epilogue variant of a launch: the accumulate modes have their own instantiation (batched loads), so do the BN statistics, the BN-backward
// sums, and the masked residual fan-in TOGETHER with the BN-backward sums (NT_EPI_ACCRED: mcn_conv2d_dgrad_addmasked_bnred)
static inline int nt_epi_of(const GemmNTParams& p) {
    if (p.stats) return NT_EPI_STATS;
    if (p.accumulate) return (p.accumulate == 2 && p.red_part) ? NT_EPI_ACCRED : NT_EPI_ACC;
    return p.red_part ? NT_EPI_BNRED : NT_EPI_STORE;
}
template <typename T>
static int launch_nt_tiles(const GemmNTParams& p, int tile, int nblocks, int mode, bool reduce, hipStream_t st) {
    const NtTile t = kNtCand[tile];
    // a K loop of one step (K <= 128 bytes: the 64-channel 1x1 layers in bf16) never touches the second LDS buffer: declaring
    // one lets twice as many workgroups share a CU where registers allow (MCN_NT_LDS1=0 restores the two-buffer launch)
    static const int lds1 = [] { const char* e = getenv("MCN_NT_LDS1"); return e ? atoi(e) : 1; }();
    const bool one_step = lds1 && p.sk_mode == 0 && ((p.nchunks + 7) >> 3) <= 1;
    const int lds = reduce ? 0 : (one_step ? 1 : 2) * (t.bm + t.bn) * 128;
    const dim3 grid(nblocks), block(t.nw * 64);
#define MCN_LAUNCH_NT_S(BMV, BNV, NWV, MODEV, EPIV)                                   \
    do {                                                                             \
        static bool once = (allow_lds(conv_gemm_nt<T, BMV, BNV, MODEV, NWV, EPIV>, 2 * (BMV + BNV) * 128), true); \
        (void)once;                                                                  \
        hipLaunchKernelGGL((conv_gemm_nt<T, BMV, BNV, MODEV, NWV, EPIV>), grid, block, lds, st, p); \
    } while (0)
#define MCN_LAUNCH_NT(BMV, BNV, NWV, MODEV)                                           \
    do {                                                                             \
        if (epi == NT_EPI_STATS) MCN_LAUNCH_NT_S(BMV, BNV, NWV, MODEV, NT_EPI_STATS);  \
        else if (epi == NT_EPI_ACC) MCN_LAUNCH_NT_S(BMV, BNV, NWV, MODEV, NT_EPI_ACC); \
        else if (epi == NT_EPI_BNRED) MCN_LAUNCH_NT_S(BMV, BNV, NWV, MODEV, NT_EPI_BNRED); \
        else if (epi == NT_EPI_ACCRED) MCN_LAUNCH_NT_S(BMV, BNV, NWV, MODEV, NT_EPI_ACCRED); \
        else MCN_LAUNCH_NT_S(BMV, BNV, NWV, MODEV, NT_EPI_STORE);                      \
    } while (0)
#define MCN_LAUNCH_NT_MODE(BMV, BNV, NWV)                                                 \
    do {                                                                                  \
        if (reduce) {                                                                     \
            if (epi == NT_EPI_STATS) hipLaunchKernelGGL((conv_nt_sk_reduce<T, BMV, BNV, NWV, NT_EPI_STATS>), grid, block, 0, st, p);    \
            else if (epi == NT_EPI_ACC) hipLaunchKernelGGL((conv_nt_sk_reduce<T, BMV, BNV, NWV, NT_EPI_ACC>), grid, block, 0, st, p);   \
            else if (epi == NT_EPI_BNRED) hipLaunchKernelGGL((conv_nt_sk_reduce<T, BMV, BNV, NWV, NT_EPI_BNRED>), grid, block, 0, st, p); \
            else if (epi == NT_EPI_ACCRED) hipLaunchKernelGGL((conv_nt_sk_reduce<T, BMV, BNV, NWV, NT_EPI_ACCRED>), grid, block, 0, st, p); \
            else hipLaunchKernelGGL((conv_nt_sk_reduce<T, BMV, BNV, NWV, NT_EPI_STORE>), grid, block, 0, st, p);                        \
        } else if (mode_nt == NT_LINEAR) MCN_LAUNCH_NT(BMV, BNV, NWV, NT_LINEAR);         \
        else if (mode_nt == NT_UNIFORM) MCN_LAUNCH_NT(BMV, BNV, NWV, NT_UNIFORM);         \
        else MCN_LAUNCH_NT(BMV, BNV, NWV, NT_GENERIC);                                    \
    } while (0)
    // epilogue variant: the accumulate modes have their own instantiation (batched loads), so do the BN statistics
    const int epi = nt_epi_of(p);
    if (mode == NT_WINDOW && !reduce) {
        const int wlds = nt_window_lds(p.win_rows - t.bm, t);
#define MCN_LAUNCH_WIN_E(BMV, BNV, EPIV)                                                                  \
    do {                                                                                                  \
        static bool once = (allow_lds(conv_gemm_nt_win<T, BMV, BNV, 4, EPIV>, 64 * 1024), true);          \
        (void)once;                                                                                       \
        hipLaunchKernelGGL((conv_gemm_nt_win<T, BMV, BNV, 4, EPIV>), grid, block, wlds, st, p);           \
    } while (0)
#define MCN_LAUNCH_WIN(BMV, BNV)                                                                          \
    do {                                                                                                  \
        if (epi == NT_EPI_STATS) MCN_LAUNCH_WIN_E(BMV, BNV, NT_EPI_STATS);                                \
        else if (epi == NT_EPI_ACC) MCN_LAUNCH_WIN_E(BMV, BNV, NT_EPI_ACC);                               \
        else if (epi == NT_EPI_BNRED) MCN_LAUNCH_WIN_E(BMV, BNV, NT_EPI_BNRED);                           \
        else if (epi == NT_EPI_ACCRED) MCN_LAUNCH_WIN_E(BMV, BNV, NT_EPI_ACCRED);                         \
        else MCN_LAUNCH_WIN_E(BMV, BNV, NT_EPI_STORE);                                                    \
    } while (0)
        if (t.bm == 128 && t.bn == 128) MCN_LAUNCH_WIN(128, 128);
        else if (t.bm == 128) MCN_LAUNCH_WIN(128, 64);
        else MCN_LAUNCH_WIN(64, 64);
#undef MCN_LAUNCH_WIN
#undef MCN_LAUNCH_WIN_E
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    const int mode_nt = mode == NT_WINDOW ? NT_UNIFORM : mode;          // (the stream-K reduce pass of a window launch: epilogue only)
    if (t.nw == 8) {
        if constexpr (sizeof(T) == 2) {
            if (t.bm == 256) MCN_LAUNCH_NT_MODE(256, 128, 8);
            else MCN_LAUNCH_NT_MODE(128, 128, 8);
        } else {
            MCN_FAIL(MCN_E_UNSUPPORTED, "conv: the 8-wave tiles are for the 2-byte types only");
        }
    } else if (t.bm == 128 && t.bn == 128) MCN_LAUNCH_NT_MODE(128, 128, 4);
    else if (t.bm == 128) MCN_LAUNCH_NT_MODE(128, 64, 4);
    else MCN_LAUNCH_NT_MODE(64, 64, 4);
#undef MCN_LAUNCH_NT_MODE
#undef MCN_LAUNCH_NT
#undef MCN_LAUNCH_NT_S
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// conv_gemm_nt_pers (persistent workgroups with the next tile's first K-step prefetched under the epilogue): the 1x1 /
// stride-1 launches without bias on the 4-wave tiles that run unsplit.  MCN_NT_PERS: 0 = off, 1 = forward with statistics (default), 2 = all.
// Measured per layer (B = 256, us): bf16 128x128 56x56 64->256 fwd+stats 151 -> 127, 28x28 128->512 94 -> 84, 7x7 512->2048 45 -> 37,
// dgrad+residual 14x14 72 -> 67, 7x7 56 -> 51; the 128x64 tile (Nn = 64 layers) loses 1-10 %, the 8-wave tile loses on the
// statistics epilogue (its VALU work doubles per pixel row); fp32 64x64 gains 2-9 % with the store / statistics epilogues and
// loses 2-12 % with the accumulate one (its loads wait behind the prefetch).
// In the training step only the forward launches (statistics epilogue) use it: the backward's dgrad launches share the chip with
// the wgrad GEMMs of the side stream, and workgroups with a fixed tile list cannot rebalance around them — with every eligible
// launch persistent the bf16 step was 0.16 ms SLOWER although its kernels summed to 0.35 ms less (MCN_NT_PERS=2: all of them).
static bool nt_pers_tile(const NtTile& t, size_t es, int epi) {
    static const int level = [] { const char* e = getenv("MCN_NT_PERS"); return e ? atoi(e) : 1; }();
    if (level <= 0 || t.nw != 4 || (level == 1 && epi != NT_EPI_STATS)) return false;
    return es == 4 ? (t.bm == 64 && epi != NT_EPI_ACC) : (t.bm == 128 && t.bn == 128);
}
// introspection twin of launch_nt's decision (assumes the caller hands over the stream-K workspace, as the executor does)
static bool nt_pers_geom(int tile, long M, int Nn, int nchunks, size_t es, int tile_hint, int epi = NT_EPI_STORE) {
    const NtTile t = kNtCand[tile];
    if (!nt_pers_tile(t, es, epi)) return false;
    const long W = ((M + t.bm - 1) / t.bm) * ((Nn + t.bn - 1) / t.bn);
    return (tile_hint & MCN_TILE_NOSPLIT) || sk_plan(tile, W, (nchunks + 7) >> 3, es).slices < 2;
}
template <typename K>
static int pers_slots(K kernel, int threads, int lds) {
    allow_lds(kernel, lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kernel), threads, (size_t)lds) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
    }
    return n;
}
// resident workgroups of a persistent instantiation (slots per CU x CUs), queried once per instantiation
template <typename T, int BMV, int BNV, int EPIV>
static long nt_pers_cap_of() {
    static const int slots = pers_slots(conv_gemm_nt_pers<T, BMV, BNV, 4, EPIV>, 256, 2 * (BMV + BNV) * 128);
    // MCN_PERS_CUS (experiment, profiles/probes/rccl_interference.py): size the persistent grids for fewer CUs, i.e. leave the others to a
    // collective's reduction kernel that runs beside them (a multiple of 32 keeps the counted statistics rows: grid / 8 % N tiles == 0)
    static const int cus = [] { const char* e = getenv("MCN_PERS_CUS"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= MCN_NUM_CU ? v : MCN_NUM_CU; }();
    return (long)slots * cus;
}
template <typename T>
static long nt_pers_cap(int tile, int epi) {
    const NtTile t = kNtCand[tile];
#define MCN_PERS_CAP(BMV, BNV)                                                                  \
    (epi == NT_EPI_STATSC ? nt_pers_cap_of<T, BMV, BNV, NT_EPI_STATSC>()                         \
     : epi == NT_EPI_STATS ? nt_pers_cap_of<T, BMV, BNV, NT_EPI_STATS>()                         \
     : epi == NT_EPI_ACC ? nt_pers_cap_of<T, BMV, BNV, NT_EPI_ACC>() : nt_pers_cap_of<T, BMV, BNV, NT_EPI_STORE>())
    if (t.bm == 128 && t.bn == 128) return MCN_PERS_CAP(128, 128);
    return MCN_PERS_CAP(64, 64);
#undef MCN_PERS_CAP
}
// Counted statistics rows (NT_EPI_STATSC): the forward launch that runs persistent AND whose workgroups each stay on one
// channel block (tile v -> v + grid keeps the N tile when grid / 8 is a multiple of the N-tile count: xcd_remap adds v / 8
// to a per-XCD base).  Geometry only, so that mcn_conv2d_bnstats_rows() and the launch agree.  MCN_NT_STATSC=0: off.
template <typename T>
static bool nt_stats_counted(int mode, long M, int Nn, int nchunks, int tile, int tile_hint, long* grid_out) {
    static const int on = [] { const char* e = getenv("MCN_NT_STATSC"); return e ? atoi(e) : 1; }();
    const NtTile t = kNtCand[tile];
    if (!on || mode != NT_LINEAR || !nt_pers_tile(t, sizeof(T), NT_EPI_STATS)) return false;
    const long ntn = (Nn + t.bn - 1) / t.bn;
    const long W = ((M + t.bm - 1) / t.bm) * ntn;
    if (!(tile_hint & MCN_TILE_NOSPLIT) && sk_plan(tile, W, (nchunks + 7) >> 3, sizeof(T)).slices >= 2) return false;
    const long cap = nt_pers_cap<T>(tile, NT_EPI_STATSC);
    if (W > cap && (cap % 8 || (cap / 8) % ntn)) return false;
    if (grid_out) *grid_out = W < cap ? W : cap;
    return true;
}
template <typename T>
static int launch_nt_pers(const GemmNTParams& p, int tile, long W, hipStream_t st, int epi) {
    const NtTile t = kNtCand[tile];
    // (round 3, measured and dropped: a one-time start offset of 1.7 / 3.4 / 5.1 us between the residency slots of a CU, so that some
    // workgroups sit in their K loop while the others store — bf16 21.55 -> 21.60-21.64 ms, fp32 68.4-68.8 -> 68.7-69.0 ms per step, two
    // rounds each: the workgroups of a CU do not stay in lockstep long enough for a deliberate stagger to matter)
    const int lds = 2 * (t.bm + t.bn) * 128;
    const long cap = nt_pers_cap<T>(tile, epi);
    const dim3 grid((unsigned)(W < cap ? W : cap)), block(256);
#define MCN_LAUNCH_PERS(BMV, BNV)                                                                                              \
    do {                                                                                                                       \
        if (epi == NT_EPI_STATSC) hipLaunchKernelGGL((conv_gemm_nt_pers<T, BMV, BNV, 4, NT_EPI_STATSC>), grid, block, lds, st, p);      \
        else if (epi == NT_EPI_STATS) hipLaunchKernelGGL((conv_gemm_nt_pers<T, BMV, BNV, 4, NT_EPI_STATS>), grid, block, lds, st, p);   \
        else if (epi == NT_EPI_ACC) hipLaunchKernelGGL((conv_gemm_nt_pers<T, BMV, BNV, 4, NT_EPI_ACC>), grid, block, lds, st, p);       \
        else hipLaunchKernelGGL((conv_gemm_nt_pers<T, BMV, BNV, 4, NT_EPI_STORE>), grid, block, lds, st, p);                            \
    } while (0)
    if (t.bm == 128 && t.bn == 128) MCN_LAUNCH_PERS(128, 128);
    else if (t.bm == 64 && t.bn == 64) MCN_LAUNCH_PERS(64, 64);
    else MCN_FAIL(MCN_E_UNSUPPORTED, "conv: no persistent instantiation for this tile");
#undef MCN_LAUNCH_PERS
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// conv_gemm_nt_wpp: the geometry was admitted by nt_wpp_geom() (nt_hint) — checked again here from the GEMM parameters, loudly
template <typename T>
static int launch_nt_wpp(GemmNTParams p, const NtTile& t, long W, bool taps, hipStream_t st) {
    const bool same_grid = p.OH == p.IH && p.OW == p.IW && p.osy == 1 && p.osx == 1 && p.OHf == p.OH && p.OWf == p.OW;
    if (!taps || p.ntaps != 9 || p.cpt % 8 || p.sy != 1 || p.sx != 1 || !same_grid) MCN_FAIL(MCN_E_UNSUPPORTED, "conv: window ping-pong tile on a geometry it does not take");
    int dmin = 0x7fffffff, dmax = -0x7fffffff;
    for (int k = 0; k < p.ntaps; ++k) {
        const int d = (int)(short)(p.tap[k] & 0xffff) * p.IW + (p.tap[k] >> 16);
        if (d < dmin) dmin = d;
        if (d > dmax) dmax = d;
    }
    if (dmax - dmin > MCN_WPP_MAX_SPAN) MCN_FAIL(MCN_E_UNSUPPORTED, "conv: window ping-pong tile: the taps span more rows than the window holds");
    p.win_dmin = dmin;
    p.win_rows = 256 + (dmax - dmin);
    const int epi = nt_epi_of(p);
    const dim3 grid((unsigned)W), block(512);
#define MCN_LAUNCH_WPP_E(BNV, EPIV)                                                                                      \
    do {                                                                                                                 \
        constexpr int lds = 2 * WPP_WROWS * 128 + 3 * BNV * 128;                                                         \
        static const bool ok = allow_lds(conv_gemm_nt_wpp<T, BNV, EPIV>, lds);                                           \
        if (!ok) MCN_FAIL(MCN_E_UNSUPPORTED, "conv: the runtime refused the window ping-pong kernel's LDS request");         \
        hipLaunchKernelGGL((conv_gemm_nt_wpp<T, BNV, EPIV>), grid, block, lds, st, p);                                   \
    } while (0)
#define MCN_LAUNCH_WPP(BNV)                                                          \
    do {                                                                             \
        if (epi == NT_EPI_STATS) MCN_LAUNCH_WPP_E(BNV, NT_EPI_STATS);                \
        else if (epi == NT_EPI_ACC) MCN_LAUNCH_WPP_E(BNV, NT_EPI_ACC);               \
        else if (epi == NT_EPI_BNRED) MCN_LAUNCH_WPP_E(BNV, NT_EPI_BNRED);           \
        else if (epi == NT_EPI_ACCRED) MCN_LAUNCH_WPP_E(BNV, NT_EPI_ACCRED);         \
        else MCN_LAUNCH_WPP_E(BNV, NT_EPI_STORE);                                    \
    } while (0)
    if (t.bn == 128) MCN_LAUNCH_WPP(128);
    else MCN_LAUNCH_WPP(64);
#undef MCN_LAUNCH_WPP
#undef MCN_LAUNCH_WPP_E
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// sk_ws: scratch for the stream-K partials (may be null / too small: the conv then runs unsplit)
template <typename T>
static int launch_nt(GemmNTParams p, bool taps, int tile_hint, hipStream_t st, void* sk_ws = nullptr, size_t sk_ws_bytes = 0) {
    if (p.M <= 0 || p.Nn <= 0) return MCN_OK;
    const int tile = pick_nt_tile<T>(p.M, p.Nn, tile_hint & 0xff);
    const NtTile t = kNtCand[tile];
    p.m_begin = 0;
    p.m_end = p.M;
    // the epilogue addresses the output through a buffer descriptor: images x full output grid x channel stride
    p.out_bytes = (unsigned)((size_t)(p.M / (p.OH * p.OW)) * p.OHf * p.OWf * p.ldo * sizeof(T));
    const long W = (long)((p.M + t.bm - 1) / t.bm) * ((p.Nn + t.bn - 1) / t.bn);
    static const int epi_flags = [] { const char* e = getenv("MCN_NT_EPI_FLAGS"); return e ? atoi(e) : 0; }();
    p.epi_flags = epi_flags;
    if (t.wpp) {
        if constexpr (sizeof(T) == 2) return launch_nt_wpp<T>(p, t, W, taps, st);
        else MCN_FAIL(MCN_E_UNSUPPORTED, "conv: the window ping-pong tiles are for the 2-byte types only");
    }
    int mode = !taps ? NT_LINEAR : ((p.cpt % 8 == 0) ? NT_UNIFORM : NT_GENERIC);
    if (mode == NT_UNIFORM) {
        const bool same_grid = p.OH == p.IH && p.OW == p.IW && p.osy == 1 && p.osx == 1 && p.OHf == p.OH && p.OWf == p.OW;
        int dmin = 0;
        const int span = nt_window_span(sizeof(T), p.ntaps, p.cpt, p.sy, p.sx, same_grid, p.tap, p.IW, &dmin);
        if (span >= 0 && nt_window_lds(span, t)) {
            mode = NT_WINDOW;
            p.win_dmin = dmin;
            p.win_rows = t.bm + span;
        }
    }
    const SkPlan sp = sk_plan(tile, W, (p.nchunks + 7) >> 3, sizeof(T));
    if ((tile_hint & MCN_TILE_NOSPLIT) || sp.slices < 2 || !sk_ws || sk_ws_bytes < sp.bytes) {
        const int epi = nt_epi_of(p);
        // (counted statistics rows are a property of the geometry — mcn_conv2d_bnstats_rows() promised them to the BN side — so
        // that launch is persistent with or without a bias)
        if (p.stats && nt_stats_counted<T>(mode, p.M, p.Nn, p.nchunks, tile, tile_hint, nullptr)) return launch_nt_pers<T>(p, tile, W, st, NT_EPI_STATSC);
        if (mode == NT_LINEAR && !p.bias && epi != NT_EPI_BNRED && epi != NT_EPI_ACCRED && nt_pers_tile(t, sizeof(T), epi)) return launch_nt_pers<T>(p, tile, W, st, epi);
        return launch_nt_tiles<T>(p, tile, (int)W, mode, false, st);
    }
    p.sk_mode = 1;
    p.sk_body = sp.body;
    p.sk_slices = sp.slices;
    p.partial = (float*)sk_ws;
    int rc = launch_nt_tiles<T>(p, tile, sp.body + sp.tail * sp.slices, mode, false, st);
    if (rc) return rc;
    return launch_nt_tiles<T>(p, tile, sp.tail, mode, true, st);
}

// The 128x128 wgrad tile on 8 waves: measured +0.7 % of the bf16 step (23.73 -> 23.56 ms, two A/B pairs), -0.2 % in fp32 (whose
// wgrad overlaps the BN backward on the side stream anyway): 2-byte types only.  MCN_TN_NW8 = 0 / 1 forces it off / on.
static bool tn_nw8(size_t es) {
    static const int v = [] { const char* e = getenv("MCN_TN_NW8"); return e ? atoi(e) : -1; }();
    return v < 0 ? es == 2 : v != 0;
}
template <typename T>
static int launch_tn(const GemmTNParams& p_in, bool linear, int splits, int forced_tile, hipStream_t st) {
    int BR, BN;
    tn_tile(p_in.rows, p_in.Nn, DtypeOf<T>::value, linear, forced_tile, &BR, &BN);
    const int tiles = ((p_in.rows + BR - 1) / BR) * ((p_in.Nn + BN - 1) / BN);
    const bool ring = tn_ring(sizeof(T), BR, BN, linear);
    const bool nw8 = BR == 128 && BN == 128 && tn_nw8(sizeof(T));
    const dim3 grid(tiles, splits), block(nw8 || (ring && BR == 128) || BR == 256 ? 512 : 256);
    const int KP = sizeof(T) == 4 ? 32 : 64;
    // XCD-aware order (conv_gemm_tn): the tiles of a split run next to each other on one XCD.  Measured per layer (B = 256, serial
    // launches): bf16 28x28 128ch 3x3 123 -> 83 us, 56x56 128ch 3x3 / 2 130 -> 99, the 1x1 layers with 4-16 tiles per split -12...-30 %,
    // fp32 1x1 layers with 4-8 tiles -3...-7 %; the 3x3 layers with 256+ channels (36+ tiles per split, 15-30 long splits) LOSE 10-12 % in
    // both types, also with the tiles in sub-groups of 8 / 16 / 32 (MCN_TN_GRP), and the 1x1 layers with 64+ tiles are a wash: so the
    // order is used up to 16 tiles per split (32 for one-tap filters).  MCN_TN_GRP=0: off; N: sub-groups of N wherever splits > 1.
    static const int grp_env = [] { const char* e = getenv("MCN_TN_GRP"); return e ? atoi(e) : -1; }();
    GemmTNParams p = p_in;
    p.grp = 0;
    if (splits > 1) {
        if (grp_env > 0) {
            const int nch = (tiles + grp_env - 1) / grp_env;
            p.grp = (tiles + nch - 1) / nch;
        } else if (grp_env < 0 && tiles <= (p.ntaps == 1 ? 32 : 16)) {
            p.grp = tiles;
        }
    }
    // MCN_TN_LDS_KB (experiment): request at least this much LDS per wgrad workgroup — caps the workgroups per CU (81: one, 54: two) so
    // that the side stream's wgrad leaves LDS to the main stream's kernels
    static const int lds_floor = [] { const char* e = getenv("MCN_TN_LDS_KB"); return e ? atoi(e) * 1024 : 0; }();
#define MCN_LAUNCH_TN(BRV, BNV, LINV, NWV)                                           \
    do {                                                                             \
        int lds = 2 * KP * (BRV + BNV) * (int)sizeof(T);                             \
        if (lds < lds_floor) lds = lds_floor;                                        \
        static bool once = (allow_lds(conv_gemm_tn<T, BRV, BNV, LINV, NWV>, 160 * 1024), true); \
        (void)once;                                                                  \
        hipLaunchKernelGGL((conv_gemm_tn<T, BRV, BNV, LINV, NWV>), grid, block, lds, st, p); \
    } while (0)
#define MCN_LAUNCH_TN_LIN(BRV, BNV, NWV)                             \
    do {                                                             \
        if (linear) MCN_LAUNCH_TN(BRV, BNV, true, NWV); else MCN_LAUNCH_TN(BRV, BNV, false, NWV); \
    } while (0)
    const int ring2 = tn_ring2(sizeof(T), BR, BN, linear);
    if (ring2) {
        if constexpr (sizeof(T) == 2) {
            const int lds = ring2 * KP * (128 + 128) * (int)sizeof(T);
#define MCN_LAUNCH_TNR(KERN, LINV)                                                          \
    do {                                                                                    \
        static bool once = (allow_lds(KERN<T, 128, 128, LINV, 8>, 160 * 1024), true);       \
        (void)once;                                                                         \
        hipLaunchKernelGGL((KERN<T, 128, 128, LINV, 8>), grid, dim3(512), lds, st, p);      \
    } while (0)
            if (ring2 == 3) { if (linear) MCN_LAUNCH_TNR(conv_gemm_tn3, true); else MCN_LAUNCH_TNR(conv_gemm_tn3, false); }
            else { if (linear) MCN_LAUNCH_TNR(conv_gemm_tn4, true); else MCN_LAUNCH_TNR(conv_gemm_tn4, false); }
#undef MCN_LAUNCH_TNR
        }
    } else if (ring && BR == 64) {
        if constexpr (sizeof(T) == 4) {
            const int lds = 3 * KP * (64 + 64) * (int)sizeof(T);
            if (linear) hipLaunchKernelGGL((conv_gemm_tn3<T, 64, 64, true, 4>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((conv_gemm_tn3<T, 64, 64, false, 4>), grid, dim3(256), lds, st, p);
        }
    } else if (ring) {
        if constexpr (sizeof(T) == 4) {
            const int lds = 3 * KP * (128 + 128) * (int)sizeof(T);
            if (linear) {
                static bool once = (allow_lds(conv_gemm_tn3<T, 128, 128, true, 8>, lds), true);
                (void)once;
                hipLaunchKernelGGL((conv_gemm_tn3<T, 128, 128, true, 8>), grid, block, lds, st, p);
            } else {
                static bool once = (allow_lds(conv_gemm_tn3<T, 128, 128, false, 8>, lds), true);
                (void)once;
                hipLaunchKernelGGL((conv_gemm_tn3<T, 128, 128, false, 8>), grid, block, lds, st, p);
            }
        }
    } else if (BR == 256) {
        if constexpr (sizeof(T) == 2) MCN_LAUNCH_TN_LIN(256, 256, 8);
        else MCN_FAIL(MCN_E_UNSUPPORTED, "conv: the 256x256 wgrad tile is for the 2-byte types");
    } else if (nw8) MCN_LAUNCH_TN_LIN(128, 128, 8);
    else if (BR == 128 && BN == 128) MCN_LAUNCH_TN_LIN(128, 128, 4);
    else if (BR == 128) MCN_LAUNCH_TN_LIN(128, 64, 4);
    else if (BN == 128) MCN_LAUNCH_TN_LIN(64, 128, 4);
    else MCN_LAUNCH_TN_LIN(64, 64, 4);
#undef MCN_LAUNCH_TN_LIN
#undef MCN_LAUNCH_TN
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

template <typename T>
static int launch_pack(const PackParams& p, hipStream_t st) {
    if (p.mode >= 2) {                                   // Winograd filter transform (fp32 only)
        const int Kin = p.mode == 3 ? p.Cout : p.Cin, Kout = p.mode == 3 ? p.Cin : p.Cout;
        const long tot = (long)((Kout + 63) / 64) * (Kin / 32) * 2048;
        hipLaunchKernelGGL(wino_filter_transform_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, p.w, (float*)p.out, p.Cin, p.Cout, p.mode == 3 ? 1 : 0);
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    const long total = (long)p.rows * p.ntaps * p.Cp;
    if (total <= 0) return MCN_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((pack_weights_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

static NaiveConvParams naive_params(const Geo& g) {
    NaiveConvParams p;
    memset(&p, 0, sizeof(p));
    p.N = g.N; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.Cout = g.Cout; p.KH = g.KH; p.KW = g.KW; p.SH = g.SH; p.SW = g.SW;
    p.DH = g.DH; p.DW = g.DW; p.padT = g.pT; p.padL = g.pL; p.OH = g.OH; p.OW = g.OW; p.x_cs = g.xcs;
    p.scale = 1.f;
    return p;
}
static inline unsigned nblocks(long total, int cap = 8192) {
    long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// K-sliced tail of the Winograd forward / dgrad (launch_wino): slices per block of the short last round, 1 = unsplit.  Shared by the launch
// and by the introspection entry points (mcn_conv2d_kslices / mcn_conv2d_launch_list), which assume the executor's workspace
// (mcn_conv2d_workspace_bytes: MCN_SK_MAX_BYTES behind the packed operand).
static int wino_tail_slices(int total, int Kin, bool nosplit, bool have_ws, size_t sk_ws_bytes) {
    static const int tail_on = [] { const char* e = getenv("MCN_WINO_TAIL"); return e ? atoi(e) : 1; }();
    const int ns = Kin / 32, rem = total % MCN_NUM_CU;
    int slices = 1;
    if (tail_on && !nosplit && total > MCN_NUM_CU && rem > 0 && rem <= MCN_NUM_CU / 2 && ns >= 2 && have_ws) {
        slices = MCN_NUM_CU / rem;
        if (slices > ns) slices = ns;
        if (slices > 8) slices = 8;
        while (slices > 1 && (size_t)rem * slices * 8 * 128 * 64 * sizeof(float) > sk_ws_bytes) --slices;
        const int per = (ns + slices - 1) / slices;
        slices = (ns + per - 1) / per;                     // no empty slices
    }
    return slices;
}
static int wino_geom_slices(int N, int H, int W, int Kin, int Kout, bool nosplit) {
    const int total = (((long)N * ((H + 1) / 2) * ((W + 1) / 2) + 63) / 64) * ((Kout + 63) / 64);
    return wino_tail_slices(total, Kin, nosplit, true, MCN_SK_MAX_BYTES);
}

// ---- forward -------------------------------------------------------------------------------------------
// Winograd launch: `in` [N][H][W][Cs] (Kin channels) -> `out` [N][H][W][Kout]; exactly one of stats / red_part may be set
static int launch_wino(const void* in, const void* u, void* out, const float* bias, int N, int H, int W, int Cs, int Kin, int Kout, float* stats,
                       const void* red_x, const unsigned char* red_mask, float* red_part, hipStream_t st, bool accumulate = false, void* sk_ws = nullptr,
                       size_t sk_ws_bytes = 0, bool nosplit = false) {
    WinoParams p;
    memset(&p, 0, sizeof(p));
    p.in = (const float*)in; p.u = (const float*)u; p.out = (float*)out; p.bias = bias;
    p.H = H; p.W = W; p.Cs = Cs; p.Cin = Kin; p.TH = (H + 1) / 2; p.TW = (W + 1) / 2; p.ntiles = N * p.TH * p.TW;
    p.Nn = Kout; p.ldo = Kout;
    p.in_bytes = (unsigned)((size_t)N * H * W * Cs * sizeof(float));
    p.u_bytes = (unsigned)wino_u_bytes(Kin, Kout);
    p.out_bytes = (unsigned)((size_t)N * H * W * Kout * sizeof(float));
    p.stats = stats; p.red_x = (const float*)red_x; p.red_mask = red_mask; p.red_part = red_part; p.red_row0 = 0;
    const int lds = 2 * WINO_STAGE;
    const int total = ((p.ntiles + 63) / 64) * ((Kout + 63) / 64);
    // K-sliced tail: one workgroup per CU, so `total` blocks run in ceil(total / 256) rounds; when the last round is short its blocks are cut
    // along the 32-channel super-steps into slices that fill the chip once more (a second launch parks their accumulators, a third sums them in a
    // fixed order and runs the epilogue: deterministic).  14 x 14, 256 -> 256, B = 256: 784 blocks = 3 rounds + 16 blocks -> 64 slices of a quarter block.
    const int rem = total % MCN_NUM_CU;
    const int slices = wino_tail_slices(total, Kin, nosplit, sk_ws != nullptr, sk_ws_bytes);
    p.sk_slices = slices;
    p.sk_body = slices > 1 ? total - rem : total;
    p.partial = (float*)sk_ws;
    const dim3 grid((unsigned)p.sk_body), sgrid((unsigned)(rem * slices)), rgrid((unsigned)rem);
    // (the attribute is set ONCE per instantiation — 40 Winograd calls per step, also under graph capture — and checked; every
    // launch asks for the same WINO_LDS_MAX-bounded size class, so the one-time request covers the largest)
#define MCN_WINO_LAUNCH(EPIV)                                                                                          \
    do {                                                                                                               \
        static const bool ok0 = allow_lds(conv_wino_f2k3_w8<0, EPIV>, 160 * 1024);                                     \
        static const bool ok1 = allow_lds(conv_wino_f2k3_w8<WINO_SLICE, NT_EPI_STORE>, 160 * 1024);                    \
        static const bool ok2 = allow_lds(conv_wino_f2k3_w8<WINO_REDUCE, EPIV>, 160 * 1024);                           \
        if (!ok0 || !ok1 || !ok2) MCN_FAIL(MCN_E_LAUNCH, "conv (Winograd): the runtime refused %d bytes of LDS", lds); \
        hipLaunchKernelGGL((conv_wino_f2k3_w8<0, EPIV>), grid, dim3(512), lds, st, p);                                 \
        if (slices > 1) {                                                                                              \
            hipLaunchKernelGGL((conv_wino_f2k3_w8<WINO_SLICE, NT_EPI_STORE>), sgrid, dim3(512), lds, st, p);           \
            hipLaunchKernelGGL((conv_wino_f2k3_w8<WINO_REDUCE, EPIV>), rgrid, dim3(512), lds, st, p);                  \
        }                                                                                                              \
    } while (0)
    if (stats) MCN_WINO_LAUNCH(NT_EPI_STATS);
    else if (accumulate) MCN_WINO_LAUNCH(NT_EPI_ACC);
    else if (red_part) MCN_WINO_LAUNCH(NT_EPI_BNRED);
    else MCN_WINO_LAUNCH(NT_EPI_STORE);
#undef MCN_WINO_LAUNCH
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

template <typename T>
static int conv_fwd_t(const void* x, const float* w, const void* w_packed, const float* bias, void* y, const Geo& g, mcn_dtype dt,
                      void* ws, size_t ws_bytes, hipStream_t st, float* stats = nullptr) {
    const long M = (long)g.N * g.OH * g.OW;
    if (M == 0) return MCN_OK;
    if (!mfma_path_ok(g, dt)) {
        if (stats) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_fwd_bnstats: geometry takes the fallback kernel (mcn_conv2d_bnstats_rows() == 0)");
        if (skinny_ok(g, dt, MCN_SKINNY_MAX_CO) && ws && ws_bytes >= skinny_w_bytes(g)) {
            const int CO = skinny_co(g);
            float* wp = (float*)ws;
            hipLaunchKernelGGL((skinny_pack_w<T>), dim3((g.Cin * CO + 255) / 256), dim3(256), 0, st, w, wp, g.Cin, g.Cout, CO, 0);
            if (skinny_split(M, g.Cin / ce_of(dt))) {
                const dim3 grid((unsigned)((M + 3) / 4));
#define MCN_SKINNY_FWD(COV) hipLaunchKernelGGL((skinny_conv_fwd_split<T, COV>), grid, dim3(256), 0, st, (const T*)x, (const float*)wp, bias, (T*)y, M, g.Cin, g.Cout, 0)
                if (CO == 8) MCN_SKINNY_FWD(8); else if (CO == 16) MCN_SKINNY_FWD(16); else if (CO == 24) MCN_SKINNY_FWD(24); else MCN_SKINNY_FWD(32);
#undef MCN_SKINNY_FWD
                MCN_CHECK_LAUNCH();
                return MCN_OK;
            }
            const dim3 grid(nblocks(M, 4096));
#define MCN_SKINNY_FWD(COV) hipLaunchKernelGGL((skinny_conv_fwd<T, COV>), grid, dim3(256), 0, st, (const T*)x, (const float*)wp, bias, (T*)y, M, g.Cin, g.Cout, 0)
            if (CO == 8) MCN_SKINNY_FWD(8); else if (CO == 16) MCN_SKINNY_FWD(16); else if (CO == 24) MCN_SKINNY_FWD(24); else MCN_SKINNY_FWD(32);
#undef MCN_SKINNY_FWD
            MCN_CHECK_LAUNCH();
            return MCN_OK;
        }
        NaiveConvParams p = naive_params(g);
        p.x = x; p.w = w; p.y = y; p.bias = bias;
        hipLaunchKernelGGL((naive_conv_fwd<T>), dim3(nblocks(M * g.Cout)), dim3(256), 0, st, p);
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    const size_t need = fwd_pack_bytes(g, dt);
    if (!w_packed && (!ws || ws_bytes < need)) MCN_FAIL(MCN_E_WORKSPACE, "conv2d_fwd: workspace %zu < %zu", ws_bytes, need);
    if (wino_fwd_ok(g, dt)) {
        if (!w_packed) {
            PackParams wk;
            memset(&wk, 0, sizeof(wk));
            wk.w = w; wk.out = ws; wk.Cin = g.Cin; wk.Cout = g.Cout; wk.mode = 2;
            int rc = launch_pack<T>(wk, st);
            if (rc) return rc;
        }
        const size_t used = w_packed ? 0 : need;
        return launch_wino(x, w_packed ? w_packed : ws, y, bias, g.N, g.H, g.W, g.xcs, g.Cin, g.Cout, stats, nullptr, nullptr, nullptr, st, false,
                           ws ? (char*)ws + used : nullptr, ws && ws_bytes > used ? ws_bytes - used : 0, (g.tile & MCN_TILE_NOSPLIT) != 0);
    }
    const int ce = ce_of(dt), Cp = round_up(g.Cin, ce), ntaps = g.KH * g.KW;
    PackParams pk;
    memset(&pk, 0, sizeof(pk));
    pk.w = w; pk.out = ws; pk.KW = g.KW; pk.Cin = g.Cin; pk.Cout = g.Cout; pk.rows = g.Cout; pk.Cp = Cp; pk.ntaps = ntaps; pk.mode = 0;
    GemmNTParams p;
    memset(&p, 0, sizeof(p));
    for (int r = 0; r < g.KH; ++r)
        for (int s = 0; s < g.KW; ++s) {
            const int t = r * g.KW + s;
            pk.tr[t] = (signed char)r; pk.ts[t] = (signed char)s;
            p.tap[t] = ((r * g.DH - g.pT) & 0xffff) | ((s * g.DW - g.pL) << 16);
        }
    if (!w_packed) {                                     // per-use cast / re-pack of the fp32 master (convnet.py:1421-1422)
        int rc = launch_pack<T>(pk, st);
        if (rc) return rc;
    }
    p.in = x; p.wt = w_packed ? w_packed : ws; p.out = y; p.bias = bias; p.stats = stats;
    p.M = (int)M; p.OH = g.OH; p.OW = g.OW; p.IH = g.H; p.IW = g.W; p.Cs = g.xcs;
    p.cpt = Cp / ce; p.ntaps = ntaps; p.nchunks = ntaps * p.cpt; p.sy = g.SH; p.sx = g.SW; p.Nn = g.Cout;
    p.OHf = g.OH; p.OWf = g.OW; p.ldo = g.Cout; p.osy = 1; p.osx = 1; p.oy0 = 0; p.ox0 = 0; p.accumulate = 0;
    p.in_bytes = (unsigned)((size_t)g.N * g.H * g.W * g.xcs * sizeof(T));
    p.wt_bytes = (unsigned)((size_t)g.Cout * ntaps * Cp * sizeof(T));
    const bool linear = ntaps == 1 && g.SH == 1 && g.SW == 1 && g.pT == 0 && g.pL == 0 && g.OH == g.H && g.OW == g.W;
    // workspace: [packed weights unless the caller keeps them | stream-K partials]
    const size_t used = w_packed ? 0 : need;
    return launch_nt<T>(p, !linear, nt_hint(g, DtypeOf<T>::value, false), st, ws ? (char*)ws + used : nullptr, ws && ws_bytes > used ? ws_bytes - used : 0);
}

extern "C" int mcn_conv2d_fwd(const void* x, const float* w, const void* w_packed, const float* bias, void* y, const mcn_conv_geom* gg,
                              mcn_dtype dtype, mcn_layout layout, void* ws, size_t ws_bytes, void* stream) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (layout != MCN_NHWC) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_fwd: only NHWC activations (convert with mcn_input_prep)");
    if (!x || !w || !y) MCN_FAIL(MCN_E_BADARG, "conv2d_fwd: null pointer");
    if (bias && g.Cout % 4) {
        if (mfma_path_ok(g, dtype)) MCN_FAIL(MCN_E_BADARG, "conv2d_fwd: internal: bias with Cout%%4");
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return conv_fwd_t<float>(x, w, w_packed, bias, y, g, dtype, ws, ws_bytes, st);
    if (dtype == MCN_BF16) return conv_fwd_t<bf16_t>(x, w, w_packed, bias, y, g, dtype, ws, ws_bytes, st);
    else if (dtype == MCN_F16) return conv_fwd_t<f16_t>(x, w, w_packed, bias, y, g, dtype, ws, ws_bytes, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_fwd: dtype %d unsupported", (int)dtype);
}

// forward conv that also emits the batch-norm statistics partials of its output (consumed by mcn_bn_fwd_train_fused)
extern "C" int32_t mcn_conv2d_bnstats_rows(const mcn_conv_geom* gg, mcn_dtype dtype, int32_t* rows_per_partial) {
    Geo g;
    if (rows_per_partial) *rows_per_partial = 0;
    if (!gg || geo_from(gg, &g)) return 0;
    if (!mcn_dtype_ok(dtype) || !mfma_path_ok(g, dtype)) return 0;
    const long M = (long)g.N * g.OH * g.OW;
    if (M <= 0) return 0;
    if (wino_fwd_ok(g, dtype)) return wino_rows(g);      // counted rows [rows][4][Cout] (*rows_per_partial stays 0)
    const NtTile* cand = kNtCand;
    const int t = dtype == MCN_F32 ? pick_nt_tile<float>((int)M, g.Cout, nt_hint(g, dtype, false)) : pick_nt_tile<bf16_t>((int)M, g.Cout, nt_hint(g, dtype, false));
    const int wrows = cand[t].nw / 2;
    // counted rows (rows_per_partial = -BN): one row per persistent workgroup (its wave rows are merged in the flush), [4][BN] floats each
    const int ce = ce_of(dtype), cpt = round_up(g.Cin, ce) / ce;
    const int mode = conv_is_linear(g) ? NT_LINEAR : NT_UNIFORM;
    long grid = 0;
    const bool counted = dtype == MCN_F32   ? nt_stats_counted<float>(mode, M, g.Cout, g.KH * g.KW * cpt, t, nt_hint(g, dtype, false), &grid)
                         : dtype == MCN_F16 ? nt_stats_counted<f16_t>(mode, M, g.Cout, g.KH * g.KW * cpt, t, nt_hint(g, dtype, false), &grid)
                                            : nt_stats_counted<bf16_t>(mode, M, g.Cout, g.KH * g.KW * cpt, t, nt_hint(g, dtype, false), &grid);
    if (counted) {
        // compact counted rows: keyed by the channel block of the persistent workgroup (conv_kernels.h, nt_stats_flush)
        if (rows_per_partial) *rows_per_partial = -(g.Cout < cand[t].bn ? g.Cout : cand[t].bn);      // (a layer narrower than the tile: one block of Cout channels)
        return (int32_t)grid;
    }
    if (rows_per_partial) *rows_per_partial = cand[t].bm / wrows;
    return (int32_t)(wrows * ((M + cand[t].bm - 1) / cand[t].bm));
}
extern "C" int mcn_conv2d_fwd_bnstats(const void* x, const float* w, const void* w_packed, const float* bias, void* y, float* stats_partials,
                                      const mcn_conv_geom* gg, mcn_dtype dtype, mcn_layout layout, void* ws, size_t ws_bytes, void* stream) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (layout != MCN_NHWC) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_fwd_bnstats: only NHWC activations");
    if (!x || !w || !y || !stats_partials) MCN_FAIL(MCN_E_BADARG, "conv2d_fwd_bnstats: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return conv_fwd_t<float>(x, w, w_packed, bias, y, g, dtype, ws, ws_bytes, st, stats_partials);
    if (dtype == MCN_BF16) return conv_fwd_t<bf16_t>(x, w, w_packed, bias, y, g, dtype, ws, ws_bytes, st, stats_partials);
    else if (dtype == MCN_F16) return conv_fwd_t<f16_t>(x, w, w_packed, bias, y, g, dtype, ws, ws_bytes, st, stats_partials);
    MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_fwd_bnstats: dtype %d unsupported", (int)dtype);
}

// ---- dgrad -----------------------------------------------------------------------------------------------
static inline int pos_mod(int a, int m) { return ((a % m) + m) % m; }

template <typename T>
static int conv_dgrad_t(const void* dy, const float* w, const void* w_packed, void* dx, const Geo& g, int accumulate, mcn_dtype dt,
                        void* ws, size_t ws_bytes, hipStream_t st, const void* add_src = nullptr, const unsigned char* add_mask = nullptr,
                        const void* red_x = nullptr, const unsigned char* red_mask = nullptr, float* red_part = nullptr) {
    const long Min = (long)g.N * g.H * g.W;
    if (Min == 0) return MCN_OK;
    if (!mfma_dgrad_ok(g, dt)) {
        if (skinny_ok(g, dt, MCN_SKINNY_MAX_CO) && !add_src && ws && ws_bytes >= skinny_w_bytes(g)) {
            const int CO = skinny_co(g);
            float* wp = (float*)ws;
            hipLaunchKernelGGL((skinny_pack_w<T>), dim3((g.Cin * CO + 255) / 256), dim3(256), 0, st, w, wp, g.Cin, g.Cout, CO, 0);
            // few pixels (SE bottlenecks: Min = batch): spread the channel chunks over gridDim.y instead of one long loop per thread
            const unsigned bx = nblocks(Min, 4096);
            const int nch = g.Cin / ce_of(dt);
            long gy = 2048 / bx;
            if (gy > nch) gy = nch;
            if (gy < 1) gy = 1;
            const dim3 grid(bx, (unsigned)gy);
#define MCN_SKINNY_DGRAD(COV) hipLaunchKernelGGL((skinny_conv_dgrad<T, COV>), grid, dim3(256), 0, st, (const T*)dy, (const float*)wp, (T*)dx, Min, g.Cin, g.Cout, accumulate)
            if (CO == 8) MCN_SKINNY_DGRAD(8); else if (CO == 16) MCN_SKINNY_DGRAD(16); else if (CO == 24) MCN_SKINNY_DGRAD(24); else MCN_SKINNY_DGRAD(32);
#undef MCN_SKINNY_DGRAD
            MCN_CHECK_LAUNCH();
            return MCN_OK;
        }
        if (skinny_in_ok(g, dt, MCN_SKINNY_MAX_CO) && !add_src && ws && ws_bytes >= skinny_in_w_bytes(g)) {
            // few input channels: dx[pixel][c] = sum_n dy[pixel][n] * w[c][n] is the skinny FORWARD over dy with the transposed table
            const int CO = skinny_ci(g);
            float* wp = (float*)ws;
            hipLaunchKernelGGL((skinny_pack_w<T>), dim3((g.Cout * CO + 255) / 256), dim3(256), 0, st, w, wp, g.Cout, g.Cin, CO, 1);
            if (skinny_split(Min, g.Cout / ce_of(dt))) {
                const dim3 grid((unsigned)((Min + 3) / 4));
#define MCN_SKINNY_DGRAD_IN(COV) hipLaunchKernelGGL((skinny_conv_fwd_split<T, COV>), grid, dim3(256), 0, st, (const T*)dy, (const float*)wp, (const float*)nullptr, (T*)dx, Min, g.Cout, g.Cin, accumulate)
                if (CO == 8) MCN_SKINNY_DGRAD_IN(8); else if (CO == 16) MCN_SKINNY_DGRAD_IN(16); else if (CO == 24) MCN_SKINNY_DGRAD_IN(24); else MCN_SKINNY_DGRAD_IN(32);
#undef MCN_SKINNY_DGRAD_IN
                MCN_CHECK_LAUNCH();
                return MCN_OK;
            }
            const dim3 grid(nblocks(Min, 4096));
#define MCN_SKINNY_DGRAD_IN(COV) hipLaunchKernelGGL((skinny_conv_fwd<T, COV>), grid, dim3(256), 0, st, (const T*)dy, (const float*)wp, (const float*)nullptr, (T*)dx, Min, g.Cout, g.Cin, accumulate)
            if (CO == 8) MCN_SKINNY_DGRAD_IN(8); else if (CO == 16) MCN_SKINNY_DGRAD_IN(16); else if (CO == 24) MCN_SKINNY_DGRAD_IN(24); else MCN_SKINNY_DGRAD_IN(32);
#undef MCN_SKINNY_DGRAD_IN
            MCN_CHECK_LAUNCH();
            return MCN_OK;
        }
        NaiveConvParams p = naive_params(g);
        p.dy = dy; p.w = w; p.dx = dx; p.accumulate = accumulate;
        hipLaunchKernelGGL((naive_conv_dgrad<T>), dim3(nblocks(Min * g.Cin)), dim3(256), 0, st, p);
        MCN_CHECK_LAUNCH();
        return MCN_OK;
    }
    const size_t need = dgrad_pack_bytes(g, dt);
    if (!w_packed && (!ws || ws_bytes < need)) MCN_FAIL(MCN_E_WORKSPACE, "conv2d_dgrad: workspace %zu < %zu", ws_bytes, need);
    if (wino_dgrad_ok(g, dt)) {                          // (eligibility depends on the geometry alone: the packed operand is U)
        if (add_src) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad: masked fan-in on a Winograd layer (mcn_conv2d_dgrad_addmasked_ok() == 0)");
        if (accumulate && red_part) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad: accumulate + BN-backward sums");
        // the gradient of a 3x3 / stride 1 / pad 1 convolution is the same convolution of dy with the filter rotated by 180 degrees
        if (!w_packed) {
            PackParams wk;
            memset(&wk, 0, sizeof(wk));
            wk.w = w; wk.out = ws; wk.Cin = g.Cin; wk.Cout = g.Cout; wk.mode = 3;
            int rc = launch_pack<T>(wk, st);
            if (rc) return rc;
        }
        const size_t used = w_packed ? 0 : need;
        return launch_wino(dy, w_packed ? w_packed : ws, dx, nullptr, g.N, g.H, g.W, g.Cout, g.Cout, g.Cin, nullptr, red_x, red_mask, red_part, st, accumulate != 0,
                           ws ? (char*)ws + used : nullptr, ws && ws_bytes > used ? ws_bytes - used : 0, (g.tile & MCN_TILE_NOSPLIT) != 0);
    }
    const int ce = ce_of(dt), Cp = round_up(g.Cout, ce);

    // one exact sub-convolution per stride-parity class of dx
    struct Cls { int py, px, nt; signed char r[MCN_MAX_TAPS], s[MCN_MAX_TAPS], dy[MCN_MAX_TAPS], dx[MCN_MAX_TAPS]; };
    bool any_empty = false;
    Cls* cls = (Cls*)alloca(sizeof(Cls) * g.SH * g.SW);
    int ncls = 0;
    for (int py = 0; py < g.SH; ++py)
        for (int px = 0; px < g.SW; ++px) {
            if (py >= g.H || px >= g.W) continue;
            Cls& c = cls[ncls];
            c.py = py; c.px = px; c.nt = 0;
            for (int r = 0; r < g.KH; ++r) {
                const int ty = py + g.pT - r * g.DH;
                if (pos_mod(ty, g.SH)) continue;
                for (int s = 0; s < g.KW; ++s) {
                    const int tx = px + g.pL - s * g.DW;
                    if (pos_mod(tx, g.SW)) continue;
                    c.r[c.nt] = (signed char)r; c.s[c.nt] = (signed char)s;
                    c.dy[c.nt] = (signed char)(ty / g.SH); c.dx[c.nt] = (signed char)(tx / g.SW);
                    c.nt++;
                }
            }
            if (c.nt == 0) any_empty = true; else ncls++;
        }
    if (any_empty && !accumulate) {
        if (hipMemsetAsync(dx, 0, (size_t)Min * g.Cin * sizeof(T), st) != hipSuccess) MCN_FAIL(MCN_E_LAUNCH, "conv2d_dgrad: memset failed");
    }
    char* wsp = w_packed ? (char*)const_cast<void*>(w_packed) : (char*)ws;
    int red_row = 0;                                   // BN-backward partial rows: one block of rows per stride-parity launch
    for (int k = 0; k < ncls; ++k) {
        const Cls& c = cls[k];
        const int OHs = (g.H - c.py + g.SH - 1) / g.SH, OWs = (g.W - c.px + g.SW - 1) / g.SW;
        PackParams pk;
        memset(&pk, 0, sizeof(pk));
        pk.w = w; pk.out = wsp; pk.KW = g.KW; pk.Cin = g.Cin; pk.Cout = g.Cout; pk.rows = g.Cin; pk.Cp = Cp; pk.ntaps = c.nt; pk.mode = 1;
        GemmNTParams p;
        memset(&p, 0, sizeof(p));
        bool zero_off = true;
        for (int t = 0; t < c.nt; ++t) {
            pk.tr[t] = c.r[t]; pk.ts[t] = c.s[t];
            p.tap[t] = ((int)c.dy[t] & 0xffff) | ((int)c.dx[t] << 16);
            if (c.dy[t] || c.dx[t]) zero_off = false;
        }
        int rc = MCN_OK;
        if (!w_packed) {
            rc = launch_pack<T>(pk, st);
            if (rc) return rc;
        }
        p.in = dy; p.wt = wsp; p.out = dx; p.bias = nullptr;
        p.M = g.N * OHs * OWs; p.OH = OHs; p.OW = OWs; p.IH = g.OH; p.IW = g.OW; p.Cs = g.Cout;
        p.cpt = Cp / ce; p.ntaps = c.nt; p.nchunks = c.nt * p.cpt; p.sy = 1; p.sx = 1; p.Nn = g.Cin;
        p.OHf = g.H; p.OWf = g.W; p.ldo = g.Cin; p.osy = g.SH; p.osx = g.SW; p.oy0 = c.py; p.ox0 = c.px; p.accumulate = add_src ? 2 : accumulate;
        p.add_src = add_src; p.add_mask = add_mask;
        p.red_x = red_x; p.red_mask = red_mask; p.red_part = red_part; p.red_row0 = red_row;
        if (red_part) {
            const NtTile rt = kNtCand[pick_nt_tile<T>(p.M, p.Nn, nt_hint(g, DtypeOf<T>::value, true))];
            red_row += (rt.nw / 2) * ((p.M + rt.bm - 1) / rt.bm);
        }
        p.in_bytes = (unsigned)((size_t)g.N * g.OH * g.OW * g.Cout * sizeof(T));
        p.wt_bytes = (unsigned)((size_t)g.Cin * c.nt * Cp * sizeof(T));
        const bool linear = c.nt == 1 && zero_off && OHs == g.OH && OWs == g.OW;
        const size_t used = w_packed ? 0 : need;
        rc = launch_nt<T>(p, !linear, nt_hint(g, DtypeOf<T>::value, true), st, ws ? (char*)ws + used : nullptr, ws && ws_bytes > used ? ws_bytes - used : 0);
        if (rc) return rc;
        wsp += align_up((size_t)g.Cin * c.nt * Cp * sizeof(T), 256);
    }
    return MCN_OK;
}

extern "C" int mcn_conv2d_dgrad(const void* dy, const float* w, const void* w_packed, void* dx, const mcn_conv_geom* gg, int accumulate,
                                mcn_dtype dtype, mcn_layout layout, void* ws, size_t ws_bytes, void* stream) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (layout != MCN_NHWC) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad: only NHWC activations");
    if (!dy || !w || !dx) MCN_FAIL(MCN_E_BADARG, "conv2d_dgrad: null pointer");
    if (g.xcs != g.Cin) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad: dx must be dense (x_cs == Cin)");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return conv_dgrad_t<float>(dy, w, w_packed, dx, g, accumulate, dtype, ws, ws_bytes, st);
    if (dtype == MCN_BF16) return conv_dgrad_t<bf16_t>(dy, w, w_packed, dx, g, accumulate, dtype, ws, ws_bytes, st);
    else if (dtype == MCN_F16) return conv_dgrad_t<f16_t>(dy, w, w_packed, dx, g, accumulate, dtype, ws, ws_bytes, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad: dtype %d unsupported", (int)dtype);
}

// dgrad that also adds the masked gradient of a residual block's output: dx = dgrad(dy) + add_src * [add_mask bit].
// Stride-1 geometries on the MFMA path only (one parity class, every dx element written exactly once, dx dense).
extern "C" int32_t mcn_conv2d_dgrad_addmasked_ok(const mcn_conv_geom* gg, mcn_dtype dtype) {
    Geo g;
    if (!gg || geo_from(gg, &g)) return 0;
    if (!mcn_dtype_ok(dtype)) return 0;
    if (wino_dgrad_ok(g, dtype)) return 0;              // (the Winograd dgrad has no masked fan-in epilogue; the block's first conv is 1x1 anyway)
    return (g.SH == 1 && g.SW == 1 && g.xcs == g.Cin && mfma_dgrad_ok(g, dtype) && g.Cin % ce_of(dtype) == 0) ? 1 : 0;
}
extern "C" int mcn_conv2d_dgrad_addmasked(const void* dy, const float* w, const void* w_packed, void* dx, const void* add_src,
                                          const uint8_t* add_mask, const mcn_conv_geom* gg, mcn_dtype dtype, mcn_layout layout, void* ws,
                                          size_t ws_bytes, void* stream) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (layout != MCN_NHWC) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad_addmasked: only NHWC activations");
    if (!dy || !w || !dx || !add_src || !add_mask) MCN_FAIL(MCN_E_BADARG, "conv2d_dgrad_addmasked: null pointer");
    if (!mcn_conv2d_dgrad_addmasked_ok(gg, dtype)) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad_addmasked: geometry not eligible (see mcn_conv2d_dgrad_addmasked_ok)");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return conv_dgrad_t<float>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, add_src, add_mask);
    if (dtype == MCN_F16) return conv_dgrad_t<f16_t>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, add_src, add_mask);
    return conv_dgrad_t<bf16_t>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, add_src, add_mask);
}

// dgrad whose output dx IS the gradient of a BN + ReLU's output (the conv is that BN's only reader): the epilogue also accumulates the
// BN's backward sums — per (M tile, wave row) partial rows [rows][2][Cin]: sum dy', sum dy' * x with dy' = the stored dx where the
// forward's ReLU bit is set, x = the BN's input — so that the BN backward needs no reduction pass over (dy, x)
// (mcn_bn_bwd_from_partials).  rows: mcn_conv2d_dgrad_bnred_rows() (0 = geometry not eligible: use mcn_conv2d_dgrad).
extern "C" int32_t mcn_conv2d_dgrad_bnred_rows(const mcn_conv_geom* gg, mcn_dtype dtype) {
    Geo g;
    if (!gg || geo_from(gg, &g) || !mcn_dtype_ok(dtype) || !mfma_dgrad_ok(g, dtype) || g.xcs != g.Cin) return 0;
    if (wino_dgrad_ok(g, dtype)) return wino_rows(g);
    long rows = 0;
    for (int py = 0; py < g.SH && py < g.H; ++py)
        for (int px = 0; px < g.SW && px < g.W; ++px) {
            int nt = 0;
            for (int r = 0; r < g.KH; ++r)
                for (int s2 = 0; s2 < g.KW; ++s2)
                    if (!pos_mod(py + g.pT - r * g.DH, g.SH) && !pos_mod(px + g.pL - s2 * g.DW, g.SW)) nt++;
            if (!nt) continue;
            const int OHs = (g.H - py + g.SH - 1) / g.SH, OWs = (g.W - px + g.SW - 1) / g.SW;
            const long M = (long)g.N * OHs * OWs;
            const int t = dtype == MCN_F32 ? pick_nt_tile<float>((int)M, g.Cin, nt_hint(g, dtype, true)) : pick_nt_tile<bf16_t>((int)M, g.Cin, nt_hint(g, dtype, true));
            rows += (long)(kNtCand[t].nw / 2) * ((M + kNtCand[t].bm - 1) / kNtCand[t].bm);
        }
    return rows > 0x7fffffffl ? 0 : (int32_t)rows;
}
extern "C" int mcn_conv2d_dgrad_bnred(const void* dy, const float* w, const void* w_packed, void* dx, const void* bn_x, const uint8_t* relu_mask,
                                      float* red_partials, const mcn_conv_geom* gg, mcn_dtype dtype, mcn_layout layout, void* ws, size_t ws_bytes,
                                      void* stream) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (layout != MCN_NHWC) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad_bnred: only NHWC activations");
    if (!dy || !w || !dx || !bn_x || !relu_mask || !red_partials) MCN_FAIL(MCN_E_BADARG, "conv2d_dgrad_bnred: null pointer");
    if (mcn_conv2d_dgrad_bnred_rows(gg, dtype) <= 0) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad_bnred: geometry not eligible (mcn_conv2d_dgrad_bnred_rows() == 0)");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return conv_dgrad_t<float>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, nullptr, nullptr, bn_x, relu_mask, red_partials);
    if (dtype == MCN_F16) return conv_dgrad_t<f16_t>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, nullptr, nullptr, bn_x, relu_mask, red_partials);
    return conv_dgrad_t<bf16_t>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, nullptr, nullptr, bn_x, relu_mask, red_partials);
}

// Both at once (round 4): dx = dgrad(dy) + add_src * [add_mask bit] is the COMPLETE gradient of a residual unit's output y_b = relu(bn(x_b) + skip)
// (its two readers are this conv and the next unit's residual add), so the backward sums of that unit's output BN ride in the same
// epilogue: red_partials [rows][2][Cin] = sum dy', sum dy' * bn_x over the pixel rows of a wave row, dy' = the stored dx where relu_mask
// (the BN's own [y_b > 0] bytes) is set.  mcn_bn_bwd_from_partials then runs the BN backward without a reduction pass.  Eligible when both
// mcn_conv2d_dgrad_addmasked_ok() and mcn_conv2d_dgrad_bnred_rows() say so (rows = mcn_conv2d_dgrad_bnred_rows()).
extern "C" int mcn_conv2d_dgrad_addmasked_bnred(const void* dy, const float* w, const void* w_packed, void* dx, const void* add_src, const uint8_t* add_mask,
                                                const void* bn_x, const uint8_t* relu_mask, float* red_partials, const mcn_conv_geom* gg, mcn_dtype dtype,
                                                mcn_layout layout, void* ws, size_t ws_bytes, void* stream) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (layout != MCN_NHWC) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad_addmasked_bnred: only NHWC activations");
    if (!dy || !w || !dx || !add_src || !add_mask || !bn_x || !relu_mask || !red_partials) MCN_FAIL(MCN_E_BADARG, "conv2d_dgrad_addmasked_bnred: null pointer");
    if (!mcn_conv2d_dgrad_addmasked_ok(gg, dtype) || mcn_conv2d_dgrad_bnred_rows(gg, dtype) <= 0)
        MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_dgrad_addmasked_bnred: geometry not eligible (mcn_conv2d_dgrad_addmasked_ok / mcn_conv2d_dgrad_bnred_rows)");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return conv_dgrad_t<float>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, add_src, add_mask, bn_x, relu_mask, red_partials);
    if (dtype == MCN_F16) return conv_dgrad_t<f16_t>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, add_src, add_mask, bn_x, relu_mask, red_partials);
    return conv_dgrad_t<bf16_t>(dy, w, w_packed, dx, g, 0, dtype, ws, ws_bytes, st, add_src, add_mask, bn_x, relu_mask, red_partials);
}

// ---- wgrad -----------------------------------------------------------------------------------------------
template <typename T>
static int colsum_t(const void* x, float* out, long M, int C, float scale, void* ws, hipStream_t st) {
    const int parts = (int)colsum_parts(M);
    const int rpb = (int)((M + parts - 1) / parts);
    float* part = (float*)ws;
    int TX = 8;
    while (TX < C && TX < 256) TX *= 2;
    const dim3 grid((C + TX - 1) / TX, parts);
    hipLaunchKernelGGL((colsum_partial_kernel<T>), grid, dim3(256), 0, st, (const T*)x, part, M, C, rpb, TX);
    MCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 31) / 32), dim3(256), 0, st, (const float*)part, out, parts, C, scale);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// bias gradient on its own (tf.nn.bias_add's BiasAddGrad, convnet.py:1694, behind a convolution that has no dbias output of its own: the
// depthwise convolution): dbias[c] = grad_scale * sum_m dy[m][c], the two-stage column sum of the conv wgrad (fixed order: bit-reproducible)
extern "C" size_t mcn_bias_grad_workspace_bytes(int64_t M, int32_t C) {
    if (M <= 0 || C <= 0) return 0;
    return align_up(colsum_parts((long)M) * (size_t)C * 4, 256);
}
extern "C" int mcn_bias_grad(const void* dy, float* dbias, int64_t M, int32_t C, float grad_scale, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !dbias || M <= 0 || C <= 0) MCN_FAIL(MCN_E_BADARG, "bias_grad: bad argument (M=%ld C=%d)", (long)M, C);
    if (!ws || ws_bytes < mcn_bias_grad_workspace_bytes(M, C)) MCN_FAIL(MCN_E_WORKSPACE, "bias_grad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return colsum_t<float>(dy, dbias, (long)M, C, grad_scale, ws, st);
    if (dtype == MCN_BF16) return colsum_t<bf16_t>(dy, dbias, (long)M, C, grad_scale, ws, st);
    if (dtype == MCN_F16) return colsum_t<f16_t>(dy, dbias, (long)M, C, grad_scale, ws, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "bias_grad: dtype %d unsupported", (int)dtype);
}

template <typename T>
static int conv_wgrad_t(const void* x, const void* dy, float* dw, float* dbias, const Geo& g, float scale, mcn_dtype dt, void* ws,
                        size_t ws_bytes, hipStream_t st) {
    const long M = (long)g.N * g.OH * g.OW;
    const size_t need = wgrad_ws_bytes(g, dt);
    if (need && (!ws || ws_bytes < need)) MCN_FAIL(MCN_E_WORKSPACE, "conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    char* wsp = (char*)ws;
    if (!mfma_path_ok(g, dt) && skinny_ok(g, dt, MCN_SKINNY_MAX_CO_WGRAD)) {
        const int CO = skinny_co(g), ce = ce_of(dt);
        const long slab = skinny_wgrad_slab(M, g.Cin / ce), total = (long)g.Cin * g.Cout;
        const int slabs = (int)((M + slab - 1) / slab);
        float* part = (float*)wsp;
        wsp += align_up((size_t)(slabs + 1) * total * 4, 256);
        const dim3 grid((g.Cin / ce + 7) / 8, slabs);
#define MCN_SKINNY_WGRAD(COV) hipLaunchKernelGGL((skinny_conv_wgrad<T, COV>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, part, M, g.Cin, g.Cout, slab, 0)
        if (CO == 8) MCN_SKINNY_WGRAD(8); else if (CO == 16) MCN_SKINNY_WGRAD(16); else MCN_SKINNY_WGRAD(24);
#undef MCN_SKINNY_WGRAD
        MCN_CHECK_LAUNCH();
        hipLaunchKernelGGL(naive_wgrad_reduce, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const float*)part, dw, total, slabs, scale);
        MCN_CHECK_LAUNCH();
    } else if (!mfma_path_ok(g, dt) && skinny_in_ok(g, dt, MCN_SKINNY_MAX_CO_WGRAD)) {
        // few input channels: the same kernel with the operands swapped (dy is the chunked one), partials stored transposed
        const int CO = skinny_ci(g), ce = ce_of(dt);
        const long slab = skinny_wgrad_slab(M, g.Cout / ce), total = (long)g.Cin * g.Cout;
        const int slabs = (int)((M + slab - 1) / slab);
        float* part = (float*)wsp;
        wsp += align_up((size_t)(slabs + 1) * total * 4, 256);
        const dim3 grid((g.Cout / ce + 7) / 8, slabs);
#define MCN_SKINNY_WGRAD_IN(COV) hipLaunchKernelGGL((skinny_conv_wgrad<T, COV>), grid, dim3(256), 0, st, (const T*)dy, (const T*)x, part, M, g.Cout, g.Cin, slab, 1)
        if (CO == 8) MCN_SKINNY_WGRAD_IN(8); else if (CO == 16) MCN_SKINNY_WGRAD_IN(16); else MCN_SKINNY_WGRAD_IN(24);
#undef MCN_SKINNY_WGRAD_IN
        MCN_CHECK_LAUNCH();
        hipLaunchKernelGGL(naive_wgrad_reduce, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const float*)part, dw, total, slabs, scale);
        MCN_CHECK_LAUNCH();
    } else if (!mfma_path_ok(g, dt)) {
        NaiveConvParams p = naive_params(g);
        p.x = x; p.dy = dy; p.dw = dw; p.scale = scale;
        const long total = (long)g.KH * g.KW * g.Cin * g.Cout;
        const int sl = naive_wgrad_slices(g);
        float* part = (float*)wsp;
        if (sl > 1) wsp += align_up((size_t)sl * total * 4, 256);
        hipLaunchKernelGGL((naive_conv_wgrad<T>), dim3(nblocks(total * 8), sl), dim3(256), 0, st, p, part);
        MCN_CHECK_LAUNCH();
        if (sl > 1) {
            hipLaunchKernelGGL(naive_wgrad_reduce, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const float*)part, dw, total, sl, scale);
            MCN_CHECK_LAUNCH();
        }
    } else if (wino_wgrad_ok(g, dt)) {
        // Winograd F(3x3, 2x2): the slab layout is the direct wgrad's ([split][tap * Cin + c][n], Cp == Cin), so is the reduce
        WinoWgradParams p;
        memset(&p, 0, sizeof(p));
        int tps = 0;
        const int splits = wino_wgrad_splits(g, &tps);
        p.x = (const float*)x; p.dy = (const float*)dy; p.slab = (float*)wsp;
        p.H = g.H; p.W = g.W; p.Cs = g.xcs; p.Cin = g.Cin; p.ldy = g.Cout; p.Nn = g.Cout;
        p.TH = (g.H + 1) / 2; p.TW = (g.W + 1) / 2; p.ntiles = g.N * p.TH * p.TW; p.tiles_per_split = tps;
        p.nbc = (g.Cin + 63) / 64; p.nbn = (g.Cout + 63) / 64;
        p.x_bytes = (unsigned)((size_t)g.N * g.H * g.W * g.xcs * sizeof(float));
        p.dy_bytes = (unsigned)((size_t)M * g.Cout * sizeof(float));
        if (M > 0) {
            const int lds = WINO_WG_LDS_W8;
            static const bool lds_ok = allow_lds(conv_wino_wgrad_f3k2_w8, WINO_WG_LDS_W8);
            if (!lds_ok) MCN_FAIL(MCN_E_LAUNCH, "conv (Winograd wgrad): the runtime refused %d bytes of LDS", lds);
            hipLaunchKernelGGL(conv_wino_wgrad_f3k2_w8, dim3((unsigned)(p.nbc * p.nbn * splits)), dim3(512), lds, st, p);
            MCN_CHECK_LAUNCH();
        }
        const long total = 9L * g.Cin * g.Cout;
        const int nsp = M > 0 ? splits : 0;
        const long t4 = total / 4;
        const float* sl = (const float*)wsp;
        if ((((uintptr_t)dw) & 15) == 0) {
            if (t4 >= 256 * 256 || nsp < 8) hipLaunchKernelGGL((wgrad_reduce_linear_kernel<1>), dim3(nblocks(t4, 2048)), dim3(256), 0, st, sl, dw, nsp, t4, scale);
            else if (t4 >= 64 * 256 || nsp < 32) hipLaunchKernelGGL((wgrad_reduce_linear_kernel<4>), dim3(nblocks(t4 * 4, 2048)), dim3(256), 0, st, sl, dw, nsp, t4, scale);
            else hipLaunchKernelGGL((wgrad_reduce_linear_kernel<16>), dim3(nblocks(t4 * 16, 2048)), dim3(256), 0, st, sl, dw, nsp, t4, scale);
        } else
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, sl, dw, nsp, 9, g.Cin, g.Cin, g.Cout, scale);
        MCN_CHECK_LAUNCH();
        wsp += align_up((size_t)splits * total * 4, 256);
    } else {
        int nsteps, sps;
        const int splits = wgrad_splits(g, dt, &nsteps, &sps);
        const int ce = ce_of(dt), Cp = round_up(g.Cin, ce), ntaps = g.KH * g.KW;
        GemmTNParams p;
        memset(&p, 0, sizeof(p));
        for (int r = 0; r < g.KH; ++r)
            for (int s = 0; s < g.KW; ++s) {
                p.tdy[r * g.KW + s] = (signed char)(r * g.DH - g.pT);
                p.tdx[r * g.KW + s] = (signed char)(s * g.DW - g.pL);
            }
        p.x = x; p.dy = dy; p.slab = (float*)wsp;
        p.M = (int)M; p.OH = g.OH; p.OW = g.OW; p.IH = g.H; p.IW = g.W; p.Cs = g.xcs; p.Cp = Cp; p.ntaps = ntaps; p.rows = ntaps * Cp;
        p.sy = g.SH; p.sx = g.SW; p.Nn = g.Cout; p.ldy = g.Cout; p.nsteps = nsteps; p.steps_per_split = sps;
        p.x_bytes = (unsigned)((size_t)g.N * g.H * g.W * g.xcs * sizeof(T));
        p.dy_bytes = (unsigned)((size_t)M * g.Cout * sizeof(T));
        const bool linear = conv_is_linear(g);
        static const int tn_dbg = [] { const char* e = getenv("MCN_TN_DBG"); return e ? atoi(e) : 0; }();
        p.dbg = tn_dbg;
        if (M > 0) {
            int rc = launch_tn<T>(p, linear, splits, g.tile, st);
            if (rc) return rc;
        }
        const long total = (long)ntaps * g.Cin * g.Cout;
        const int nsp = M > 0 ? splits : 0;
        if (Cp == g.Cin && total % 4 == 0 && (((uintptr_t)dw) & 15) == 0) {
            // lanes per output vector: enough workgroups to cover the chip even for the smallest filters
            const long t4 = total / 4;
            const float* sl = (const float*)wsp;
            if (t4 >= 256 * 256 || nsp < 8) hipLaunchKernelGGL((wgrad_reduce_linear_kernel<1>), dim3(nblocks(t4, 2048)), dim3(256), 0, st, sl, dw, nsp, t4, scale);
            else if (t4 >= 64 * 256 || nsp < 32) hipLaunchKernelGGL((wgrad_reduce_linear_kernel<4>), dim3(nblocks(t4 * 4, 2048)), dim3(256), 0, st, sl, dw, nsp, t4, scale);
            else hipLaunchKernelGGL((wgrad_reduce_linear_kernel<16>), dim3(nblocks(t4 * 16, 2048)), dim3(256), 0, st, sl, dw, nsp, t4, scale);
        } else
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, (const float*)wsp, dw, nsp, ntaps, Cp, g.Cin, g.Cout, scale);
        MCN_CHECK_LAUNCH();
        wsp += align_up((size_t)splits * p.rows * g.Cout * 4, 256);
    }
    if (dbias) return colsum_t<T>(dy, dbias, M, g.Cout, scale, wsp, st);
    return MCN_OK;
}

extern "C" int mcn_conv2d_wgrad(const void* x, const void* dy, float* dw, float* dbias, const mcn_conv_geom* gg, float grad_scale,
                                mcn_dtype dtype, mcn_layout layout, void* ws, size_t ws_bytes, void* stream) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (layout != MCN_NHWC) MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_wgrad: only NHWC activations");
    if (!x || !dy || !dw) MCN_FAIL(MCN_E_BADARG, "conv2d_wgrad: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MCN_F32) return conv_wgrad_t<float>(x, dy, dw, dbias, g, grad_scale, dtype, ws, ws_bytes, st);
    if (dtype == MCN_BF16) return conv_wgrad_t<bf16_t>(x, dy, dw, dbias, g, grad_scale, dtype, ws, ws_bytes, st);
    else if (dtype == MCN_F16) return conv_wgrad_t<f16_t>(x, dy, dw, dbias, g, grad_scale, dtype, ws, ws_bytes, st);
    MCN_FAIL(MCN_E_UNSUPPORTED, "conv2d_wgrad: dtype %d unsupported", (int)dtype);
}

// ---- packed operands kept by the caller: sizes, one-launch batch packing, kernel introspection --------------------------------
extern "C" size_t mcn_conv2d_packed_bytes(mcn_conv_op op, const mcn_conv_geom* gg, mcn_dtype dtype) {
    Geo g;
    if (geo_from(gg, &g) != MCN_OK || !mcn_dtype_ok(dtype)) return 0;
    if (op == MCN_CONV_FWD) return mfma_path_ok(g, dtype) ? fwd_pack_bytes(g, dtype) : 0;
    if (op == MCN_CONV_DGRAD) return mfma_dgrad_ok(g, dtype) ? dgrad_pack_bytes(g, dtype) : 0;
    return 0;
}

// jobs per descriptor of the batch kernel (64 workgroups each): 32x32 transpose tiles (forward) / filter rows (dgrad)
#define MCN_PACK_SLICE_TILES 128
#define MCN_PACK_SLICE_ROWS 512
// appends pk cut into slices of at most `slice` jobs; returns the number of descriptors written
static int pack_njobs(const PackParams& pk) {
    if (pk.mode >= 2) {                                  // Winograd filter transform: 256 elements of U per job
        const int Kin = pk.mode == 3 ? pk.Cout : pk.Cin, Kout = pk.mode == 3 ? pk.Cin : pk.Cout;
        return ((Kout + 63) / 64) * (Kin / 32) * 8;
    }
    return pk.mode == 0 ? pk.ntaps * ((pk.Cp + 31) / 32) * ((pk.rows + 31) / 32) : pk.rows * pk.ntaps;
}
static int pack_emit(const PackParams& pk, PackParams* out) {
    const int njobs = pack_njobs(pk);
    const int slice = pk.mode == 0 ? MCN_PACK_SLICE_TILES : MCN_PACK_SLICE_ROWS;
    int n = 0;
    for (int j = 0; j < njobs; j += slice) {
        out[n] = pk;
        out[n].job0 = j;
        out[n].job1 = j + slice < njobs ? j + slice : njobs;
        ++n;
    }
    return n;
}
// upper bound of the descriptors one job needs (any dtype)
static size_t pack_desc_bound(const mcn_conv_geom& g, int op) {
    if (op == MCN_CONV_DGRAD)
        return (size_t)g.SH * g.SW + (size_t)g.Cin * g.KH * g.KW / MCN_PACK_SLICE_ROWS + 1 + (size_t)((g.Cin + 63) / 64) * (g.Cout / 32 + 1) * 8 / MCN_PACK_SLICE_ROWS;
    return (size_t)g.KH * g.KW * ((g.Cin + 31) / 32) * ((g.Cout + 31) / 32) / MCN_PACK_SLICE_TILES + 1;
}
// descriptors for one job; returns how many (0 = the op does not use a packed operand)
static int pack_descs(const Geo& g, mcn_dtype dt, mcn_conv_op op, const float* w, void* packed, PackParams* out) {
    const int ce = ce_of(dt);
    const size_t es = mcn_dtype_size(dt);
    if ((op == MCN_CONV_FWD && wino_fwd_ok(g, dt)) || (op == MCN_CONV_DGRAD && wino_dgrad_ok(g, dt))) {
        PackParams pk;
        memset(&pk, 0, sizeof(pk));
        pk.w = w; pk.out = packed; pk.Cin = g.Cin; pk.Cout = g.Cout; pk.mode = op == MCN_CONV_FWD ? 2 : 3;
        return pack_emit(pk, out);
    }
    if (op == MCN_CONV_FWD) {
        if (!mfma_path_ok(g, dt)) return 0;
        PackParams& pk = out[0];
        memset(&pk, 0, sizeof(pk));
        pk.w = w; pk.out = packed; pk.KW = g.KW; pk.Cin = g.Cin; pk.Cout = g.Cout; pk.rows = g.Cout; pk.Cp = round_up(g.Cin, ce);
        pk.ntaps = g.KH * g.KW; pk.mode = 0;
        for (int r = 0; r < g.KH; ++r)
            for (int s = 0; s < g.KW; ++s) { pk.tr[r * g.KW + s] = (signed char)r; pk.ts[r * g.KW + s] = (signed char)s; }
        const PackParams whole = pk;
        return pack_emit(whole, out);
    }
    if (op != MCN_CONV_DGRAD || !mfma_dgrad_ok(g, dt)) return 0;
    const int Cp = round_up(g.Cout, ce);
    char* dst = (char*)packed;
    int n = 0;
    for (int py = 0; py < g.SH; ++py)
        for (int px = 0; px < g.SW; ++px) {
            if (py >= g.H || px >= g.W) continue;
            PackParams pk;
            memset(&pk, 0, sizeof(pk));
            for (int r = 0; r < g.KH; ++r) {
                if (pos_mod(py + g.pT - r * g.DH, g.SH)) continue;
                for (int s = 0; s < g.KW; ++s) {
                    if (pos_mod(px + g.pL - s * g.DW, g.SW)) continue;
                    pk.tr[pk.ntaps] = (signed char)r; pk.ts[pk.ntaps] = (signed char)s;
                    pk.ntaps++;
                }
            }
            if (pk.ntaps == 0) continue;
            pk.w = w; pk.out = dst; pk.KW = g.KW; pk.Cin = g.Cin; pk.Cout = g.Cout; pk.rows = g.Cin; pk.Cp = Cp; pk.mode = 1;
            n += pack_emit(pk, out + n);
            dst += align_up((size_t)g.Cin * pk.ntaps * Cp * es, 256);
        }
    return n;
}

extern "C" size_t mcn_conv2d_pack_table_bytes(const mcn_pack_job* jobs, int32_t njobs) {
    size_t n = 0;
    for (int i = 0; i < njobs; ++i) n += pack_desc_bound(jobs[i].geom, jobs[i].op);
    return n * sizeof(PackParams);
}
extern "C" int mcn_conv2d_pack_table_build(const mcn_pack_job* jobs, int32_t njobs, mcn_dtype dtype, void* host_table, size_t bytes,
                                           int32_t* ndesc) {
    if (!jobs || !host_table || !ndesc || njobs < 0) MCN_FAIL(MCN_E_BADARG, "pack_table_build: bad argument");
    if (bytes < mcn_conv2d_pack_table_bytes(jobs, njobs)) MCN_FAIL(MCN_E_WORKSPACE, "pack_table_build: table buffer too small");
    if (!mcn_dtype_ok(dtype)) MCN_FAIL(MCN_E_UNSUPPORTED, "pack_table_build: dtype %d unsupported", (int)dtype);
    PackParams* out = (PackParams*)host_table;
    int n = 0;
    for (int i = 0; i < njobs; ++i) {
        Geo g;
        int rc = geo_from(&jobs[i].geom, &g);
        if (rc) return rc;
        if (!jobs[i].w_hwio || !jobs[i].packed) MCN_FAIL(MCN_E_BADARG, "pack_table_build: job %d has a null pointer", i);
        n += pack_descs(g, dtype, (mcn_conv_op)jobs[i].op, jobs[i].w_hwio, jobs[i].packed, out + n);
    }
    *ndesc = n;
    return MCN_OK;
}
// ---- pixel-pair form of a stride-2 convolution on few input channels (the stem) ----------------------------------------
// A 2-byte type keeps 8 elements per 16-byte chunk: a 3-channel image padded to 8 channels wastes 5/8 of every operand byte and
// MFMA (K = taps x 8).  With the image stored 4 channels per pixel, two horizontally adjacent pixels are one chunk, and a
// convolution with horizontal stride 2 reads whole pairs: output ox covers input pixels 2*ox - padL .. + KW - 1, i.e. pairs
// ox - padL' .. + KW' - 1 with padL' = ceil(padL / 2), KW' = floor((KW - 1 - padL) / 2) + padL' + 1.  That IS a convolution of
// the [N, H, W/2, 8] view with a [KH, KW', 8, Cout] filter, horizontal stride 1: filter element (kp, parity, c) is the original
// (kx = 2*kp + parity - (padL & 1), c), zero where kx falls outside 0..KW-1 or c >= Cin.  ResNet stem 7x7 / 2: K 392 -> 224;
// EfficientNet stem 3x3 / 2: 72 -> 48.  The weight gradient of the paired filter is gathered back to the HWIO layout.
static bool pair_plan(const mcn_conv_geom* g, mcn_dtype dtype, mcn_conv_geom* out) {
    if (!g || mcn_dtype_size(dtype) != 2 || g->Cin < 1 || g->Cin > 4 || g->SW != 2 || g->DW != 1 || g->DH != 1 || g->W < 2 || (g->W & 1)) return false;
    if (g->x_cs != 0 && g->x_cs != 4) return false;                       // the caller stores the image 4 channels per pixel
    if (g->KW < 2 || g->padL < 0 || g->padL > g->KW - 1) return false;
    const int OW = (g->W + g->padL + g->padR - g->KW) / 2 + 1;
    if (OW < 1) return false;
    const int pl = (g->padL + 1) / 2;
    const int kw = (g->KW - 1 - g->padL) / 2 + pl + 1;
    const int Wp = g->W / 2;
    const int pr = OW - 1 + kw - Wp - pl;                                 // so that the paired geometry yields the same OW
    if (pr < 0) return false;
    if (out) {
        *out = *g;
        out->W = Wp; out->Cin = 8; out->KW = kw; out->SW = 1; out->padL = pl; out->padR = pr; out->x_cs = 8;
    }
    return true;
}
extern "C" int mcn_conv2d_pair_geom(const mcn_conv_geom* g, mcn_dtype dtype, mcn_conv_geom* paired) { return pair_plan(g, dtype, paired) ? 1 : 0; }
__global__ __launch_bounds__(256) void pair_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int KH, int KW, int Cin, int Cout, int KWp, int shift) {
    const long total = (long)KH * KWp * 8 * Cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % Cout);
        long r = i / Cout;
        const int e = (int)(r % 8);
        r /= 8;
        const int kp = (int)(r % KWp), ky = (int)(r / KWp);
        const int c = e & 3, kx = 2 * kp + (e >> 2) - shift;
        wp[i] = (c < Cin && kx >= 0 && kx < KW) ? w[(((long)ky * KW + kx) * Cin + c) * Cout + n] : 0.f;
    }
}
__global__ __launch_bounds__(256) void pair_wgrad_fold_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int KH, int KW, int Cin, int Cout, int KWp, int shift) {
    const long total = (long)KH * KW * Cin * Cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % Cout);
        long r = i / Cout;
        const int c = (int)(r % Cin);
        r /= Cin;
        const int kx = (int)(r % KW), ky = (int)(r / KW);
        const int ks = kx + shift;
        dw[i] = dwp[(((long)ky * KWp + (ks >> 1)) * 8 + (ks & 1) * 4 + c) * Cout + n];
    }
}
extern "C" int mcn_conv2d_pair_weights(const float* w_hwio, float* w_paired, const mcn_conv_geom* g, mcn_dtype dtype, void* stream) {
    mcn_conv_geom pg;
    if (!w_hwio || !w_paired || !pair_plan(g, dtype, &pg)) MCN_FAIL(MCN_E_BADARG, "conv2d_pair_weights: geometry has no pixel-pair form");
    const long total = (long)g->KH * pg.KW * 8 * g->Cout;
    hipLaunchKernelGGL(pair_weights_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)stream, w_hwio, w_paired,
                       g->KH, g->KW, g->Cin, g->Cout, pg.KW, g->padL & 1);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}
extern "C" int mcn_conv2d_pair_wgrad_fold(const float* dw_paired, float* dw_hwio, const mcn_conv_geom* g, mcn_dtype dtype, void* stream) {
    mcn_conv_geom pg;
    if (!dw_paired || !dw_hwio || !pair_plan(g, dtype, &pg)) MCN_FAIL(MCN_E_BADARG, "conv2d_pair_wgrad_fold: geometry has no pixel-pair form");
    const long total = (long)g->KH * g->KW * g->Cin * g->Cout;
    hipLaunchKernelGGL(pair_wgrad_fold_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)stream, dw_paired, dw_hwio,
                       g->KH, g->KW, g->Cin, g->Cout, pg.KW, g->padL & 1);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

extern "C" int mcn_conv2d_pack_run(const void* dev_table, int32_t ndesc, mcn_dtype dtype, void* stream) {
    if (ndesc <= 0) return MCN_OK;
    if (!dev_table) MCN_FAIL(MCN_E_BADARG, "pack_run: null table");
    const dim3 grid(64, ndesc), block(256);
    if (dtype == MCN_F32) hipLaunchKernelGGL((pack_weights_batch_kernel<float>), grid, block, 0, (hipStream_t)stream, (const PackParams*)dev_table);
    else if (dtype == MCN_BF16) hipLaunchKernelGGL((pack_weights_batch_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const PackParams*)dev_table);
    else if (dtype == MCN_F16) hipLaunchKernelGGL((pack_weights_batch_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, (const PackParams*)dev_table);
    else MCN_FAIL(MCN_E_UNSUPPORTED, "pack_run: dtype %d unsupported", (int)dtype);
    MCN_CHECK_LAUNCH();
    return MCN_OK;
}

// name of the GEMM kernel a conv call launches and how many times (profiling aid: matches rocprofv3's kernel names)
extern "C" int mcn_conv2d_kernel_name(mcn_conv_op op, const mcn_conv_geom* gg, mcn_dtype dtype, char* buf, size_t buflen) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (!buf || buflen < 64) MCN_FAIL(MCN_E_BADARG, "kernel_name: buffer too small");
    const char* tn = dtype == MCN_F32 ? "float" : (dtype == MCN_F16 ? "_Float16" : "bf16");
    const int ce = ce_of(dtype);
    const NtTile* cand = kNtCand;
    if (op == MCN_CONV_FWD) {
        if (!mfma_path_ok(g, dtype) && skinny_ok(g, dtype, MCN_SKINNY_MAX_CO)) {
            snprintf(buf, buflen, "skinny_conv_fwd%s<%s, %d>", skinny_split((long)g.N * g.OH * g.OW, g.Cin / ce) ? "_split" : "", tn, skinny_co(g));
            return 1;
        }
        if (!mfma_path_ok(g, dtype)) { snprintf(buf, buflen, "naive_conv_fwd<%s>", tn); return 1; }
        if (wino_fwd_ok(g, dtype)) { snprintf(buf, buflen, "conv_wino_f2k3_w8<0, 0>"); return 1; }      /* (the trailing parameter is the epilogue: 1 = BN statistics) */
        const long M = (long)g.N * g.OH * g.OW;
        const int t = dtype == MCN_F32 ? pick_nt_tile<float>((int)M, g.Cout, nt_hint(g, dtype, false)) : pick_nt_tile<bf16_t>((int)M, g.Cout, nt_hint(g, dtype, false));
        const int cpt = round_up(g.Cin, ce) / ce;
        const int mode = conv_is_linear(g) ? NT_LINEAR : (cpt % 8 == 0 ? NT_UNIFORM : NT_GENERIC);
        if (cand[t].wpp) snprintf(buf, buflen, "conv_gemm_nt_wpp<%s, %d, 0>", tn, cand[t].bn);
        else if (mode == NT_UNIFORM && nt_window_geom(g, mcn_dtype_size(dtype), cpt, cand[t])) snprintf(buf, buflen, "conv_gemm_nt_win<%s, %d, %d, %d, 0>", tn, cand[t].bm, cand[t].bn, cand[t].nw);
        else if (mode == NT_LINEAR && nt_pers_geom(t, M, g.Cout, cpt, mcn_dtype_size(dtype), nt_hint(g, dtype, false), NT_EPI_STATS)) {
            /* (answers for mcn_conv2d_fwd_bnstats; a biased launch, and by default a forward without statistics: conv_gemm_nt.  Epilogue 3 = counted statistics rows) */
            const bool counted = dtype == MCN_F32   ? nt_stats_counted<float>(mode, M, g.Cout, cpt, t, nt_hint(g, dtype, false), nullptr)
                                 : dtype == MCN_F16 ? nt_stats_counted<f16_t>(mode, M, g.Cout, cpt, t, nt_hint(g, dtype, false), nullptr)
                                                    : nt_stats_counted<bf16_t>(mode, M, g.Cout, cpt, t, nt_hint(g, dtype, false), nullptr);
            snprintf(buf, buflen, "conv_gemm_nt_pers<%s, %d, %d, %d, %d>", tn, cand[t].bm, cand[t].bn, cand[t].nw, counted ? 3 : 0);
        }
        else snprintf(buf, buflen, "conv_gemm_nt<%s, %d, %d, %d, %d, 0>", tn, cand[t].bm, cand[t].bn, mode, cand[t].nw);
        return 1;
    }
    if (op == MCN_CONV_DGRAD) {
        if (!mfma_dgrad_ok(g, dtype) && skinny_ok(g, dtype, MCN_SKINNY_MAX_CO)) { snprintf(buf, buflen, "skinny_conv_dgrad<%s, %d>", tn, skinny_co(g)); return 1; }
        if (!mfma_dgrad_ok(g, dtype) && skinny_in_ok(g, dtype, MCN_SKINNY_MAX_CO)) {
            snprintf(buf, buflen, "skinny_conv_fwd%s<%s, %d>", skinny_split((long)g.N * g.H * g.W, g.Cout / ce) ? "_split" : "", tn, skinny_ci(g));
            return 1;
        }
        if (!mfma_dgrad_ok(g, dtype)) { snprintf(buf, buflen, "naive_conv_dgrad<%s>", tn); return 1; }
        if (wino_dgrad_ok(g, dtype)) { snprintf(buf, buflen, "conv_wino_f2k3_w8<0, 0>"); return 1; }    /* (... 4 = BN-backward sums, 2 = accumulate) */
        int ncls = 0, nt0 = 0;
        for (int py = 0; py < g.SH && py < g.H; ++py)
            for (int px = 0; px < g.SW && px < g.W; ++px) {
                int nt = 0;
                for (int r = 0; r < g.KH; ++r)
                    for (int s = 0; s < g.KW; ++s)
                        if (!pos_mod(py + g.pT - r * g.DH, g.SH) && !pos_mod(px + g.pL - s * g.DW, g.SW)) nt++;
                if (nt) { if (!ncls) nt0 = nt; ncls++; }
            }
        const int OHs = (g.H + g.SH - 1) / g.SH, OWs = (g.W + g.SW - 1) / g.SW;
        const long M = (long)g.N * OHs * OWs;
        const int t = dtype == MCN_F32 ? pick_nt_tile<float>((int)M, g.Cin, nt_hint(g, dtype, true)) : pick_nt_tile<bf16_t>((int)M, g.Cin, nt_hint(g, dtype, true));
        const int cpt = round_up(g.Cout, ce) / ce;
        const bool lin = g.KH * g.KW == 1 && nt0 == 1;
        const int mode = lin ? NT_LINEAR : (cpt % 8 == 0 ? NT_UNIFORM : NT_GENERIC);
        if (cand[t].wpp) snprintf(buf, buflen, "conv_gemm_nt_wpp<%s, %d, 0>", tn, cand[t].bn);
        else if (mode == NT_UNIFORM && nt_window_geom(g, mcn_dtype_size(dtype), cpt, cand[t])) snprintf(buf, buflen, "conv_gemm_nt_win<%s, %d, %d, %d, 0>", tn, cand[t].bm, cand[t].bn, cand[t].nw);
        else if (mode == NT_LINEAR && nt_pers_geom(t, M, g.Cin, nt0 * cpt, mcn_dtype_size(dtype), nt_hint(g, dtype, true))) snprintf(buf, buflen, "conv_gemm_nt_pers<%s, %d, %d, %d, 0>", tn, cand[t].bm, cand[t].bn, cand[t].nw);
        else snprintf(buf, buflen, "conv_gemm_nt<%s, %d, %d, %d, %d, 0>", tn, cand[t].bm, cand[t].bn, mode, cand[t].nw);
        return ncls;
    }
    if (!mfma_path_ok(g, dtype) && skinny_ok(g, dtype, MCN_SKINNY_MAX_CO_WGRAD)) { snprintf(buf, buflen, "skinny_conv_wgrad<%s, %d>", tn, skinny_co(g)); return 1; }
    if (!mfma_path_ok(g, dtype) && skinny_in_ok(g, dtype, MCN_SKINNY_MAX_CO_WGRAD)) { snprintf(buf, buflen, "skinny_conv_wgrad<%s, %d>", tn, skinny_ci(g)); return 1; }
    if (!mfma_path_ok(g, dtype)) { snprintf(buf, buflen, "naive_conv_wgrad<%s>", tn); return 1; }
    if (wino_wgrad_ok(g, dtype)) { snprintf(buf, buflen, "conv_wino_wgrad_f3k2_w8"); return 1; }
    int br, bn;
    tn_tile(g.KH * g.KW * round_up(g.Cin, ce), g.Cout, dtype, conv_is_linear(g), g.tile, &br, &bn);
    if (tn_ring(mcn_dtype_size(dtype), br, bn, conv_is_linear(g))) snprintf(buf, buflen, "conv_gemm_tn3<%s, %d, %d, %s, %d>", tn, br, bn, conv_is_linear(g) ? "true" : "false", br == 128 ? 8 : 4);
    else snprintf(buf, buflen, "conv_gemm_tn<%s, %d, %d, %s, %d>", tn, br, bn, conv_is_linear(g) ? "true" : "false", br == 256 || (br == 128 && bn == 128 && tn_nw8(mcn_dtype_size(dtype))) ? 8 : 4);
    return 1;
}

// one line per GEMM launch of a conv call: "<kernel symbol as rocprofv3 prints it>:<filter taps of that launch>\n".
// A strided dgrad launches once per stride-parity class of dx, and the classes differ in taps, pixel count (tile choice)
// and addressing mode, so they are not always the same symbol.  Returns the number of launches (lines).
extern "C" int mcn_conv2d_launch_list(mcn_conv_op op, const mcn_conv_geom* gg, mcn_dtype dtype, char* buf, size_t buflen) {
    Geo g;
    int rc = geo_from(gg, &g);
    if (rc) return rc;
    if (!buf || buflen < 96) MCN_FAIL(MCN_E_BADARG, "launch_list: buffer too small");
    if (op != MCN_CONV_DGRAD || !mfma_dgrad_ok(g, dtype) || wino_dgrad_ok(g, dtype)) {
        char one[96];
        const int n = mcn_conv2d_kernel_name(op, gg, dtype, one, sizeof(one));
        if (n < 0) return n;
        const bool wf = op == MCN_CONV_FWD && mfma_path_ok(g, dtype) && wino_fwd_ok(g, dtype);
        const bool wd = op == MCN_CONV_DGRAD && mfma_dgrad_ok(g, dtype) && wino_dgrad_ok(g, dtype);
        if ((wf || wd) && !(g.tile & MCN_TILE_NOSPLIT)
            && (wf ? wino_geom_slices(g.N, g.H, g.W, g.Cin, g.Cout, false) : wino_geom_slices(g.N, g.H, g.W, g.Cout, g.Cin, false)) > 1) {
            // K-sliced tail (launch_wino): the body launch, the slices of the short last round (plain store: template argument 128 = WINO_SLICE) and
            // the reduce launch that runs the epilogue (64 = WINO_REDUCE; its trailing 0 is replaced by the caller's epilogue like the body's)
            snprintf(buf, buflen, "%s:%d\nconv_wino_f2k3_w8<%d, 0>:%d\nconv_wino_f2k3_w8<%d, 0>:%d\n", one, g.KH * g.KW, (int)WINO_SLICE, g.KH * g.KW, (int)WINO_REDUCE, 0);
            return 3;
        }
        snprintf(buf, buflen, "%s:%d\n", one, g.KH * g.KW);
        return 1;
    }
    const char* tn = dtype == MCN_F32 ? "float" : (dtype == MCN_F16 ? "_Float16" : "bf16");
    const int ce = ce_of(dtype), cpt = round_up(g.Cout, ce) / ce;
    size_t used = 0;
    int nl = 0;
    buf[0] = 0;
    for (int py = 0; py < g.SH && py < g.H; ++py)
        for (int px = 0; px < g.SW && px < g.W; ++px) {
            int nt = 0;
            bool zero_off = true;
            for (int r = 0; r < g.KH; ++r)
                for (int s = 0; s < g.KW; ++s) {
                    const int ty = py + g.pT - r * g.DH, tx = px + g.pL - s * g.DW;
                    if (pos_mod(ty, g.SH) || pos_mod(tx, g.SW)) continue;
                    nt++;
                    if (ty / g.SH || tx / g.SW) zero_off = false;
                }
            if (!nt) continue;
            const int OHs = (g.H - py + g.SH - 1) / g.SH, OWs = (g.W - px + g.SW - 1) / g.SW;
            const long M = (long)g.N * OHs * OWs;
            const int t = dtype == MCN_F32 ? pick_nt_tile<float>((int)M, g.Cin, nt_hint(g, dtype, true)) : pick_nt_tile<bf16_t>((int)M, g.Cin, nt_hint(g, dtype, true));
            const bool lin = nt == 1 && zero_off && OHs == g.OH && OWs == g.OW;
            const int mode = lin ? NT_LINEAR : (cpt % 8 == 0 ? NT_UNIFORM : NT_GENERIC);
            const int w = kNtCand[t].wpp ? snprintf(buf + used, buflen - used, "conv_gemm_nt_wpp<%s, %d, 0>:%d\n", tn, kNtCand[t].bn, nt)
                          : (mode == NT_UNIFORM && nt_window_geom(g, mcn_dtype_size(dtype), cpt, kNtCand[t]))
                              ? snprintf(buf + used, buflen - used, "conv_gemm_nt_win<%s, %d, %d, %d, 0>:%d\n", tn, kNtCand[t].bm, kNtCand[t].bn, kNtCand[t].nw, nt)
                          : (mode == NT_LINEAR && nt_pers_geom(t, M, g.Cin, nt * cpt, mcn_dtype_size(dtype), nt_hint(g, dtype, true)))
                              ? snprintf(buf + used, buflen - used, "conv_gemm_nt_pers<%s, %d, %d, %d, 0>:%d\n", tn, kNtCand[t].bm, kNtCand[t].bn, kNtCand[t].nw, nt)
                              : snprintf(buf + used, buflen - used, "conv_gemm_nt<%s, %d, %d, %d, %d, 0>:%d\n", tn, kNtCand[t].bm, kNtCand[t].bn, mode, kNtCand[t].nw, nt);
            if (w < 0 || (size_t)w >= buflen - used) MCN_FAIL(MCN_E_BADARG, "launch_list: buffer too small");
            used += (size_t)w;
            nl++;
        }
    return nl;
}

// K-slices per tail tile of the stream-K split the (first) GEMM launch of this conv uses when the workspace has room
// (1 = every tile runs its whole K loop; tests and the bench use it to know which layers are split)
extern "C" int32_t mcn_conv2d_kslices(mcn_conv_op op, const mcn_conv_geom* gg, mcn_dtype dtype) {
    Geo g;
    if (!gg || geo_from(gg, &g) || !mcn_dtype_ok(dtype) || (g.tile & MCN_TILE_NOSPLIT)) return 1;
    const int ce = ce_of(dtype);
    long M;
    int Nn, nchunks;
    if (op == MCN_CONV_FWD) {
        if (!mfma_path_ok(g, dtype)) return 1;
        if (wino_fwd_ok(g, dtype)) return wino_geom_slices(g.N, g.H, g.W, g.Cin, g.Cout, false);      // K-sliced tail of the Winograd launch
        M = (long)g.N * g.OH * g.OW; Nn = g.Cout; nchunks = g.KH * g.KW * (round_up(g.Cin, ce) / ce);
    } else if (op == MCN_CONV_DGRAD) {
        if (!mfma_dgrad_ok(g, dtype)) return 1;
        if (wino_dgrad_ok(g, dtype)) return wino_geom_slices(g.N, g.H, g.W, g.Cout, g.Cin, false);
        int nt0 = 0;                                      // taps of the first non-empty stride-parity class
        for (int py = 0; py < g.SH && py < g.H && !nt0; ++py)
            for (int px = 0; px < g.SW && px < g.W && !nt0; ++px)
                for (int r = 0; r < g.KH; ++r)
                    for (int s = 0; s < g.KW; ++s)
                        if (!pos_mod(py + g.pT - r * g.DH, g.SH) && !pos_mod(px + g.pL - s * g.DW, g.SW)) nt0++;
        M = (long)g.N * ((g.H + g.SH - 1) / g.SH) * ((g.W + g.SW - 1) / g.SW); Nn = g.Cin; nchunks = nt0 * (round_up(g.Cout, ce) / ce);
    } else {
        return 1;
    }
    if (M <= 0) return 1;
    const int t = dtype == MCN_F32 ? pick_nt_tile<float>((int)M, Nn, nt_hint(g, dtype, op == MCN_CONV_DGRAD)) : pick_nt_tile<bf16_t>((int)M, Nn, nt_hint(g, dtype, op == MCN_CONV_DGRAD));
    const long W = (long)((M + kNtCand[t].bm - 1) / kNtCand[t].bm) * ((Nn + kNtCand[t].bn - 1) / kNtCand[t].bn);
    return sk_plan(t, W, (nchunks + 7) >> 3, mcn_dtype_size(dtype)).slices;
}

// ---- fully connected = 1x1 convolution on a [B][1][1][In] tensor ------------------------------------------
static mcn_conv_geom fc_geom(int B, int In, int Out) {
    mcn_conv_geom g;
    memset(&g, 0, sizeof(g));
    g.N = B; g.H = 1; g.W = 1; g.Cin = In; g.Cout = Out; g.KH = g.KW = g.SH = g.SW = g.DH = g.DW = 1;
    return g;
}
extern "C" size_t mcn_fc_workspace_bytes(int32_t B, int32_t In, int32_t Out, mcn_dtype dtype) {
    const mcn_conv_geom g = fc_geom(B, In, Out);
    size_t a = mcn_conv2d_workspace_bytes(MCN_CONV_FWD, &g, dtype);
    size_t b = mcn_conv2d_workspace_bytes(MCN_CONV_DGRAD, &g, dtype);
    size_t c = mcn_conv2d_workspace_bytes(MCN_CONV_WGRAD, &g, dtype);
    size_t m = a > b ? a : b;
    return m > c ? m : c;
}
extern "C" int mcn_fc_fwd(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t In, int32_t Out, mcn_dtype dtype,
                          void* ws, size_t ws_bytes, void* stream) {
    const mcn_conv_geom g = fc_geom(B, In, Out);
    return mcn_conv2d_fwd(x, w, nullptr, bias, y, &g, dtype, MCN_NHWC, ws, ws_bytes, stream);
}
extern "C" int mcn_fc_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias, float grad_scale, int32_t B,
                          int32_t In, int32_t Out, mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream) {
    const mcn_conv_geom g = fc_geom(B, In, Out);
    int rc = MCN_OK;
    if (dx) rc = mcn_conv2d_dgrad(dy, w, nullptr, dx, &g, 0, dtype, MCN_NHWC, ws, ws_bytes, stream);
    if (rc) return rc;
    if (dw) rc = mcn_conv2d_wgrad(x, dy, dw, dbias, &g, grad_scale, dtype, MCN_NHWC, ws, ws_bytes, stream);
    return rc;
}
