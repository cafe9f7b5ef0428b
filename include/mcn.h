/*
 * mcn.h — C-ABI of libmcn_hip.so: the MI355X (gfx950 / CDNA4) implementation of MyConvNet's
 * conv / batch-norm / ReLU / pooling / loss / momentum training hot path.
 *
 * The reference (dooyounggo/MyConvNet) is pure Python on TensorFlow 1.x and has no FFI of its
 * own; its boundary for this path is the set of TensorFlow ops that convnet.py / optimizers.py
 * call.  Each entry point below replaces exactly one of those call sites (cited as
 * reference `file:line`) and is what a maintainer would bind from Python with ctypes
 * (INTEGRATION.md shows the stub).  The ABI carries plain device pointers, integer sizes and
 * a hipStream_t passed as void*; no torch / C++ types.
 *
 * Conventions
 *   - activations NHWC (the reference's default layout, convnet.py:1619-1623; TF-1.x CPU conv
 *     kernels accept only NHWC); MCN_NCHW is accepted by mcn_input_prep only.
 *   - conv filters are the reference's fp32 master variables in HWIO order [KH][KW][Cin][Cout]
 *     (convnet.py:1655); the library re-packs / casts them into caller-provided workspace on
 *     every call, mirroring weight_variable()'s per-use cast (convnet.py:1421-1422).
 *   - dtype = storage type of activations and activation gradients.  All accumulation,
 *     batch-norm statistics, parameter gradients, loss and optimizer state are fp32.
 *   - every function returns MCN_OK (0) or a negative status; the message is available from
 *     mcn_last_error() (thread-local).  Nothing throws across the ABI.
 *   - the caller owns every buffer including workspaces; the library keeps no device memory and
 *     no mutable global state, never synchronises the device, and orders work only by `stream`.
 */
#ifndef MCN_H_
#define MCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCN_VERSION 100 /* 0.1.0 */

/* storage type of activations / activation gradients.  MCN_F16 is the reference's own low precision (tf.float16,
 * convnet.py:63) and needs its loss scaling (optimizers.py:102-111: loss_scale / grad_scale arguments); MCN_BF16 is the
 * MI355X-native choice (fp32 exponent range, no scaling).  Same kernels, same MFMA rate. */
typedef enum { MCN_F32 = 0, MCN_BF16 = 1, MCN_F16 = 2 } mcn_dtype;
typedef enum { MCN_NHWC = 0, MCN_NCHW = 1 } mcn_layout;
typedef enum { MCN_OK = 0, MCN_E_BADARG = -1, MCN_E_UNSUPPORTED = -2, MCN_E_LAUNCH = -3, MCN_E_WORKSPACE = -4 } mcn_status;

/* activation fused into a producer kernel */
typedef enum {
    MCN_ACT_NONE = 0,
    MCN_ACT_RELU = 1,
    MCN_ACT_SWISH = 2,  /* x*sigmoid(x), convnet.py:2553-2556 */
    MCN_ACT_SIGMOID = 3, /* convnet.py:2550; element-wise entry points only */
    MCN_ACT_RELU6 = 4,   /* tf.nn.relu6, convnet.py:2539-2540; element-wise entry points only */
    MCN_ACT_LRELU = 5,   /* tf.nn.leaky_relu(alpha), convnet.py:2542-2545 (alpha = param, reference default 0.2); element-wise only */
    MCN_ACT_TANH = 6     /* tf.nn.tanh, convnet.py:2547; element-wise entry points only */
} mcn_act;

int mcn_version(void);
const char* mcn_last_error(void);
/* hex digest of the sources (csrc/ *.hip *.h + this header) the library was built from; the binding compares it with the
 * tree it is imported from and refuses a stale binary (myconvnet_amd/_ffi.py, build.py) */
const char* mcn_build_id(void);

/* ---- convolution ------------------------------------------------------------------------
 * Geometry shared by the three conv entry points.  x:[N][H][W][x_cs] (x_cs = channel stride in
 * elements, >= Cin; pass 0 for Cin), w:[KH][KW][Cin][Cout] fp32, y:[N][OH][OW][Cout] where
 * OH/OW follow from the explicit pads: OH = (H + padT + padB - (KH-1)*DH - 1)/SH + 1.
 * TF "SAME" pads are computed by the caller (asymmetric: before = total/2).           */
typedef struct {
    int32_t N, H, W, Cin, Cout;
    int32_t KH, KW, SH, SW, DH, DW;
    int32_t padT, padB, padL, padR;
    int32_t x_cs; /* channel stride of x (0 => Cin).  Channels >= Cin must be finite (they meet zero weights) */
    int32_t tile; /* low byte: 0 = library heuristic; k > 0 = force tile candidate k-1 of this op
                     (1..mcn_conv2d_tile_candidates); results are identical up to fp32 summation order — used by callers that
                     time the candidates.  | MCN_TILE_NOSPLIT: fwd / dgrad run every tile's whole K loop in one workgroup
                     (default: the tiles of the last, partially filled round of workgroups are cut along K when the workspace
                     has room for the partial sums — deterministic, again identical up to fp32 summation order) */
} mcn_conv_geom;
#define MCN_TILE_NOSPLIT 0x100
/* | MCN_TILE_NOWINO: keep an fp32 3x3 / stride 1 / pad 1 convolution on the direct kernels (default: Winograd F(2x2, 3x3)).
 * The packed operand of such a layer is the transformed filter: pack job and call must agree on this flag. */
#define MCN_TILE_NOWINO 0x200

typedef enum { MCN_CONV_FWD = 0, MCN_CONV_DGRAD = 1, MCN_CONV_WGRAD = 2 } mcn_conv_op;

/* number of tile shapes the given op can be forced to through mcn_conv_geom.tile (0: op takes no hint).  Forward / dgrad: 1-3 = 128x128, 128x64, 64x64 on 4 waves;
 * 4-5 = 256x128, 128x128 on 8 waves and 6 = the 3x3 window ping-pong kernel (conv_gemm_nt_wpp), 2-byte types only — a hint the geometry or dtype cannot take runs the
 * library's own choice, never an error. */
int mcn_conv2d_tile_candidates(mcn_conv_op op);
/* bytes of workspace the given op needs for this geometry/dtype (0 is a valid answer) */
size_t mcn_conv2d_workspace_bytes(mcn_conv_op op, const mcn_conv_geom* g, mcn_dtype dtype);
/* K-slices per tile of the last, partially filled round of workgroups of the op's (first) GEMM launch: 1 = unsplit
 * (always for MCN_CONV_WGRAD, whose split over pixels is mcn_conv2d_workspace_bytes' business) */
int32_t mcn_conv2d_kslices(mcn_conv_op op, const mcn_conv_geom* g, mcn_dtype dtype);

/* replaces tf.nn.conv2d (reference convnet.py:1659) [+ tf.nn.bias_add, convnet.py:1694, when
 * bias != NULL].  w_packed (may be NULL): the operand produced by mcn_conv2d_pack_* from the same
 * w_hwio for MCN_CONV_FWD; when NULL the call re-packs w_hwio into `workspace` itself. */
int mcn_conv2d_fwd(const void* x, const float* w_hwio, const void* w_packed, const float* bias, void* y,
                   const mcn_conv_geom* g, mcn_dtype dtype, mcn_layout layout, void* workspace,
                   size_t workspace_bytes, void* stream);

/* replaces Conv2DBackpropInput (autograd of convnet.py:1659 via optimizers.py:106).
 * dx:[N][H][W][Cin] (dense, channel stride Cin).  accumulate != 0 => dx += result.
 * w_packed (may be NULL): operand packed for MCN_CONV_DGRAD. */
int mcn_conv2d_dgrad(const void* dy, const float* w_hwio, const void* w_packed, void* dx, const mcn_conv_geom* g,
                     int accumulate, mcn_dtype dtype, mcn_layout layout, void* workspace, size_t workspace_bytes,
                     void* stream);

/* The per-use cast of the fp32 master weights (reference weight_variable(), convnet.py:1421-1422) done ONCE per
 * step for every convolution of a model in a single launch: the caller keeps one packed operand per (conv, op),
 * builds a descriptor table on the host, uploads it once, and runs it whenever the masters changed.
 *   bytes  = mcn_conv2d_packed_bytes(op, geom, dtype)           (0: this op takes no packed operand)
 *   mcn_conv2d_pack_table_build(jobs, n, dtype, host_table, mcn_conv2d_pack_table_bytes(jobs, n), &ndesc)
 *   copy host_table to the device;  mcn_conv2d_pack_run(dev_table, ndesc, dtype, stream)
 * The table is opaque; pack_table_bytes() is an upper bound (big operands are cut into several descriptors so that the one
 * launch stays balanced), ndesc is the number of descriptors actually written.                     */
typedef struct {
    const float* w_hwio; /* device pointer: fp32 master [KH][KW][Cin][Cout] */
    void* packed;        /* device pointer: mcn_conv2d_packed_bytes() bytes */
    mcn_conv_geom geom;
    int32_t op;          /* MCN_CONV_FWD or MCN_CONV_DGRAD */
    int32_t reserved;
} mcn_pack_job;
size_t mcn_conv2d_packed_bytes(mcn_conv_op op, const mcn_conv_geom* g, mcn_dtype dtype);
size_t mcn_conv2d_pack_table_bytes(const mcn_pack_job* jobs, int32_t njobs);
int mcn_conv2d_pack_table_build(const mcn_pack_job* jobs, int32_t njobs, mcn_dtype dtype, void* host_table,
                                size_t host_table_bytes, int32_t* ndesc);
int mcn_conv2d_pack_run(const void* dev_table, int32_t ndesc, mcn_dtype dtype, void* stream);

/* Pixel-pair form of a horizontally stride-2 convolution on <= 4 input channels in a 2-byte type (the stem conv of
 * models/resnet_v1_5.py:25 / efficientnet.py:96): the image is stored 4 channels per pixel ([N,H,W,4], x_cs = 4), two adjacent
 * pixels are one 16-byte chunk, and the conv is the stride-(SH,1) convolution of the [N,H,W/2,8] view with a [KH,KW',8,Cout]
 * filter (7x7/2: KW' = 4, K 392 -> 224).  mcn_conv2d_pair_geom returns 1 and fills `paired` (to be used with the ordinary
 * mcn_conv2d_* entry points on the same buffers) when the form exists, else 0.  mcn_conv2d_pair_weights builds the paired filter
 * (fp32, [KH][KW'][8][Cout]) from the HWIO master; mcn_conv2d_pair_wgrad_fold gathers the paired filter's gradient back to HWIO.
 * `g` is the ORIGINAL geometry in all three. */
int mcn_conv2d_pair_geom(const mcn_conv_geom* g, mcn_dtype dtype, mcn_conv_geom* paired);
int mcn_conv2d_pair_weights(const float* w_hwio, float* w_paired, const mcn_conv_geom* g, mcn_dtype dtype, void* stream);
int mcn_conv2d_pair_wgrad_fold(const float* dw_paired, float* dw_hwio, const mcn_conv_geom* g, mcn_dtype dtype, void* stream);

/* profiling aid: writes the name of the GEMM kernel a conv call launches (as rocprofv3 prints it) into buf
 * (>= 64 bytes) and returns how many launches of it the call makes (stride-2 dgrad: one per parity class).  For
 * The last template argument of the printed name is the epilogue variant: 0 as printed; 1 for mcn_conv2d_fwd_bnstats;
 * 2 for a dgrad that accumulates (accumulate != 0, mcn_conv2d_dgrad_addmasked). */
int mcn_conv2d_kernel_name(mcn_conv_op op, const mcn_conv_geom* g, mcn_dtype dtype, char* buf, size_t buflen);
/* the same per launch: one line "<kernel symbol>:<filter taps of that launch>\n" for every GEMM launch of the call (the
 * stride-parity classes of a strided dgrad differ in taps, tile and addressing mode); returns the number of lines.
 * buf >= 96 bytes per launch. */
int mcn_conv2d_launch_list(mcn_conv_op op, const mcn_conv_geom* g, mcn_dtype dtype, char* buf, size_t buflen);

/* dgrad fused with the gradient fan-in of an identity shortcut (models/resnet_v1_5.py:66-70: x + skip, relu): the input
 * of a residual block receives dgrad(conv_0) + [y_block > 0] * dy_block.  add_src = dy_block (same shape and dtype as dx),
 * add_mask = the ReLU byte mask the block's last BN wrote (mcn_bn_fwd_train, relu_mask): the masked tensor is never
 * materialised (one write + one read of a block-output-sized tensor less per identity block).  Stride-1 geometries only:
 * mcn_conv2d_dgrad_addmasked_ok() != 0. */
int32_t mcn_conv2d_dgrad_addmasked_ok(const mcn_conv_geom* geom, mcn_dtype dtype);
int mcn_conv2d_dgrad_addmasked(const void* dy, const float* w_hwio, const void* w_packed, void* dx, const void* add_src,
                               const uint8_t* add_mask, const mcn_conv_geom* geom, mcn_dtype dtype, mcn_layout layout,
                               void* workspace, size_t workspace_bytes, void* stream);

/* replaces Conv2DBackpropFilter.  dw:[KH][KW][Cin][Cout] fp32 (overwritten; deterministic
 * split-K through workspace slabs).  dbias (optional, fp32 [Cout]) = column sums of dy
 * (BiasAddGrad). grad_scale multiplies both (loss un-scaling, optimizers.py:109-111). */
int mcn_conv2d_wgrad(const void* x, const void* dy, float* dw_hwio, float* dbias, const mcn_conv_geom* g,
                     float grad_scale, mcn_dtype dtype, mcn_layout layout, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---- batch normalisation ----------------------------------------------------------------
 * x,y:[M][C] (M = N*H*W).  replaces tf.nn.fused_batch_norm(is_training=True) and the running
 * statistics update (reference convnet.py:1883-1888, 1898-1914):
 *   mean = E[x], var = biased variance, y = act(gamma*(x-mean)/sqrt(var+eps) + beta [+ skip])
 *   batch_mean = mean, batch_var = var * M/max(M-1,1)
 *   if running_mean != NULL: running <- momentum*running + (1-momentum)*batch  (single tower)
 * save_mean / save_invstd (fp32 [C]) are kept for the backward pass.  `skip` (same shape as y,
 * may be NULL) is the residual branch of stochastic_depth(drop_rate=0) (convnet.py:2511) and
 * `act` the following tf.nn.relu (convnet.py:2537), both fused.
 * relu_mask (may be NULL; mcn_bn_relu_mask_bytes(M, C, dtype) bytes, needs C % chunk == 0): with act == RELU the apply
 * pass also writes [y > 0] as one byte per 16-byte chunk; mcn_bn_bwd reads it instead of y (1/16 of the bytes) — meant
 * for the BNs with a fused residual, whose mask cannot be recomputed from x alone.
 * workspace: mcn_bn_workspace_bytes(M, C). */
size_t mcn_bn_workspace_bytes(int64_t M, int32_t C);
size_t mcn_bn_relu_mask_bytes(int64_t M, int32_t C, mcn_dtype dtype);
int mcn_bn_fwd_train(const void* x, const float* gamma, const float* beta, const void* skip, void* y,
                     uint8_t* relu_mask, float* save_mean, float* save_invstd, float* batch_mean, float* batch_var,
                     float* running_mean, float* running_var, float momentum, int64_t M, int32_t C, float eps,
                     mcn_act act, mcn_dtype dtype, void* workspace, size_t workspace_bytes, void* stream);
/* mcn_bn_fwd_train (no residual, no ReLU mask) whose apply pass ALSO leaves the per-image channel means of the stored output
 * (round 4): y [N][HW][C] = act(bn(x)), gap [N][C] = mean over HW of y as stored — the squeeze-excite block's
 * tf.reduce_mean(x, axis=[1, 2]) right behind the depthwise conv's BN + swish (models/efficientnet.py:150-152, 183), which was a
 * second full read of y (mcn_global_avgpool_fwd).  Sums in fp32, one rounding to the storage type. */
int mcn_bn_fwd_train_gap(const void* x, const float* gamma, const float* beta, void* y, void* gap, float* save_mean,
                         float* save_invstd, float* batch_mean, float* batch_var, float* running_mean, float* running_var,
                         float momentum, int32_t N, int32_t HW, int32_t C, float eps, mcn_act act, mcn_dtype dtype,
                         void* workspace, size_t workspace_bytes, void* stream);

/* conv -> batch-norm fusion: the forward conv accumulates, in its epilogue, the column sums and sums of squares of the
 * values it stores and writes them as mcn_conv2d_bnstats_rows() partial rows — [rows][3][Cout] fp32: sum(y-p),
 * sum((y-p)^2) and the shift p over rows_per_partial consecutive pixels each (0 rows = geometry not eligible, use the
 * plain calls); mcn_bn_fwd_train_fused merges them in double precision and skips the statistics pass over x (one read
 * of the layer output less per BN; same tf.nn.fused_batch_norm semantics, convnet.py:1883-1914).  Caller-owned buffer.
 * *rows_per_partial < 0 on return announces COUNTED rows of BN = -*rows_per_partial output channels (BN <= Cout): the launch runs as persistent
 * workgroups, each walks tiles of ONE block of BN output channels, keeps its sums across them and writes one row at the end (a
 * row's pixels are not a contiguous range): [Cout/BN blocks][rows/blocks][4][BN] fp32 — sum(y-p), sum((y-p)^2), p and the number of
 * pixel rows summed, keyed by the channel block, so a channel's partials are the rows/blocks rows of its block.
 * (*rows_per_partial == 0, accepted by the BN entry points: the same four planes as [rows][4][Cout] with count 0 for foreign
 * blocks.)  Size the buffer for rows x 4 x Cout floats and hand rows_per_partial through to mcn_bn_fwd_train_fused unchanged. */
int32_t mcn_conv2d_bnstats_rows(const mcn_conv_geom* geom, mcn_dtype dtype, int32_t* rows_per_partial);
int mcn_conv2d_fwd_bnstats(const void* x, const float* w_hwio, const void* w_packed, const float* bias, void* y,
                           float* stats_partials, const mcn_conv_geom* geom, mcn_dtype dtype, mcn_layout layout,
                           void* workspace, size_t workspace_bytes, void* stream);
int mcn_bn_fwd_train_fused(const void* x, const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma,
                           const float* beta, const void* skip, void* y, uint8_t* relu_mask, float* save_mean,
                           float* save_invstd, float* batch_mean, float* batch_var, float* running_mean, float* running_var, float momentum,
                           int64_t M, int32_t C, float eps, mcn_act act, mcn_dtype dtype, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Residual unit with a projection shortcut, y = relu(bn(x) + bn_s(xs)) (models/resnet_v1_5.py:63-75): the shortcut's BN needs
 * no apply pass of its own — mcn_bn_fwd_train_fused_stats finalizes its statistics (saved / batch / running values as
 * mcn_bn_fwd_train_fused) and leaves its affine in scale_shift [2][C] (caller-owned), mcn_bn_fwd_train_fused_affskip is
 * mcn_bn_fwd_train_fused(act = ReLU) taking the shortcut conv's raw output xs and that affine as the residual.  Same y and ReLU mask,
 * bit for bit, as applying the shortcut BN to a tensor first; that tensor is never written. */
int mcn_bn_fwd_train_fused_stats(const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma, const float* beta,
                                 float* save_mean, float* save_invstd, float* batch_mean, float* batch_var, float* running_mean,
                                 float* running_var, float momentum, int64_t M, int32_t C, float eps, float* scale_shift, void* workspace,
                                 size_t workspace_bytes, void* stream);
int mcn_bn_fwd_train_fused_affskip(const void* x, const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma,
                                   const float* beta, const void* skip_x, const float* skip_scale_shift, void* y, uint8_t* relu_mask,
                                   float* save_mean, float* save_invstd, float* batch_mean, float* batch_var, float* running_mean,
                                   float* running_var, float momentum, int64_t M, int32_t C, float eps, mcn_dtype dtype, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* conv -> BN(train) -> ReLU -> max-pool (the stem of models/resnet_v1_5.py:25-31) with the statistics from the conv epilogue: the
 * finalize step of mcn_bn_fwd_train_fused, then ONE pass that normalises, rectifies and pools x [N,H,W,C] into pooled
 * [N,OH,OW,C] + argmax (same values, ties and arg-max as mcn_bn_fwd_train_fused followed by mcn_maxpool_fwd; the normalised
 * tensor is never stored).  mcn_maxpool_fwd_affine_relu is that pass on its own (scale / shift: fp32 [C]). */
int mcn_bn_fwd_train_fused_maxpool(const void* x, const float* stats_partials, int32_t nparts, int32_t rows_per_partial, const float* gamma,
                                   const float* beta, void* pooled, int8_t* argmax, float* save_mean, float* save_invstd, float* batch_mean,
                                   float* batch_var, float* running_mean, float* running_var, float momentum, int32_t N, int32_t H, int32_t W,
                                   int32_t C, float eps, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH,
                                   int32_t OW, mcn_dtype dtype, void* workspace, size_t workspace_bytes, void* stream);
int mcn_maxpool_fwd_affine_relu(const void* x, const float* scale, const float* shift, void* y, int8_t* argmax, int32_t N, int32_t H, int32_t W,
                                int32_t C, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW,
                                mcn_dtype dtype, void* stream);

/* replaces tf.nn.fused_batch_norm(is_training=False) (convnet.py:1889-1896, 1916-1923) */
int mcn_bn_fwd_infer(const void* x, const float* gamma, const float* beta, const float* mean, const float* var,
                     const void* skip, void* y, int64_t M, int32_t C, float eps, mcn_act act, mcn_dtype dtype,
                     void* stream);

/* replaces FusedBatchNormGrad [+ ReluGrad + the add's gradient fan-out].
 * dy: gradient w.r.t. the (activated) output y.  If act == RELU the mask [y > 0] is applied first:
 *   relu_mask != NULL : mask from the byte mask mcn_bn_fwd_train wrote (y is ignored);
 *   y != NULL : mask from the stored forward output (required when the forward fused a residual `skip`);
 *   y == NULL : mask recomputed from x as [fma(x, gamma*invstd, beta - mean*gamma*invstd) > 0], the expression the
 *               forward apply pass evaluated (saves one read of y per pass; forward without `skip` only).
 * If act == SWISH (EfficientNet, models/efficientnet.py:66,142,150) z = bn(x) is recomputed the same way and
 *   dz = dy * (s + z*s*(1-s)), s = sigmoid(z); no fused residual (dskip must be NULL), y is ignored.
 * dskip (may be NULL): receives the masked dy, i.e. the gradient of the residual branch.
 * dgamma/dbeta fp32 [C], multiplied by grad_scale. */
int mcn_bn_bwd(const void* dy, const void* x, const void* y, const uint8_t* relu_mask, const float* gamma, const float* beta,
               const float* save_mean, const float* save_invstd, void* dx, void* dskip, float* dgamma, float* dbeta,
               float grad_scale, int64_t M, int32_t C, mcn_act act, mcn_dtype dtype, void* workspace,
               size_t workspace_bytes, void* stream);

/* BN-backward reduction in the epilogue of the dgrad that produces dy (the bottleneck's inner BNs: conv -> BN -> ReLU -> conv): when a
 * conv is the ONLY reader of a training-mode BN + ReLU's output, its dgrad mcn_conv2d_dgrad_bnred also accumulates, per (M tile,
 * wave row), sum dy' and sum dy' * x (dy' = the gradient it stores where the forward's ReLU bit is set, bn_x = the BN's input, relu_mask =
 * the byte mask of mcn_bn_fwd_train*) into red_partials [rows][2][Cin], rows = mcn_conv2d_dgrad_bnred_rows() (0: not eligible);
 * mcn_bn_bwd_from_partials then finalizes from those rows and runs only the apply pass: the BN backward's reduction pass over (dy, x)
 * is gone.  dx / dgamma / dbeta as mcn_bn_bwd up to the order of fp32 sums. */
int32_t mcn_conv2d_dgrad_bnred_rows(const mcn_conv_geom* geom, mcn_dtype dtype);
int mcn_conv2d_dgrad_bnred(const void* dy, const float* w_hwio, const void* w_packed, void* dx, const void* bn_x, const uint8_t* relu_mask,
                           float* red_partials, const mcn_conv_geom* geom, mcn_dtype dtype, mcn_layout layout, void* workspace,
                           size_t workspace_bytes, void* stream);
/* mcn_conv2d_dgrad_addmasked and mcn_conv2d_dgrad_bnred in ONE launch (round 4): dx = dgrad(dy) + add_src * [add_mask bit] is the complete
 * gradient of a residual unit's output y_b = relu(bn(x_b) + skip_b) (models/resnet_v1_5.py:176-183: its readers are the next unit's conv_0 —
 * this launch — and the next unit's residual add), so the backward sums of that unit's output BN (convnet.py:1883; tf.gradients of
 * fused_batch_norm) ride in the epilogue: red_partials [rows][2][Cin] as in mcn_conv2d_dgrad_bnred with bn_x = x_b and relu_mask = the
 * [y_b > 0] byte mask, and mcn_bn_bwd_from_partials runs that BN's backward without its reduction pass over (dy, x_b).  Eligible when
 * mcn_conv2d_dgrad_addmasked_ok() != 0 and mcn_conv2d_dgrad_bnred_rows() > 0 (= rows). */
int mcn_conv2d_dgrad_addmasked_bnred(const void* dy, const float* w, const void* w_packed, void* dx, const void* add_src,
                                     const uint8_t* add_mask, const void* bn_x, const uint8_t* relu_mask, float* red_partials,
                                     const mcn_conv_geom* g, mcn_dtype dtype, mcn_layout layout, void* workspace,
                                     size_t workspace_bytes, void* stream);
int mcn_bn_bwd_from_partials(const void* dy, const void* x, const uint8_t* relu_mask, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_invstd, const float* red_partials, int32_t nparts, void* dx,
                             float* dgamma, float* dbeta, float grad_scale, int64_t M, int32_t C, mcn_dtype dtype, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Squeeze-excite backward without the gradient tensor of the block's input (models/efficientnet.py:152-163): x_se (the output of a
 * BN + swish) is read by the SE branch's global average pool and by y = x_se * m; its gradient round(round(dy*m) + dgap/HW) is
 * composed inside the BN's two backward passes (mcn_bn_bwd_se; dy = gradient of y, se_mask = m [N,C], dgap [N,C] = gradient of the
 * pooled tensor), mcn_channel_scale_bwd_dm is the reduction half of mcn_channel_scale_bwd (dm[n,c] = sum_hw dy * x_se).  Bit-identical
 * to mcn_channel_scale_bwd + mcn_global_avgpool_bwd_acc + mcn_bn_bwd(act = swish). */
int mcn_channel_scale_bwd_dm(const void* dy, const void* x, void* dm, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void* stream);
int mcn_bn_bwd_se(const void* dy, const void* se_mask, const void* dgap, const void* x, const float* gamma, const float* beta, const float* save_mean,
                  const float* save_invstd, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int64_t HW, int32_t C,
                  mcn_dtype dtype, void* workspace, size_t workspace_bytes, void* stream);
/* Round 4: the same pair with ONE pass fewer over the widest activations of an MBConv block.  mcn_channel_scale_bwd_dm_bnsums reads the BN's INPUT x (not the
 * stored x_se: x_se = round(swish(bn(x))) is rebuilt on the fly, dm as above) and also writes per-image(-slice) sums, fp32, mcn_se_bwd_sums_floats() of them (dm partial, sum dy s', sum dy s' xh,
 * sum s', sum s' xh; s' = swish'(bn(x)), xh = (x - mean) * invstd) from which mcn_bn_bwd_se_sums forms the BN-backward sums in a loop over N instead of its
 * reduction pass (sum g s' = sum_n m A + dgap / HW B ...; g unrounded there: dgamma / dbeta equal mcn_bn_bwd_se's to fp32 summation accuracy, dx up to
 * one rounding of the storage type around the two coefficients they feed).  Replaces efficientnet.py:152-163's backward like the pair above. */
/* ... and its forward half: with mcn_bn_fwd_train_gap(y = NULL) (the pooled means only) and y = round(round(act(bn(x))) * m[n,c]) from the BN's INPUT here, the BN + swish
 * output x_se of a squeeze-excite block is never written or read (models/efficientnet.py:150-163; the backward pair above does not read it either). */
int mcn_bn_act_scale_fwd(const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, const void* m, void* y, int32_t N, int64_t HW,
                         int32_t C, mcn_act act, mcn_dtype dtype, void* stream);
size_t mcn_se_bwd_sums_floats(int32_t N, int64_t HW, int32_t C, mcn_dtype dtype);      /* floats of `sums` for this shape (the pass slices every image's pixels over several workgroups) */
int mcn_channel_scale_bwd_dm_bnsums(const void* dy, const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dm,
                                    float* sums, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype, void* stream);
int mcn_bn_bwd_se_sums(const void* dy, const void* se_mask, const void* dgap, const void* x, const float* gamma, const float* beta, const float* save_mean,
                       const float* save_invstd, const float* sums, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int64_t HW, int32_t C,
                       mcn_dtype dtype, void* ws, size_t ws_bytes, void* stream);

/* mcn_bn_bwd(act = ReLU, no fused residual) for a BN whose output feeds ONLY a 3x3 / stride-2 max-pool (the stem): takes the
 * pooled gradient [N,OH,OW,C] and the pool's arg-max and routes it inside its two passes; the full-resolution gradient of the BN
 * output (the largest gradient tensor of the network) is never written.  Same dx as mcn_maxpool_bwd + mcn_bn_bwd, bit for bit. */
int mcn_bn_bwd_maxpool(const void* dy_pooled, const int8_t* argmax, const void* x, const float* gamma, const float* beta, const float* save_mean,
                       const float* save_invstd, void* dx, float* dgamma, float* dbeta, float grad_scale, int32_t N, int32_t H, int32_t W, int32_t C,
                       int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/* gradient of fused_batch_norm(is_training=False) used INSIDE a training graph: the frozen-statistics BN of
 * update_batch_norm=False / blocks_to_train (convnet.py:1781-1789, 1915-1923).  mean / var: the running statistics the
 * forward pass (mcn_bn_fwd_infer) normalised with.  dx = dz*gamma*invstd, dgamma = sum(dz*xhat), dbeta = sum(dz), with
 * dz = dy masked / scaled by the activation as in mcn_bn_bwd (ReLU: from the stored y, required; swish: recomputed). */
int mcn_bn_bwd_frozen(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* mean,
                      const float* var, float eps, void* dx, void* dskip, float* dgamma, float* dbeta, float grad_scale,
                      int64_t M, int32_t C, mcn_act act, mcn_dtype dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---- depthwise convolution ---------------------------------------------------
 * replaces tf.nn.depthwise_conv2d (convnet.py:1645) with channel multiplier 1 (every EfficientNet call site,
 * models/efficientnet.py:145) and its two gradients.  geom: Cin == Cout == C (multiple of the 16-byte chunk);
 * w / dw: fp32 [KH][KW][C] (the reference's [kh,kw,cin,1] filter, rounded per use in bf16 mode, convnet.py:1421);
 * x, y and their gradients NHWC in `dtype`.  HBM-bound: one pass over x and y.
 * dgrad: accumulate != 0 adds into dx.  wgrad: deterministic two-stage reduction through `workspace`. */
int mcn_dwconv2d_fwd(const void* x, const float* w, void* y, const mcn_conv_geom* geom, mcn_dtype dtype, void* stream);
int mcn_dwconv2d_dgrad(const void* dy, const float* w, void* dx, const mcn_conv_geom* geom, int32_t accumulate,
                       mcn_dtype dtype, void* stream);
size_t mcn_dwconv2d_workspace_bytes(const mcn_conv_geom* geom, mcn_dtype dtype);
int mcn_dwconv2d_wgrad(const void* x, const void* dy, float* dw, const mcn_conv_geom* geom, float grad_scale,
                       mcn_dtype dtype, void* workspace, size_t workspace_bytes, void* stream);

/* depthwise channel multiplier != 1 and biased depthwise convolution (convnet.py:1634-1650 filter [kh, kw, cin, mult], output channel c * mult + q;
 * bias: tf.nn.bias_add, convnet.py:1678-1694).  A multiplier-`mult` depthwise convolution = mcn_dwconv2d_* on C * mult channels applied to the input
 * with every channel repeated `mult` times (the filter buffer [kh][kw][cin][mult] read as [kh][kw][cin * mult] is already in that order):
 *   mcn_channel_repeat_fwd: y[m][c * mult + q] = x[m][c];   mcn_channel_repeat_bwd: dx[m][c] = sum_q dy[m][c * mult + q] (fp32 sum, one rounding).
 * The bias is added by mcn_channel_affine(scale = 1, shift = bias); its gradient dbias[c] = grad_scale * sum_m dy[m][c] by mcn_bias_grad
 * (deterministic two-stage column sum; workspace: mcn_bias_grad_workspace_bytes(M, C)). */
int mcn_channel_repeat_fwd(const void* x, void* y, int64_t M, int32_t C, int32_t mult, mcn_dtype dtype, void* stream);
int mcn_channel_repeat_bwd(const void* dy, void* dx, int64_t M, int32_t C, int32_t mult, mcn_dtype dtype, void* stream);
size_t mcn_bias_grad_workspace_bytes(int64_t M, int32_t C);
int mcn_bias_grad(const void* dy, float* dbias, int64_t M, int32_t C, float grad_scale, mcn_dtype dtype, void* workspace, size_t workspace_bytes,
                  void* stream);

/* squeeze-excite scale (models/efficientnet.py:161 `x = x*se_mask`): y[n,h,w,c] = x[n,h,w,c] * m[n,c];
 * backward: dx = dy * m, dm[n,c] = sum_hw dy * x.  x/y/m in `dtype`, HW = H*W. */
int mcn_channel_scale_fwd(const void* x, const void* m, void* y, int32_t N, int64_t HW, int32_t C, mcn_dtype dtype,
                          void* stream);
int mcn_channel_scale_bwd(const void* dy, const void* x, const void* m, void* dx, void* dm, int32_t N, int64_t HW,
                          int32_t C, mcn_dtype dtype, void* stream);

/* per-channel affine y[m][c] = x[m][c]*scale[c] + shift[c] over [M][C] (the VGG input re-scaling,
 * reference models/vggnet.py:25; also the apply pass of batch norm). scale/shift fp32 [C]. */
int mcn_channel_affine(const void* x, const float* scale, const float* shift, void* y, int64_t M, int32_t C,
                       mcn_dtype dtype, void* stream);

/* ---- element-wise -----------------------------------------------------------------------
 * tf.nn.relu / ReluGrad (convnet.py:2537); x + skip followed by relu (convnet.py:2511, 2537). */
int mcn_relu_fwd(const void* x, void* y, int64_t n, mcn_dtype dtype, void* stream);
int mcn_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, mcn_dtype dtype, void* stream);
int mcn_add_relu_fwd(const void* a, const void* b, void* y, int64_t n, mcn_act act, mcn_dtype dtype, void* stream);
/* dx = dy * [y > 0] (written once; both branches of the add read it) */
int mcn_add_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, mcn_act act, mcn_dtype dtype, void* stream);
/* element-wise swish / sigmoid / relu on small tensors (the squeeze-excite branch, models/efficientnet.py:186-195;
 * tf.nn.sigmoid convnet.py:2550, swish convnet.py:2553-2556).  Backward: relu and sigmoid differentiate through the
 * stored output y (x may be NULL), swish through its input x (y may be NULL). */
int mcn_act_fwd(const void* x, void* y, int64_t n, mcn_act act, mcn_dtype dtype, void* stream);
int mcn_act_bwd(const void* dy, const void* x, const void* y, void* dx, int64_t n, mcn_act act, mcn_dtype dtype,
                void* stream);
/* the same with the activation's parameter (ConvNet.activation(x, type, params), convnet.py:2514-2534): every entry of the
 * reference's dispatcher — relu, relu6, lrelu (param = alpha), tanh, sigmoid, swish.  Backward: relu / relu6 / tanh / sigmoid
 * differentiate through y (Relu6Grad: 0 < y < 6; TanhGrad: 1 - y^2), swish / lrelu through x (LeakyReluGrad: x > 0 ? dy : alpha*dy).
 * mcn_act_fwd / mcn_act_bwd are these with param = 0.2 (the reference's default alpha). */
int mcn_act_fwd_p(const void* x, void* y, int64_t n, mcn_act act, float param, mcn_dtype dtype, void* stream);
int mcn_act_bwd_p(const void* dy, const void* x, const void* y, void* dx, int64_t n, mcn_act act, float param, mcn_dtype dtype,
                  void* stream);
/* a += b (gradient accumulation at fan-out points) */
int mcn_accumulate(void* a, const void* b, int64_t n, mcn_dtype dtype, void* stream);
/* dtype conversion (tf.cast, convnet.py:469-471, 477-480) */
int mcn_cast(const void* src, mcn_dtype src_dtype, void* dst, mcn_dtype dst_dtype, int64_t n, void* stream);

/* input preparation (reference convnet.py:452-471): y = (x - image_mean) * scale_factor, cast to
 * dtype, channels padded with zeros to out_cs (>= C), src layout NHWC or NCHW -> NHWC. x is fp32. */
int mcn_input_prep(const float* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t out_cs,
                   float image_mean, float scale_factor, mcn_layout src_layout, mcn_dtype dtype, void* stream);
/* labels (fp32 class ids; NaN or out-of-range => all-zero row) -> one-hot fp32 [B][C] (convnet.py:441-449) */
int mcn_one_hot(const float* labels, float* onehot, int32_t B, int32_t C, void* stream);

/* ---- pooling -----------------------------------------------------------------------------
 * tf.nn.max_pool (convnet.py:1509): padded cells never win; argmax (int8 window-local index of
 * the first maximum in row-major window order) is stored for the backward pass. */
int mcn_maxpool_fwd(const void* x, void* y, int8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH,
                    int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW,
                    mcn_dtype dtype, void* stream);
int mcn_maxpool_bwd(const void* dy, const int8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                    int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH,
                    int32_t OW, mcn_dtype dtype, void* stream);
/* tf.nn.avg_pool (convnet.py:1548): SAME divides by the number of valid cells */
int mcn_avgpool_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                    int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype,
                    void* stream);
int mcn_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                    int32_t SH, int32_t SW, int32_t padT, int32_t padL, int32_t OH, int32_t OW, mcn_dtype dtype,
                    void* stream);
/* tf.reduce_mean(x, axis=[1,2]) (models/resnet_v1_5.py:72-73): [N][HW][C] -> [N][C] */
int mcn_global_avgpool_fwd(const void* x, void* y, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void* stream);
int mcn_global_avgpool_bwd(const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void* stream);
/* dx += broadcast(dy) / HW: the pooled branch's contribution when dx already holds another one (the input of a
 * squeeze-excite block feeds both the pool and the channel scale, models/efficientnet.py:179-197) */
int mcn_global_avgpool_bwd_acc(const void* dy, void* dx, int32_t N, int32_t HW, int32_t C, mcn_dtype dtype, void* stream);

/* ---- fully connected ----------------------------------------------------------------------
 * tf.matmul(x, W) + b (convnet.py:1743): x:[B][In], w:[In][Out] fp32 master, y:[B][Out]. */
size_t mcn_fc_workspace_bytes(int32_t B, int32_t In, int32_t Out, mcn_dtype dtype);
int mcn_fc_fwd(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t In, int32_t Out,
               mcn_dtype dtype, void* workspace, size_t workspace_bytes, void* stream);
int mcn_fc_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias, float grad_scale,
               int32_t B, int32_t In, int32_t Out, mcn_dtype dtype, void* workspace, size_t workspace_bytes,
               void* stream);

/* ---- loss ----------------------------------------------------------------------------------
 * tf.nn.softmax (models/resnet_v1_5.py:78) + softmax_cross_entropy_with_logits_v2
 * (convnet.py:600) + valid mask / class weights / label smoothing / mean over the batch
 * (convnet.py:552-594).  logits fp32 [B][C]; labels fp32 [B][C] (one-hot or soft);
 * class_w fp32 [C] or NULL.  Outputs: pred [B][C], ce [B] (per-sample cross-entropy),
 * coef [B] (batch_weight*valid), dlogits [B][C] = loss_scale * d(mean(coef*ce))/dlogits,
 * loss[1] = mean(coef*ce). */
int mcn_softmax_xent_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce,
                             float* coef, float* dlogits, float* loss, int32_t B, int32_t C, float label_smoothing,
                             float loss_scale, void* stream);
/* the same loss over many short rows — SegNet's per-pixel cross-entropy (segmentation/segnet.py:74-78 -> convnet.py:528-597
 * with Y of shape [N,H,W,classes]): B = N*H*W rows, one thread per row; workspace >= 1024 floats for the mean. */
int mcn_softmax_xent_rows_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce,
                                  float* coef, float* dlogits, float* loss, int64_t B, int32_t C, float label_smoothing,
                                  float loss_scale, void* workspace, size_t workspace_bytes, void* stream);
/* the same with SegNet's label smoothing (segmentation/segnet.py:117-122): the cross-entropy runs against
 * (1 - ls) * labels + ls * avg_labels, avg_labels = tf.nn.avg_pool2d(labels, 5x5, stride 1, SAME) of the one-hot map
 * (mcn_avgpool_fwd), while batch weights and the valid mask keep reading the raw labels (convnet.py:552, 567-573). */
int mcn_softmax_xent_rows_soft_fwd_bwd(const float* logits, const float* labels, const float* avg_labels, const float* class_w,
                                       float* pred, float* ce, float* coef, float* dlogits, float* loss, int64_t B, int32_t C,
                                       float label_smoothing, float loss_scale, void* workspace, size_t workspace_bytes,
                                       void* stream);
/* the focal variants (convnet.py:581-592, round 4; focal_gamma = sigmoid_focal_alpha = 0: the entry points above):
 *   focal_gamma > 0         : softmax_losses *= (1 - p_t)^gamma, p_t = sum_c Y_c * pred_c — differentiated through the softmax as tf.gradients does;
 *   sigmoid_focal_alpha > 0 : softmax_losses *= stop_gradient(1 - sigmoid(alpha (p_t - 0.5))) / (1 - sigmoid(-alpha / 2)).
 * `ce` receives the row's FACTORED cross-entropy, so loss = mean(coef * ce) as before. */
int mcn_softmax_xent_focal_fwd_bwd(const float* logits, const float* labels, const float* class_w, float* pred, float* ce,
                                   float* coef, float* dlogits, float* loss, int32_t B, int32_t C, float label_smoothing,
                                   float loss_scale, float focal_gamma, float sigmoid_focal_alpha, void* stream);
int mcn_softmax_xent_rows_focal_fwd_bwd(const float* logits, const float* labels, const float* avg_labels, const float* class_w,
                                        float* pred, float* ce, float* coef, float* dlogits, float* loss, int64_t B, int32_t C,
                                        float label_smoothing, float loss_scale, float focal_gamma, float sigmoid_focal_alpha,
                                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- segmentation path -----------------------------------------------------
 * tf.image.resize_bilinear (convnet.py:2396; align_corners=True at models/deeplabv3plus.py:64,74) and its gradient
 * (gather form, deterministic); x/y NHWC in `dtype`, interpolation in fp32. */
int mcn_resize_bilinear_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                            int32_t align_corners, mcn_dtype dtype, void* stream);
int mcn_resize_bilinear_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                            int32_t align_corners, mcn_dtype dtype, void* stream);
/* tf.concat along channels (models/deeplabv3plus.py:101,110) = one call per input; its gradient = one call per slice:
 * dst[m][dst_offset + c] = src[m][src_offset + c] for c < C, m < M (strides in elements). */
int mcn_copy_channels(const void* src, int32_t src_stride, int32_t src_offset, void* dst, int32_t dst_stride,
                      int32_t dst_offset, int64_t M, int32_t C, mcn_dtype dtype, void* stream);
/* SegNet label encoding (segmentation/segnet.py:31-50): NaN -> 0, class = round(label - 1), one-hot of depth C; label 0
 * (class -1) and classes >= C give an all-zero row = ignored pixel.  labels fp32 [P], onehot fp32 [P][C]. */
int mcn_one_hot_seg(const float* labels, float* onehot, int64_t P, int32_t C, void* stream);

/* l2_factor * sum tf.nn.l2_loss(w) over a flat range (convnet.py:560-563): out[0] += factor*sum(w^2)/2 */
int mcn_l2_loss(const float* w, int64_t n, float factor, float* out, void* workspace, size_t workspace_bytes,
                void* stream);
/* L1 regulariser (convnet.py:553-557: l1_factor * sum_w sum|w|): out[0] += factor * sum |w|; workspace >= 1024 floats */
int mcn_l1_loss(const float* w, int64_t n, float factor, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* ... and its gradient: g += (l1 / hyper[3]) * sign(w), launched in front of mcn_sgd_nesterov_fused_h, which scales g by
 * hyper[3] = 1 / towers (the regulariser is not a tower mean) */
int mcn_l1_grad_h(float* g, const float* w, int64_t n, float l1, const float* hyper, void* stream);

/* ---- optimizer -------------------------------------------------------------------------------
 * tf.train.MomentumOptimizer(lr, momentum, use_nesterov=True).apply_gradients (optimizers.py:676,
 * 160, 176) fused with: gradient averaging scale (optimizers.py:138), the L2 term of the loss
 * (convnet.py:563), ExponentialMovingAverage.apply on the PRE-update value (convnet.py:1401;
 * control dependency optimizers.py:159,175) and the optional decoupled decay (optimizers.py:169):
 *   ema <- d*ema + (1-d)*w                         (if ema != NULL)
 *   g   <- grad_scale*g + l2*w
 *   a   <- momentum*a + g ;  w <- w - lr*g - lr*momentum*a ;  w <- w - wd*w
 * over n contiguous fp32 elements. */
int mcn_sgd_nesterov_fused(float* w, const float* g, float* accum, float* ema, int64_t n, float lr, float momentum,
                           float l2, float wd, float ema_decay, float grad_scale, void* stream);
/* The same update with its per-step scalars read from DEVICE memory, hyper = {lr, wd, ema_decay, grad_scale} (fp32 x 4): the
 * launch arguments are then identical every step, so the whole step can be captured once in a hipGraph and replayed (the
 * reference's optimizers.py:590-594 is one session.run per step; here: one graph launch).  use_wd: 0 = no decoupled decay
 * for this range. */
int mcn_sgd_nesterov_fused_h(float* w, const float* g, float* accum, float* ema, int64_t n, const float* hyper, float momentum,
                             float l2, int32_t use_wd, void* stream);
/* the reference's other decoupled decays, applied after apply_gradients (optimizers.py:163-170; the default
 * w -= wd*w rides in mcn_sgd_nesterov_fused): mode 0: w -= wd*w; 1 (l1_weight_decay): w -= wd*sign(w);
 * 2 (huber_decay_delta): w -= wd*w/sqrt(1 + (w/delta)^2).  n contiguous fp32 elements. */
typedef enum { MCN_DECAY_L2 = 0, MCN_DECAY_L1 = 1, MCN_DECAY_HUBER = 2 } mcn_decay_mode;
int mcn_decoupled_decay(float* w, int64_t n, float wd, int32_t mode, float delta, void* stream);
int mcn_decoupled_decay_h(float* w, int64_t n, const float* hyper /* wd = hyper[1] */, int32_t mode, float delta, void* stream);
/* replaces tf.clip_by_global_norm(grads, gradient_threshold) (optimizers.py:112-113) over the flat gradient buffer:
 * first g[i] += l2 * w[i] for i < n_l2 (the gradient of the L2 term, which the reference's loss contains and which
 * otherwise rides in mcn_sgd_nesterov_fused — pass l2 = 0 there when clipping), then g *= t / max(||g||_2, t).
 * norm_out (device, may be NULL) receives the pre-clip norm.  workspace >= 1028 floats. */
int mcn_clip_by_global_norm(float* g, const float* w, int64_t n, int64_t n_l2, float l2, float threshold, float* norm_out,
                            void* workspace, size_t workspace_bytes, void* stream);
/* the same over the TRAINABLE runs of the flat buffer only (gradient clipping together with blocks_to_train: the reference
 * clips the gradients of update_vars = tf.trainable_variables(), optimizers.py:53,106,112-113 — frozen variables are in
 * neither the norm nor the L2 fold).  runs: HOST array of nruns x {start, end, l2_end} element offsets (start <= l2_end
 * <= end: [start, l2_end) receives the L2 gradient); one norm over all runs.  workspace >= (nruns * 1024 + 4) floats. */
int mcn_clip_by_global_norm_runs(float* g, const float* w, const int64_t* runs, int32_t nruns, float l2, float threshold, float* norm_out,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* shadow <- d*shadow + (1-d)*v (EMA of BN running statistics, convnet.py:1812,1826) */
int mcn_ema_update(float* shadow, const float* v, int64_t n, float decay, void* stream);
int mcn_ema_update_h(float* shadow, const float* v, int64_t n, const float* hyper /* decay = hyper[2] */, void* stream);
/* chained running-statistics update over `towers` ranks (convnet.py:1899-1909):
 * running <- m*running + (1-m)*batch[k] for k = 0..towers-1; batch:[towers][n] */
int mcn_bn_running_chain(float* running, const float* batch, int32_t towers, int64_t n, float momentum, void* stream);
/* the same over a SUB-RANGE of the statistics vector: batch row k starts at batch + k*tower_stride.  The host runs it over
 * the maximal runs of BNs that update their statistics (update_batch_norm / blocks_to_train, convnet.py:1781-1795,
 * 1915-1923: a frozen BN has no update op in the reference), so frozen running statistics are never rewritten — nor
 * touched at all while the backward pass of a frozen BN reads them. */
int mcn_bn_running_chain_strided(float* running, const float* batch, int32_t towers, int64_t n, int64_t tower_stride, float momentum,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCN_H_ */
