"""
ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

CPU (NumPy) restatement of the arithmetic of MyConvNet's conv / batch-norm / ReLU / pooling /
loss / momentum hot path.  The reference executes this arithmetic inside TensorFlow 1.14-1.15
(third-party, pinned only as "tensorflow-gpu >= 1.14.0" in the reference README.md:57, not
vendored under /root/reference and not installable here), so every function below restates the
*documented TF-1.15 op semantics* at the reference's call site, which is cited per function.
The reference ships no tests, golden vectors or fixtures for this path and
TensorFlow cannot be imported in this environment (ordinary ModuleNotFoundError), therefore the
oracle cannot be pinned against the real reference: "parity unpinned".  It is cross-checked
against torch-CPU (tests/test_oracle_vs_torch.py), which is an independent implementation of
the same published op definitions, not the oracle of record.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (myconvnet_amd) never imports it.

Conventions: activations NHWC, conv filters HWIO (reference convnet.py:1655), all math in the
dtype of the inputs (use float64 for tight checks, float32 for the timed CPU baseline).
"""
import numpy as np


# --------------------------------------------------------------------------------------------
# padding  (TF "SAME"/"VALID"; reference call sites convnet.py:1625-1628, 1493-1496)
# --------------------------------------------------------------------------------------------
def out_size(in_size, k, s, padding, d=1):
    """TF output size.  SAME: ceil(in/s).  VALID: ceil((in - (k-1)*d)/s)."""
    if padding.upper() == 'SAME':
        return -(-in_size // s)
    eff = (k - 1) * d + 1
    return -(-(in_size - eff + 1) // s)


def same_pads(in_size, k, s, d=1):
    """TF SAME padding: total = max((out-1)*s + (k-1)*d + 1 - in, 0); before = total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    before = total // 2
    return before, total - before


def resolve_pads(h, w, kh, kw, sh, sw, padding, dh=1, dw=1):
    if padding.upper() == 'SAME':
        pt, pb = same_pads(h, kh, sh, dh)
        pl, pr = same_pads(w, kw, sw, dw)
    else:
        pt = pb = pl = pr = 0
    return pt, pb, pl, pr


def _pair(v):
    if isinstance(v, (list, tuple)):
        return (v[0], v[0]) if len(v) == 1 else (v[0], v[1])
    return (v, v)


# --------------------------------------------------------------------------------------------
# conv2d  (tf.nn.conv2d at convnet.py:1659; gradients via optimizer.compute_gradients,
#          optimizers.py:106 -> Conv2DBackpropInput / Conv2DBackpropFilter)
# --------------------------------------------------------------------------------------------
def _tap_view(xp, r, s, oh, ow, sh, sw, dh, dw):
    return xp[:, r * dh: r * dh + (oh - 1) * sh + 1: sh, s * dw: s * dw + (ow - 1) * sw + 1: sw, :]


def conv2d_fwd(x, w, stride=1, padding='SAME', dilation=1):
    """y[n,oy,ox,k] = sum_{r,s,c} x[n, oy*sh + r*dh - pt, ox*sw + s*dw - pl, c] * w[r,s,c,k]
    (cross-correlation, zero padding)."""
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    n, h, wd, c = x.shape
    kh, kw, c2, k = w.shape
    assert c == c2
    pt, pb, pl, pr = resolve_pads(h, wd, kh, kw, sh, sw, padding, dh, dw)
    oh = out_size(h, kh, sh, padding, dh)
    ow = out_size(wd, kw, sw, padding, dw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, oh, ow, k), dtype=x.dtype)
    for r in range(kh):
        for s in range(kw):
            v = _tap_view(xp, r, s, oh, ow, sh, sw, dh, dw)
            y += (v.reshape(-1, c) @ w[r, s]).reshape(n, oh, ow, k)
    return y


def conv2d_dgrad(dy, w, x_shape, stride=1, padding='SAME', dilation=1):
    """dx = d(sum(dy*y))/dx."""
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    n, h, wd, c = x_shape
    kh, kw, _, k = w.shape
    pt, pb, pl, pr = resolve_pads(h, wd, kh, kw, sh, sw, padding, dh, dw)
    oh, ow = dy.shape[1:3]
    dxp = np.zeros((n, h + pt + pb, wd + pl + pr, c), dtype=dy.dtype)
    dy2 = dy.reshape(-1, k)
    for r in range(kh):
        for s in range(kw):
            v = _tap_view(dxp, r, s, oh, ow, sh, sw, dh, dw)
            v += (dy2 @ w[r, s].T).reshape(n, oh, ow, c)
    return np.ascontiguousarray(dxp[:, pt:pt + h, pl:pl + wd, :])


def conv2d_wgrad(x, dy, w_shape, stride=1, padding='SAME', dilation=1):
    """dw[r,s,c,k] = sum_{n,oy,ox} x[n, oy*sh + r*dh - pt, ox*sw + s*dw - pl, c] * dy[n,oy,ox,k]."""
    sh, sw = _pair(stride)
    dh, dw_ = _pair(dilation)
    n, h, wd, c = x.shape
    kh, kw, _, k = w_shape
    pt, pb, pl, pr = resolve_pads(h, wd, kh, kw, sh, sw, padding, dh, dw_)
    oh, ow = dy.shape[1:3]
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dwt = np.zeros(w_shape, dtype=x.dtype)
    dy2 = dy.reshape(-1, k)
    for r in range(kh):
        for s in range(kw):
            v = _tap_view(xp, r, s, oh, ow, sh, sw, dh, dw_)
            dwt[r, s] = v.reshape(-1, c).T @ dy2
    return dwt


def bias_add_fwd(x, b):
    """tf.nn.bias_add (convnet.py:1694)."""
    return x + b


def bias_add_bwd(dy):
    return dy.reshape(-1, dy.shape[-1]).sum(axis=0)


# --------------------------------------------------------------------------------------------
# depthwise conv2d  (tf.nn.depthwise_conv2d at convnet.py:1645; filter [kh,kw,cin,mult], output channel
# c*mult + m; gradients DepthwiseConv2dNativeBackpropInput / BackpropFilter) — SURVEY §8f-2
# --------------------------------------------------------------------------------------------
def depthwise_conv2d_fwd(x, w, stride=1, padding='SAME', dilation=1):
    """y[n,oy,ox,c*mult+m] = sum_{r,s} x[n, oy*sh + r*dh - pt, ox*sw + s*dw - pl, c] * w[r,s,c,m]."""
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    n, h, wd, c = x.shape
    kh, kw, c2, mult = w.shape
    assert c == c2
    pt, pb, pl, pr = resolve_pads(h, wd, kh, kw, sh, sw, padding, dh, dw)
    oh = out_size(h, kh, sh, padding, dh)
    ow = out_size(wd, kw, sw, padding, dw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, oh, ow, c, mult), dtype=x.dtype)
    for r in range(kh):
        for s in range(kw):
            v = _tap_view(xp, r, s, oh, ow, sh, sw, dh, dw)
            y += v[..., None] * w[r, s]
    return y.reshape(n, oh, ow, c * mult)


def depthwise_conv2d_dgrad(dy, w, x_shape, stride=1, padding='SAME', dilation=1):
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    n, h, wd, c = x_shape
    kh, kw, _, mult = w.shape
    pt, pb, pl, pr = resolve_pads(h, wd, kh, kw, sh, sw, padding, dh, dw)
    oh, ow = dy.shape[1:3]
    dxp = np.zeros((n, h + pt + pb, wd + pl + pr, c), dtype=dy.dtype)
    dy5 = dy.reshape(n, oh, ow, c, mult)
    for r in range(kh):
        for s in range(kw):
            v = _tap_view(dxp, r, s, oh, ow, sh, sw, dh, dw)
            v += (dy5 * w[r, s]).sum(axis=-1)
    return np.ascontiguousarray(dxp[:, pt:pt + h, pl:pl + wd, :])


def depthwise_conv2d_wgrad(x, dy, w_shape, stride=1, padding='SAME', dilation=1):
    sh, sw = _pair(stride)
    dh, dw_ = _pair(dilation)
    n, h, wd, c = x.shape
    kh, kw, _, mult = w_shape
    pt, pb, pl, pr = resolve_pads(h, wd, kh, kw, sh, sw, padding, dh, dw_)
    oh, ow = dy.shape[1:3]
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dwt = np.zeros(w_shape, dtype=x.dtype)
    dy5 = dy.reshape(n, oh, ow, c, mult)
    for r in range(kh):
        for s in range(kw):
            v = _tap_view(xp, r, s, oh, ow, sh, sw, dh, dw_)
            dwt[r, s] = (v[..., None] * dy5).sum(axis=(0, 1, 2))
    return dwt


# --------------------------------------------------------------------------------------------
# bilinear resize, channel concat  (DeepLabv3+ row, SURVEY §8f-3: tf.image.resize_bilinear at convnet.py:2396 with
# align_corners=True as called from models/deeplabv3plus.py:64,74; tf.concat at :101,110)
# --------------------------------------------------------------------------------------------
def _resize_coords(in_size, out_size, align_corners=True):
    """Source coordinate, lower index, upper index and upper weight per output index (TF ResizeBilinear:
    align_corners -> scale = (in-1)/(out-1); otherwise half_pixel_centers -> (o+0.5)*in/out - 0.5 clamped at 0)."""
    o = np.arange(out_size, dtype=np.float64)
    if align_corners:
        scale = (in_size - 1) / (out_size - 1) if out_size > 1 else 0.0
        src = o * scale
    else:
        src = np.maximum((o + 0.5) * (in_size / out_size) - 0.5, 0.0)
    lo = np.minimum(np.floor(src).astype(np.int64), in_size - 1)
    hi = np.minimum(lo + 1, in_size - 1)
    return lo, hi, src - lo


def resize_bilinear_fwd(x, out_hw, align_corners=True):
    n, h, w, c = x.shape
    oh, ow = out_hw
    ylo, yhi, fy = _resize_coords(h, oh, align_corners)
    xlo, xhi, fx = _resize_coords(w, ow, align_corners)
    fy = fy.reshape(1, oh, 1, 1).astype(x.dtype)
    fx = fx.reshape(1, 1, ow, 1).astype(x.dtype)
    top = x[:, ylo][:, :, xlo] * (1 - fx) + x[:, ylo][:, :, xhi] * fx
    bot = x[:, yhi][:, :, xlo] * (1 - fx) + x[:, yhi][:, :, xhi] * fx
    return top * (1 - fy) + bot * fy


def resize_bilinear_bwd(dy, x_shape, align_corners=True):
    n, h, w, c = x_shape
    oh, ow = dy.shape[1:3]
    ylo, yhi, fy = _resize_coords(h, oh, align_corners)
    xlo, xhi, fx = _resize_coords(w, ow, align_corners)
    dx = np.zeros(x_shape, dtype=dy.dtype)
    fy = fy.reshape(1, oh, 1, 1)
    fx = fx.reshape(1, 1, ow, 1)
    for ys, wy in ((ylo, 1 - fy), (yhi, fy)):
        for xs, wx in ((xlo, 1 - fx), (xhi, fx)):
            contrib = (dy * (wy * wx)).astype(dy.dtype)
            np.add.at(dx, (slice(None), ys[:, None], xs[None, :]), contrib)
    return dx


def concat_fwd(xs):
    return np.concatenate(xs, axis=-1)


def concat_bwd(dy, channels):
    out, o = [], 0
    for c in channels:
        out.append(np.ascontiguousarray(dy[..., o:o + c]))
        o += c
    return out


def seg_one_hot_labels(y, num_classes, dtype=np.float32):
    """SegNet labels (segmentation/segnet.py:31-50): NaN -> 0, class = round(y - 1), -1 (label 0) and out-of-range
    classes give an all-zero row = ignored pixel."""
    y = np.where(np.isnan(y), 0.0, y)
    idx = np.rint(y - 1.0).astype(np.int64)
    oh = np.zeros(y.shape + (num_classes,), dtype=dtype)
    valid = (idx >= 0) & (idx < num_classes)
    np.put_along_axis(oh, np.clip(idx, 0, num_classes - 1)[..., None], valid[..., None].astype(dtype), axis=-1)
    return oh


# --------------------------------------------------------------------------------------------
# batch norm  (tf.nn.fused_batch_norm, convnet.py:1883-1896; running stats convnet.py:1898-1914)
# --------------------------------------------------------------------------------------------
def bn_fwd_train(x, gamma, beta, eps=1e-3):
    """Returns y, batch_mean, batch_var_unbiased, save_mean, save_invstd.
    TF semantics: normalise with the *biased* batch variance, eps inside the sqrt; the returned
    batch variance is Bessel-corrected (x n/max(n-1,1))."""
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    n = x2.shape[0]
    mean = x2.mean(axis=0)
    var = ((x2 - mean) ** 2).mean(axis=0)
    invstd = 1.0 / np.sqrt(var + eps)
    y = ((x2 - mean) * (invstd * gamma) + beta).reshape(x.shape)
    var_unbiased = var * (n / max(n - 1, 1))
    return y.astype(x.dtype), mean, var_unbiased, mean, invstd


def bn_fwd_infer(x, gamma, beta, mean, var, eps=1e-3):
    """fused_batch_norm(is_training=False) (convnet.py:1889-1896, 1916-1923)."""
    invstd = 1.0 / np.sqrt(var + eps)
    return ((x - mean) * (invstd * gamma) + beta).astype(x.dtype)


def bn_bwd_frozen(dy, x, gamma, mean, var, eps=1e-3):
    """Gradient of fused_batch_norm(is_training=False) (the frozen-statistics BN of update_batch_norm=False /
    blocks_to_train, convnet.py:1915-1923): mean and variance are constants, so dx = dy*gamma*invstd,
    dgamma = sum(dy * xhat), dbeta = sum(dy)."""
    c = x.shape[-1]
    invstd = 1.0 / np.sqrt(var + eps)
    dy2 = dy.reshape(-1, c)
    xhat = (x.reshape(-1, c) - mean) * invstd
    return (dy * (gamma * invstd)).astype(x.dtype), (dy2 * xhat).sum(0), dy2.sum(0)


def bn_bwd(dy, x, gamma, save_mean, save_invstd):
    """FusedBatchNormGrad: gradient through the batch statistics."""
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    dy2 = dy.reshape(-1, c)
    n = x2.shape[0]
    xhat = (x2 - save_mean) * save_invstd
    dbeta = dy2.sum(axis=0)
    dgamma = (dy2 * xhat).sum(axis=0)
    dx = (gamma * save_invstd) * (dy2 - dbeta / n - xhat * (dgamma / n))
    return dx.reshape(x.shape).astype(x.dtype), dgamma, dbeta


def bn_running_update(mu, sigma, batch_mean, batch_var_unbiased, momentum=0.99):
    """convnet.py:1898-1901: mu <- m*mu + (1-m)*batch_mean (same for sigma = running variance)."""
    r = 1.0 - momentum
    return momentum * mu + r * batch_mean, momentum * sigma + r * batch_var_unbiased


def bn_running_update_chain(mu, sigma, batch_means, batch_vars, momentum=0.99):
    """Multi-tower chain of convnet.py:1899-1909: tower k's update starts from tower k-1's."""
    for bm, bv in zip(batch_means, batch_vars):
        mu, sigma = bn_running_update(mu, sigma, bm, bv, momentum)
    return mu, sigma


# --------------------------------------------------------------------------------------------
# activations / residual  (convnet.py:2536-2537, 2500-2512)
# --------------------------------------------------------------------------------------------
def relu_fwd(x):
    return np.maximum(x, 0)


def relu_bwd(dy, y):
    """ReluGrad: dy * [y > 0]."""
    return dy * (y > 0)


def relu6_fwd(x):
    """tf.nn.relu6 (convnet.py:2539-2540): min(max(x, 0), 6)."""
    return np.minimum(np.maximum(x, 0), 6.0)


def relu6_bwd(dy, x):
    """Relu6Grad: dy * [0 < x < 6]."""
    return dy * ((x > 0) & (x < 6))


def lrelu_fwd(x, alpha=0.2):
    """tf.nn.leaky_relu (convnet.py:2542-2545; the reference's default alpha is 0.2)."""
    return np.where(x > 0, x, alpha * x)


def lrelu_bwd(dy, x, alpha=0.2):
    """LeakyReluGrad: x > 0 ? dy : alpha * dy."""
    return np.where(x > 0, dy, alpha * dy)


def tanh_fwd(x):
    """tf.nn.tanh (convnet.py:2547)."""
    return np.tanh(x)


def tanh_bwd(dy, y):
    """TanhGrad: dy * (1 - y^2)."""
    return dy * (1.0 - y * y)


def sigmoid_fwd(x):
    """tf.nn.sigmoid (convnet.py:2550)."""
    return 1.0 / (1.0 + np.exp(-x))


def sigmoid_bwd(dy, y):
    return dy * y * (1.0 - y)


def swish_fwd(x):
    """x * sigmoid(x) (convnet.py:2553-2556)."""
    return x * sigmoid_fwd(x)


def swish_bwd(dy, x):
    """d/dx [x*s(x)] = s + x*s*(1-s)  (what TF autodiff of x*sigmoid(x) evaluates to)."""
    s = sigmoid_fwd(x)
    return dy * (s + x * s * (1.0 - s))


def channel_scale_fwd(x, m):
    """x * se_mask with the mask broadcast over H, W (models/efficientnet.py:161); m: [N,1,1,C] or [N,C]."""
    return x * m.reshape(x.shape[0], 1, 1, x.shape[-1])


def channel_scale_bwd(dy, x, m):
    """Returns dx, dm ([N,C])."""
    dx = dy * m.reshape(x.shape[0], 1, 1, x.shape[-1])
    dm = (dy * x).sum(axis=(1, 2))
    return dx, dm


def sample_scale_fwd(x, s):
    """x * survived[n] (stochastic depth, convnet.py:2503-2509); s: [N]."""
    return x * s.reshape(-1, 1, 1, 1)


def add_fwd(x, skip):
    """stochastic_depth with drop_rate == 0 (convnet.py:2511)."""
    return x + skip


# --------------------------------------------------------------------------------------------
# pooling  (tf.nn.max_pool convnet.py:1509, tf.nn.avg_pool convnet.py:1548,
#           tf.reduce_mean models/resnet_v1_5.py:73)
# --------------------------------------------------------------------------------------------
def maxpool_fwd(x, k, s, padding='SAME'):
    """Padded cells never win.  Returns y and the window-local arg-max (r*kw + s_) of the FIRST
    maximum in row-major window scan order (the TF-CPU / Eigen tie rule: strict '>' update)."""
    kh, kw = _pair(k)
    sh, sw = _pair(s)
    n, h, w, c = x.shape
    pt, pb, pl, pr = resolve_pads(h, w, kh, kw, sh, sw, padding)
    oh = out_size(h, kh, sh, padding)
    ow = out_size(w, kw, sw, padding)
    neg = np.finfo(x.dtype).min if np.issubdtype(x.dtype, np.floating) else np.iinfo(x.dtype).min
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    y = np.full((n, oh, ow, c), -np.inf, dtype=x.dtype)
    arg = np.zeros((n, oh, ow, c), dtype=np.int8)
    for r in range(kh):
        for s_ in range(kw):
            v = _tap_view(xp, r, s_, oh, ow, sh, sw, 1, 1)
            upd = v > y
            y = np.where(upd, v, y)
            arg = np.where(upd, np.int8(r * kw + s_), arg)
    del neg
    return y, arg


def maxpool_bwd(dy, arg, x_shape, k, s, padding='SAME'):
    kh, kw = _pair(k)
    sh, sw = _pair(s)
    n, h, w, c = x_shape
    pt, pb, pl, pr = resolve_pads(h, w, kh, kw, sh, sw, padding)
    oh, ow = dy.shape[1:3]
    dxp = np.zeros((n, h + pt + pb, w + pl + pr, c), dtype=dy.dtype)
    for r in range(kh):
        for s_ in range(kw):
            v = _tap_view(dxp, r, s_, oh, ow, sh, sw, 1, 1)
            v += dy * (arg == r * kw + s_)
    return np.ascontiguousarray(dxp[:, pt:pt + h, pl:pl + w, :])


def avgpool_fwd(x, k, s, padding='SAME'):
    """SAME divides by the number of valid (un-padded) cells."""
    kh, kw = _pair(k)
    sh, sw = _pair(s)
    n, h, w, c = x.shape
    pt, pb, pl, pr = resolve_pads(h, w, kh, kw, sh, sw, padding)
    oh = out_size(h, kh, sh, padding)
    ow = out_size(w, kw, sw, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    ones = np.pad(np.ones((1, h, w, 1), dtype=x.dtype), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, oh, ow, c), dtype=x.dtype)
    cnt = np.zeros((1, oh, ow, 1), dtype=x.dtype)
    for r in range(kh):
        for s_ in range(kw):
            y += _tap_view(xp, r, s_, oh, ow, sh, sw, 1, 1)
            cnt += _tap_view(ones, r, s_, oh, ow, sh, sw, 1, 1)
    return y / cnt


def avgpool_bwd(dy, x_shape, k, s, padding='SAME'):
    kh, kw = _pair(k)
    sh, sw = _pair(s)
    n, h, w, c = x_shape
    pt, pb, pl, pr = resolve_pads(h, w, kh, kw, sh, sw, padding)
    oh, ow = dy.shape[1:3]
    ones = np.pad(np.ones((1, h, w, 1), dtype=dy.dtype), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cnt = np.zeros((1, oh, ow, 1), dtype=dy.dtype)
    for r in range(kh):
        for s_ in range(kw):
            cnt += _tap_view(ones, r, s_, oh, ow, sh, sw, 1, 1)
    g = dy / cnt
    dxp = np.zeros((n, h + pt + pb, w + pl + pr, c), dtype=dy.dtype)
    for r in range(kh):
        for s_ in range(kw):
            v = _tap_view(dxp, r, s_, oh, ow, sh, sw, 1, 1)
            v += g
    return np.ascontiguousarray(dxp[:, pt:pt + h, pl:pl + w, :])


def global_avgpool_fwd(x):
    """tf.reduce_mean(x, axis=[1,2]) (models/resnet_v1_5.py:72-73)."""
    return x.mean(axis=(1, 2))


def global_avgpool_bwd(dy, x_shape):
    n, h, w, c = x_shape
    return np.broadcast_to((dy / (h * w))[:, None, None, :], x_shape).astype(dy.dtype).copy()


# --------------------------------------------------------------------------------------------
# fc  (tf.matmul(x, W) + b, convnet.py:1743)
# --------------------------------------------------------------------------------------------
def fc_fwd(x, w, b=None):
    y = x @ w
    return y if b is None else y + b


def fc_bwd(dy, x, w):
    return dy @ w.T, x.T @ dy, dy.sum(axis=0)


# --------------------------------------------------------------------------------------------
# input preparation and labels  (convnet.py:438-471)
# --------------------------------------------------------------------------------------------
def input_prep(x, image_mean=0.5, scale_factor=2.0):
    """(X - image_mean) * scale_factor (convnet.py:452, 466); zero_pad/center_crop are no-ops at
    equal size (convnet.py:714-726, 1137-1149)."""
    return (x - x.dtype.type(image_mean)) * x.dtype.type(scale_factor)


def one_hot_labels(y, num_classes, dtype=np.float32):
    """convnet.py:441-449: NaN -> -1; cast int32; one_hot (out-of-range rows are all zero)."""
    y = np.where(np.isnan(y), -1.0, y).astype(np.int32)
    oh = np.zeros((y.shape[0], num_classes), dtype=dtype)
    ok = (y >= 0) & (y < num_classes)
    oh[np.nonzero(ok)[0], y[ok]] = 1
    return oh


# --------------------------------------------------------------------------------------------
# loss  (convnet.py:528-607)
# --------------------------------------------------------------------------------------------
def softmax(logits):
    z = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def softmax_xent_fwd_bwd(logits, onehot, class_weights=None, label_smoothing=0.0, loss_scale=1.0, avg_labels=None, focal_gamma=0.0, sigmoid_focal_alpha=0.0):
    """pred = softmax(logits) (models/resnet_v1_5.py:78);
    valid = |sum(Y) - 1| < 1e-5 (convnet.py:567-573); batch_w = sum(Y * w) (convnet.py:552);
    labels = Y*(1-ls) + ls/C (convnet.py:603-607), or with `avg_labels` SegNet's Y*(1-ls) + ls*avg_pool2d(Y, 5x5, SAME)
    (segmentation/segnet.py:117-122: avg_labels = avgpool_fwd(Y, 5, 1, 'SAME') flattened like Y);
    CE_i = -sum_c labels_ic * log_softmax_ic (softmax_cross_entropy_with_logits_v2, convnet.py:600);
    focal factors (convnet.py:581-592): CE_i *= (1 - p_t)^gamma, p_t = sum_c Y_ic pred_ic, differentiated through the softmax (tf.gradients does);
    CE_i *= stop_gradient(1 - sigmoid(alpha (p_t - 0.5))) / (1 - sigmoid(-alpha / 2));
    softmax_loss = mean_i(batch_w_i * valid_i * CE_i) over ALL rows (convnet.py:594).
    Returns pred, softmax_loss, per-sample CE, dlogits (= d(loss_scale*softmax_loss)/dlogits)."""
    b, c = logits.shape
    dt = logits.dtype
    w = np.ones(c, dtype=dt) if class_weights is None else np.asarray(class_weights, dtype=dt)
    bw = (onehot * w).sum(axis=-1)
    sumy = onehot.sum(axis=-1)
    valid = ((sumy > 1.0 - 1e-5) & (sumy < 1.0 + 1e-5)).astype(dt)
    labels = onehot * (1.0 - label_smoothing) + label_smoothing / c if label_smoothing > 0 else onehot
    if label_smoothing > 0 and avg_labels is not None:
        labels = onehot * (1.0 - label_smoothing) + label_smoothing * np.asarray(avg_labels, dtype=dt)
    z = logits - logits.max(axis=-1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=-1, keepdims=True))
    logsm = z - lse
    pred = np.exp(logsm)
    ce = -(labels * logsm).sum(axis=-1)
    coef = bw * valid
    dce = pred * labels.sum(axis=-1, keepdims=True) - labels                 # d CE_i / d logits_i
    if focal_gamma > 0 or sigmoid_focal_alpha > 0:
        pt = (onehot * pred).sum(axis=-1)
        F, dF, S = np.ones_like(pt), np.zeros_like(pt), np.ones_like(pt)
        if focal_gamma > 0:
            om = np.maximum(1.0 - pt, 0.0)
            F = om ** focal_gamma
            dF = np.where(om > 0, -focal_gamma * np.maximum(om, 1e-300) ** (focal_gamma - 1.0), 0.0)
        if sigmoid_focal_alpha > 0:
            sig = lambda v: 1.0 / (1.0 + np.exp(-v))                          # noqa: E731
            S = (1.0 - sig(sigmoid_focal_alpha * (pt - 0.5))) / (1.0 - sig(-0.5 * sigmoid_focal_alpha))
        dpt = onehot * pred - pt[:, None] * pred                              # d p_t / d logits (softmax Jacobian applied to Y)
        dce = (F * S)[:, None] * dce + (ce * dF * S)[:, None] * dpt
        ce = ce * F * S
    loss = (coef * ce).mean()
    dlogits = dce * (coef * (loss_scale / b))[:, None]
    return pred.astype(dt), dt.type(loss), ce.astype(dt), dlogits.astype(dt)


def l1_reg_loss(weights, l1_factor):
    """l1_factor * sum_w sum |w| (convnet.py:553-557); its gradient is l1_factor * sign(w)."""
    return l1_factor * sum(float(np.abs(w.astype(np.float64)).sum()) for w in weights)


def l2_reg_loss(weights, l2_factor=1e-4):
    """l2_factor * sum_w tf.nn.l2_loss(w) = l2_factor * sum_w sum(w^2)/2 (convnet.py:560-563)."""
    return l2_factor * sum(float((w.astype(np.float64) ** 2).sum()) / 2.0 for w in weights)


# --------------------------------------------------------------------------------------------
# optimizer  (optimizers.py:668-677 -> TF ApplyMomentum(use_nesterov=True); EMA convnet.py:183-184)
# --------------------------------------------------------------------------------------------
def ema_decay(decay, step):
    """tf.train.ExponentialMovingAverage(decay, num_updates=step): min(decay, (1+t)/(10+t))."""
    return min(decay, (1.0 + step) / (10.0 + step))


def decoupled_decay(w, wd, l1=False, huber_delta=None):
    """The decay applied after apply_gradients (optimizers.py:163-170): w - wd*w, or w - wd*sign(w) (l1_weight_decay),
    or the pseudo-Huber form w - wd*w/sqrt(1 + (w/delta)^2) (huber_decay_delta, which wins over l1)."""
    if huber_delta is not None:
        return w - wd * w / np.sqrt(1.0 + (w / huber_delta) ** 2)
    if l1:
        return w - wd * np.sign(w)
    return w - wd * w


def sgd_nesterov_step(w, g, accum, lr, momentum=0.9, l2=0.0, ema=None, ema_d=None, wd=0.0, grad_scale=1.0, l1_decay=False, huber_delta=None):
    """One update of one tensor, in the reference's order:
      1. EMA of the PRE-update value (update_ops are control dependencies of apply_gradients,
         optimizers.py:159,175): ema <- d*ema + (1-d)*w
      2. g_total = grad_scale*g + l2*w           (l2 term of the loss, convnet.py:563)
      3. accum <- momentum*accum + g_total ; w <- w - lr*g_total - lr*momentum*accum
      4. optional decoupled decay w <- w - wd*w, or its L1 / pseudo-Huber variants  (optimizers.py:163-170)
    Returns w, accum, ema."""
    if ema is not None:
        ema = ema_d * ema + (1.0 - ema_d) * w
    gt = grad_scale * g + l2 * w
    accum = momentum * accum + gt
    w = w - lr * gt - lr * momentum * accum
    if wd > 0.0:
        w = decoupled_decay(w, wd, l1_decay, huber_delta)
    return w, accum, ema


def clip_by_global_norm(grads, threshold):
    """tf.clip_by_global_norm (optimizers.py:113): g * t / max(||g||, t) with ||g|| over ALL arrays of the dict."""
    norm = np.sqrt(sum(float((np.asarray(g, dtype=np.float64) ** 2).sum()) for g in grads.values()))
    f = threshold / max(norm, threshold)
    return {k: g * f for k, g in grads.items()}, norm


def lr_multiplier(curr_step, steps_per_epoch, num_epochs, warmup_epoch=1.0, decay_method=None,
                  decay_params=(0.94, 2), curr_epoch=1):
    """optimizers.py:608-632."""
    warmup_steps = np.around(warmup_epoch * steps_per_epoch)
    if curr_step < warmup_steps:
        return (curr_step + 1) / warmup_steps
    if decay_method is None:
        return 1.0
    m = decay_method.lower()
    if m == 'step':
        mult = 1.0
        for n in range(len(decay_params) - 1):
            mult *= np.power(decay_params[0], np.maximum(np.sign(curr_epoch - decay_params[n + 1]), 0.0))
        return mult
    if m == 'exponential':
        return decay_params[0] ** ((curr_step - warmup_steps) / steps_per_epoch / decay_params[1])
    total_steps = steps_per_epoch * num_epochs - warmup_steps
    if m in ('poly', 'polynomial'):
        power = decay_params[0] if isinstance(decay_params, (list, tuple)) else decay_params
        return (1 - (curr_step - warmup_steps) / total_steps) ** power
    anneal = decay_params[0] if isinstance(decay_params, (list, tuple)) else decay_params
    anneal = 0 if anneal is None else int(anneal)
    curr_prog = ((anneal + 1) * (curr_step - warmup_steps) / total_steps) % 1.0
    return 0.5 * (1 + np.cos(curr_prog * np.pi))


def accuracy_score(y_true, y_pred):
    """evaluators.py:86-109: per-image accuracy over the valid positions (an image without one scores 1), mean over images;
    labels / predictions one-hot `[..., C]` or class ids `[..., 1]` (id < 0 = ignored).  Pinned to the reference's own
    output by tests/golden/evaluator.npz."""
    if y_true.shape[-1] == 1:
        y_t = y_true[..., 0].astype(int)
        valid = y_t >= 0
    else:
        y_t = y_true.argmax(axis=-1)
        valid = np.isclose(y_true.sum(axis=-1), 1)
    y_p = y_pred[..., 0].astype(int) if y_pred.shape[-1] == 1 else y_pred.argmax(axis=-1)
    right = np.equal(y_t, y_p) & valid
    scores = []
    for r, v in zip(right.reshape(len(right), -1), valid.reshape(len(valid), -1)):
        nv = int(v.sum())
        scores.append(1.0 if nv == 0 else int(r.sum()) / nv)
    return float(np.mean(scores))


# ---- Winograd F(2x2, 3x3) restated (test infrastructure): the identities csrc/wino_kernels.h implements, in float64 -----------------
# Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 2x2 output tile / 4x4 input patch (Lavin & Gray 2016); the weight gradient is the transpose of the
# same bilinear map: dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G.  Not used by the reference (TensorFlow picks its algorithm inside
# cuDNN); used by tests/test_oracle_winograd.py to pin the transform matrices and the tile / padding bookkeeping against conv2d_fwd / _wgrad.
WINO_G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
WINO_BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
WINO_AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def _wino_patches(x):
    """[N, TH, TW, 4, 4, C] input patches of the 2x2 output tiles of a 3x3 / stride 1 / pad 1 convolution (zeros outside the image)"""
    n, h, w, c = x.shape
    th, tw = (h + 1) // 2, (w + 1) // 2
    xp = np.zeros((n, 2 * th + 2, 2 * tw + 2, c), np.float64)
    xp[:, 1:h + 1, 1:w + 1] = x
    out = np.empty((n, th, tw, 4, 4, c), np.float64)
    for a in range(4):
        for b in range(4):
            out[:, :, :, a, b] = xp[:, a:a + 2 * th:2, b:b + 2 * tw:2]
    return out


def winograd_conv2d_fwd(x, w):
    """3x3 / stride 1 / SAME convolution through F(2x2, 3x3); x [N,H,W,C], w [3,3,C,K] -> [N,H,W,K] (float64)"""
    n, h, wd, c = x.shape
    u = np.einsum('ar,rsck,bs->abck', WINO_G, np.asarray(w, np.float64), WINO_G)
    v = np.einsum('ar,nijrsc,bs->nijabc', WINO_BT, _wino_patches(np.asarray(x, np.float64)), WINO_BT)
    m = np.einsum('nijabc,abck->nijabk', v, u)
    y = np.einsum('pa,nijabk,qb->nijpqk', WINO_AT, m, WINO_AT)
    th, tw = y.shape[1], y.shape[2]
    return y.transpose(0, 1, 3, 2, 4, 5).reshape(n, 2 * th, 2 * tw, -1)[:, :h, :wd]


def winograd_conv2d_wgrad(x, dy):
    """weight gradient of the same convolution through F(3x3, 2x2): [3,3,C,K] (float64); outputs outside the image count as zero"""
    n, h, wd, c = x.shape
    k = dy.shape[-1]
    th, tw = (h + 1) // 2, (wd + 1) // 2
    dyp = np.zeros((n, 2 * th, 2 * tw, k), np.float64)
    dyp[:, :h, :wd] = dy
    dyt = dyp.reshape(n, th, 2, tw, 2, k).transpose(0, 1, 3, 2, 4, 5)                  # [n, th, tw, 2, 2, k]
    z = np.einsum('pa,nijpqk,qb->nijabk', WINO_AT, dyt, WINO_AT)                      # A dY A^T  (A = WINO_AT^T)
    v = np.einsum('ar,nijrsc,bs->nijabc', WINO_BT, _wino_patches(np.asarray(x, np.float64)), WINO_BT)
    du = np.einsum('nijabc,nijabk->abck', v, z)
    return np.einsum('ar,abck,bs->rsck', WINO_G, du, WINO_G)
