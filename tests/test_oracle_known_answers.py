"""Known-answer tests of the oracle: cases small enough to be worked out by hand from the documented semantics of the TensorFlow ops the
reference calls (each test cites the reference line whose op it pins).  The reference ships no fixtures of its own for the floating-point
path and TensorFlow cannot run in this image, so these do not make the oracle "pinned" — they fix the conventions a restatement
most easily gets wrong (SAME padding on even sizes and strides, which variance goes where in batch norm, the order of the Nesterov
update, the first-maximum rule of the pooling arg-max, what the loss is averaged over) against answers that do not come from the oracle or
from torch."""
import math

import numpy as np
import pytest

from oracle import ops


# ---- padding and convolution (tf.nn.conv2d, padding='SAME': convnet.py:1659) -------------------------------------------------------------
def test_same_padding_puts_the_odd_cell_at_the_end():
    # TF: out = ceil(in / s), total = max((out - 1) s + k - in, 0), before = total // 2 (the extra cell goes to the bottom / right)
    assert ops.same_pads(4, 3, 2) == (0, 1)          # 4 -> 2 outputs: windows [0,1,2], [2,3,pad]
    assert ops.same_pads(5, 3, 2) == (1, 1)
    assert ops.same_pads(224, 7, 2) == (2, 3)        # the ResNet stem (models/resnet_v1_5.py:42)
    assert ops.same_pads(112, 3, 2) == (0, 1)        # its max-pool (models/resnet_v1_5.py:50)
    assert ops.same_pads(7, 3, 1, d=2) == (2, 2)     # dilated 3x3 (models/resnet_v1_5_dilated.py)
    assert ops.out_size(4, 3, 2, 'SAME') == 2 and ops.out_size(4, 3, 2, 'VALID') == 1


def test_conv_of_ones_counts_the_taps_inside_the_image():
    x = np.ones((1, 4, 4, 1), np.float64)
    w = np.ones((3, 3, 1, 1), np.float64)
    y = ops.conv2d_fwd(x, w, 1, 'SAME')[0, :, :, 0]
    np.testing.assert_array_equal(y, [[4, 6, 6, 4], [6, 9, 9, 6], [6, 9, 9, 6], [4, 6, 6, 4]])
    # stride 2 on an even size: windows start at 0 and 2, only the bottom / right one hangs over the edge
    y2 = ops.conv2d_fwd(x, w, 2, 'SAME')[0, :, :, 0]
    np.testing.assert_array_equal(y2, [[9, 6], [6, 4]])


def test_conv_is_a_cross_correlation_and_its_gradients_are_its_transposes():
    x = np.zeros((1, 3, 3, 1))
    x[0, 1, 1, 0] = 1.0                               # an impulse returns the FLIPPED filter under cross-correlation
    w = np.arange(9, dtype=np.float64).reshape(3, 3, 1, 1)
    y = ops.conv2d_fwd(x, w, 1, 'SAME')[0, :, :, 0]
    np.testing.assert_array_equal(y, w[::-1, ::-1, 0, 0])
    # <dy, conv(x, w)> is linear in x and in w: dgrad / wgrad are the two adjoints
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 4, 3))
    w = rng.standard_normal((3, 3, 3, 2))
    dy = rng.standard_normal((2, 3, 2, 2))
    lhs = float((dy * ops.conv2d_fwd(x, w, 2, 'SAME')).sum())
    assert math.isclose(lhs, float((ops.conv2d_dgrad(dy, w, x.shape, 2, 'SAME') * x).sum()), rel_tol=1e-12)
    assert math.isclose(lhs, float((ops.conv2d_wgrad(x, dy, w.shape, 2, 'SAME') * w).sum()), rel_tol=1e-12)


def test_depthwise_conv_is_one_filter_per_channel():
    # tf.nn.depthwise_conv2d with channel multiplier 1 (convnet.py:1634-1650): channel c sees only filter w[:, :, c, 0]
    x = np.ones((1, 3, 3, 2))
    w = np.zeros((3, 3, 2, 1))
    w[:, :, 0, 0] = 1.0
    w[1, 1, 1, 0] = 5.0
    y = ops.depthwise_conv2d_fwd(x, w, 1, 'SAME')
    np.testing.assert_array_equal(y[0, :, :, 0], [[4, 6, 4], [6, 9, 6], [4, 6, 4]])
    np.testing.assert_array_equal(y[0, :, :, 1], np.full((3, 3), 5.0))


# ---- pooling (tf.nn.max_pool / avg_pool: convnet.py:1472-1530) ------------------------------------------------------------------------------
def test_max_pool_same_ignores_padding_and_keeps_the_first_maximum():
    x = -np.ones((1, 4, 4, 1))                       # all negative: a zero pad cell must not win
    y, arg = ops.maxpool_fwd(x, 3, 2, 'SAME')
    np.testing.assert_array_equal(y[0, :, :, 0], [[-1, -1], [-1, -1]])
    np.testing.assert_array_equal(arg[0, :, :, 0], [[0, 0], [0, 0]])      # ties: the first cell of the row-major window scan
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    y, arg = ops.maxpool_fwd(x, 3, 2, 'SAME')
    np.testing.assert_array_equal(y[0, :, :, 0], [[10, 11], [14, 15]])
    np.testing.assert_array_equal(arg[0, :, :, 0], [[8, 7], [5, 4]])      # window-local r * 3 + s of cells (2,2), (2,3), (3,2), (3,3)
    dx = ops.maxpool_bwd(np.ones((1, 2, 2, 1)), arg, x.shape, 3, 2, 'SAME')
    want = np.zeros((4, 4))
    want[2, 2] = want[2, 3] = want[3, 2] = want[3, 3] = 1.0
    np.testing.assert_array_equal(dx[0, :, :, 0], want)


def test_avg_pool_same_divides_by_the_cells_inside_the_image():
    x = np.ones((1, 3, 3, 1))
    y = ops.avgpool_fwd(x, 3, 1, 'SAME')
    np.testing.assert_allclose(y[0, :, :, 0], np.ones((3, 3)), rtol=0, atol=1e-15)    # TF excludes the padding from the count
    np.testing.assert_allclose(ops.global_avgpool_fwd(np.arange(8.0).reshape(1, 2, 2, 2)).reshape(-1), [3.0, 4.0])


# ---- batch norm (tf.nn.fused_batch_norm, convnet.py:1883-1901) -----------------------------------------------------------------------------
def test_batch_norm_uses_the_biased_variance_inside_and_returns_the_unbiased_one():
    x = np.array([1.0, 3.0]).reshape(2, 1, 1, 1)     # mean 2, biased variance 1, unbiased 2
    y, bm, bv, sm, si = ops.bn_fwd_train(x, np.array([2.0]), np.array([0.5]), eps=0.0)
    np.testing.assert_allclose(y.reshape(-1), [-1.5, 2.5])
    assert bm[0] == 2.0 and bv[0] == 2.0 and sm[0] == 2.0 and si[0] == 1.0
    eps = 1e-3
    y, _, _, _, si = ops.bn_fwd_train(x, np.array([1.0]), np.array([0.0]), eps=eps)
    assert math.isclose(si[0], 1.0 / math.sqrt(1.0 + eps), rel_tol=1e-15)              # eps under the root
    mu, sigma = ops.bn_running_update(np.array([10.0]), np.array([1.0]), bm, bv, momentum=0.9)
    np.testing.assert_allclose([mu[0], sigma[0]], [0.9 * 10 + 0.1 * 2, 0.9 * 1 + 0.1 * 2])   # the Bessel-corrected variance is what moves the average


def test_batch_norm_gradient_is_orthogonal_to_constants_and_to_xhat():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 3, 3, 2))
    dy = rng.standard_normal(x.shape)
    g = np.array([1.5, -0.5])
    _, _, _, sm, si = ops.bn_fwd_train(x, g, np.zeros(2), eps=1e-3)
    dx, dg, db = ops.bn_bwd(dy, x, g, sm, si)
    xhat = (x - sm) * si
    np.testing.assert_allclose(dx.reshape(-1, 2).sum(0), 0.0, atol=1e-12)            # y does not change when x is shifted
    np.testing.assert_allclose(db, dy.reshape(-1, 2).sum(0))
    np.testing.assert_allclose(dg, (dy * xhat).reshape(-1, 2).sum(0))
    # and against the central difference of the forward
    h = 1e-6
    e = np.zeros_like(x)
    e[1, 2, 0, 1] = h
    f = lambda v: float((dy * ops.bn_fwd_train(v, g, np.zeros(2), eps=1e-3)[0]).sum())      # noqa: E731
    assert math.isclose((f(x + e) - f(x - e)) / (2 * h), dx[1, 2, 0, 1], rel_tol=1e-6)


# ---- activations (convnet.py:2536-2553) --------------------------------------------------------------------------------------------------
def test_activation_values_at_hand_points():
    assert ops.swish_fwd(np.array([0.0]))[0] == 0.0
    assert math.isclose(ops.swish_bwd(np.array([1.0]), np.array([0.0]))[0], 0.5)              # d/dx x sigmoid(x) at 0
    assert math.isclose(ops.swish_fwd(np.array([1.0]))[0], 1.0 / (1.0 + math.exp(-1.0)))
    np.testing.assert_array_equal(ops.relu6_fwd(np.array([-1.0, 3.0, 7.0])), [0.0, 3.0, 6.0])
    np.testing.assert_array_equal(ops.relu6_bwd(np.ones(3), np.array([-1.0, 3.0, 7.0])), [0.0, 1.0, 0.0])
    np.testing.assert_array_equal(ops.relu_bwd(np.ones(3), ops.relu_fwd(np.array([-2.0, 0.0, 2.0]))), [0.0, 0.0, 1.0])   # gradient 0 AT 0


# ---- loss (convnet.py:552-607) -----------------------------------------------------------------------------------------------------------
def test_softmax_loss_of_uniform_logits_is_log_c_and_its_gradient_is_p_minus_y_over_b():
    c, b = 5, 4
    logits = np.zeros((b, c))
    y = np.eye(c)[[0, 1, 2, 3]]
    pred, loss, ce, dl = ops.softmax_xent_fwd_bwd(logits, y)
    np.testing.assert_allclose(pred, 1.0 / c)
    assert math.isclose(float(loss), math.log(c), rel_tol=1e-12)
    np.testing.assert_allclose(dl, (pred - y) / b)
    # label smoothing: labels = y (1 - ls) + ls / C; on uniform logits the loss is still log C
    _, loss_s, _, dl_s = ops.softmax_xent_fwd_bwd(logits, y, label_smoothing=0.1)
    assert math.isclose(float(loss_s), math.log(c), rel_tol=1e-12)
    np.testing.assert_allclose(dl_s, (pred - (y * 0.9 + 0.1 / c)) / b)


def test_rows_without_exactly_one_label_are_dropped_but_still_counted_in_the_mean():
    logits = np.zeros((2, 4))
    y = np.zeros((2, 4))
    y[0, 1] = 1.0                                     # row 1 has no label: valid = 0 (convnet.py:567-573), the mean runs over both rows (594)
    _, loss, _, dl = ops.softmax_xent_fwd_bwd(logits, y)
    assert math.isclose(float(loss), math.log(4) / 2, rel_tol=1e-12)
    np.testing.assert_array_equal(dl[1], np.zeros(4))
    w = np.array([1.0, 3.0, 1.0, 1.0])                 # class weights scale the sample's loss by the weight of ITS label (552)
    _, loss_w, _, _ = ops.softmax_xent_fwd_bwd(logits, y, class_weights=w)
    assert math.isclose(float(loss_w), 3 * math.log(4) / 2, rel_tol=1e-12)


def test_regularisers():
    w = [np.array([3.0, -4.0]), np.array([[1.0]])]
    assert math.isclose(ops.l2_reg_loss(w, 0.5), 0.5 * (9 + 16 + 1) / 2)               # tf.nn.l2_loss = sum(w^2) / 2 (convnet.py:560-563)
    assert math.isclose(ops.l1_reg_loss(w, 0.5), 0.5 * 8)
    g, norm = ops.clip_by_global_norm({'a': np.array([3.0]), 'b': np.array([4.0])}, 1.0)   # optimizers.py:113
    assert norm == 5.0 and math.isclose(g['a'][0], 0.6) and math.isclose(g['b'][0], 0.8)
    g, _ = ops.clip_by_global_norm({'a': np.array([3.0]), 'b': np.array([4.0])}, 10.0)
    assert g['a'][0] == 3.0                           # below the threshold: untouched


# ---- optimizer (optimizers.py:668-677: tf.train.MomentumOptimizer(use_nesterov=True); EMA convnet.py:183-184) ---------------------------
def test_nesterov_update_by_hand():
    # ApplyMomentum(use_nesterov): accum = m accum + g; var -= lr g + lr m accum
    w, acc, ema = ops.sgd_nesterov_step(np.array([1.0]), np.array([0.5]), np.array([0.2]), lr=0.1, momentum=0.9, l2=0.0, ema=np.array([2.0]), ema_d=0.5)
    assert math.isclose(acc[0], 0.9 * 0.2 + 0.5)
    assert math.isclose(w[0], 1.0 - 0.1 * 0.5 - 0.1 * 0.9 * (0.9 * 0.2 + 0.5))
    assert math.isclose(ema[0], 0.5 * 2.0 + 0.5 * 1.0)                                 # the shadow averages the value BEFORE the update (optimizers.py:159,175)
    w2, _, _ = ops.sgd_nesterov_step(np.array([1.0]), np.array([0.0]), np.array([0.0]), lr=0.1, momentum=0.9, l2=0.01)
    assert math.isclose(w2[0], 1.0 - 0.1 * 0.01 - 0.1 * 0.9 * 0.01)                   # the L2 term enters as a gradient, through the momentum
    w3, _, _ = ops.sgd_nesterov_step(np.array([2.0]), np.array([0.0]), np.array([0.0]), lr=0.1, momentum=0.9, wd=0.25)
    assert math.isclose(w3[0], 2.0 * 0.75)                                             # decoupled decay after the step (optimizers.py:163-170)


def test_ema_decay_warm_up_and_learning_rate_schedule():
    assert ops.ema_decay(0.999, 0) == 0.1 and ops.ema_decay(0.999, 90) == 0.91        # min(decay, (1 + t) / (10 + t))
    assert ops.ema_decay(0.999, 10 ** 6) == 0.999
    # optimizers.py:608-632: linear warm-up over warmup_epoch epochs, then the schedule on the remaining steps
    assert ops.lr_multiplier(0, 100, 10, warmup_epoch=1.0) == 1.0 / 100
    assert ops.lr_multiplier(49, 100, 10, warmup_epoch=1.0) == 0.5
    assert ops.lr_multiplier(100, 100, 10, warmup_epoch=1.0, decay_method='cosine', decay_params=(0,)) == 1.0
    assert math.isclose(ops.lr_multiplier(550, 100, 10, warmup_epoch=1.0, decay_method='cosine', decay_params=(0,)), 0.5, abs_tol=1e-12)
    assert math.isclose(ops.lr_multiplier(550, 100, 10, warmup_epoch=1.0, decay_method='poly', decay_params=(2,)), 0.25, abs_tol=1e-12)
    assert math.isclose(ops.lr_multiplier(300, 100, 10, warmup_epoch=1.0, decay_method='exponential', decay_params=(0.5, 2)), 0.5, abs_tol=1e-12)


# ---- input pipeline boundary (convnet.py:452, 466) and resize (tf.image.resize_bilinear(align_corners=True), convnet.py:2378) -----------
def test_input_prep_and_one_hot():
    np.testing.assert_allclose(ops.input_prep(np.array([0.0, 0.5, 1.0])), [-1.0, 0.0, 1.0])      # (x - 0.5) * 2
    oh = ops.one_hot_labels(np.array([2, 0]), 3)
    np.testing.assert_array_equal(oh, [[0, 0, 1], [1, 0, 0]])


def test_bilinear_resize_align_corners_hits_the_corners_and_midpoints():
    x = np.array([[0.0, 2.0], [4.0, 6.0]]).reshape(1, 2, 2, 1)
    y = ops.resize_bilinear_fwd(x, (3, 3), align_corners=True)[0, :, :, 0]
    np.testing.assert_allclose(y, [[0, 1, 2], [2, 3, 4], [4, 5, 6]])
    dx = ops.resize_bilinear_bwd(np.ones((1, 3, 3, 1)), x.shape, align_corners=True)[0, :, :, 0]
    np.testing.assert_allclose(dx, np.full((2, 2), 9 / 4))                             # the adjoint spreads 9 unit gradients evenly over 4 corners


@pytest.mark.parametrize('h,k,s', [(7, 3, 2), (8, 3, 2), (9, 5, 2), (6, 1, 2), (5, 3, 1)])
def test_output_size_is_ceil_of_in_over_stride(h, k, s):
    x = np.ones((1, h, h, 1))
    assert ops.conv2d_fwd(x, np.ones((k, k, 1, 1)), s, 'SAME').shape[1] == -(-h // s)
    assert ops.maxpool_fwd(x, k, s, 'SAME')[0].shape[1] == -(-h // s)
