"""GPU parity, whole network, segmentation row: DeepLabv3+ on the dilated ResNet-50 backbone (dilated /
multi-grid 3x3 convs, ASPP, align_corners bilinear upsampling, channel concat, per-pixel cross-entropy with ignored
pixels) — HIP kernels through the C-ABI against the oracle's stand-alone restatement of models/deeplabv3plus.py +
models/resnet_v1_5_dilated.py + segmentation/segnet.py.  Tolerances as in test_gpu_net.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import net as ON  # noqa: E402
from oracle import ops as O   # noqa: E402
from test_gpu_net import rel_l2, cosine, bf16q  # noqa: E402
from test_gpu_efficientnet import worst_grad  # noqa: E402  (BN -> conv -> BN chains without activation: exact-zero dbeta)

B, SIZE, CLASSES = 4, 97, 6


def make(dtype, fuse=True, spec_kw=None, **kw):
    import myconvnet_amd as M
    spec = ON.DeepLabSpec(CLASSES, width_div=8, depth_div=3, aspp_dilations=(1, 2, 3), **(spec_kw or {}))
    model = M.DeepLabV3PlusResNet50([SIZE, SIZE, 3], CLASSES, batch_size=B, width_div=8, depth_div=3, aspp_dilations=[1, 2, 3], fuse=fuse,
                                    half_precision=(dtype == 'bfloat16'), num_gpus=1, **kw)
    params, stats = ON.init_variables(spec.variables(), seed=6, dtype=np.float32)
    rng = np.random.default_rng(12)
    for k in params:
        if k.endswith('gamma'):
            lo, span = (0.1, 0.2) if 'conv_2/bn' in k else (0.8, 0.4)
            params[k] = (lo + span * rng.random(params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('biases'):
            params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    model.set_variables(dict(params, **stats))
    return model, spec, params, stats


def batch(rng):
    x = rng.random((B, SIZE, SIZE, 3)).astype(np.float32)
    y = rng.integers(0, CLASSES + 1, (B, SIZE, SIZE)).astype(np.float32)      # 0 = ignore
    y[0, :3, :] = np.nan                                                      # NaN labels are ignored too
    return x, y


def f64(d):
    return {k: v.astype(np.float64) for k, v in d.items()}


@pytest.mark.parametrize('fuse', [True, False])
def test_deeplab_two_steps_fp32(fuse):
    import myconvnet_amd as M
    rng = np.random.default_rng(51)
    model, spec, params, stats = make('float32', fuse)
    assert set(model.variables) == set(params) | set(stats)
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state = ON.TrainState(f64(params), f64(stats))
    for step in range(2):
        x, y = batch(rng)
        model.feed(x, y)
        loss, y_true, y_pred = opt._step(None)
        rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B)
        assert y_pred.shape == (B, SIZE, SIZE, CLASSES)
        assert abs(loss - rloss) <= 1e-4 * abs(rloss), (loss, rloss)
        assert rel_l2(y_pred, rpred) <= 1e-4
        mism = (y_pred.argmax(-1) != rpred.argmax(-1)).mean()
        assert mism <= 1e-4, 'arg-max mismatch on {:.2e} of the pixels'.format(mism)        # ties within fp32 resolution only
        np.testing.assert_array_equal(y_true, O.seg_one_hot_labels(y, CLASSES))
        grads = model.get_variables('grad')
        worst = worst_grad(grads, rgrads)
        assert worst[0] <= 1e-3, 'step {}: worst gradient {}'.format(step, worst)
        got = model.get_variables('data')
        worst = max((rel_l2(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert worst[0] <= 1e-4, 'step {}: worst variable {}'.format(step, worst)


def test_deeplab_label_smoothing_is_the_5x5_average_of_the_label_map():
    """SegNet._label_smoothing (segmentation/segnet.py:117-122): labels <- (1 - f) * Y + f * avg_pool2d(Y, 5x5, 1, SAME); batch weights and
    the valid mask keep reading Y (convnet.py:552, 567-573).  One mcn_avgpool_fwd over the one-hot map + mcn_softmax_xent_rows_soft_fwd_bwd."""
    import myconvnet_amd as M
    rng = np.random.default_rng(54)
    model, spec, params, stats = make('float32', label_smoothing=0.2)
    fns = [getattr(fn, '__name__', '') for fn, _ in model._train_low.fwd.calls]
    assert 'mcn_softmax_xent_rows_focal_fwd_bwd' in fns and 'mcn_avgpool_fwd' in fns
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state, plain = ON.TrainState(f64(params), f64(stats)), ON.TrainState(f64(params), f64(stats))
    x, y = batch(rng)
    y[1, 40:60, 40:60] = 3.0                                                  # a uniform region (average == label) next to noisy labels
    model.feed(x, y)
    loss, y_true, y_pred = opt._step(None)
    rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B, hp=dict(label_smoothing=0.2))
    ploss, _, pgrads = ON.train_step(spec, plain, x.astype(np.float64), y.astype(np.float64), batch_total=B)
    assert abs(rloss - ploss) > 1e-2 * abs(ploss)                             # the smoothing is visible at this size
    assert abs(loss - rloss) <= 1e-4 * abs(rloss), (loss, rloss)
    assert rel_l2(y_pred, rpred) <= 1e-4
    np.testing.assert_array_equal(y_true, O.seg_one_hot_labels(y, CLASSES))   # Y itself stays the raw one-hot map
    worst = worst_grad(model.get_variables('grad'), rgrads)
    assert worst[0] <= 1e-3, 'worst gradient {}'.format(worst)


def test_deeplab_image_level_feature_and_stop_gradient():
    """The reference's other branch of two head switches (off in its defaults): the ASPP image-level feature — global mean -> 1x1 conv -> norm ->
    resize back, concatenated as a fifth ASPP branch (models/deeplabv3plus.py:90-99) — and tf.stop_gradient on the low-level backbone feature
    (feature_gradients[i] = False, :50-53): block_1 then receives its gradient through the deeper blocks only."""
    import myconvnet_amd as M
    rng = np.random.default_rng(55)
    head = dict(aspp_level_feature=True, feature_gradients=[None, False])
    model, spec, params, stats = make('float32', spec_kw=head, **head)
    assert 'block_5/aspp/conv_pool/weights' in params and any(n.op == 'stopgrad' for n in model.graph.nodes)
    sg = [n for n in model.graph.nodes if n.op == 'stopgrad'][0]
    assert sg.outputs[0].buf.data_ptr() == sg.inputs[0].buf.data_ptr() and not sg.outputs[0].needs_grad      # shared storage, no gradient path
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state, plain = ON.TrainState(f64(params), f64(stats)), ON.TrainState(f64(params), f64(stats))
    x, y = batch(rng)
    model.feed(x, y)
    loss, _, y_pred = opt._step(None)
    rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B)
    assert abs(loss - rloss) <= 1e-4 * abs(rloss), (loss, rloss)
    assert rel_l2(y_pred, rpred) <= 1e-4
    worst = worst_grad(model.get_variables('grad'), rgrads)
    assert worst[0] <= 2e-3, 'worst gradient {}'.format(worst)               # (the image-level branch normalises over M = B = 4 rows)
    # the stop is visible: with the gradient flowing (same weights) block_1's gradients differ
    spec2 = ON.DeepLabSpec(CLASSES, width_div=8, depth_div=3, aspp_dilations=(1, 2, 3), aspp_level_feature=True)
    _, _, g2 = ON.train_step(spec2, plain, x.astype(np.float64), y.astype(np.float64), batch_total=B)
    k = 'block_1/res_0/conv_0/weights'
    assert rel_l2(g2[k], rgrads[k]) > 1e-3


def test_deeplab_eval_on_ema():
    import myconvnet_amd as M
    rng = np.random.default_rng(52)
    model, spec, params, stats = make('float32')
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state = ON.TrainState(f64(params), f64(stats))
    x, y = batch(rng)
    model.feed(x, y)
    opt._step(None)
    ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B)
    ds = M.DataSet(x, y, batch_size=B)
    _, y_true, y_pred, loss = model.predict(ds, return_images=False)
    _, _, pred, _, _ = ON.forward_loss(spec, state, x.astype(np.float64), y.astype(np.float64), train=False, use_ema=True)
    assert rel_l2(y_pred, pred) <= 1e-4
    assert (y_pred.argmax(-1) != pred.argmax(-1)).mean() <= 1e-4


def test_deeplab_step_bf16():
    import myconvnet_amd as M
    rng = np.random.default_rng(53)
    model, spec, params, stats = make('bfloat16')
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state, state64 = ON.TrainState(f64(params), f64(stats)), ON.TrainState(f64(params), f64(stats))
    x, y = batch(rng)
    model.feed(x, y)
    loss, _, y_pred = opt._step(None)
    rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B, quant=bf16q)
    xloss, xpred, xgrads = ON.train_step(spec, state64, x.astype(np.float64), y.astype(np.float64), batch_total=B)
    assert abs(loss - rloss) <= 2e-2 * abs(rloss), (loss, rloss)
    assert rel_l2(y_pred, xpred) <= 1.5 * rel_l2(rpred, xpred) + 0.02
    grads = model.get_variables('grad')
    keys = sorted(k for k in rgrads if k.endswith('weights'))
    g = np.concatenate([grads[k].ravel() for k in keys])
    r = np.concatenate([rgrads[k].ravel() for k in keys])
    e = np.concatenate([xgrads[k].ravel() for k in keys])
    err_dev, err_emu = rel_l2(g, e), rel_l2(r, e)
    print('bf16 gradient error vs float64: device {:.3f}, bf16-emulating oracle {:.3f}; cos {:.3f}'.format(err_dev, err_emu, cosine(g, r)))
    assert err_dev <= 1.5 * err_emu + 0.02, (err_dev, err_emu)
    assert cosine(g, e) >= 0.9
