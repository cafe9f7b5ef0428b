"""GPU parity tests, whole network: the product's ResNet / VGG graphs (HIP kernels through the C-ABI) against the
oracle's stand-alone NumPy restatement of the same reference model files, same injected weights, same batch.

fp32 tolerance: north_star's 1e-3 relative (asserted tighter where the network is well conditioned); arg-max of the
predictions bit-exact.  bf16 is checked against the float64 oracle with bf16-sized tolerances.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import os     # noqa: E402
import sys    # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from flip_util import assert_flip_budget, flip_aware_step  # noqa: E402
from oracle import net as ON  # noqa: E402
from oracle import ops as O   # noqa: E402

RNG = np.random.default_rng(5)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


BATCH = 8
LABELS = np.array([1, 7, 3, 9, 0, 4, 4, 2], dtype=np.float32)


def make_resnet(kind, dtype, fuse, batch=BATCH, size=64, classes=10, **kw):
    import myconvnet_amd as M
    cls = M.ResNet50 if kind == 50 else M.ResNet18
    spec = ON.ResNetSpec.resnet50(classes, 8) if kind == 50 else ON.ResNetSpec.resnet18(classes, 8)
    model = cls([size, size, 3], classes, batch_size=batch, width_div=8, fuse=fuse, half_precision=(dtype != 'float32'),
                half_precision_dtype=(dtype if dtype != 'float32' else 'bfloat16'), num_gpus=1, **kw)
    params, stats = ON.init_variables(spec.variables(), seed=3, dtype=np.float32)
    rng = np.random.default_rng(9)
    # Non-trivial BN parameters (zero-init gammas would hide the residual branches) chosen so that the tiny test network
    # is well conditioned: with batch 8 a float32 evaluation of the oracle itself stays within 3e-5 of float64 on every
    # gradient tensor (large residual gammas make the 16-unit net chaotic: even NumPy-fp32 vs NumPy-fp64 then differ by >1e-2).
    for k in params:
        if k.endswith('gamma'):
            lo, span = (0.1, 0.2) if ('conv_2/bn' in k or (kind == 18 and 'conv_1/bn' in k)) else (0.8, 0.4)
            params[k] = (lo + span * rng.random(params[k].shape)).astype(np.float32)
        if k.endswith('beta'):
            params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    model.set_variables(dict(params, **stats))
    return model, spec, params, stats


@pytest.mark.parametrize('kind,fuse', [(50, True), (50, False), (18, True)])
def test_resnet_two_steps_fp32(kind, fuse):
    import myconvnet_amd as M
    model, spec, params, stats = make_resnet(kind, 'float32', fuse)
    assert set(model.variables) == set(params) | set(stats)
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    for step in range(2):
        x = RNG.random((BATCH, 64, 64, 3)).astype(np.float32)
        y = LABELS
        model.feed(x, y)
        loss, y_true, y_pred = opt._step(None)
        rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=BATCH)
        assert abs(loss - rloss) <= 1e-4 * abs(rloss), (loss, rloss)
        assert rel_l2(y_pred, rpred) <= 1e-4
        np.testing.assert_array_equal(y_pred.argmax(-1), rpred.argmax(-1))
        np.testing.assert_array_equal(y_true, O.one_hot_labels(y, 10))
        grads = model.get_variables('grad')
        worst = max((rel_l2(grads[k], rgrads[k]), k) for k in rgrads)
        assert worst[0] <= 1e-3, 'step {}: worst gradient {}'.format(step, worst)
        got = model.get_variables('data')
        worst = max((rel_l2(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert worst[0] <= 1e-4, 'step {}: worst variable {}'.format(step, worst)
        ema = model.get_variables('ema')
        worst = max((rel_l2(ema[k], v), k) for k, v in list(state.ema.items()) + list(state.ema_stats.items()))
        assert worst[0] <= 1e-4, 'step {}: worst EMA {}'.format(step, worst)


@pytest.mark.parametrize('variant', [dict(base_weight_decay=0.3), dict(base_weight_decay=0.3, l1_weight_decay=True),
                                     dict(base_weight_decay=0.3, huber_decay_delta=0.05),
                                     dict(base_weight_decay=0.3, huber_decay_delta=0.05, bias_norm_decay=True, weight_decay_scheduling=False),
                                     dict(base_weight_decay=0.3, bias_norm_decay=True, gradient_threshold=0.5)])
def test_resnet_decoupled_decay_variants(variant):
    """optimizers.py:149-173: w -= wd*w / wd*sign(w) / pseudo-Huber after the update, wd = base*B/256 [* lr multiplier];
    bias_norm_decay widens the L2 term (convnet.py:536) and the decay to biases and BN gamma / beta."""
    import myconvnet_amd as M
    rng = np.random.default_rng(77)
    mk = {k: v for k, v in variant.items() if k == 'bias_norm_decay'}
    model, spec, params, stats = make_resnet(18, 'float32', True, **mk)
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=4, learning_warmup_epochs=1.0, **variant)
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    hp = dict(variant)
    for step in range(3):
        x = rng.random((BATCH, 64, 64, 3)).astype(np.float32)
        model.feed(x, LABELS)
        opt.curr_step = step
        opt._update_learning_rate()                                          # the train loop's per-step call (optimizers.py:408)
        loss, _, y_pred = opt._step(None)
        mult = O.lr_multiplier(step, 4, opt.num_epochs, warmup_epoch=1.0)
        assert abs(opt.curr_multiplier - mult) <= 1e-12
        rloss, rpred, _ = ON.train_step(spec, state, x.astype(np.float64), LABELS.astype(np.float64), hp=hp, lr_mult=mult, batch_total=BATCH)
        assert abs(loss - rloss) <= 1e-4 * abs(rloss), (step, loss, rloss)
        got = model.get_variables('data')
        worst = max((rel_l2(got[k], v), k) for k, v in state.params.items())
        assert worst[0] <= 1e-4, 'step {}: worst variable {}'.format(step, worst)
    # the decay was visible at this size: the oracle's own run without it ends somewhere else
    plain = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    rng = np.random.default_rng(77)
    for step in range(3):
        x = rng.random((BATCH, 64, 64, 3))
        ON.train_step(spec, plain, x.astype(np.float32).astype(np.float64), LABELS.astype(np.float64),
                      hp={k: v for k, v in hp.items() if k == 'gradient_threshold'}, lr_mult=O.lr_multiplier(step, 4, 100, warmup_epoch=1.0), batch_total=BATCH)
    assert max(rel_l2(state.params[k], plain.params[k]) for k in params if k.endswith('/weights')) > 1e-3


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('freeze', [dict(blocks_to_train=[4, None]),            # fine-tune the last stage + logits: no gradient below
                                    dict(blocks_to_train=[None]),               # logits only
                                    dict(blocks_to_train=[0, None]),            # gradient flows back THROUGH frozen blocks
                                    dict(update_batch_norm=False),              # everything trains, statistics frozen
                                    dict(blocks_to_train=[1, 2], update_batch_norm=True)])
def test_resnet_frozen_blocks_and_frozen_batch_norm(freeze, dtype):
    """blocks_to_train / update_batch_norm (convnet.py:1384-1389, 1781-1795; optimizers.py:53): frozen variables get no
    gradient, no update and no decay but keep their EMA shadow moving; a BN outside the trained blocks normalises with
    its running statistics also in training mode and back-propagates as an affine map (mcn_bn_bwd_frozen)."""
    import myconvnet_amd as M
    rng = np.random.default_rng(123)
    model, spec, params, stats = make_resnet(18, dtype, True, **freeze)
    for k in stats:                                                           # non-trivial running statistics
        stats[k] = (0.2 * rng.standard_normal(stats[k].shape) if k.endswith('/mu') else 0.5 + rng.random(stats[k].shape)).astype(np.float32)
    model.set_variables(dict(params, **stats))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0, base_weight_decay=0.01)
    fns = [getattr(fn, '__name__', '') for fn, _ in model._train_low.bwd.calls]
    btt = freeze.get('blocks_to_train')
    if btt == [None]:
        assert not any(f.startswith('mcn_conv2d') or f.startswith('mcn_bn') for f in fns), fns   # backward = the fc layer alone
    if freeze.get('update_batch_norm') is False or btt == [0, None]:
        assert 'mcn_bn_bwd_frozen' in fns
    bf = dtype == 'bfloat16'
    quant = bf16q if bf else None
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    hp = dict(freeze, base_weight_decay=0.01)
    for step in range(2):
        x = rng.random((BATCH, 64, 64, 3)).astype(np.float32)
        model.feed(x, LABELS)
        loss, _, y_pred = opt._step(None)
        grads = model.get_variables('grad')
        if bf:
            rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), LABELS.astype(np.float64), hp=hp, batch_total=BATCH, quant=quant)
        else:
            # fp32 against float64: every gradient tensor within 1e-3 (north_star's bar), with the ReLU decisions at the oracle's own
            # near-ties (|z| <= 4e-6 rms) taken from the device where that is what separates the two — tests/flip_util.py.  (Round 2 had a
            # blanket 1e-2 here for the second step: one such decision moves a 16-channel beta gradient by 3e-3.)
            rloss, rpred, rgrads, flips = flip_aware_step(spec, state, [(x.astype(np.float64), LABELS.astype(np.float64))], grads, hp=hp, batch_total=BATCH, tol=1e-3)
            print('step {}: {} ReLU near-tie decision(s) taken from the device'.format(step, assert_flip_budget(flips, '{} step {}'.format(freeze, step))))
        assert abs(loss - rloss) <= (3e-2 if bf else 1e-4) * abs(rloss), (step, loss, rloss)
        trainable = set(v.name for v in model.store.variables if v.trainable)
        assert trainable == set(rgrads), (sorted(trainable ^ set(rgrads))[:4])
        if bf:
            # (bf16 storage: small gamma / beta gradients are noisy tensor by tensor; test_resnet_step_bf16 has the calibration)
            worst = min((cosine(grads[k], rgrads[k]), k) for k in rgrads if np.linalg.norm(rgrads[k]) > 1e-6)
            assert worst[0] >= 0.9, 'step {}: worst gradient {}'.format(step, worst)
            flat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in sorted(rgrads)])
            assert cosine(flat(grads), flat(rgrads)) >= 0.98
        else:
            worst = max((rel_l2(grads[k], rgrads[k]), k) for k in rgrads if np.linalg.norm(rgrads[k]) > 1e-9)
            assert worst[0] <= 1e-3, 'step {}: worst gradient {} ({} near-tie decisions inverted)'.format(step, worst, len(flips))
            if step == 0 and not flips:
                assert worst[0] <= 1e-5, 'step 0 differs from the float64 oracle by rounding alone: {}'.format(worst)
            flat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in sorted(rgrads)])
            assert rel_l2(flat(grads), flat(rgrads)) <= 1e-3
        got, ema = model.get_variables('data'), model.get_variables('ema')
        tol = 2e-2 if bf else 1e-4
        worst = max((rel_l2(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert worst[0] <= tol, 'step {}: worst variable {}'.format(step, worst)
        worst = max((rel_l2(ema[k], v), k) for k, v in list(state.ema.items()) + list(state.ema_stats.items()))
        assert worst[0] <= tol, 'step {}: worst EMA {}'.format(step, worst)
        for k in params:                                                      # frozen variables did not move, bit for bit
            if k not in trainable:
                np.testing.assert_array_equal(got[k], params[k])
        for k in stats:                                                       # frozen statistics neither
            frozen_bn = (freeze.get('update_batch_norm') is False) or (freeze.get('update_batch_norm') is None and btt is not None
                                                                       and not ON.trainable_name(k, btt))
            if frozen_bn:
                np.testing.assert_array_equal(got[k], stats[k])


@pytest.mark.parametrize('freeze', [dict(blocks_to_train=[0, None]), dict(blocks_to_train=[3, 4, None])])
def test_resnet_frozen_blocks_with_gradient_clipping(freeze):
    """gradient_threshold together with blocks_to_train (optimizers.py:53,106,112-113 + convnet.py:1384-1389): tf.clip_by_global_norm sees
    the gradients of update_vars only, so the norm and the folded L2 gradient cover the trainable runs of the flat store
    (mcn_clip_by_global_norm_runs); frozen variables stay bit-identical and the clip is active (norm > threshold)."""
    import myconvnet_amd as M
    rng = np.random.default_rng(321)
    model, spec, params, stats = make_resnet(18, 'float32', True, **freeze)
    thr = 0.05
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0, gradient_threshold=thr)
    assert 'mcn_clip_by_global_norm_runs' in [getattr(fn, '__name__', '') for fn, _ in opt._clip.calls]
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    hp = dict(freeze, gradient_threshold=thr)
    trainable = set(v.name for v in model.store.variables if v.trainable)
    for step in range(2):
        x = rng.random((BATCH, 64, 64, 3)).astype(np.float32)
        model.feed(x, LABELS)
        loss, _, y_pred = opt._step(None)
        rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), LABELS.astype(np.float64), hp=hp, batch_total=BATCH)
        assert set(rgrads) == trainable
        assert abs(loss - rloss) <= 1e-4 * abs(rloss), (step, loss, rloss)
        norm = float(opt.grad_norm.item())
        assert norm > thr                                                     # the clip did something
        grads = model.get_variables('grad')                                   # the flat gradient holds the clipped full-loss gradient
        gn = np.sqrt(sum(float(np.sum(np.asarray(grads[k], np.float64) ** 2)) for k in trainable))
        assert abs(gn - thr) <= 1e-4 * thr, (gn, thr)
        worst = max((rel_l2(grads[k], rgrads[k]), k) for k in rgrads if np.linalg.norm(rgrads[k]) > 1e-9)
        assert worst[0] <= 1e-3, 'step {}: worst clipped gradient {}'.format(step, worst)
        got = model.get_variables('data')
        worst = max((rel_l2(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert worst[0] <= 1e-4, 'step {}: worst variable {}'.format(step, worst)
        for k in params:
            if k not in trainable:
                np.testing.assert_array_equal(got[k], params[k])
                assert not np.any(grads[k])                                   # nothing was folded into a frozen variable's slot


def test_resnet_l1_regulariser_and_focal_losses():
    """l1_reg, focal_loss_factor and sigmoid_focal_loss_factor together (convnet.py:529-533, 553-557, 581-592; all 0 in the reference's defaults): the
    loss value, the gradients (the flat buffer holds data gradient + l1 * sign(w) once the update has run) and the updated variables."""
    import myconvnet_amd as M
    rng = np.random.default_rng(91)
    hp = dict(l1_reg=1e-5, focal_loss_factor=2.0, sigmoid_focal_loss_factor=3.0)
    model, spec, params, stats = make_resnet(18, 'float32', True, **hp)
    names = [getattr(fn, '__name__', '') for fn, _ in model._train_low.fwd.calls]
    assert 'mcn_softmax_xent_focal_fwd_bwd' in names and 'mcn_l1_loss' in names
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    assert 'mcn_l1_grad_h' in [getattr(fn, '__name__', '') for fn, _ in opt.optimization_operation.calls]
    state, plain = (ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()}) for _ in range(2))
    for step in range(2):
        x = rng.random((BATCH, 64, 64, 3)).astype(np.float32)
        model.feed(x, LABELS)
        loss, _, y_pred = opt._step(None)
        rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), LABELS.astype(np.float64), hp=hp, batch_total=BATCH)
        if step == 0:
            ploss, _, _ = ON.train_step(spec, plain, x.astype(np.float64), LABELS.astype(np.float64), batch_total=BATCH)
            assert abs(rloss - ploss) > 5e-2 * abs(ploss)                      # the three terms are visible at this size
        assert abs(loss - rloss) <= 1e-4 * abs(rloss), (step, loss, rloss)
        assert rel_l2(y_pred, rpred) <= 1e-4
        grads = model.get_variables('grad')
        worst = max((rel_l2(grads[k], rgrads[k]), k) for k in rgrads if np.linalg.norm(rgrads[k]) > 1e-9)
        assert worst[0] <= 1e-3, 'step {}: worst gradient {}'.format(step, worst)
        got = model.get_variables('data')
        worst = max((rel_l2(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert worst[0] <= 1e-4, 'step {}: worst variable {}'.format(step, worst)


@pytest.mark.parametrize('freeze', [dict(), dict(blocks_to_train=[3, 4, None])])
def test_resnet_l1_regulariser_with_gradient_clipping(freeze):
    """l1_reg together with gradient_threshold (convnet.py:553-557 + optimizers.py:106-113): the reference differentiates the FULL loss — cross-entropy +
    L2 + L1 terms — and clips that gradient per tower by its global norm, so l1 * sign(w) must be inside the clipped vector (and inside the norm), also
    when only some blocks train."""
    import myconvnet_amd as M
    rng = np.random.default_rng(93)
    hp = dict(l1_reg=3e-4, gradient_threshold=0.5)
    model, spec, params, stats = make_resnet(18, 'float32', True, l1_reg=hp['l1_reg'], **freeze)
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0, gradient_threshold=hp['gradient_threshold'])
    assert 'mcn_l1_grad_h' in [getattr(fn, '__name__', '') for fn, _ in opt._clip.calls]
    assert 'mcn_l1_grad_h' not in [getattr(fn, '__name__', '') for fn, _ in opt.optimization_operation.calls]
    ohp = dict(hp, **freeze)
    state, noclip, nol1 = (ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()}) for _ in range(3))
    for step in range(2):
        x = rng.random((BATCH, 64, 64, 3)).astype(np.float32)
        model.feed(x, LABELS)
        loss, _, _ = opt._step(None)
        rloss, _, rgrads = ON.train_step(spec, state, x.astype(np.float64), LABELS.astype(np.float64), hp=ohp, batch_total=BATCH)
        assert abs(loss - rloss) <= 1e-4 * abs(rloss), (step, loss, rloss)
        grads = model.get_variables('grad')
        worst = max((rel_l2(grads[k], rgrads[k]), k) for k in rgrads if np.linalg.norm(rgrads[k]) > 1e-9)
        assert worst[0] <= 1e-3, 'step {}: worst gradient {}'.format(step, worst)
        # the clip was active, and the L1 term weighs in the clipped vector: both reference runs without one of the two end somewhere else
        total = np.sqrt(sum(float((rgrads[k] ** 2).sum()) for k in rgrads))
        assert abs(total - hp['gradient_threshold']) <= 1e-6 * hp['gradient_threshold'], total
        got = model.get_variables('data')
        worst = max((rel_l2(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert worst[0] <= 1e-4, 'step {}: worst variable {}'.format(step, worst)
        if step == 0:
            _, _, g1 = ON.train_step(spec, noclip, x.astype(np.float64), LABELS.astype(np.float64), hp=dict(freeze, l1_reg=hp['l1_reg']), batch_total=BATCH)
            _, _, g2 = ON.train_step(spec, nol1, x.astype(np.float64), LABELS.astype(np.float64), hp=dict(freeze, gradient_threshold=hp['gradient_threshold']), batch_total=BATCH)
            k = 'block_4/res_1/conv_1/weights'
            assert rel_l2(g1[k], rgrads[k]) > 1e-2 and rel_l2(g2[k], rgrads[k]) > 1e-2


def test_resnet_autotuned_tiles_step_matches_oracle():
    """ConvNet.autotune() (bench.py --autotune) pins a tile per conv launch — including the dgrads that carry the masked fan-in and the
    unit-output BN's backward sums (mcn_conv2d_dgrad_addmasked_bnred), whose partial-row count follows the tile.  The step behind the pinned
    tiles is the step of the library's own choice up to the order of fp32 sums (every gradient tensor, 2e-5), and the oracle's in loss and
    predictions (the per-tensor oracle bar of the untuned step: test_resnet_two_steps_fp32 / ..._bn_statistics_from_conv_epilogue)."""
    import myconvnet_amd as M
    x = np.random.default_rng(63).random((BATCH, 64, 64, 3)).astype(np.float32)
    res = {}
    for tuned in (False, True):
        model, spec, params, stats = make_resnet(50, 'float32', True, fuse_bn_stats=True, fuse_bn_bwd_red=True, fuse_bn_out_red=True)
        names = [getattr(fn, '__name__', '') for fn, _ in model._train_low.bwd.calls]
        assert 'mcn_conv2d_dgrad_addmasked_bnred' in names and 'mcn_conv2d_dgrad_bnred' in names
        opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
        model.feed(x, LABELS)
        if tuned:
            model.forward(train=True)                            # fill the buffers the timed launches read
            model.backward()
            chosen = model.autotune()
            seen = {id(a._obj) for fn, args in model._train_low.bwd.calls if getattr(fn, '__name__', '') == 'mcn_conv2d_dgrad_addmasked_bnred'
                    for a in args if hasattr(a, '_obj')}
            assert seen and seen <= set(chosen), 'the fan-in + BN-sums dgrads were not tile-tuned'
            # pin a tile that is NOT the library's choice on every such dgrad: the partial-row count of the BN backward behind it must follow
            for fn, args in model._train_low.bwd.calls:
                if getattr(fn, '__name__', '') == 'mcn_conv2d_dgrad_addmasked_bnred':
                    [a for a in args if hasattr(a, '_obj')][0]._obj.tile = 2            # candidate 1: 128 x 64
            model._train_low.sync_partial_rows()
            model.set_variables(dict(params, **stats))           # (forward() moved the running statistics: a fresh step)
        loss, _, y_pred = opt._step(None)
        res[tuned] = (loss, y_pred, model.get_variables('grad'))
        if not tuned:
            state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
            rloss, rpred, _ = ON.train_step(spec, state, x.astype(np.float64), LABELS.astype(np.float64), batch_total=BATCH)
            assert abs(loss - rloss) <= 1e-4 * abs(rloss)
            assert rel_l2(y_pred, rpred) <= 1e-4
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * abs(res[False][0])
    worst = max((rel_l2(res[True][2][k], v), k) for k, v in res[False][2].items() if np.linalg.norm(v) > 1e-6)
    assert worst[0] <= 2e-5, worst


def test_resnet_fp32_bn_statistics_from_conv_epilogue():
    """fuse_bn_stats (default on for bf16 only) / defer_dskip: the fp32 network with the BN statistics taken in the conv
    epilogues and the residual fan-in applied in the dgrad epilogue / projection BN backward."""
    import myconvnet_amd as M
    rng = np.random.default_rng(61)
    model, spec, params, stats = make_resnet(50, 'float32', True, fuse_bn_stats=True, defer_dskip=True)
    assert any(getattr(fn, '__name__', '') == 'mcn_bn_fwd_train_fused' for fn, _ in model._train_low.fwd.calls)
    # (the masked fan-in of the identity shortcuts: with or — round 4 — together with the unit-output BN's backward sums)
    assert any(getattr(fn, '__name__', '') in ('mcn_conv2d_dgrad_addmasked', 'mcn_conv2d_dgrad_addmasked_bnred') for fn, _ in model._train_low.bwd.calls)
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    for step in range(2):
        x = rng.random((BATCH, 64, 64, 3)).astype(np.float32)
        model.feed(x, LABELS)
        loss, _, y_pred = opt._step(None)
        rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), LABELS.astype(np.float64), batch_total=BATCH)
        assert abs(loss - rloss) <= 1e-4 * abs(rloss)
        assert rel_l2(y_pred, rpred) <= 1e-4
        grads = model.get_variables('grad')
        worst = max((rel_l2(grads[k], rgrads[k]), k) for k in rgrads)
        assert worst[0] <= 1e-3, 'step {}: worst gradient {}'.format(step, worst)
        got = model.get_variables('data')
        worst = max((rel_l2(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert worst[0] <= 1e-4, 'step {}: worst variable {}'.format(step, worst)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_resnet_bn_backward_sums_from_dgrad_epilogue(dtype):
    """fuse_bn_bwd_red (default: fp32 only): the two inner BNs of every bottleneck take their backward sums from the epilogue of the
    dgrad that produces their output gradient.  Same step with the switch off: gradients agree up to the order of the fp32 sums
    (fp32) / to bf16 rounding of dx around unchanged sums (bf16)."""
    import myconvnet_amd as M
    res = {}
    x = np.random.default_rng(67).random((BATCH, 64, 64, 3)).astype(np.float32)
    for on in (True, False):
        model, spec, params, stats = make_resnet(50, dtype, True, fuse_bn_bwd_red=on, fuse_bn_out_red=on)
        names = [getattr(fn, '__name__', '') for fn, _ in model._train_low.bwd.calls]
        assert ('mcn_conv2d_dgrad_bnred' in names) == on and ('mcn_bn_bwd_from_partials' in names) == on
        assert ('mcn_conv2d_dgrad_addmasked_bnred' in names) == on
        if on:
            assert names.count('mcn_conv2d_dgrad_bnred') == 32                                                  # 16 bottlenecks x (BN0, BN1)
            # round 4: the unit-output BNs whose gradient is completed by the next unit's conv_0 dgrad + masked fan-in: 16 units - the last of
            # each of the four stages (their readers are two convs / the global mean)
            assert names.count('mcn_conv2d_dgrad_addmasked_bnred') == 12 and 'mcn_conv2d_dgrad_addmasked' not in names
            assert names.count('mcn_bn_bwd_from_partials') == 32 + 12
        opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
        model.feed(x, LABELS)
        loss, _, _ = opt._step(None)
        res[on] = (loss, model.get_variables('grad'))
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * abs(res[False][0])          # the forward pass is the same launches
    if dtype == 'float32':
        worst = max((rel_l2(res[True][1][k], v), k) for k, v in res[False][1].items() if np.linalg.norm(v) > 1e-6)
        assert worst[0] <= 2e-5, worst
    else:
        # bf16 storage: a 1e-7 change of a BN's coefficients re-rounds a few of its dx elements by one bf16 ulp, and 50 layers amplify
        # that about as much as they amplify fp32 rounding above (x 200).  Measured: whole gradient 4.5e-3, tensors <= 1e-2 except the
        # stem's beta (a sum that nearly cancels) at 8e-2 — hence the aggregate criteria of test_resnet_step_bf16.
        flat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in sorted(res[False][1])])
        assert rel_l2(flat(res[True][1]), flat(res[False][1])) <= 2e-2
        worst = min((cosine(res[True][1][k], v), k) for k, v in res[False][1].items() if np.linalg.norm(v) > 1e-6)
        assert worst[0] >= 0.99, worst


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))


def bf16q(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


def _low_precision_step(kind, dtype, quant, seed, **opt_kw):
    """One training step of the width/8 test network in a 2-byte storage type on the device and in the oracle with the device's
    rounding points (oracle.net.Tape(fused_rounding=True): BN + add + ReLU round once, gradient contributions are added unrounded):
      free   — the oracle runs on its own: forward agreement (loss, predictions) and aggregate gradient agreement.  These tiny nets
               amplify one storage ulp 20-50x over their depth, and fp32-vs-float64 sums put a few roundings per layer on the other side,
               so tensor-by-tensor agreement of the free run depends on the seed (measured: 1e-3 ... 2e-1);
      forced — the oracle takes the device's stored activation and activation gradient at every residual-unit boundary
               (Tape.barrier), so a rounding difference cannot travel further than one unit: EVERY gradient tensor must then agree
               to 2e-2 (bf16) / 3e-3 (fp16).  A dropped or wrong-sign contribution inside a unit is O(1) there.
      exact  — float64 without storage rounding."""
    import myconvnet_amd as M
    model, spec, params, stats = make_resnet(kind, dtype, True)
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0, **opt_kw)
    x = np.random.default_rng(seed).random((BATCH, 64, 64, 3)).astype(np.float32)
    model.feed(x, LABELS)
    loss, _, y_pred = opt._step(None)
    fresh = lambda: ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})  # noqa: E731
    hp = dict(opt_kw) or None
    x64, y64 = x.astype(np.float64), LABELS.astype(np.float64)
    out = dict(model=model, loss=loss, pred=y_pred, grads=model.get_variables('grad'), state_free=fresh())
    out['free'] = ON.train_step(spec, out['state_free'], x64, y64, batch_total=BATCH, quant=quant, fused_rounding=True, hp=hp)
    names = ['block_0'] + ['block_{}/res_{}'.format(i, j) for i in range(1, 5) for j in range(spec.res_units[i])]
    host = lambda t: t.float().cpu().numpy().astype(np.float64)  # noqa: E731
    probe = {'force_act': {n: host(model.d[n].buf) for n in names}, 'force_grad': {n: host(model.d[n].grad) for n in names}}
    out['forced'] = ON.train_step(spec, fresh(), x64, y64, batch_total=BATCH, quant=quant, fused_rounding=True, hp=hp, probe=probe)
    out['exact'] = ON.train_step(spec, fresh(), x64, y64, batch_total=BATCH)
    return out


def _weights_vector(d):
    keys = sorted(k for k in d if k.endswith('weights'))
    return np.concatenate([np.asarray(d[k], np.float64).ravel() for k in keys])


@pytest.mark.parametrize('kind', [50, 18])
@pytest.mark.parametrize('seed', [5, 6])
def test_resnet_step_bf16(kind, seed):
    """bf16 storage (8 significant bits), like with like: the float64 oracle with every activation / activation gradient / per-use
    weight rounded to bf16 at the device's rounding points (fp32 masters, fp32 BN statistics, fp32 accumulation: the reference's
    half-precision structure).  See _low_precision_step for the oracle runs."""
    r = _low_precision_step(kind, 'bfloat16', bf16q, seed)
    loss, grads = r['loss'], r['grads']
    rloss, rpred, rgrads = r['free']
    floss, _, fgrads = r['forced']
    # forward pass, free run: same rounding points => the loss agrees to 5e-3 (measured 0 ... 1.7e-3 over nets and seeds; round 2, op-by-op
    # rounding in the oracle: 2e-2) and the predictions to 2e-2 (5e-2)
    assert abs(loss - rloss) <= 5e-3 * abs(rloss), (loss, rloss)
    assert rel_l2(r['pred'], rpred) <= 2e-2
    # every gradient tensor, forced run
    worst = max((rel_l2(grads[k], fgrads[k]), k) for k in fgrads if np.linalg.norm(fgrads[k]) > 1e-6)
    print('bf16, teacher-forced at the unit boundaries: worst gradient tensor', worst)
    assert worst[0] <= 2e-2, worst
    assert abs(loss - floss) <= 1e-5 * abs(floss)
    # free run, aggregate: the device is as accurate against exact arithmetic as the bf16-emulating oracle
    g, rr, e = _weights_vector(grads), _weights_vector(rgrads), _weights_vector(r['exact'][2])
    err_dev, err_emu = rel_l2(g, e), rel_l2(rr, e)
    print('bf16 gradient error vs float64: device {:.3f}, bf16-emulating oracle {:.3f}; cos(device, emu) {:.3f}'.format(err_dev, err_emu, cosine(g, rr)))
    assert err_dev <= 1.5 * err_emu + 0.02, (err_dev, err_emu)
    assert cosine(g, e) >= 0.9
    assert abs(np.linalg.norm(g) / np.linalg.norm(e) - 1.0) <= 0.1
    # running statistics come from fp32 sums of the bf16 activations
    got = r['model'].get_variables('data')
    assert rel_l2(got['block_0/conv_0/bn/mu'], r['state_free'].stats['block_0/conv_0/bn/mu']) <= 1e-3


def fp16q(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(torch.float16).float().numpy().astype(np.float64)


@pytest.mark.parametrize('kind', [50, 18])
def test_resnet_step_fp16_with_loss_scaling(kind):
    """fp16 storage + loss_scaling_factor = 128: the reference's own low-precision recipe (convnet.py:63, optimizers.py:102-111).
    Like with like: the float64 oracle with fp16 rounding at the device's rounding points and the same scale on the loss gradient.
    fp16 keeps 11 significant bits: every gradient tensor of the teacher-forced run within 3e-3; integer arg-max of the prediction
    agrees with the fp16-emulating oracle wherever its top-2 margin exceeds the storage resolution."""
    r = _low_precision_step(kind, 'float16', fp16q, 5, loss_scaling_factor=128.0)
    model = r['model']
    assert model.dtype == 'float16' and model.loss_scale == 128.0
    loss, y_pred, grads = r['loss'], r['pred'], r['grads']
    rloss, rpred, rgrads = r['free']
    floss, _, fgrads = r['forced']
    assert abs(loss - rloss) <= 2e-3 * abs(rloss), (loss, rloss)
    assert rel_l2(y_pred, rpred) <= 5e-3
    top2 = np.sort(rpred, axis=-1)
    sure = (top2[:, -1] - top2[:, -2]) > 2e-2
    np.testing.assert_array_equal(y_pred.argmax(-1)[sure], rpred.argmax(-1)[sure])
    worst = max((rel_l2(grads[k], fgrads[k]), k) for k in fgrads if np.linalg.norm(fgrads[k]) > 1e-6)
    print('fp16, teacher-forced at the unit boundaries: worst gradient tensor', worst)
    assert worst[0] <= 3e-3, worst
    g, rr, e = _weights_vector(grads), _weights_vector(rgrads), _weights_vector(r['exact'][2])
    err_dev, err_emu = rel_l2(g, e), rel_l2(rr, e)
    print('fp16 gradient error vs float64: device {:.4f}, fp16-emulating oracle {:.4f}; cos(device, emu) {:.5f}'.format(err_dev, err_emu, cosine(g, rr)))
    assert err_dev <= 1.5 * err_emu + 0.004, (err_dev, err_emu)
    assert cosine(g, e) >= 0.98 and cosine(g, rr) >= 0.98
    assert abs(np.linalg.norm(g) / np.linalg.norm(e) - 1.0) <= 0.05
    got = model.get_variables('data')
    assert rel_l2(got['block_0/conv_0/bn/mu'], r['state_free'].stats['block_0/conv_0/bn/mu']) <= 1e-3
    worst = max((rel_l2(got[k], v), k) for k, v in r['state_free'].params.items() if k.endswith('weights'))
    assert worst[0] <= 5e-3, worst                                        # one Nesterov step from identical masters


def test_resnet_eval_uses_ema_and_running_stats():
    import myconvnet_amd as M
    model, spec, params, stats = make_resnet(50, 'float32', True)
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    x = RNG.random((BATCH, 64, 64, 3)).astype(np.float32)
    y = LABELS
    model.feed(x, y)
    opt._step(None)
    ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=BATCH)
    ds = M.DataSet(x, y, batch_size=BATCH)
    _, y_true, y_pred, loss = model.predict(ds, return_images=False)
    t, out, pred, rloss, _ = ON.forward_loss(spec, state, x.astype(np.float64), y.astype(np.float64), train=False, use_ema=True)
    # the L2 term of the reported loss always reads the master variables (convnet.py:535)
    rloss = rloss - O.l2_reg_loss([v for k, v in state.ema.items() if k.endswith('/weights')]) + \
        O.l2_reg_loss([v for k, v in state.params.items() if k.endswith('/weights')])
    assert rel_l2(y_pred, pred) <= 1e-4
    assert abs(loss - rloss) <= 1e-4 * abs(rloss)
    np.testing.assert_array_equal(y_pred.argmax(-1), pred.argmax(-1))


@pytest.mark.parametrize('dtype,width_div', [('float32', 4), ('bfloat16', 4), ('float32', 1)])
def test_vgg16_trunk_config1(dtype, width_div):
    """BASELINE config #1: VGG-16 on 8x8x3 random images, batch 4 (trunk only: the head needs 224x224).
    width_div = 1 is the configuration as BASELINE.json states it (64 ... 512 channels: models/vggnet.py:29-103); width_div = 4 keeps the
    2-byte variant and a quick case."""
    import myconvnet_amd as M
    spec = ON.VGGSpec(16, 10, backbone_only=True, width_div=width_div)
    model = M.VGG16([8, 8, 3], 10, batch_size=4, backbone_only=True, width_div=width_div, half_precision=(dtype == 'bfloat16'), num_gpus=1)
    params, _ = ON.init_variables(spec.variables(), seed=1, dtype=np.float32)
    rng = np.random.default_rng(2)
    for k in params:
        if k.endswith('biases'):
            params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
        else:
            params[k] = (params[k] * 0.3).astype(np.float32)      # keep activations O(1) after the *255 input scaling
    model.set_variables(params)
    x = RNG.random((4, 8, 8, 3)).astype(np.float32)
    model.feed(x, np.zeros(4, np.float32))
    model.forward(train=True)
    out = model.d['block_4']
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {})
    tape, rout, _, _, _ = ON.forward_loss(spec, state, x.astype(np.float64), None, quant=(bf16q if dtype == 'bfloat16' else None))
    got = model.fetch(out)
    tol = 1e-4 if dtype == 'float32' else 5e-2
    assert got.shape == rout.a.shape == (4, 1, 1, 512 // width_div)
    assert rel_l2(got, rout.a) <= tol
    # backward from an injected output gradient
    dy = RNG.standard_normal(rout.a.shape).astype(np.float32)
    out.grad.copy_(torch.as_tensor(dy).to(out.grad.dtype).to(out.grad.device))
    model.backward()
    rout.g = torch.as_tensor(dy).to(out.grad.dtype).float().numpy().astype(np.float64)
    rgrads = tape.backward()
    grads = model.get_variables('grad')
    if dtype == 'float32':
        worst = max((rel_l2(grads[k], rgrads[k]), k) for k in rgrads)
        assert worst[0] <= 1e-3, worst
    else:                                              # bf16: direction and size of the whole gradient
        keys = sorted(rgrads)
        g = np.concatenate([grads[k].ravel() for k in keys])
        r = np.concatenate([rgrads[k].ravel() for k in keys])
        assert cosine(g, r) >= 0.98 and abs(np.linalg.norm(g) / np.linalg.norm(r) - 1.0) <= 0.1, (cosine(g, r), np.linalg.norm(g) / np.linalg.norm(r))


def test_channel_first_input_matches_nhwc():
    """channel_first=True: the batch is fed NCHW (reference convnet.py:467-468) and must give the same logits."""
    import myconvnet_amd as M
    a, spec, params, stats = make_resnet(50, 'float32', True)
    b = M.ResNet50([64, 64, 3], 10, batch_size=BATCH, width_div=8, channel_first=True, num_gpus=1)
    b.set_variables(dict(params, **stats))
    x = RNG.random((BATCH, 64, 64, 3)).astype(np.float32)
    y = LABELS
    a.feed(x, y)
    a.forward(train=True)
    b.feed(np.ascontiguousarray(x.transpose(0, 3, 1, 2)), y)
    b.forward(train=True)
    np.testing.assert_array_equal(a.fetch(a.logits), b.fetch(b.logits))
    assert b.d['block_1'].get_shape() == [BATCH, 32, 16, 16]
