"""GPU tests of the training LOOP (reference optimizers.py:179-563 `train`, convnet.py:609-665 `predict`) and of the
loss-scaling path (optimizers.py:102-111) — the host code around `_step` that round 1 never executed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import net as ON  # noqa: E402
from oracle import ops as O   # noqa: E402
from test_gpu_net import make_resnet, rel_l2  # noqa: E402


def _state(params, stats):
    return ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})


def test_train_loop_two_epochs_matches_manual_steps_and_oracle_validation():
    """Optimizer.train(): feed / next_batch(shard=rank), LR schedule, epoch bookkeeping, validation through predict()
    (short last batch: 20 examples, batch 8) and is_better — against (i) the oracle stepping on the same batches and
    (ii) the oracle's evaluation on EMA weights + EMA running statistics."""
    import myconvnet_amd as M
    B, classes = 8, 10
    model, spec, params, stats = make_resnet(50, 'float32', True)
    xs, ys = M.synthetic(24, (64, 64, 3), classes, seed=11)
    xv, yv = M.synthetic(20, (64, 64, 3), classes, seed=12)
    yv[3] = np.nan                                                        # an ignored validation sample
    train_set = M.DataSet(xs, ys, batch_size=B)
    val_set = M.DataSet(xv, yv, batch_size=B)
    ev = M.AccuracyEvaluator()
    hp = dict(base_learning_rate=0.05, num_epochs=2, learning_warmup_epochs=1.0, learning_rate_decay_method='cosine', learning_rate_decay_params=(0,))
    opt = M.MomentumOptimizer(model, train_set, ev, val_set=val_set, **hp)
    d = opt.train(details=True, verbose=False)
    assert opt.curr_step == 6 and opt.curr_epoch == 3 and opt.steps_per_epoch == 3 and model.global_step == 6
    assert [len(d[k]) for k in ('train_losses', 'train_scores', 'eval_losses', 'eval_scores')] == [2, 2, 2, 2]

    # (i) the oracle on the same six batches with the same LR multipliers
    state = _state(params, stats)
    ref = M.MomentumOptimizer.__new__(M.MomentumOptimizer)               # only the LR schedule of a second instance
    ref.warmup_epoch, ref.decay_method, ref.decay_params = 1.0, 'cosine', (0,)
    ref.steps_per_epoch, ref.num_epochs, ref.curr_step, ref.curr_epoch, ref.curr_multiplier = 3, 2, 0, 1, 1.0
    losses, scores, evals = [], [], []
    for i in range(6):
        ref._update_learning_rate()
        idx = np.arange(i * B, (i + 1) * B) % 24
        rloss, rpred, _ = ON.train_step(spec, state, xs[idx].astype(np.float64), ys[idx].astype(np.float64), batch_total=B,
                                        hp=dict(base_learning_rate=0.05), lr_mult=ref.curr_multiplier)
        losses.append(rloss)
        scores.append(O.accuracy_score(O.one_hot_labels(ys[idx], classes), rpred))
        ref.curr_step += 1
        if (i + 1) % 3 == 0:
            ref.curr_epoch += 1
            # (ii) validation: EMA weights / EMA running statistics; the reported L2 term reads the master variables
            el, preds = [], []
            for s in range(0, 20, B):
                xb, yb = xv[s:s + B].astype(np.float64), yv[s:s + B].astype(np.float64)
                _, _, pred, l, _ = ON.forward_loss(spec, state, xb, yb, train=False, use_ema=True)
                l = l - O.l2_reg_loss([v for k, v in state.ema.items() if k.endswith('/weights')]) + \
                    O.l2_reg_loss([v for k, v in state.params.items() if k.endswith('/weights')])
                el.append(l)
                preds.append(pred)
            evals.append((float(np.mean(el)), O.accuracy_score(O.one_hot_labels(yv, classes), np.concatenate(preds))))
    for e in range(2):
        assert abs(d['train_losses'][e] - np.mean(losses[3 * e:3 * e + 3])) <= 2e-4 * abs(np.mean(losses[3 * e:3 * e + 3]))
        assert d['train_scores'][e] == pytest.approx(np.mean(scores[3 * e:3 * e + 3]), abs=1e-12)      # arg-max exact
        assert abs(d['eval_losses'][e] - evals[e][0]) <= 2e-4 * abs(evals[e][0]), (e, d['eval_losses'][e], evals[e][0])
        assert d['eval_scores'][e] == pytest.approx(evals[e][1], abs=1e-12)
    best = 0.0
    for _, sc in evals:
        if ev.is_better(sc, best):
            best = sc
    assert opt.best_score == pytest.approx(best, abs=1e-12)
    got = model.get_variables('data')
    worst = max((rel_l2(got[k], v), k) for k, v in state.params.items())
    assert worst[0] <= 3e-4, worst


def test_predict_visits_every_example_once_and_drops_padding():
    """predict() on 20 examples with batch 8 (2 full batches + 4): outputs in dataset order, padded rows dropped, the short
    batch's loss averaged over the rows that exist; max_examples truncates like the reference (convnet.py:612-616)."""
    import myconvnet_amd as M
    model, spec, params, stats = make_resnet(50, 'float32', True)
    xv, yv = M.synthetic(20, (64, 64, 3), 10, seed=13)
    state = _state(params, stats)
    ds = M.DataSet(xv, yv, batch_size=8)
    X, y_true, y_pred, loss = model.predict(ds, return_images=True)
    assert X.shape == (20, 64, 64, 3) and y_true.shape == y_pred.shape == (20, 10)
    np.testing.assert_array_equal(X, xv)
    np.testing.assert_array_equal(y_true, O.one_hot_labels(yv, 10))
    rl = []
    for s in range(0, 20, 8):
        _, _, pred, l, _ = ON.forward_loss(spec, state, xv[s:s + 8].astype(np.float64), yv[s:s + 8].astype(np.float64), train=False, use_ema=True)
        assert rel_l2(y_pred[s:s + 8], pred) <= 1e-4
        np.testing.assert_array_equal(y_pred[s:s + 8].argmax(-1), pred.argmax(-1))
        rl.append(l)
    assert abs(loss - np.mean(rl)) <= 1e-4 * abs(np.mean(rl))
    _, yt5, yp5, _ = model.predict(ds, return_images=False, max_examples=5)
    assert yt5.shape == (5, 10)
    np.testing.assert_array_equal(yp5, y_pred[:5])


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_loss_scaling_keeps_injected_variables_and_changes_nothing(dtype):
    """Optimizer(loss_scaling_factor=128) re-lowers the model but must keep injected variables / state (round-1 bug: it
    re-initialised them), scale the loss gradient by 128 and every parameter gradient by 1/128 (optimizers.py:102-111):
    a power of two, so the step equals the unscaled one up to rounding, and the oracle's."""
    import myconvnet_amd as M
    res = {}
    for scale in (1.0, 128.0, 0.5):
        model, spec, params, stats = make_resnet(50, dtype, True)
        before = model.get_variables('data')
        opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0, loss_scaling_factor=scale)
        assert model.loss_scale == max(scale, 1.0)                       # the reference scales only for a factor > 1
        after = model.get_variables('data')
        for k in before:
            np.testing.assert_array_equal(before[k], after[k])           # not re-initialised by the re-lowering
        rng = np.random.default_rng(77)
        x = rng.random((8, 64, 64, 3)).astype(np.float32)
        y = rng.integers(0, 10, 8).astype(np.float32)
        model.feed(x, y)
        loss, _, pred = opt._step(None)
        res[scale] = (loss, pred, model.get_variables('grad'), model.get_variables('data'))
    state = _state(params, stats)
    rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=8)
    l1, p1, g1, d1 = res[1.0]
    l128, p128, g128, d128 = res[128.0]
    assert l128 == l1 and res[0.5][0] == l1                               # the reported loss is the unscaled one
    np.testing.assert_array_equal(p128, p1)
    tol_g, tol_d = (2e-5, 1e-6) if dtype == 'float32' else (3e-2, 1e-3)
    worst = max((rel_l2(g128[k], g1[k]), k) for k in g1 if np.linalg.norm(g1[k]) > 0)
    assert worst[0] <= tol_g, worst
    assert max(rel_l2(d128[k], d1[k]) for k in d1) <= tol_d
    for k in d1:
        np.testing.assert_array_equal(res[0.5][3][k], d1[k])             # factor <= 1: no scaling at all
    if dtype == 'float32':
        assert abs(l128 - rloss) <= 1e-4 * abs(rloss)
        worst = max((rel_l2(g128[k], rgrads[k]), k) for k in rgrads)
        assert worst[0] <= 1e-3, worst


@pytest.mark.parametrize('kind', ['resnet_fp32', 'resnet_bf16', 'effnet_bf16'])
def test_graph_replay_is_bit_identical_to_eager_launch_lists(kind):
    """One hipGraph launch per step (Optimizer._capture_step: forward, backward with the wgrad side stream, update; per-step
    scalars from the device hyper-parameter buffer) against the eager launch lists: 6 steps with a moving learning rate, EMA
    decay and — EfficientNet — fresh stochastic-depth / dropout masks every step; losses and every variable bit for bit."""
    import myconvnet_amd as M
    import torch
    res = {}
    for graph in (False, True):
        if kind.startswith('resnet'):
            model, spec, params, stats = make_resnet(50, 'float32' if kind.endswith('fp32') else 'bfloat16', True)
        else:
            from test_gpu_efficientnet import make_effnet
            model, spec, params, stats = make_effnet('bfloat16', True, initial_drop_rate=0.1, final_drop_rate=0.3, dropout_rate=0.2)
        model._mask_rng = np.random.default_rng(4242)
        opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=4, num_epochs=3, learning_warmup_epochs=1.0,
                                  learning_rate_decay_method='cosine', learning_rate_decay_params=(0,), base_weight_decay=0.01, use_graph=graph)
        rng = np.random.default_rng(321)
        losses = []
        for step in range(6):
            x = rng.random((8, 64, 64, 3)).astype(np.float32)
            y = rng.integers(0, 10, 8).astype(np.float32)
            model.feed(x, y)
            opt._update_learning_rate()
            loss, _, pred = opt._step(None)
            opt.curr_step += 1
            losses.append(loss)
        assert (opt._graph is not None) == graph                          # steps 3-6 were graph replays
        torch.cuda.synchronize()
        res[graph] = (losses, model.get_variables('data'), model.get_variables('ema'), pred)
    assert res[False][0] == res[True][0]
    for which in (1, 2):
        for k, v in res[False][which].items():
            np.testing.assert_array_equal(v, res[True][which][k], err_msg=k)
    np.testing.assert_array_equal(res[False][3], res[True][3])
