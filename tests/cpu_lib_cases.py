"""Cases of tests/test_cpu_lib.py, run as `MCN_LIB_PATH=.../libmcn_cpu.so python tests/cpu_lib_cases.py <case>` in a fresh process (the binding
loads ONE library per process; the test runner itself holds libmcn_hip.so).  Each case drives the HOST code of the product — graph
recording, fusion lowering, launch lists, optimizer, data parallel exchange — through the C-ABI with libmcn_cpu.so behind it and checks the
result against the oracle.  Exit code 0 = pass; the last line printed is a short summary."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))

import torch  # noqa: E402

import myconvnet_amd as M  # noqa: E402
from myconvnet_amd import _ffi  # noqa: E402
from oracle import net as ON  # noqa: E402

assert _ffi.IS_CPU_LIB, 'run with MCN_LIB_PATH=<repo>/myconvnet_amd/libmcn_cpu.so'


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def conditioned(params, kind):
    rng = np.random.default_rng(9)
    for k in params:
        if k.endswith('gamma'):
            lo, span = (0.1, 0.2) if ('conv_2/bn' in k or (kind == 18 and 'conv_1/bn' in k)) else (0.8, 0.4)
            params[k] = (lo + span * rng.random(params[k].shape)).astype(np.float32)
        if k.endswith('beta'):
            params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    return params


def fresh(params, stats):
    return ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})


def resnet(kind, dtype='float32', steps=2, model_kw=None, **opt_kw):
    B = 8
    spec = ON.ResNetSpec.resnet50(10, 8) if kind == 50 else ON.ResNetSpec.resnet18(10, 8)
    params, stats = ON.init_variables(spec.variables(), seed=3, dtype=np.float32)
    params = conditioned(params, kind)
    cls = M.ResNet50 if kind == 50 else M.ResNet18
    model = cls([64, 64, 3], 10, batch_size=B, width_div=8, num_gpus=1, device='cpu', half_precision=(dtype != 'float32'),
                half_precision_dtype=(dtype if dtype != 'float32' else 'bfloat16'), **(model_kw or {}))
    model.set_variables(dict(params, **stats))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0, **opt_kw)
    if model_kw and 'blocks_to_train' in model_kw and opt_kw.get('gradient_threshold') is not None:
        assert 'mcn_clip_by_global_norm_runs' in [getattr(fn, '__name__', '') for fn, _ in opt._clip.calls]
    names = [getattr(fn, '__name__', '') for fn, _ in model._train_low.fwd.calls + model._train_low.bwd.calls]
    assert 'mcn_conv2d_fwd' in names and 'mcn_bn_fwd_train' in names and 'mcn_conv2d_fwd_bnstats' not in names      # the plain lowering
    state = fresh(params, stats)
    rng = np.random.default_rng(5)
    y = np.array([1, 7, 3, 9, 0, 4, 4, 2], dtype=np.float32)
    quant = None
    if dtype != 'float32':
        tdt = {'bfloat16': torch.bfloat16, 'float16': torch.float16}[dtype]
        quant = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(tdt).float().numpy().astype(np.float64)  # noqa: E731
    worst = 0.0
    for step in range(steps):
        x = rng.random((B, 64, 64, 3)).astype(np.float32)
        model.feed(x, y)
        loss, _, pred = opt._step(None)
        rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B, quant=quant,
                                             fused_rounding=quant is not None, hp=dict(opt_kw, **(model_kw or {})) or None)
        grads = model.get_variables('grad')
        if quant is None:
            assert abs(loss - rloss) <= 1e-5 * abs(rloss), (step, loss, rloss)
            assert np.abs(pred - rpred).max() <= 1e-5
            if opt_kw.get('gradient_threshold') is not None:      # the device's flat gradient holds the CLIPPED full-loss gradient; the oracle
                w = (0.0, '')                                      # returns the tower mean of the same: compared through the update below
            else:
                w = max((rel(grads[k], rgrads[k]), k) for k in rgrads if np.linalg.norm(rgrads[k]) > 1e-9)
            assert w[0] <= 1e-3, (step, w)
            worst = max(worst, w[0])
        else:
            # same rounding points as the oracle's fused mode, fp32 arithmetic here against float64 there: forward agreement
            assert abs(loss - rloss) <= 5e-3 * abs(rloss), (step, loss, rloss)
            assert rel(pred, rpred) <= 2e-2
    if quant is None:
        got = model.get_variables('data')
        w = max((rel(got[k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert w[0] <= 1e-4, w
        ema = model.get_variables('ema')
        w = max((rel(ema[k], v), k) for k, v in state.ema.items())
        assert w[0] <= 1e-4, w
    print('resnet{} {} x{} steps ok, worst gradient tensor {:.2e}'.format(kind, dtype, steps, worst))


def deeplab(label_smoothing=0.0, **head_kw):
    """DeepLabv3+ on the dilated ResNet-50 (width/8, depth/3): dilated convs, ASPP, bilinear resize, concat, per-pixel loss with ignored pixels"""
    from oracle import ops as O
    B, SIZE, CLASSES = 2, 65, 6
    spec = ON.DeepLabSpec(CLASSES, width_div=8, depth_div=3, aspp_dilations=(1, 2, 3), **head_kw)
    model = M.DeepLabV3PlusResNet50([SIZE, SIZE, 3], CLASSES, batch_size=B, width_div=8, depth_div=3, aspp_dilations=[1, 2, 3], num_gpus=1, device='cpu',
                                    label_smoothing=label_smoothing, **{k: (list(v) if isinstance(v, tuple) else v) for k, v in head_kw.items()})
    params, stats = ON.init_variables(spec.variables(), seed=6, dtype=np.float32)
    rng = np.random.default_rng(12)
    for k in params:
        if k.endswith('gamma'):
            lo, span = (0.1, 0.2) if 'conv_2/bn' in k else (0.8, 0.4)
            params[k] = (lo + span * rng.random(params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('biases'):
            params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    model.set_variables(dict(params, **stats))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state = fresh(params, stats)
    x = rng.random((B, SIZE, SIZE, 3)).astype(np.float32)
    y = rng.integers(0, CLASSES + 1, (B, SIZE, SIZE)).astype(np.float32)
    y[0, :3, :] = np.nan
    model.feed(x, y)
    loss, y_true, y_pred = opt._step(None)
    if label_smoothing:
        assert 'mcn_softmax_xent_rows_focal_fwd_bwd' in [getattr(fn, '__name__', '') for fn, _ in model._train_low.fwd.calls]
    rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B,
                                         hp=dict(label_smoothing=label_smoothing) if label_smoothing else None)
    assert abs(loss - rloss) <= 1e-4 * abs(rloss), (loss, rloss)
    assert rel(y_pred, rpred) <= 1e-4
    np.testing.assert_array_equal(y_true, O.seg_one_hot_labels(y, CLASSES))
    grads = model.get_variables('grad')
    assert set(rgrads) <= set(grads) and (not head_kw.get('aspp_level_feature') or 'block_5/aspp/conv_pool/weights' in rgrads)
    scale = max(np.linalg.norm(v) for v in rgrads.values())
    w = max((np.linalg.norm(np.asarray(grads[k], np.float64) - rgrads[k]) / max(np.linalg.norm(rgrads[k]), 1e-6 * scale), k) for k in rgrads)
    # (the image-level branch normalises over M = B = 2 rows: its beta gradient is the least well conditioned number of the net)
    assert w[0] <= (4e-3 if head_kw.get('aspp_level_feature') else 2e-3), w
    print('deeplabv3+ (w/8, d/3, 65x65{}) step ok, worst gradient tensor {:.2e} {}'.format((', 5x5 label smoothing' if label_smoothing else '') + (', ' + str(head_kw) if head_kw else ''), *w))


def train_loop():
    """Optimizer.train() for two epochs with a validation set on the CPU library: loop bookkeeping, LR schedule, predict()"""
    from myconvnet_amd.dataset import DataSet
    from myconvnet_amd.evaluators import AccuracyEvaluator
    rng = np.random.default_rng(11)
    B = 8
    xs = rng.random((3 * B, 32, 32, 3)).astype(np.float32)
    ys = rng.integers(0, 10, 3 * B).astype(np.float32)
    train = DataSet(xs, ys, batch_size=B)
    val = DataSet(xs[:B + 3], ys[:B + 3], batch_size=B)
    model = M.ResNet18([32, 32, 3], 10, batch_size=B, width_div=8, num_gpus=1, device='cpu')
    opt = M.MomentumOptimizer(model, train, AccuracyEvaluator(), val_set=val, base_learning_rate=0.05, num_epochs=2, learning_warmup_epochs=1.0)
    d = opt.train(details=True, verbose=False)
    assert len(d['train_losses']) == 2 and len(d['eval_scores']) == 2 and opt.curr_epoch == 3 and opt.curr_step == 6, (d, opt.curr_epoch, opt.curr_step)
    assert all(np.isfinite(d['train_losses'])) and all(0.0 <= s <= 1.0 for s in d['eval_scores'])
    print('train loop ok: losses {}'.format(['%.4f' % v for v in d['train_losses']]))


def efficientnet():
    import make_golden as MG
    spec = ON.EfficientNetSpec.b0(10, width_div=2, depth_div=2)
    params, stats = MG.effnet_params(spec)
    B = 4
    model = M.EfficientNetB0([64, 64, 3], 10, batch_size=B, width_div=2, depth_div=2, num_gpus=1, device='cpu')
    model.set_variables(dict(params, **stats))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    state = fresh(params, stats)
    rng = np.random.default_rng(3)
    x = rng.random((B, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 10, B).astype(np.float32)
    model.feed(x, y)
    loss, _, pred = opt._step(None)
    rloss, rpred, rgrads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=B)
    assert abs(loss - rloss) <= 1e-4 * abs(rloss), (loss, rloss)
    grads = model.get_variables('grad')
    w = max((rel(grads[k], rgrads[k]), k) for k in rgrads if np.linalg.norm(rgrads[k]) > 1e-8)
    assert w[0] <= 2e-3, w
    print('efficientnet-b0 (w/2, d/2) step ok, worst gradient tensor {:.2e} {}'.format(*w))


def dist_stats(stats, freeze):
    if freeze:                                                                # non-trivial running statistics for the frozen BNs
        rng = np.random.default_rng(77)
        for k in stats:
            stats[k] = (0.2 * rng.standard_normal(stats[k].shape) if k.endswith('/mu') else 0.5 + rng.random(stats[k].shape)).astype(np.float32)
    return stats


def dist_worker(rank, world, port, q, freeze=None):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MCN_DIST_BACKEND='gloo')
    B = 4
    spec = ON.ResNetSpec.resnet18(10, 8)
    params, stats = ON.init_variables(spec.variables(), seed=3, dtype=np.float32)
    params = conditioned(params, 18)
    stats = dist_stats(stats, freeze)
    model = M.ResNet18([32, 32, 3], 10, batch_size=B * world, width_div=8, num_gpus=world, device='cpu', seed=rank, **(freeze or {}))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0, allreduce_bucket_mb=0.02)
    model.set_variables(dict(params, **stats))
    rng = np.random.default_rng(123)
    out = []
    for step in range(2):
        x = rng.random((B * world, 32, 32, 3)).astype(np.float32)
        y = rng.integers(0, 10, B * world).astype(np.float32)
        model.feed(x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B])
        loss, _, pred = opt._step(None)
        out.append((loss, pred))
    if freeze:
        assert 'mcn_bn_bwd_frozen' in [getattr(fn, '__name__', '') for fn, _ in model._train_low.bwd.calls]
        chain = [a for fn, a in opt._post_fwd.calls if getattr(fn, '__name__', '') == 'mcn_bn_running_chain_strided']
        assert chain and sum(a[3] for a in chain) < model.stats.size           # the chain leaves the frozen ranges alone
    q.put((rank, out, model.get_variables('data'), len(opt.dp.reducer.plan)))
    torch.distributed.destroy_process_group()


def dist(world=2, freeze=None):
    """the data-parallel step (bucketed all-reduce hooks in the backward list, BN statistics all-gather + chained running update, 1/N in
    the update) executed by `world` CPU ranks over gloo against the oracle's multi-tower restatement"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=dist_worker, args=(r, world, port, q, freeze)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    B = 4
    spec = ON.ResNetSpec.resnet18(10, 8)
    params, stats = ON.init_variables(spec.variables(), seed=3, dtype=np.float32)
    params = conditioned(params, 18)
    stats = dist_stats(stats, freeze)
    state = fresh(params, stats)
    rng = np.random.default_rng(123)
    for step in range(2):
        x = rng.random((B * world, 32, 32, 3)).astype(np.float32)
        y = rng.integers(0, 10, B * world).astype(np.float32)
        towers = [(x[r * B:(r + 1) * B].astype(np.float64), y[r * B:(r + 1) * B].astype(np.float64)) for r in range(world)]
        rloss, rpred, _ = ON.train_step(spec, state, None, None, batch_total=B * world, tower_batches=towers, hp=dict(freeze) if freeze else None)
        for r in range(world):
            loss, pred = res[r][1][step]
            assert abs(loss - rloss) <= 1e-4 * abs(rloss), (step, r, loss, rloss)
            np.testing.assert_allclose(pred, rpred, rtol=0, atol=2e-4)
    for r in range(world):
        w = max((rel(res[r][2][k], v), k) for k, v in list(state.params.items()) + list(state.stats.items()))
        assert w[0] <= 3e-4, w
        assert res[r][3] >= (1 if freeze else 2)
    for k in res[0][2]:
        for r in range(1, world):
            np.testing.assert_array_equal(res[0][2][k], res[r][2][k])
    if freeze:                       # ADVICE r3 (dist.py:156): the chained update rewrote frozen running statistics to r * m^world every step
        btt, ubn = freeze.get('blocks_to_train'), freeze.get('update_batch_norm')
        n_frozen = 0
        for k in stats:
            if ubn is False or (ubn is None and btt is not None and not ON.trainable_name(k, btt)):
                np.testing.assert_array_equal(res[0][2][k], stats[k])
                n_frozen += 1
        assert n_frozen > 0
    print('{}-rank data-parallel step ok (gloo, libmcn_cpu.so{})'.format(world, ', frozen: {}'.format(freeze) if freeze else ''))


if __name__ == '__main__':
    case = sys.argv[1]
    if case == 'resnet50':
        resnet(50)
    elif case == 'resnet18':
        resnet(18)
    elif case == 'resnet18_bf16':
        resnet(18, 'bfloat16', steps=1)
    elif case == 'resnet50_fp16':
        resnet(50, 'float16', steps=1, loss_scaling_factor=128.0)
    elif case == 'resnet18_decay_clip':
        resnet(18, steps=2, base_weight_decay=0.01)
    elif case == 'resnet18_frozen_clip':                   # clipping with blocks_to_train (optimizers.py:112-113 + convnet.py:1384-1389)
        resnet(18, steps=2, model_kw=dict(blocks_to_train=[0, None]), gradient_threshold=0.05)
    elif case == 'resnet18_l1_focal':                      # l1_reg + both focal factors (convnet.py:553-557, 581-592)
        resnet(18, steps=2, model_kw=dict(l1_reg=1e-5, focal_loss_factor=2.0, sigmoid_focal_loss_factor=3.0))
    elif case == 'resnet18_l1_clip':                       # l1_reg inside the clipped full-loss gradient (convnet.py:553-557 + optimizers.py:106-113)
        resnet(18, steps=2, model_kw=dict(l1_reg=3e-4), gradient_threshold=0.5)
    elif case == 'dw_mult_bias':                           # conv_layer(depthwise=True) with a channel multiplier / a bias (convnet.py:1634-1650, 1678-1694)
        import dw_mult_case
        from test_gpu_ops import check, q
        for c in dw_mult_case.CASES:
            dw_mult_case.run_case(M, c, 'float32', check, q, device='cpu')
        print('depthwise multiplier / bias cases ok')
    elif case == 'train_loop':
        train_loop()
    elif case == 'efficientnet':
        efficientnet()
    elif case == 'deeplab':
        deeplab()
    elif case == 'deeplab_level':                          # the reference's other branches: ASPP image-level feature + tf.stop_gradient on the low-level feature
        deeplab(aspp_level_feature=True, feature_gradients=(None, False))
    elif case == 'deeplab_ls':                             # SegNet's label smoothing: 5x5 average of the label map (segnet.py:117-122)
        deeplab(label_smoothing=0.2)
    elif case == 'dist2':
        dist(2)
    elif case == 'dist4':
        dist(4)
    elif case == 'dist2_frozen':
        dist(2, freeze=dict(blocks_to_train=[0, None]))
    else:
        raise SystemExit('unknown case ' + case)
