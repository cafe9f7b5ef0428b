"""Generates tests/golden/*.npz from the NumPy oracle (float64 math, stored as float32/float64).

    python tests/golden/make_golden.py [ops] [resnet50_w8] [mbconv]

PARITY UNPINNED: the reference has no golden vectors and cannot run here, so these fixtures freeze
the oracle's answers — they guard the oracle against regressions and give the GPU tests fixed inputs/outputs that
travel to the GPU box.  Every fixture stores inputs AND expected outputs; nothing here is reference source text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import net as ON  # noqa: E402
from oracle import ops as O   # noqa: E402

# name: (n, h, w, cin, cout, k, stride, padding, dilation)
CONV_CASES = {
    'c1x1_s1': (2, 8, 8, 8, 16, 1, 1, 'SAME', 1),
    'c1x1_s2': (2, 8, 8, 8, 16, 1, 2, 'SAME', 1),
    'c3x3_s1': (2, 7, 7, 8, 8, 3, 1, 'SAME', 1),
    'c3x3_s2_even': (2, 16, 16, 8, 8, 3, 2, 'SAME', 1),
    'c3x3_s2_odd': (2, 15, 15, 8, 8, 3, 2, 'SAME', 1),
    'c7x7_s2_stem': (2, 16, 16, 3, 16, 7, 2, 'SAME', 1),
    'c3x3_biased_vgg': (2, 8, 8, 8, 16, 3, 1, 'SAME', 1),
    'c3x3_dilated': (1, 14, 14, 8, 8, 3, 1, 'SAME', 2),
}


def f32(a):
    return np.asarray(a, dtype=np.float32)


def golden_ops():
    rng = np.random.default_rng(20261003)
    out = {}
    for name, (n, h, w, cin, cout, k, s, pad, dil) in CONV_CASES.items():
        x = f32(rng.standard_normal((n, h, w, cin)))
        wt = f32(rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin))
        y = O.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), s, pad, dil)
        dy = f32(rng.standard_normal(y.shape))
        out[name + '/x'], out[name + '/w'], out[name + '/dy'] = x, wt, dy
        out[name + '/y'] = f32(y)
        out[name + '/dx'] = f32(O.conv2d_dgrad(dy.astype(np.float64), wt.astype(np.float64), x.shape, s, pad, dil))
        out[name + '/dw'] = f32(O.conv2d_wgrad(x.astype(np.float64), dy.astype(np.float64), wt.shape, s, pad, dil))
        if 'biased' in name:
            b = f32(rng.standard_normal(cout))
            out[name + '/b'] = b
            out[name + '/y'] = f32(y + b)
            out[name + '/db'] = f32(O.bias_add_bwd(dy.astype(np.float64)))
    # batch norm incl. running update and backward
    x = f32(rng.standard_normal((4, 5, 5, 8)) * 2 + 1)
    g, b = f32(0.5 + rng.random(8)), f32(rng.standard_normal(8))
    y, bm, bv, sm, si = O.bn_fwd_train(x.astype(np.float64), g.astype(np.float64), b.astype(np.float64), 1e-3)
    dy = f32(rng.standard_normal(x.shape))
    dx, dg, db = O.bn_bwd(dy.astype(np.float64), x.astype(np.float64), g.astype(np.float64), sm, si)
    rm, rv = O.bn_running_update(np.zeros(8), np.ones(8), bm, bv, 0.99)
    out.update({'bn/x': x, 'bn/gamma': g, 'bn/beta': b, 'bn/dy': dy, 'bn/y': f32(y), 'bn/batch_mean': f32(bm), 'bn/batch_var': f32(bv),
                'bn/invstd': f32(si), 'bn/dx': f32(dx), 'bn/dgamma': f32(dg), 'bn/dbeta': f32(db), 'bn/running_mean': f32(rm), 'bn/running_var': f32(rv)})
    # max-pool 3x3/2 SAME and 2x2/2 with ties
    for nm, (h, k, s) in {'mp3x3_s2': (16, 3, 2), 'mp2x2_s2': (8, 2, 2)}.items():
        x = f32(np.maximum(rng.standard_normal((2, h, h, 4)), 0))          # post-ReLU: many exact ties at 0
        y, arg = O.maxpool_fwd(x.astype(np.float64), k, s, 'SAME')
        dy = f32(rng.standard_normal(y.shape))
        out.update({nm + '/x': x, nm + '/y': f32(y), nm + '/arg': arg, nm + '/dy': dy,
                    nm + '/dx': f32(O.maxpool_bwd(dy.astype(np.float64), arg, x.shape, k, s, 'SAME'))})
    # softmax cross-entropy with an invalid (NaN) label
    logits = f32(rng.standard_normal((5, 10)) * 2)
    labels = np.array([3, 0, np.nan, 9, 7], dtype=np.float32)
    oh = O.one_hot_labels(labels, 10)
    pred, loss, ce, dl = O.softmax_xent_fwd_bwd(logits.astype(np.float64), oh.astype(np.float64))
    out.update({'xent/logits': logits, 'xent/labels': labels, 'xent/onehot': oh, 'xent/pred': f32(pred), 'xent/loss': np.float64(loss),
                'xent/ce': f32(ce), 'xent/dlogits': f32(dl)})
    # Nesterov + L2 + EMA, 3 steps
    w = f32(rng.standard_normal(64))
    ws, a, e = w.astype(np.float64), np.zeros(64), w.astype(np.float64)
    gs = f32(rng.standard_normal((3, 64)))
    traj = []
    for t in range(3):
        ws, a, e = O.sgd_nesterov_step(ws, gs[t].astype(np.float64), a, 0.1, 0.9, l2=1e-4, ema=e, ema_d=O.ema_decay(0.99, t))
        traj.append(np.stack([ws, a, e]))
    out.update({'sgd/w0': w, 'sgd/grads': gs, 'sgd/traj': np.stack(traj)})
    return out


def net_params(spec, seed=3):
    params, stats = ON.init_variables(spec.variables(), seed=seed, dtype=np.float32)
    rng = np.random.default_rng(9)
    for k in params:
        if k.endswith('gamma'):
            lo, span = (0.1, 0.2) if 'conv_2/bn' in k else (0.8, 0.4)
            params[k] = (lo + span * rng.random(params[k].shape)).astype(np.float32)
        if k.endswith('beta'):
            params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    return params, stats


def golden_net():
    """ResNet-50 topology, width/8, 10 classes, N=8, 64x64 (SURVEY §8c whole-net fixture, conditioned as tests/test_gpu_net.py)."""
    spec = ON.ResNetSpec.resnet50(10, 8)
    params, stats = net_params(spec)
    rng = np.random.default_rng(77)
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    out = {'param_checksum': np.array([sum(float(v.astype(np.float64).sum()) for v in params.values()),
                                       sum(float((v.astype(np.float64) ** 2).sum()) for v in params.values())])}
    keys = sorted(params)
    for step in range(2):
        x8 = rng.integers(0, 256, (8, 64, 64, 3)).astype(np.uint8)      # stored as bytes; x = x8/255 in float32
        x = f32(x8) / np.float32(255)
        y = rng.integers(0, 10, 8).astype(np.float32)
        t, logits, _, _, _ = ON.forward_loss(spec, state, x.astype(np.float64), y.astype(np.float64))
        loss, pred, grads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=8)
        p = 'step{}/'.format(step)
        out.update({p + 'x_u8': x8, p + 'y': y, p + 'logits': f32(logits.a), p + 'loss': np.float64(loss), p + 'pred': f32(pred),
                    p + 'argmax': pred.argmax(-1).astype(np.int32), p + 'grad_norms': np.array([np.linalg.norm(grads[k]) for k in keys]),
                    p + 'param_norms': np.array([np.linalg.norm(state.params[k]) for k in keys])})
        if step == 0:
            for k in ('block_0/conv_0/weights', 'block_2/res_0/conv_skip/weights', 'block_None/logits/weights'):
                out[p + 'grad/' + k] = f32(grads[k])
    out['keys'] = np.array(keys)
    out['final/block_4_mu'] = f32(state.stats['block_4/res_2/conv_2/bn/mu'])
    out['final/ema_logits_w'] = f32(state.ema['block_None/logits/weights'])
    return out


# name: (n, h, w, c, k, stride, padding)  — EfficientNet row
DW_CASES = {
    'dw3x3_s1': (2, 8, 8, 16, 3, 1, 'SAME'),
    'dw3x3_s2_even': (2, 12, 12, 24, 3, 2, 'SAME'),
    'dw5x5_s2_odd': (2, 9, 9, 16, 5, 2, 'SAME'),
    'dw5x5_s1': (1, 7, 7, 40, 5, 1, 'SAME'),
}


def effnet_params(spec, seed=4):
    params, stats = ON.init_variables(spec.variables(), seed=seed, dtype=np.float32)
    rng = np.random.default_rng(10)
    for k in params:
        if k.endswith('gamma'):
            lo, span = (0.1, 0.2) if 'conv_2/norm' in k else (0.8, 0.4)
            params[k] = (lo + span * rng.random(params[k].shape)).astype(np.float32)
        if k.endswith('beta') or k.endswith('biases'):
            params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    return params, stats


def golden_mbconv():
    """Depthwise conv, swish / sigmoid, squeeze-excite scale and two training steps of EfficientNet-B0 (width/2, depth/2,
    10 classes, N=8, 64x64) with stochastic depth and dropout masks stored in the fixture."""
    rng = np.random.default_rng(20261004)
    out = {}
    for name, (n, h, w, c, k, s, pad) in DW_CASES.items():
        x = f32(rng.standard_normal((n, h, w, c)))
        wt = f32(rng.standard_normal((k, k, c, 1)) / k)
        y = O.depthwise_conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), s, pad)
        dy = f32(rng.standard_normal(y.shape))
        out.update({name + '/x': x, name + '/w': wt, name + '/dy': dy, name + '/y': f32(y),
                    name + '/dx': f32(O.depthwise_conv2d_dgrad(dy.astype(np.float64), wt.astype(np.float64), x.shape, s, pad)),
                    name + '/dw': f32(O.depthwise_conv2d_wgrad(x.astype(np.float64), dy.astype(np.float64), wt.shape, s, pad))})
    x = f32(3 * rng.standard_normal((4, 40)))
    dy = f32(rng.standard_normal(x.shape))
    out.update({'act/x': x, 'act/dy': dy, 'act/swish': f32(O.swish_fwd(x.astype(np.float64))), 'act/swish_dx': f32(O.swish_bwd(dy.astype(np.float64), x.astype(np.float64))),
                'act/sigmoid': f32(O.sigmoid_fwd(x.astype(np.float64)))})
    x, m, dy = f32(rng.standard_normal((3, 5, 5, 16))), f32(rng.random((3, 1, 1, 16))), f32(rng.standard_normal((3, 5, 5, 16)))
    dx, dm = O.channel_scale_bwd(dy.astype(np.float64), x.astype(np.float64), m.astype(np.float64))
    out.update({'se/x': x, 'se/m': m, 'se/dy': dy, 'se/y': f32(O.channel_scale_fwd(x.astype(np.float64), m.astype(np.float64))), 'se/dx': f32(dx), 'se/dm': f32(dm)})
    # whole network
    spec = ON.EfficientNetSpec.b0(10, width_div=2, depth_div=2)
    spec.final_drop_rate = 0.3
    params, stats = effnet_params(spec)
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    out['net/param_checksum'] = np.array([sum(float(v.astype(np.float64).sum()) for v in params.values()),
                                          sum(float((v.astype(np.float64) ** 2).sum()) for v in params.values())])
    keys = sorted(params)
    units = [(name, dr) for name, k, s, cin, cout, mult, dr in spec.units() if s == 1 and cin == cout]
    for step in range(2):
        x8 = rng.integers(0, 256, (8, 64, 64, 3)).astype(np.uint8)
        x = f32(x8) / np.float32(255)
        y = rng.integers(0, 10, 8).astype(np.float32)
        surv = np.stack([(rng.random(8) >= dr).astype(np.float32) / np.float32(1.0 - dr) for _, dr in units])
        drop = (rng.random((8, spec.channels[-1])) >= 0.25).astype(np.float32) / np.float32(0.75)
        spec.survival = {name: surv[i].astype(np.float64) for i, (name, _) in enumerate(units)}
        spec.dropout_mask = drop.astype(np.float64)
        loss, pred, grads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=8)
        p = 'net/step{}/'.format(step)
        out.update({p + 'x_u8': x8, p + 'y': y, p + 'survival': surv, p + 'dropout': drop, p + 'loss': np.float64(loss), p + 'pred': f32(pred),
                    p + 'argmax': pred.argmax(-1).astype(np.int32), p + 'grad_norms': np.array([np.linalg.norm(grads[k]) for k in keys]),
                    p + 'param_norms': np.array([np.linalg.norm(state.params[k]) for k in keys])})
    out['net/keys'] = np.array(keys)
    out['net/units'] = np.array([u for u, _ in units])
    return out


if __name__ == '__main__':
    which = sys.argv[1:] or ['ops', 'resnet50_w8', 'mbconv']
    makers = {'ops': golden_ops, 'resnet50_w8': golden_net, 'mbconv': golden_mbconv}
    for w in which:
        np.savez_compressed(os.path.join(HERE, w + '.npz'), **makers[w]())
    for f in [w + '.npz' for w in which]:
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')
