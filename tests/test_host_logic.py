"""CPU tests of the host side: graph construction through the reference's block-method surface, fusion, the
reference's printed known answers, parameter naming, LR schedule, EMA rule, data sharding, bucket planner."""
import numpy as np
import pytest

import myconvnet_amd as M
from myconvnet_amd import dist as D
from oracle import net as ON
from oracle import ops as O


def test_resnet50_known_answers_and_names():
    net = M.ResNet50([224, 224, 3], 1000, batch_size=256, auto_compile=False)
    assert net.params == 25557032                       # reference prints this (convnet.py:212)
    assert net.conv_macs == 4087136256
    fc = [li for li in net.layer_info if li['name'] == 'block_None/logits'][0]
    assert fc['flops'] == 2048 * 1000 + 1000
    spec = ON.ResNetSpec.resnet50(1000)
    assert [(v.name, v.shape, v.kind if v.kind != 'gamma' else v.kind) for v in net._var_order] == \
        [(n, tuple(s), k.replace('gamma0', 'gamma')) for n, s, k in spec.variables()]
    # 16 zero-initialised gammas (models/resnet_v1_5.py:179-181)
    assert sum(1 for n, s, k in spec.variables() if k == 'gamma0') == 16
    assert net.block_list == [None, 0, 1, 2, 3, 4]      # None first: _init_model sets _curr_block = None (convnet.py:433)
    assert len(net.get_collection('weight_variables')) == 54
    assert len(net.get_collection('norm_statistics')) == 106
    d = net.d
    assert d['block_0'].shape == (256, 56, 56, 64) and d['block_4'].shape == (256, 7, 7, 2048)
    assert d['block_2/res_0/conv_1'].shape == (256, 28, 28, 128)      # v1.5: the 3x3 carries the stride
    assert d['logits'].shape == (256, 1000)


def test_same_padding_geometry_of_the_53_convs():
    net = M.ResNet50([224, 224, 3], 1000, batch_size=2, auto_compile=False)
    geoms = {n.scope: n.attrs['geom'] for n in net.graph.nodes if n.op == 'conv'}
    g = geoms['block_0/conv_0']
    assert (g.padT, g.padB, g.padL, g.padR, g.x_cs) == (2, 3, 2, 3, 4)
    g = geoms['block_2/res_0/conv_1']
    assert (g.KH, g.SH, g.padT, g.padB) == (3, 2, 0, 1)
    g = geoms['block_2/res_0/conv_skip']
    assert (g.KH, g.SH, g.padT, g.padB) == (1, 2, 0, 0)
    g = geoms['block_1/res_0/conv_1']
    assert (g.padT, g.padB) == (1, 1)
    assert len(geoms) == 53


def test_fusion_collapses_relu_and_residual_add():
    net = M.ResNet50([64, 64, 3], 10, batch_size=2, width_div=8, auto_compile=False)
    before = [n.op for n in net.graph.nodes]
    assert before.count('relu') == 49 and before.count('add') == 16 and before.count('bn') == 53
    net.graph.fuse()
    after = [n.op for n in net.graph.nodes]
    assert after.count('relu') == 0 and after.count('add') == 0 and after.count('bn') == 53
    fused = [n for n in net.graph.nodes if n.op == 'bn' and n.attrs.get('skip') is not None]
    assert len(fused) == 16 and all(n.attrs['act'] == 1 for n in fused)
    assert all(n.scope.endswith('conv_2/bn') for n in fused)
    # the block output is now produced by the fused bn
    assert net.d['block_1/res_0'].producer.op == 'bn'


def test_bf16_mode_and_channel_padding():
    net = M.ResNet50([32, 32, 3], 10, batch_size=2, width_div=8, half_precision=True, auto_compile=False)
    assert net.dtype == 'bfloat16' and net.X.cs == 8
    assert net.logits.dtype == 'float32'                 # cast after the head (convnet.py:477-480)
    assert [n.op for n in net.graph.nodes].count('cast') == 1


def test_vgg_trunk_shapes_config1():
    net = M.VGG16([8, 8, 3], 10, batch_size=4, backbone_only=True, auto_compile=False)
    assert net.d['block_0'].shape == (4, 4, 4, 64)
    assert net.d['block_4'].shape == (4, 1, 1, 512)
    assert [v.name for v in net._var_order][:2] == ['block_0/conv_0/weights', 'block_0/conv_0/biases']
    assert len(net._var_order) == 26
    with pytest.raises(AssertionError):
        M.VGG16([8, 8, 3], 10, batch_size=4, auto_compile=False)       # the head needs 224x224 (vggnet.py:108)


def test_unbuilt_features_fail_loudly():
    with pytest.raises(NotImplementedError):
        M.ResNet50([32, 32, 3], 10, batch_size=2, norm_type='group', auto_compile=False)
    with pytest.raises(NotImplementedError):
        M.ResNet50([32, 32, 3], 10, batch_size=2, dropout_rate=0.3, dropout_weights=True, auto_compile=False)
    net = M.ResNet50([32, 32, 3], 10, batch_size=2, dropout_rate=0.3, auto_compile=False)     # feature dropout is built
    assert [n.attrs['kind'] for n in net._random_nodes] == ['element']
    net = M.ResNet50([32, 32, 3], 10, batch_size=2, auto_compile=False)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        net.compile()


def test_efficientnet_b0_inventory_matches_reference_counts():
    """EfficientNet-B0: 5,288,548 parameters is the published count for the architecture the reference
    builds (models/efficientnet.py:10-17); variable names / shapes are the oracle's independent restatement."""
    from oracle import net as ON
    m = M.EfficientNetB0([224, 224, 3], 1000, batch_size=4, auto_compile=False, device='cpu')
    assert m.params == 5288548
    spec = ON.EfficientNetSpec.b0(1000)
    assert sorted((n, tuple(s)) for n, s, _ in spec.variables()) == sorted((v.name, tuple(v.shape)) for v in m._var_order)
    assert m.block_list == [None, 0, 1, 2, 3, 4, 5, 6, 7, 8]
    ops = [n.op for n in m.graph.nodes]
    assert ops.count('dwconv') == 16 and ops.count('chscale') == 16 and ops.count("conv") == 1 + 15 + 16 + 32 + 1
    b3 = M.EfficientNetB3([300, 300, 3], 1000, batch_size=2, auto_compile=False, device='cpu')
    assert b3.channels == [40, 24, 32, 48, 96, 136, 232, 384, 1536] and b3.conv_units == [None, 2, 3, 3, 5, 5, 6, 2, None]
    # stochastic depth: linearly increasing drop rate, one per-sample mask per residual unit
    sd = M.EfficientNetB0([64, 64, 3], 10, batch_size=4, final_drop_rate=0.2, auto_compile=False, device='cpu')
    rates = [n.attrs['rate'] for n in sd._random_nodes]
    assert len(rates) == 9 and rates[0] == pytest.approx(0.2 * 2 / 7) and rates[-1] == pytest.approx(0.2 * 6 / 7)


def test_deeplab_inventory_and_graph():
    """DeepLabv3+ / dilated ResNet / SegNet: variable names and shapes equal the oracle's independent
    restatement; 39,170,995 parameters for the ResNet-50 variant with 19 classes; per-pixel labels and loss."""
    from oracle import net as ON
    m = M.DeepLabV3PlusResNet50([513, 513, 3], 19, batch_size=2, auto_compile=False, device='cpu')
    spec = ON.DeepLabSpec(19)
    assert sorted((n, tuple(s)) for n, s, _ in spec.variables()) == sorted((v.name, tuple(v.shape)) for v in m._var_order)
    assert m.params == 39170995
    assert m.logits.shape == (2, 513, 513, 19) and m.Y.shape == (2, 513, 513, 19) and m._label_shape == (2, 513, 513)
    assert m.block_list == [None, 0, 1, 2, 3, 4, 5, 6]
    dil = sorted({(n.attrs['geom'].DH, n.attrs['geom'].SH) for n in m.graph.nodes if n.op == 'conv' and n.attrs['geom'].KH == 3})
    assert dil == [(1, 1), (1, 2), (2, 2), (4, 1), (6, 1), (8, 1), (12, 1), (18, 1)]      # multi-grid 2,4,8 in block 4 (stride 2 on the first), ASPP 6,12,18
    r101 = M.DeepLabV3PlusResNet([65, 65, 3], 5, batch_size=2, auto_compile=False, device='cpu')
    assert r101.res_units == [None, 3, 4, 23, 3] and r101.strides == [2, 1, 2, 2, 1]
    # SegNet label smoothing (segmentation/segnet.py:117-122): a 5x5 / stride 1 SAME average of the one-hot map feeds the loss beside the raw map
    ls = M.DeepLabV3PlusResNet50([65, 65, 3], 5, batch_size=2, label_smoothing=0.1, auto_compile=False, device='cpu')
    loss = [n for n in ls.graph.nodes if n.op == 'loss'][0]
    assert len(loss.inputs) == 3 and loss.inputs[1] is ls.Y and loss.attrs['label_smoothing'] == 0.1
    pool = loss.inputs[2].producer
    assert pool.op == 'avgpool' and pool.inputs[0] is ls.Y and (pool.attrs['kh'], pool.attrs['kw'], pool.attrs['sh'], pool.attrs['pt']) == (5, 5, 1, 2)


def test_lr_schedule_matches_reference_formulas():
    class Fake(M.Optimizer):
        def __init__(self, **kw):
            self.warmup_epoch = kw.get('warm', 1.0)
            self.decay_method = kw.get('method')
            self.decay_params = kw.get('params', (0.94, 2))
            self.steps_per_epoch = 100
            self.num_epochs = 10
            self.curr_step, self.curr_epoch, self.curr_multiplier = 0, 1, 1.0
    for method, params in [(None, (0.94, 2)), ('cosine', (0,)), ('poly', (2,)), ('exponential', (0.94, 2)), ('step', (0.1, 3, 6))]:
        f = Fake(method=method, params=params)
        for step in [0, 50, 99, 100, 450, 999]:
            f.curr_step = step
            f.curr_epoch = step // 100 + 1
            f._update_learning_rate()
            ref = O.lr_multiplier(step, 100, 10, 1.0, method, params, f.curr_epoch)
            assert f.curr_multiplier == pytest.approx(ref), (method, step)


def test_dataset_shards_like_the_reference():
    x = np.arange(40, dtype=np.float32).reshape(40, 1, 1, 1)
    y = np.arange(40, dtype=np.float32)
    a, b = M.DataSet(x, y, batch_size=8, num_shards=2), M.DataSet(x, y, batch_size=8, num_shards=2)
    xa, ya = a.next_batch(4, shard=0)
    xb, yb = b.next_batch(4, shard=1)
    assert list(ya) == [0, 1, 2, 3] and list(yb) == [4, 5, 6, 7]
    _, ya = a.next_batch(4, shard=0)
    assert list(ya) == [8, 9, 10, 11]
    xs, ys = M.synthetic(16, (8, 8, 3), 10)
    assert xs.dtype == np.float32 and 0 <= xs.min() and xs.max() < 1 and ys.max() < 10


def test_bucket_planner_covers_every_gradient_once_in_completion_order():
    variables = [('w%d' % i, i * 100, 100) for i in range(10)]            # creation order
    ready = {'w%d' % i: (9 - i) * 3 + 2 for i in range(10)}               # backward finishes the last layer first
    plan = D.plan_buckets(variables, ready, 250 * 4)
    covered = sorted(r for _, spans in plan for r in spans)
    assert covered == [(100, 400), (400, 700), (700, 1000), (0, 100)][::1] or sum(e - s for s, e in covered) == 1000
    assert sum(e - s for _, spans in plan for s, e in spans) == 1000
    idxs = [i for i, _ in plan]
    assert idxs == sorted(idxs)
    assert plan[0][1] == [(700, 1000)] and plan[0][0] == ready['w7']     # bucket fires once its slowest member is done
    for _, spans in plan:
        assert len(spans) == 1                                           # reverse-order layers are contiguous ranges


def test_accuracy_evaluator_matches_reference_rule():
    ev = M.AccuracyEvaluator()
    y_true = O.one_hot_labels(np.array([1, 2, np.nan, 0]), 3)
    y_pred = np.array([[0.1, 0.8, 0.1], [0.5, 0.2, 0.3], [0.3, 0.3, 0.4], [0.9, 0.05, 0.05]])
    assert ev.score(y_true, y_pred) == pytest.approx(O.accuracy_score(y_true, y_pred)) == pytest.approx(0.75)


def test_dataset_shards_visit_every_example_exactly_once():
    """Shard-per-device contract (reference dataset.py:113-129): global batch k = examples [k*B, (k+1)*B), rank r reads
    its r-th contiguous piece; ranks advance the cursor in lock step, so one pass touches every example exactly once and
    the rank-order concatenation (Y_all, convnet.py:504) is the dataset order — what predict() relies on."""
    x = np.arange(24, dtype=np.float32).reshape(24, 1)
    y = np.arange(24, dtype=np.float32)
    world, b = 2, 4
    sets = [M.DataSet(x, y, batch_size=b * world, num_shards=world) for _ in range(world)]     # one DataSet object per process
    seen = []
    for step in range(3):
        parts = [sets[r].next_batch(b, shard=r)[1] for r in range(world)]
        seen.append(np.concatenate(parts))
        np.testing.assert_array_equal(seen[-1], np.arange(step * 8, step * 8 + 8))
    assert sorted(np.concatenate(seen).tolist()) == list(range(24))


def test_loss_scale_rule_and_relower_contract():
    """optimizers.py:102-111 scales only for a factor > 1; compile() twice must not reallocate the stores (host-side check
    of the split: _allocate runs once)."""
    import inspect
    src = inspect.getsource(M.ConvNet.compile)
    assert '_allocated' in src and 'initialize_variables' not in src


def test_build_identity_covers_every_kernel_source(tmp_path):
    """wino_kernels.h was in neither source_digest() nor the mtime check.  Every file under csrc/ must be part of the
    build identity, every `#include "..."` must resolve into it, and touching any one file must change the digest."""
    import importlib.util
    import os
    import re
    import shutil
    pkg = os.path.dirname(M.__file__)
    spec = importlib.util.spec_from_file_location('_mcn_build_t', os.path.join(pkg, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    covered = {os.path.normpath(os.path.join(mod.CSRC, f)) for f in mod.SOURCES + mod.HEADERS}
    on_disk = {os.path.normpath(os.path.join(mod.CSRC, f)) for f in os.listdir(mod.CSRC) if os.path.isfile(os.path.join(mod.CSRC, f))}
    assert on_disk <= covered, sorted(on_disk - covered)
    assert os.path.normpath(os.path.join(pkg, '..', 'include', 'mcn.h')) in covered
    for f in sorted(on_disk):
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(f).read(), re.M):
            assert os.path.normpath(os.path.join(os.path.dirname(f), inc)) in covered, (f, inc)
    # a one-byte edit of ANY covered file changes the digest (run on a copy of the tree's csrc/ + include/)
    root = tmp_path / 'pkg'
    shutil.copytree(mod.CSRC, root / 'csrc', ignore=shutil.ignore_patterns('_obj'))
    os.makedirs(tmp_path / 'include')
    shutil.copy(os.path.join(pkg, '..', 'include', 'mcn.h'), tmp_path / 'include' / 'mcn.h')
    shutil.copy(os.path.join(pkg, 'build.py'), root / 'build.py')
    spec = importlib.util.spec_from_file_location('_mcn_build_c', str(root / 'build.py'))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    base = cp.source_digest()
    assert base == mod.source_digest()
    for f in cp.SOURCES + cp.HEADERS:
        path = os.path.join(cp.CSRC, f)
        orig = open(path, 'rb').read()
        open(path, 'ab').write(b'\n')
        assert cp.source_digest() != base, f
        open(path, 'wb').write(orig)
    assert cp.source_digest() == base
