"""conv_layer(depthwise=True) with a channel multiplier != 1 and / or a bias (convnet.py:1634-1650, 1678-1694) through the layer API against
oracle.ops — shared by the GPU test (tests/test_gpu_mbconv_ops.py) and the host-code case on libmcn_cpu.so (tests/cpu_lib_cases.py)."""
import numpy as np

from oracle import ops as O


def _dw_mult_model(M, shape, k, stride, mult, biased, dtype, **kw):
    """input -> 1x1 conv (to a chunk multiple of channels) -> depthwise conv with channel multiplier [+ bias]: the reference's conv_layer(depthwise=True,
    out_channels=cin * mult, biased=...) call (convnet.py:1634-1650, 1678-1694)"""
    class Net(M.ConvNet):
        def _init_params(self, **kwargs):
            pass

        def _build_model(self):
            d = dict()
            self._curr_block = 0
            with self.variable_scope('block_0'):
                x = self.conv_layer(self.X, 1, 1, out_channels=8, biased=False, scope='pre')
                x = self.conv_layer(x, k, stride, out_channels=8 * mult, depthwise=True, biased=biased, scope='dw')
            d['block_0'] = x
            return d
    n, h, w_, _ = shape
    return Net([h, w_, 3], 10, batch_size=n, backbone_only=True, half_precision=(dtype != 'float32'),
               half_precision_dtype=(dtype if dtype != 'float32' else 'bfloat16'), num_gpus=1, **kw)



def run_case(M, case, dtype, check, q, **model_kw):
    import torch
    shape, k, stride, mult, biased = case
    model = _dw_mult_model(M, shape, k, stride, mult, biased, dtype, **model_kw)
    names = [getattr(fn, '__name__', '') for fn, _ in model._train_low.fwd.calls]
    assert ('mcn_channel_repeat_fwd' in names) == (mult != 1) and ('mcn_channel_affine' in names) == biased and 'mcn_dwconv2d_fwd' in names
    rng = np.random.default_rng(5 + k + mult)
    wp = (rng.standard_normal((1, 1, 3, 8)) * 0.7).astype(np.float32)
    wd = (rng.standard_normal((k, k, 8, mult)) / k).astype(np.float32)
    vals = {'block_0/pre/weights': wp, 'block_0/dw/weights': wd}
    if biased:
        bd = (0.5 * rng.standard_normal(8 * mult)).astype(np.float32)
        vals['block_0/dw/biases'] = bd
    assert set(model.variables) == set(vals)
    assert model.variables['block_0/dw/weights'].shape == (k, k, 8, mult)
    model.set_variables(vals)
    x = rng.random(shape).astype(np.float32)
    model.feed(x, np.zeros(shape[0], np.float32))
    model.forward(train=True)
    out = model.d['block_0']
    qf = (lambda a: q(a, dtype).astype(np.float64))
    x0 = qf(O.input_prep(x.astype(np.float64)))
    a1 = qf(O.conv2d_fwd(x0, qf(wp), 1, 'SAME', 1))
    a2 = qf(O.depthwise_conv2d_fwd(a1, qf(wd), stride, 'SAME', 1))
    ref = qf(O.bias_add_fwd(a2, bd.astype(np.float64))) if biased else a2
    got = model.fetch(out)
    assert got.shape == ref.shape and got.shape[-1] == 8 * mult
    check(got, ref, dtype, 'depthwise x{} {} forward'.format(mult, '+ bias' if biased else ''))
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    out.grad.copy_(torch.as_tensor(dy).to(out.grad.dtype).to(out.grad.device))
    model.backward()
    dyq = qf(dy)
    grads = model.get_variables('grad')
    if biased:
        check(grads['block_0/dw/biases'], O.bias_add_bwd(dyq), 'float32', 'bias gradient', rel=1e-4 if dtype == 'float32' else 2e-3)
    check(grads['block_0/dw/weights'], O.depthwise_conv2d_wgrad(a1, dyq, wd.shape, stride, 'SAME', 1), 'float32', 'depthwise filter gradient', rel=1e-4 if dtype == 'float32' else 2e-3)
    da1 = qf(O.depthwise_conv2d_dgrad(dyq, qf(wd), a1.shape, stride, 'SAME', 1))
    check(grads['block_0/pre/weights'], O.conv2d_wgrad(x0, da1, wp.shape, 1, 'SAME', 1), 'float32', 'gradient behind the channel repeat',
          rel=1e-4 if dtype == 'float32' else 2e-2, mx=1e-3 if dtype == 'float32' else 5e-2)


CASES = [((3, 9, 11, 3), 3, 1, 2, True), ((2, 12, 12, 3), 5, 2, 3, True), ((2, 8, 8, 3), 3, 1, 1, True), ((2, 10, 7, 3), 3, 2, 4, False)]
