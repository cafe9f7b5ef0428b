"""CPU: the Winograd identities the fp32 3x3 kernels implement (csrc/wino_kernels.h), restated in float64 (oracle/ops.py) and checked against
the direct restatement of the convolution — transform matrices, tile / padding bookkeeping on even and odd maps, the rotated-filter form of
the gradient w.r.t. the input, and the F(3x3, 2x2) weight gradient."""
import numpy as np
import pytest

from oracle import ops as O

RNG = np.random.default_rng(77)


@pytest.mark.parametrize('shape', [(2, 8, 8, 5, 7), (3, 7, 7, 4, 6), (1, 1, 1, 3, 2), (2, 9, 12, 6, 3), (2, 14, 10, 8, 8)])
def test_winograd_restatement_matches_direct_convolution(shape):
    n, h, w, c, k = shape
    x = RNG.standard_normal((n, h, w, c))
    wt = RNG.standard_normal((3, 3, c, k))
    dy = RNG.standard_normal((n, h, w, k))
    y = O.conv2d_fwd(x, wt, 1, 'SAME', 1)
    np.testing.assert_allclose(O.winograd_conv2d_fwd(x, wt), y, rtol=0, atol=1e-12 * np.abs(y).max())
    # gradient w.r.t. the input = the same convolution of dy with the filter rotated by 180 degrees and the channel roles swapped
    dx = O.conv2d_dgrad(dy, wt, x.shape, 1, 'SAME', 1)
    wr = wt[::-1, ::-1].transpose(0, 1, 3, 2)
    np.testing.assert_allclose(O.winograd_conv2d_fwd(dy, wr), dx, rtol=0, atol=1e-12 * np.abs(dx).max())
    dw = O.conv2d_wgrad(x, dy, wt.shape, 1, 'SAME', 1)
    np.testing.assert_allclose(O.winograd_conv2d_wgrad(x, dy), dw, rtol=0, atol=1e-12 * np.abs(dw).max())


def test_winograd_transform_matrices():
    """F(2, 3) in one dimension: y_i = sum_k d_{i+k} g_k for i = 0, 1 equals A^T[(G g) (.) (B^T d)] for every d, g"""
    d, g = RNG.standard_normal(4), RNG.standard_normal(3)
    want = np.array([d[0] * g[0] + d[1] * g[1] + d[2] * g[2], d[1] * g[0] + d[2] * g[1] + d[3] * g[2]])
    np.testing.assert_allclose(O.WINO_AT @ ((O.WINO_G @ g) * (O.WINO_BT @ d)), want, atol=1e-14)
    assert np.count_nonzero(O.WINO_BT) == 8 and set(np.unique(O.WINO_BT)) <= {-1.0, 0.0, 1.0}      # the input transform is additions only
