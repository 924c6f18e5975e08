"""Pin the oracle: analytic known-answer tests, numpy-vs-torch backend agreement,
independent torch library cross-checks."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ops as N
from oracle import torch_ops as T

rng = np.random.default_rng(0)


def r(*shape):
    return rng.standard_normal(shape)


def test_same_pad_asymmetric():
    # even size, stride 2, k=3: out=4, total=1 -> before 0, after 1 (extra bottom/right)
    assert N.same_pad(8, 3, 2) == (4, 0, 1)
    assert N.same_pad(7, 3, 2) == (4, 1, 1)
    assert N.same_pad(5, 3, 1) == (5, 1, 1)
    assert N.same_pad(16, 9, 2) == (8, 3, 4)


def test_conv2d_delta_kernel_identity_and_shift():
    x = r(2, 6, 7, 3)
    w = np.zeros((3, 3, 3, 3))
    for c in range(3):
        w[1, 1, c, c] = 1.0
    np.testing.assert_allclose(N.conv2d(x, w), x)
    w = np.zeros((3, 3, 3, 3))
    for c in range(3):
        w[0, 2, c, c] = 1.0        # y[h,w] = x[h-1, w+1]
    y = N.conv2d(x, w)
    np.testing.assert_allclose(y[:, 1:, :-1], x[:, :-1, 1:])
    assert np.all(y[:, 0] == 0) and np.all(y[:, :, -1] == 0)


@pytest.mark.parametrize('k,stride,hw', [(3, 1, (9, 8)), (1, 1, (5, 6)), (5, 1, (7, 9)),
                                         (3, 2, (8, 8)), (3, 2, (7, 9))])
def test_conv2d_np_vs_torch(k, stride, hw):
    x = r(2, hw[0], hw[1], 5)
    w = r(k, k, 5, 4)
    b = r(4)
    a = N.conv2d(x, w, b, stride=stride)
    t = T.to_numpy(T.conv2d(T.asarray(x), T.asarray(w), T.asarray(b), stride=stride))
    np.testing.assert_allclose(a, t, rtol=1e-10, atol=1e-10)
    av = N.conv2d(x, w, b, stride=stride, padding='valid')
    tv = T.to_numpy(T.conv2d(T.asarray(x), T.asarray(w), T.asarray(b), stride=stride, padding='valid'))
    np.testing.assert_allclose(av, tv, rtol=1e-10, atol=1e-10)


def test_conv2d_5d_folds_batch():
    x = r(2, 3, 5, 6, 4)
    w = r(3, 3, 4, 2)
    y = N.conv2d(x, w)
    for t in range(3):
        np.testing.assert_allclose(y[:, t], N.conv2d(x[:, t], w))


@pytest.mark.parametrize('stride', [2, 4])
def test_conv_transpose_is_adjoint_of_same_strided_conv(stride):
    # <conv_s(y), x> == <y, convT_s(x)>  with the SAME weights (HWOI <-> HWIO transpose)
    k, ci, co = 9, 3, 2
    x = r(2, 5, 6, ci)                    # lives on the small grid
    y = r(2, 5 * stride, 6 * stride, co)  # lives on the big grid
    w_t = r(k, k, co, ci)                 # HWOI for the transposed conv (maps ci -> co)
    up = N.conv2d_transpose(x, w_t, stride)
    assert up.shape == y.shape
    # forward conv big->small uses kernel HWIO with in=co, out=ci: w_t itself is (k,k,co,ci)
    down = N.conv2d(y, w_t, None, stride=stride)
    np.testing.assert_allclose((up * y).sum(), (down * x).sum(), rtol=1e-10)
    tt = T.to_numpy(T.conv2d_transpose(T.asarray(x), T.asarray(w_t), stride))
    np.testing.assert_allclose(up, tt, rtol=1e-10, atol=1e-10)


def test_depth_to_space_formula_and_not_pixel_shuffle():
    n, h, w, r_, cp = 2, 3, 4, 2, 3
    x = np.arange(n * h * w * r_ * r_ * cp, dtype=np.float64).reshape(n, h, w, r_ * r_ * cp)
    y = N.depth_to_space(x, r_)
    for i in range(r_):
        for j in range(r_):
            for c in range(cp):
                np.testing.assert_array_equal(y[:, i::r_, j::r_, c], x[..., (i * r_ + j) * cp + c])
    t = T.to_numpy(T.depth_to_space(T.asarray(x), r_))
    np.testing.assert_array_equal(y, t)
    # torch pixel_shuffle uses the CRD order -> must differ for cp>1
    ps = F.pixel_shuffle(torch.from_numpy(x).permute(0, 3, 1, 2), r_).permute(0, 2, 3, 1).numpy()
    assert not np.array_equal(ps, y)


def test_resize_bilinear_ramp_and_library():
    x = np.tile(np.arange(8, dtype=np.float64)[None, None, :, None], (1, 6, 1, 2))
    y = N.resize_bilinear(x, 24, 32)
    # interior of a linear ramp is reproduced exactly: value = (ox+0.5)/4-0.5
    ox = np.arange(32)
    expect = np.clip((ox + 0.5) / 4 - 0.5, 0, 7)
    np.testing.assert_allclose(y[0, 5, :, 0], expect, atol=1e-12)
    z = r(2, 5, 7, 3)
    a = N.resize_bilinear(z, 20, 28)
    t = T.to_numpy(T.resize_bilinear(T.asarray(z), 20, 28))
    lib = F.interpolate(torch.from_numpy(z).permute(0, 3, 1, 2), size=(20, 28), mode='bilinear',
                        align_corners=False).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(a, t, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a, lib, rtol=1e-10, atol=1e-10)


def test_maxpool_lcb_attention_backends():
    x = r(2, 7, 6, 4)
    np.testing.assert_array_equal(N.max_pool2(x), T.to_numpy(T.max_pool2(T.asarray(x))))
    assert N.max_pool2(x).shape == (2, 3, 3, 4)
    w, b = r(7, 6, 4, 2), r(7, 6, 2)
    np.testing.assert_allclose(N.locally_connected_1x1(x, w, b),
                               T.to_numpy(T.locally_connected_1x1(T.asarray(x), T.asarray(w), T.asarray(b))),
                               rtol=1e-12)
    w1, b1, w2, b2 = r(1, 1, 4, 1), r(1), r(1, 1, 1, 4), r(4)
    a = N.channel_attention(x, w1, b1, w2, b2)
    t = T.to_numpy(T.channel_attention(*(T.asarray(v) for v in (x, w1, b1, w2, b2))))
    np.testing.assert_allclose(a, t, rtol=1e-12)
    # zero weights: scale = sigmoid(b2)
    a0 = N.channel_attention(x, w1 * 0, b1, w2 * 0, b2 * 0)
    np.testing.assert_allclose(a0, 0.5 * x)


def test_conv_lstm_backends_and_first_step():
    x = r(2, 3, 5, 6, 2)
    f = 3
    K, U, b = r(3, 3, 2, 4 * f) * 0.3, r(3, 3, f, 4 * f) * 0.3, r(4 * f) * 0.1
    a = N.conv_lstm2d(x, K, U, b)
    t = T.to_numpy(T.conv_lstm2d(*(T.asarray(v) for v in (x, K, U, b))))
    np.testing.assert_allclose(a, t, rtol=1e-10, atol=1e-12)
    # t=0: h0=c0=0 -> c1 = i*tanh(zc), h1 = o*tanh(c1)
    z = N.conv2d(x[:, 0], K, b)
    c1 = N.hard_sigmoid(z[..., :f]) * np.tanh(z[..., 2 * f:3 * f])
    h1 = N.hard_sigmoid(z[..., 3 * f:]) * np.tanh(c1)
    np.testing.assert_allclose(a[:, 0], h1, rtol=1e-12)


def test_losses_known_answers():
    y = rng.random((2, 16, 17, 1))
    assert N.dssim(y, y) == pytest.approx(0.0, abs=1e-12)
    assert N.mae(y, y + 0.25) == pytest.approx(0.25)
    assert N.mse(y, y - 0.5) == pytest.approx(0.25)
    assert N.bce(np.ones((3, 1)), np.full((3, 1), 0.5)) == pytest.approx(np.log(2.0))
    p = rng.random((2, 16, 17, 1)) - 0.3        # negative minimum -> shift branch
    for fn in ('dssim', 'dssim_mae', 'dssim_mse', 'dssim_mae_mse', 'mae', 'mse'):
        a = getattr(N, fn)(y, p)
        t = float(getattr(T, fn)(T.asarray(y), T.asarray(p)))
        assert a == pytest.approx(t, rel=1e-10), fn
    g = N._gauss_kernel()
    assert g.shape == (11, 11) and g.sum() == pytest.approx(1.0)
    assert g[5, 5] == g.max()


def test_adam_first_step_is_lr_sign():
    w = r(10)
    g = np.sign(r(10)) * (0.5 + rng.random(10))   # |g| >> eps
    w1, m, v = N.adam_step(w, g, np.zeros(10), np.zeros(10), 1, 1e-3)
    np.testing.assert_allclose(w1, w - 1e-3 * np.sign(g), atol=1e-8)
    wt, mt, vt = T.adam_step(*(T.asarray(a) for a in (w, g, np.zeros(10), np.zeros(10))), 1, 1e-3)
    np.testing.assert_allclose(w1, T.to_numpy(wt), rtol=1e-12)
    assert N.piecewise_lr(100000, 1e5, 1e-3, 1e-4) == 1e-3
    assert N.piecewise_lr(100001, 1e5, 1e-3, 1e-4) == 1e-4


def test_activations_backends():
    x = r(50)
    for k in (None, 'relu', 'sigmoid', 'tanh', 'elu', 'leaky_relu', 'selu', 'gelu'):
        np.testing.assert_allclose(N.activation(x, k), T.to_numpy(T.activation(T.asarray(x), k)),
                                   rtol=1e-7, atol=1e-12, err_msg=str(k))


def test_msdssim_numpy_matches_torch_and_known_answers():
    """tf.image.ssim_multiscale restatement: numpy vs torch (independent conv / pooling code), identical images -> 0,
    odd sizes use the edge-repeating (SYMMETRIC) padding before the 2x2 average."""
    import torch
    from oracle import torch_ops as T
    rng = np.random.default_rng(4)
    for shp in [(2, 96, 100, 1), (1, 89, 93, 2)]:
        y = rng.standard_normal(shp)
        p = y + 0.3 * rng.standard_normal(shp)
        for name in ('msdssim', 'msdssim_mae', 'msdssim_mae_mse'):
            a = getattr(N, name)(y, p)
            b = float(getattr(T, name)(torch.tensor(y), torch.tensor(p)))
            assert a == pytest.approx(b, rel=1e-10)
        assert N.msdssim(y, y) == pytest.approx(0.0, abs=1e-12)
    x = np.arange(15, dtype=np.float64).reshape(1, 3, 5, 1)
    d = N._downsample2_symmetric(x)[0, :, :, 0]
    assert d.shape == (2, 3)
    assert d[1, 2] == pytest.approx(x[0, 2, 4, 0])          # corner: the edge pixel repeated four times
    assert d[0, 2] == pytest.approx((x[0, 0, 4, 0] + x[0, 1, 4, 0]) / 2)


def test_resize_bicubic_known_properties():
    """tf.image.resize(method='bicubic') restated (oracle/np_ops.py: Keys cubic A = -0.5, 1024-step weight table,
    out-of-image taps dropped and renormalised): identity at equal sizes, rows sum to 1, linear precision in the interior,
    hand-computed weights at the quarter positions of a x2 up-sampling, numpy and torch versions agree."""
    import torch
    from oracle import np_ops as N, torch_ops as T
    np.testing.assert_allclose(N.bicubic_axis_matrix(7, 7), np.eye(7), atol=1e-7)
    for inn, out in ((5, 10), (8, 20), (9, 27), (12, 5), (6, 15)):
        M = N.bicubic_axis_matrix(inn, out)
        np.testing.assert_allclose(M.sum(axis=1), 1.0, atol=1e-6)
    # x2: src = o/2 - 0.25 -> fractions 0.75 (even o) and 0.25 (odd o); Keys A = -0.5 weights at t = 0.25:
    A, t = -0.5, 0.25
    w = [((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A, ((A + 2) * t - (A + 3)) * t * t + 1,
         ((A + 2) * (1 - t) - (A + 3)) * (1 - t) ** 2 + 1, ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A]
    np.testing.assert_allclose(w, [-0.0703125, 0.8671875, 0.2265625, -0.0234375])
    M = N.bicubic_axis_matrix(8, 16)
    np.testing.assert_allclose(M[7, 2:6], w, atol=1e-7)              # output 7: src = 3.25 -> taps 2..5
    np.testing.assert_allclose(M[6, 1:5], w[::-1], atol=1e-7)        # output 6: src = 2.75 -> taps 1..4, mirrored weights
    # first output: src = -0.25 -> i0 = -1, taps -2..1: only 0 and 1 are inside; renormalised
    first = np.array([w[::-1][2], w[::-1][3]])
    np.testing.assert_allclose(M[0, :2], first / first.sum(), atol=1e-6)
    # A = -0.5 reproduces linear ramps away from the border
    ramp = np.arange(12, dtype=np.float64)[None, :, None, None] * np.ones((1, 1, 3, 1))
    up = N.resize_bicubic(ramp, 24, 3)[0, :, 0, 0]
    np.testing.assert_allclose(up[4:20], (np.arange(24) + 0.5)[4:20] / 2 - 0.5, atol=1e-5)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 5, 3))
    np.testing.assert_allclose(T.resize_bicubic(torch.tensor(x), 15, 8).numpy(), N.resize_bicubic(x, 15, 8), atol=1e-12)


def test_resize_scale_translate_known_properties():
    """tf.image.resize(method = lanczos3 / lanczos5 / gaussian / mitchellcubic, antialias=False) restated
    (oracle/np_ops.py::scale_translate_axis_matrix): identity at equal sizes, rows sum to 1, spans no wider than 2R + 1 and
    clamped into the image, hand-computed kernel values, mirror symmetry, numpy and torch versions agree."""
    import torch
    from oracle import np_ops as N, torch_ops as T
    radius = {'lanczos3': 3.0, 'lanczos5': 5.0, 'gaussian': 1.5, 'mitchellcubic': 2.0}
    for m, R in radius.items():
        # equal sizes: sample = o + 0.5, the centre tap has |x| = 0; the Lanczos / Mitchell / Gaussian neighbours at |x| = 1, 2, ..
        M = N.scale_translate_axis_matrix(7, 7, m)
        if m.startswith('lanczos'):
            np.testing.assert_allclose(M, np.eye(7), atol=2e-7)     # sin(pi k) = 0 at the integer offsets (to float32 of pi)
        for inn, out in ((5, 10), (8, 32), (9, 27), (12, 5), (6, 15)):
            M = N.scale_translate_axis_matrix(inn, out, m)
            np.testing.assert_allclose(M.sum(axis=1), 1.0, atol=1e-6)
            assert (M != 0).sum(axis=1).max() <= 2 * R + 1
            np.testing.assert_allclose(M, M[::-1, ::-1], atol=1e-6)                      # the sampling grid is symmetric
    # x2 up-sampling, output 7: sample = 3.75; mitchellcubic taps 2..5 at |x| = 1.25, 0.25, 0.75, 1.75
    def mitchell(x):
        return ((-7 / 18 * x + 2) * x - 10 / 3) * x + 16 / 9 if x >= 1 else ((7 / 6 * x - 2) * x) * x + 8 / 9
    w = np.array([mitchell(1.25), mitchell(0.25), mitchell(0.75), mitchell(1.75)])
    np.testing.assert_allclose(w.sum(), 1.0, atol=1e-12)            # the Mitchell-Netravali filter is a partition of unity
    np.testing.assert_allclose(N.scale_translate_axis_matrix(8, 16, 'mitchellcubic')[7, 2:6], w, atol=1e-6)
    # gaussian: sigma = 0.5, radius 1.5 -> taps with |x| < 1.5: 2, 3, 4 at |x| = 1.25, 0.25, 0.75
    g = np.exp(-np.array([1.25, 0.25, 0.75]) ** 2 / (2 * 0.25))
    np.testing.assert_allclose(N.scale_translate_axis_matrix(8, 16, 'gaussian')[7, 2:5], g / g.sum(), atol=1e-6)
    # lanczos3: taps 1..6 (|x| = 2.25, 1.25, 0.25, 0.75, 1.75, 2.75), closed form, normalised
    xs = np.array([2.25, 1.25, 0.25, 0.75, 1.75, 2.75])
    l3 = np.sinc(xs) * np.sinc(xs / 3)
    np.testing.assert_allclose(N.scale_translate_axis_matrix(8, 16, 'lanczos3')[7, 1:7], l3 / l3.sum(), atol=1e-6)
    # the first output (sample 0.25): the span ceil(0.25 - 3.5) .. floor(0.25 + 2.5) = -3 .. 2 is CLAMPED to 0 .. 2 (not replicated)
    xs = np.array([0.25, 1.25, 2.25])
    l3 = np.sinc(xs) * np.sinc(xs / 3)
    M = N.scale_translate_axis_matrix(8, 16, 'lanczos3')
    np.testing.assert_allclose(M[0, :3], l3 / l3.sum(), atol=1e-6)
    assert not M[0, 3:].any()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 5, 3))
    for m in radius:
        np.testing.assert_allclose(T.resize_scale_translate(torch.tensor(x), 15, 8, m).numpy(),
                                   N.resize_scale_translate(x, 15, 8, m), atol=1e-12)


# ---- third-party-PUBLISHED known answers: the numeric examples of the public TensorFlow / Keras API documentation -------------
# The reference's arithmetic is TensorFlow's, which cannot run here (tests/golden/README.md).  The docstrings of the TF / Keras
# functions the reference calls carry worked examples; those printed values are the only numbers available in this container
# that were produced by TensorFlow itself and not by this repository's author.  Each is checked on BOTH oracle backends.
def _both(fn_name, *args, **kw):
    a = getattr(N, fn_name)(*[np.asarray(v, np.float64) if isinstance(v, (list, np.ndarray)) else v for v in args], **kw)
    b = getattr(T, fn_name)(*[T.asarray(np.asarray(v, np.float64)) if isinstance(v, (list, np.ndarray)) else v for v in args], **kw)
    return np.asarray(a, np.float64), np.asarray(T.to_numpy(b), np.float64)


def test_tf_doc_hard_sigmoid():
    """tf.keras.activations.hard_sigmoid doc (TF 2.x): hard_sigmoid([-3, -1, 0, 1, 3]) = [0, 0.3, 0.5, 0.7, 1] -- the 0.2 x + 0.5
    form ConvLSTM2D's recurrent_activation default uses (blocks.py:350-355), NOT Keras-3's x / 6 + 0.5."""
    for out in _both('hard_sigmoid', [-3.0, -1.0, 0.0, 1.0, 3.0]):
        np.testing.assert_allclose(out, [0.0, 0.3, 0.5, 0.7, 1.0], atol=1e-12)


def test_tf_doc_activations():
    """tf.keras.activations.{gelu, tanh, sigmoid, relu} doc examples on x = [-3, -1, 0, 1, 3] (gelu: the exact erf form, the
    approximate=False default; the tanh approximation would print -0.00363752 ... 2.9963627)."""
    x = [-3.0, -1.0, 0.0, 1.0, 3.0]
    want = dict(gelu=[-0.00404951, -0.15865529, 0.0, 0.8413447, 2.9959507], tanh=[-0.9950547, -0.7615942, 0.0, 0.7615942, 0.9950547],
                relu=[0.0, 0.0, 0.0, 1.0, 3.0])
    for k, w in want.items():
        for out in _both('activation', x, k):
            np.testing.assert_allclose(out, w, rtol=2e-6, atol=3e-7, err_msg=k)      # (the doc's values are fp32 prints: gelu(-3) carries
            # the cancellation of 1 + erf(-2.12) in single precision, 1.8e-7 absolute)
    assert abs(want['gelu'][0] - (-0.00363752)) > 1e-4            # the doc's two forms are distinguishable at this tolerance
    for out in _both('activation', [-20.0, -1.0, 0.0, 1.0, 20.0], 'sigmoid'):      # tf.keras.activations.sigmoid doc
        np.testing.assert_allclose(out, [2.0611535e-09, 2.6894143e-01, 5.0e-01, 7.3105860e-01, 1.0], rtol=2e-6)


def test_tf_doc_depth_to_space():
    """tf.nn.depth_to_space doc, block_size 2, NHWC: the three worked examples (1x1x4 -> 2x2x1, 1x1x12 -> 2x2x3, 2x2x4 -> 4x4x1)."""
    for out in _both('depth_to_space', [[[[1, 2, 3, 4]]]], 2):
        np.testing.assert_array_equal(out, [[[[1], [2]], [[3], [4]]]])
    for out in _both('depth_to_space', [[[[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]]]], 2):
        np.testing.assert_array_equal(out, [[[[1, 2, 3], [4, 5, 6]], [[7, 8, 9], [10, 11, 12]]]])
    x = [[[[1, 2, 3, 4], [5, 6, 7, 8]], [[9, 10, 11, 12], [13, 14, 15, 16]]]]
    want = [[[[1], [2], [5], [6]], [[3], [4], [7], [8]], [[9], [10], [13], [14]], [[11], [12], [15], [16]]]]
    for out in _both('depth_to_space', x, 2):
        np.testing.assert_array_equal(out, want)


def test_tf_doc_losses():
    """tf.keras.losses doc examples: BinaryCrossentropy()([[0, 1], [0, 0]], [[0.6, 0.4], [0.4, 0.6]]) = 0.815;
    MeanAbsoluteError()([[0, 1], [0, 0]], [[1, 1], [1, 0]]) = 0.5; MeanSquaredError() on the same = 0.5."""
    yt, yp = [[0.0, 1.0], [0.0, 0.0]], [[0.6, 0.4], [0.4, 0.6]]
    for out in _both('bce', yt, yp):
        assert float(out) == pytest.approx(0.815, abs=5e-4)
    for fn in ('mae', 'mse'):
        for out in _both(fn, yt, [[1.0, 1.0], [1.0, 0.0]]):
            assert float(out) == pytest.approx(0.5, abs=1e-12), fn


def test_tf_doc_layer_normalization_and_max_pooling():
    """tf.keras.layers.LayerNormalization(axis=1) doc: rows of arange(10).reshape(5, 2) * 10 -> [-1, 1] each (to the eps = 1e-3
    inside the square root: 5 / sqrt(25 + 1e-3)); tf.keras.layers.MaxPooling2D(2, strides=2, 'valid') doc:
    [[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12]] -> [[6, 8]] (floor(3 / 2) = 1 row: the odd last row is dropped)."""
    data = (np.arange(10).reshape(5, 2) * 10.0).reshape(5, 1, 1, 2)
    for out in _both('layer_norm', data, np.ones(2), np.zeros(2)):
        np.testing.assert_allclose(out.reshape(5, 2), np.tile([-1.0, 1.0], (5, 1)), atol=3e-5)
        np.testing.assert_allclose(out.reshape(5, 2)[:, 1], 5.0 / np.sqrt(25.0 + 1e-3), rtol=1e-12)
    x = np.asarray([[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12]], np.float64).reshape(1, 3, 4, 1)
    for out in _both('max_pool2', x):
        np.testing.assert_array_equal(out.reshape(1, 2), [[6.0, 8.0]])
