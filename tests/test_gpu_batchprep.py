"""On-device batch preparation (SURVEY section 8 "next" row f1) against the numpy port of the reference's host loop:
same seed -> same permutation, same crops, same batches (block means to fp32 rounding, copies bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields(n, h, w, c, seed):
    return np.random.default_rng(seed).standard_normal((n, h, w, c)).astype(np.float32)


CASES = [
    # upsampling, scale, H, W, C, predictors, n_static, patch, time_window, batch
    ('spc', 4, 32, 32, 1, None, 0, None, None, 4),
    ('spc', 4, 48, 64, 2, 3, 2, 16, None, 3),
    ('rc', 2, 40, 40, 1, 2, 1, 20, None, 5),
    ('pin', 4, 32, 48, 1, 2, 1, 20, None, 4),
    ('pin', 2, 24, 24, 3, None, 0, None, None, 2),
    ('spc', 4, 32, 32, 1, 1, 1, 16, 3, 2),
    ('pin', 2, 20, 20, 2, None, 1, 12, 4, 3),
]


@pytest.mark.parametrize('ups,scale,H,W,C,P,S,patch,tw,B', CASES)
def test_device_batches_equal_host_batches(ups, scale, H, W, C, P, S, patch, tw, B):
    from dl4ds_amd.dataloader import DataGenerator, DeviceDataGenerator
    n = 13
    hr = _fields(n, H, W, C, 1)
    preds = None if P is None else [_fields(n, H, W, P, 2)]
    stat = None if S == 0 else [np.random.default_rng(3 + i).standard_normal((H, W)).astype(np.float32) for i in range(S)]
    kw = dict(backbone='resnet', upsampling=ups, scale=scale, batch_size=B, patch_size=patch, time_window=tw,
              static_vars=stat, predictors=preds, interpolation='inter_area', seed=11)
    host = DataGenerator(hr, None, **kw)
    dev = DeviceDataGenerator(hr, None, **kw)
    assert len(dev) == len(host) and len(dev) >= 2
    for i in range(len(dev)):
        xs_h, ys_h = host[i]
        xs_d, ys_d = dev[i]
        assert len(xs_h) == len(xs_d)
        lr_d = xs_d[0].numpy()
        assert lr_d.shape == xs_h[0].shape, (lr_d.shape, xs_h[0].shape)
        np.testing.assert_allclose(lr_d, xs_h[0], rtol=0, atol=2e-6 * max(np.abs(xs_h[0]).max(), 1.0))
        np.testing.assert_array_equal(ys_d[0].numpy(), ys_h[0])             # crops are copies: bit-exact
        if len(xs_h) == 2:
            np.testing.assert_array_equal(xs_d[1].numpy(), xs_h[1])


def test_device_batches_feed_the_train_step_without_host_copies():
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    from dl4ds_amd.dataloader import DataGenerator, DeviceDataGenerator
    hr = _fields(16, 32, 32, 1, 5)
    kw = dict(backbone='resnet', upsampling='spc', scale=4, batch_size=4, seed=3)

    def run(gen_cls):
        model = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (8, 8), n_blocks=2, seed=2)
        eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
        gen = gen_cls(hr, None, **kw)
        losses = []
        for i in range(len(gen)):
            xs, ys = gen[i]
            if gen_cls is DeviceDataGenerator:
                losses.append(eng.step_device([xs[0].ptr], ys[0].ptr, 4, want_loss=True))
            else:
                losses.append(eng.step(xs, ys[0]))
        return losses
    l_host, l_dev = run(DataGenerator), run(DeviceDataGenerator)
    np.testing.assert_allclose(l_dev, l_host, rtol=2e-5)


ORACLE_CASES = [
    # upsampling, scale, H, W, C, n_pred, n_static, patch, time_window, batch
    ('spc', 4, 32, 32, 1, 0, 0, None, None, 4),
    ('spc', 4, 48, 64, 2, 0, 2, 16, None, 3),          # crop at any HR pixel, blocks aligned with the crop corner
    ('rc', 2, 40, 40, 1, 2, 1, 20, None, 5),           # predictors: corner drawn on the LR grid
    ('pin', 4, 32, 48, 1, 2, 1, 20, None, 4),          # whole field coarsened + replicated, then cropped anywhere
    ('pin', 2, 24, 24, 3, 0, 0, None, None, 2),
    ('spc', 4, 32, 32, 1, 1, 1, None, 3, 2),
    ('pin', 2, 20, 20, 2, 0, 1, 12, 4, 3),
]


@pytest.mark.parametrize('ups,scale,H,W,C,P,S,patch,tw,B', ORACLE_CASES)
def test_device_batches_equal_the_oracle(ups, scale, H, W, C, P, S, patch, tw, B):
    """dl4ds_batch_prepare against oracle/dataprep.py (the per-sample, per-pixel restatement of dataloader.py:11-360 with
    cv2.INTER_AREA's integer-ratio identities) -- NOT against the product's own host loader: same seeded permutation, same
    randint stream (y then x, upper bound exclusive), so both must produce the same crops and the same batches."""
    from dl4ds_amd.dataloader import DeviceDataGenerator
    from oracle import dataprep as O
    n = 13
    hr = _fields(n, H, W, C, 1)
    preds = None if P == 0 else _fields(n, H, W, P, 2)
    stat = None if S == 0 else [np.random.default_rng(3 + i).standard_normal((H, W)).astype(np.float32) for i in range(S)]
    dev = DeviceDataGenerator(hr, None, backbone='resnet', upsampling=ups, scale=scale, batch_size=B, patch_size=patch,
                              time_window=tw, static_vars=stat, predictors=None if preds is None else [preds],
                              interpolation='inter_area', seed=11)
    rng = np.random.default_rng(11)
    perm = rng.permutation(n - (tw or 0))
    np.testing.assert_array_equal(dev.indices, perm)
    assert len(dev) >= 2
    saw_unaligned = False
    for i in range(len(dev)):
        xo, yo, crops = O.create_batch_hr_lr(perm, i, hr, None, ups, scale=scale, batch_size=B, patch_size=patch,
                                             time_window=tw, static_vars=stat, predictors=preds,
                                             interpolation='inter_area', randint=lambda lo, hi: rng.integers(lo, hi))
        xd, yd = dev[i]
        lr_d = xd[0].numpy()
        assert lr_d.shape == xo[0].shape
        np.testing.assert_allclose(lr_d, xo[0], rtol=0, atol=2e-6 * max(np.abs(xo[0]).max(), 1.0))
        np.testing.assert_array_equal(yd[0].numpy(), yo[0])                 # HR crops are copies: bit-exact
        if len(xo) == 2:
            np.testing.assert_array_equal(xd[1].numpy(), xo[1])
        saw_unaligned |= any(c is not None and (c[0] % scale or c[1] % scale) for c in crops)
    if patch is not None and (ups == 'pin' or P == 0):
        assert saw_unaligned        # these paths crop at arbitrary pixels: the case the block alignment matters for


TAP_CASES = [
    # upsampling, scale, H, W, C, n_pred, n_static, patch, time_window, batch
    ('spc', 4, 32, 32, 1, 0, 0, None, None, 4),
    ('spc', 2, 48, 64, 2, 0, 2, 16, None, 3),          # scale 2: the cubic / Lanczos taps reach over the PATCH border (clamped there)
    ('rc', 2, 40, 40, 1, 2, 1, 20, None, 5),           # predictors: whole field resized, cropped on the LR grid (clamped at the FIELD border)
    ('pin', 4, 32, 48, 1, 2, 1, 20, None, 4),          # whole field down and up again, then cropped anywhere
    ('pin', 2, 24, 24, 3, 0, 0, None, None, 2),
    ('spc', 4, 32, 32, 1, 1, 1, None, 3, 2),
    ('pin', 2, 20, 20, 2, 0, 1, 12, 4, 3),
]


@pytest.mark.parametrize('interpolation', ['nearest', 'bilinear', 'bicubic', 'lanczos', 'inter_area'])
@pytest.mark.parametrize('ups,scale,H,W,C,P,S,patch,tw,B', TAP_CASES)
def test_device_batches_any_interpolation_equal_the_oracle(ups, scale, H, W, C, P, S, patch, tw, B, interpolation):
    """dl4ds_batch_prepare_taps (per-axis cv2 tap tables, csrc/batchprep.hip) against oracle/dataprep.py's per-pixel
    restatement of cv2.resize for every interpolation `resize_array` offers (utils.py:369-381)."""
    from dl4ds_amd.dataloader import DeviceDataGenerator
    from oracle import dataprep as O
    n = 9
    hr = _fields(n, H, W, C, 1)
    preds = None if P == 0 else _fields(n, H, W, P, 2)
    stat = None if S == 0 else [np.random.default_rng(3 + i).standard_normal((H, W)).astype(np.float32) for i in range(S)]
    dev = DeviceDataGenerator(hr, None, backbone='resnet', upsampling=ups, scale=scale, batch_size=B, patch_size=patch,
                              time_window=tw, static_vars=stat, predictors=None if preds is None else [preds],
                              interpolation=interpolation, seed=11, taps=True)
    rng = np.random.default_rng(11)
    perm = rng.permutation(n - (tw or 0))
    for i in range(len(dev)):
        xo, yo, _ = O.create_batch_hr_lr(perm, i, hr, None, ups, scale=scale, batch_size=B, patch_size=patch,
                                         time_window=tw, static_vars=stat, predictors=preds,
                                         interpolation=interpolation, randint=lambda lo, hi: rng.integers(lo, hi))
        xd, yd = dev[i]
        lr_d = xd[0].numpy()
        assert lr_d.shape == xo[0].shape
        np.testing.assert_allclose(lr_d, xo[0], rtol=0, atol=1e-5 * max(np.abs(xo[0]).max(), 1.0))
        np.testing.assert_array_equal(yd[0].numpy(), yo[0])
        if len(xo) == 2:
            np.testing.assert_array_equal(xd[1].numpy(), xo[1])
