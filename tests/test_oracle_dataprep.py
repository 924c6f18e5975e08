"""oracle/dataprep.py (dense per-pixel restatement of cv2.resize + dataloader.py:11-360) pinned by hand-computed values of
OpenCV's published formulas, and the product's host loader (dl4ds_amd/dataloader.py, separable gathers) against it."""
import numpy as np
import pytest

from oracle import dataprep as O
from dl4ds_amd import dataloader as D


def test_inter_area_integer_identities():
    """cv2.INTER_AREA at an integer ratio: down = block mean; up = pixel replication (OpenCV's area coefficients on the
    bilinear path give fx = k + 1 - s <= 0 for every sub-position k of an integer factor s)."""
    rng = np.random.default_rng(0)
    a = rng.random((8, 12, 2))
    down = O.cv2_resize(a, (3, 4), 'inter_area')                     # (size_x, size_y): 12 -> 3, 8 -> 4
    np.testing.assert_allclose(down, a.reshape(4, 2, 3, 4, 2).mean(axis=(1, 3)), rtol=1e-14)
    for s in (2, 3, 5):
        up = O.cv2_resize(down, (3 * s, 4 * s), 'inter_area')
        np.testing.assert_array_equal(up, np.repeat(np.repeat(down, s, axis=0), s, axis=1))
    # a non-integer up-scaling factor really interpolates: 2 -> 3 pixels, scale 2/3, dx = 1: sx = 0, fx = 2 - 1.5 = 0.5
    np.testing.assert_allclose(O.cv2_resize(np.array([[0.0, 1.0]]), (3, 1), 'inter_area'), [[0.0, 0.5, 1.0]])


def test_nearest_bilinear_bicubic_known_answers():
    row = np.array([[0.0, 10.0, 20.0, 30.0]])
    # INTER_NEAREST: sx = floor(dx * 4/8)
    np.testing.assert_array_equal(O.cv2_resize(row, (8, 1), 'nearest'), [[0, 0, 10, 10, 20, 20, 30, 30]])
    # INTER_LINEAR x2: centre (dx + 0.5) / 2 - 0.5 -> -0.25 (clamped to pixel 0), 0.25, 0.75, 1.25, ... last clamped
    np.testing.assert_allclose(O.cv2_resize(row, (8, 1), 'bilinear'), [[0, 2.5, 7.5, 12.5, 17.5, 22.5, 27.5, 30]])
    # INTER_LINEAR down x2 (no anti-aliasing): centres 0.5, 2.5 -> means of neighbours
    np.testing.assert_allclose(O.cv2_resize(row, (2, 1), 'bilinear'), [[5.0, 25.0]])
    # INTER_CUBIC: OpenCV's A = -0.75 (constants are reproduced, a ramp is NOT: only A = -0.5 has linear precision);
    # weights at t = 0.25 by hand
    np.testing.assert_allclose(O.cv2_resize(np.full((1, 8), 2.5), (16, 1), 'bicubic'), np.full((1, 16), 2.5), atol=1e-14)
    A, t = -0.75, 0.25
    w = [((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A, ((A + 2) * t - (A + 3)) * t * t + 1,
         ((A + 2) * (1 - t) - (A + 3)) * (1 - t) ** 2 + 1]
    w.append(1 - sum(w))
    np.testing.assert_allclose(w, [-0.10546875, 0.87890625, 0.26171875, -0.03515625])
    x = np.array([[3.0, -1.0, 4.0, 1.0, -5.0, 9.0]])
    np.testing.assert_allclose(O.cv2_resize(x, (12, 1), 'bicubic')[0, 5], np.dot(w, x[0, 1:5]))     # dx = 5: centre 2.25
    # left border: taps sx - 1 = -1 is clamped to pixel 0 (BORDER_REPLICATE): dx = 1, centre 0.25
    np.testing.assert_allclose(O.cv2_resize(x, (12, 1), 'bicubic')[0, 1], np.dot(w, x[0, [0, 0, 1, 2]]))


def test_lanczos4_known_answers():
    """INTER_LANCZOS4: eight taps sx-3..sx+4 of L(t) = sinc(t) sinc(t/4), normalised; centre tap alone when the fraction is 0."""
    x = np.array([[3.0, -1.0, 4.0, 1.0, -5.0, 9.0, 2.0, 6.0, -3.0, 5.0]])
    np.testing.assert_allclose(O.cv2_resize(np.full((1, 8), 2.5), (16, 1), 'lanczos'), np.full((1, 16), 2.5), atol=1e-14)
    # x2 up-scaling, dx = 9: centre 4.25 -> sx = 4, t = 0.25; weights by hand from the closed form
    t = 0.25
    arg = np.array([t + 3 - k for k in range(8)])
    w = np.sin(np.pi * arg) * np.sin(np.pi * arg / 4) / (np.pi ** 2 * arg ** 2 / 4)
    w /= w.sum()
    assert abs(w.sum() - 1) < 1e-15 and w[3] == w.max() and w[3] > 0.87 and w[2] < 0 and w[5] < 0
    np.testing.assert_allclose(O.cv2_resize(x, (20, 1), 'lanczos')[0, 9], np.dot(w, x[0, 1:9]), rtol=1e-13)
    # OpenCV's form: sin(y0 + 5 pi k / 4) / y_k^2 with y_k = -(t + 3 - k) pi / 4 -- the same eight values after normalisation
    yk = -(t + 3 - np.arange(8)) * np.pi / 4
    w_cv = np.sin(yk[0] + 5 * np.pi * np.arange(8) / 4) / yk ** 2
    np.testing.assert_allclose(w_cv / w_cv.sum(), w, rtol=1e-12)
    # borders: tap indices are clamped (dx = 1: centre 0.25, taps -3..4 -> 0,0,0,0,1,2,3,4)
    np.testing.assert_allclose(O.cv2_resize(x, (20, 1), 'lanczos')[0, 1], np.dot(w, x[0, [0, 0, 0, 0, 1, 2, 3, 4]]), rtol=1e-13)
    # an integer down-scaling by 2 samples at fraction 0.5: symmetric weights
    arg = np.array([0.5 + 3 - k for k in range(8)])
    w2 = np.sinc(arg) * np.sinc(arg / 4)
    w2 /= w2.sum()
    np.testing.assert_allclose(w2, w2[::-1], rtol=1e-13)
    np.testing.assert_allclose(O.cv2_resize(x, (5, 1), 'lanczos')[0, 2], np.dot(w2, x[0, 1:9]), rtol=1e-13)   # centre 4.5
    # same size: identity
    np.testing.assert_array_equal(O.cv2_resize(x, (10, 1), 'lanczos'), x)


def test_inter_area_with_one_growing_axis_takes_the_bilinear_path_on_both():
    """resize.cpp runs true INTER_AREA only when neither axis grows; otherwise BOTH axes use the bilinear path with the "area"
    coefficients sx = floor(dx s), fx = (dx + 1) - (sx + 1) / s clipped to [0, 1).  1 x 6 ramp -> 2 x 2 (x shrinks by 3, y doubles):
    fx = 2/3 at sx = 0 and 3 -> (0.667, 3.667), NOT the block means (1, 4) the same x axis yields when y does not grow."""
    a = np.arange(6.0)[None, :]
    np.testing.assert_allclose(O.cv2_resize(a, (2, 2), 'inter_area'), [[2 / 3, 3 + 2 / 3]] * 2, atol=1e-14)
    np.testing.assert_allclose(O.cv2_resize(a, (2, 1), 'inter_area'), [[1.0, 4.0]], atol=1e-14)
    np.testing.assert_allclose(O.cv2_resize(np.arange(4.0)[None, :], (2, 2), 'inter_area'), [[0.5, 2.5]] * 2, atol=1e-14)


def test_inter_area_non_integer_ratio_known_answers():
    """cv2.INTER_AREA at a non-integer ratio (resize.cpp computeResizeAreaTab): overlap-weighted means.  5 -> 2 pixels (scale 2.5):
    cell 0 = [0, 2.5) holds pixels 0 and 1 whole and half of pixel 2 -> weights (1, 1, .5) / 2.5; the ramp 0..4 gives (0.8, 3.2), the
    value OpenCV's documentation of the mode ("pixel area relation") implies and cv2 returns.  Rows sum to 1, a constant image stays
    constant, one integer and one non-integer axis are resampled independently, and a partial overlap below 1e-3 pixel is DROPPED
    (1001 -> 1000: cell 0 = [0, 1.001) keeps pixel 0 only, weight 1 / 1.001 -- OpenCV's table does not renormalise)."""
    W = O._axis_area_down(5, 2)
    np.testing.assert_allclose(W, [[0.4, 0.4, 0.2, 0, 0], [0, 0, 0.2, 0.4, 0.4]], atol=1e-15)
    np.testing.assert_allclose(O.cv2_resize(np.arange(5.0)[None, :], (2, 1), 'inter_area'), [[0.8, 3.2]], atol=1e-14)
    for ns, nd in ((7, 5), (100, 33), (512, 100), (9, 4)):
        Wn = O._axis_area_down(ns, nd)
        np.testing.assert_allclose(Wn.sum(1), 1.0, atol=1e-12)
        assert (Wn >= 0).all() and np.count_nonzero(Wn, axis=1).max() <= int(np.ceil(ns / nd)) + 1
        # every source pixel is used in full (column sums = n_dst / n_src each): the cells tile the axis
        np.testing.assert_allclose(Wn.sum(0), nd / ns, atol=1e-12)
    img = np.random.default_rng(0).standard_normal((9, 12))
    out = O.cv2_resize(img, (4, 4), 'inter_area')                      # x: 12 -> 4 (block means of 3), y: 9 -> 4 (ratio 2.25)
    np.testing.assert_allclose(out, O._axis_area_down(9, 4) @ img.reshape(9, 4, 3).mean(-1), atol=1e-14)
    np.testing.assert_allclose(O.cv2_resize(np.full((7, 10), 2.5), (3, 5), 'inter_area'), 2.5, atol=1e-14)
    W = O._axis_area_down(1001, 1000)
    assert np.count_nonzero(W[0]) == 1 and W[0, 0] == pytest.approx(1 / 1.001, rel=1e-12)
    assert np.count_nonzero(W[1]) == 2 and W[1, 1] == pytest.approx(0.999 / 1.001, rel=1e-9)


@pytest.mark.parametrize('interp', ['inter_area', 'nearest', 'bilinear', 'bicubic', 'lanczos'])
@pytest.mark.parametrize('shape,new', [((12, 8, 3), (4, 6)), ((6, 9, 1), (27, 12)), ((10, 10, 2), (10, 5)), ((4, 6), (18, 8)),
                                       ((10, 7, 2), (5, 3)), ((100, 50, 1), (16, 33)), ((9, 9), (4, 9))])
def test_product_resize_equals_oracle(interp, shape, new):
    a = np.random.default_rng(1).standard_normal(shape)
    sx, sy = new
    ref = O.resize_array(a, new, interp, squeezed=False)
    got = D.resize_array(a, new, interp, squeezed=False)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


CASES = [
    # upsampling, scale, H, W, C, n_pred, n_static, patch, time_window, lr_given, interpolation
    ('spc', 4, 32, 32, 1, 0, 0, None, None, False, 'inter_area'),
    ('spc', 4, 48, 64, 2, 0, 2, 16, None, False, 'inter_area'),        # HR crop at any pixel, the patch is coarsened
    ('rc', 2, 40, 40, 1, 2, 1, 20, None, False, 'inter_area'),         # predictors: crop drawn on the LR grid
    ('dc', 2, 24, 24, 1, 0, 1, 12, None, True, 'inter_area'),          # external LR array
    ('pin', 4, 32, 48, 1, 2, 1, 20, None, False, 'inter_area'),
    ('pin', 2, 24, 24, 3, 0, 0, None, None, False, 'bilinear'),
    ('pin', 4, 32, 32, 1, 1, 1, 16, None, False, 'bicubic'),
    ('pin', 2, 16, 16, 1, 0, 0, 8, None, True, 'nearest'),
    ('spc', 4, 32, 32, 1, 1, 1, None, 3, False, 'inter_area'),
    ('pin', 2, 20, 20, 2, 0, 1, 12, 4, False, 'inter_area'),
    ('pin', 4, 32, 48, 1, 2, 1, 20, None, False, 'lanczos'),
    ('rc', 2, 40, 40, 1, 2, 1, 20, None, False, 'lanczos'),            # patch border vs field border clamping differ
    ('spc', 2, 48, 64, 2, 0, 2, 16, None, False, 'bicubic'),
    ('rc', 2, 40, 40, 1, 2, 1, 20, None, False, 'bilinear'),
]


@pytest.mark.parametrize('ups,scale,H,W,C,P,S,patch,tw,lr_given,interp', CASES)
def test_product_batches_equal_oracle_batches(ups, scale, H, W, C, P, S, patch, tw, lr_given, interp):
    rng = np.random.default_rng(5)
    n = 9
    hr = rng.standard_normal((n, H, W, C)).astype(np.float32)
    lr_arr = None
    if lr_given:
        lr_arr = hr.reshape(n, H // scale, scale, W // scale, scale, C).mean(axis=(2, 4)).astype(np.float32)
    preds = None if P == 0 else rng.standard_normal((n, H, W, P)).astype(np.float32)
    stat = None if S == 0 else [rng.standard_normal((H, W)).astype(np.float32) for _ in range(S)]
    nidx = n - (tw or 0)
    idx = np.random.default_rng(1).permutation(nidx)
    r_prod, r_orac = np.random.default_rng(77), np.random.default_rng(77)
    for b in range(2):
        xp, yp = D.create_batch_hr_lr(idx, b, hr, lr_arr, ups, scale=scale, batch_size=3, patch_size=patch, time_window=tw,
                                      static_vars=stat, predictors=preds, interpolation=interp, rng=r_prod)
        xo, yo, crops = O.create_batch_hr_lr(idx, b, hr, lr_arr, ups, scale=scale, batch_size=3, patch_size=patch,
                                             time_window=tw, static_vars=stat, predictors=preds, interpolation=interp,
                                             randint=lambda lo, hi: r_orac.integers(lo, hi))
        assert len(xp) == len(xo)
        for a, o in zip(xp + yp, xo + yo):
            assert a.shape == o.shape and a.dtype == np.float32
            np.testing.assert_allclose(a, o, rtol=0, atol=1e-6)
        if patch is not None and ups in ('spc', 'rc', 'dc') and P == 0 and not lr_given:
            assert any(cy % scale or cx % scale for cy, cx in crops) or True      # any pixel is admissible
            for cy, cx in crops:
                assert 0 <= cy < H - patch and 0 <= cx < W - patch               # randint's upper bound is exclusive


def test_crop_corner_distribution_matches_np_random_randint():
    """utils.py:303-304: randint(0, n - size) never returns the last admissible corner n - size."""
    rng = np.random.default_rng(0)
    ys = {D.random_corner(10, 10, 6, rng)[0] for _ in range(400)}
    assert ys == {0, 1, 2, 3}
    assert D.random_corner(8, 8, 8, rng) == (0, 0)


def test_season_channels():
    """dataloader.py:224-245,508-542: four one-hot season channels appended to the auxiliary HR array and, for spatial
    samples, to the LR array; the month of a time window is the reference's `int(scipy.stats.mode(months).count)`."""
    assert [D.get_season(np.datetime64(f'2001-{m:02d}-15')) for m in (1, 4, 7, 10, 12)] == \
        ['winter', 'spring', 'summer', 'autumn', 'winter']
    assert D.get_season(np.array([7])) == O.get_season([7]) == 'summer'
    # a window of eight July days: mode count = 8 -> 'summer' by accident; five -> 'spring' (the reference's reading)
    july = np.array(['2001-07-%02d' % d for d in range(1, 9)], dtype='datetime64[D]')
    assert D.get_season(july, 8) == O.get_season([7] * 8, 8) == 'summer'
    assert D.get_season(july[:5], 5) == O.get_season([7] * 5, 5) == 'spring'
    np.testing.assert_array_equal(D.get_season_array('autumn', 2, 3)[..., 3], np.ones((2, 3)))
    assert D.get_season_array('autumn', 2, 3)[..., :3].sum() == 0
    with pytest.raises(ValueError):
        D.get_season_array('monsoon', 2, 2)

    rng = np.random.default_rng(0)
    hr = rng.random((16, 24, 1))
    topo = rng.random((16, 24))
    for ups in ('spc', 'pin'):
        got = D.create_pair_hr_lr(hr, None, ups, 4, None, static_vars=[topo], season='spring')
        ref = O.create_pair_hr_lr(hr, None, ups, 4, None, static_vars=[topo], season='spring')
        for a, b in zip(got, ref[:3]):
            np.testing.assert_allclose(a, b, rtol=1e-6)
        lr_hw = (4, 6) if ups == 'spc' else (16, 24)
        assert got[1].shape == lr_hw + (1 + 1 + 4,) and got[2].shape == (16, 24, 1 + 4)
        np.testing.assert_array_equal(got[1][..., 2:], D.get_season_array('spring', *lr_hw))
        np.testing.assert_array_equal(got[2][..., 1:], D.get_season_array('spring', 16, 24))
    # patches
    got = D.create_pair_hr_lr(hr, None, 'spc', 4, 8, static_vars=[topo], season='winter', rng=np.random.default_rng(1))
    assert got[0].shape == (8, 8, 1) and got[1].shape == (2, 2, 6) and got[2].shape == (8, 8, 5)
    # spatio-temporal samples without patches: auxiliary array only
    hr4 = rng.random((3, 16, 24, 1))
    got = D.create_pair_hr_lr(hr4, None, 'spc', 4, None, static_vars=[topo], season='summer')
    assert got[1].shape == (3, 4, 6, 1) and got[2].shape == (16, 24, 5)
    # like the reference: no season without static variables
    with pytest.raises(ValueError):
        D.create_pair_hr_lr(hr, None, 'spc', 4, None, season='summer')
    # through the generator (explicit option here; the reference's generator never passes time stamps)
    data = rng.random((6, 16, 24, 1)).astype('float32')
    stamps = np.array(['2001-01-01', '2001-04-01', '2001-07-01', '2001-10-01', '2001-12-01', '2001-02-01'], dtype='datetime64[D]')
    gen = D.DataGenerator(data, None, 'resnet', 'spc', 4, batch_size=3, static_vars=[topo], seed=3, time_metadata=stamps)
    (lr, aux), (hrb,) = gen[0]
    assert lr.shape == (3, 4, 6, 6) and aux.shape == (3, 16, 24, 5)
    for k, i in enumerate(gen.indices[:3]):
        want = D.SEASONS.index(D.get_season(stamps[i]))
        assert aux[k, 0, 0, 1 + want] == 1 and aux[k, ..., 1:].sum() == 16 * 24


@pytest.mark.parametrize('mode,tmode', [('bilinear', 'bilinear'), ('bicubic', 'bicubic'), ('nearest', 'nearest')])
@pytest.mark.parametrize('src,dst', [((8, 12), (16, 24)), ((8, 12), (20, 30)), ((9, 7), (31, 17)), ((16, 24), (8, 12)), ((15, 10), (6, 7)),
                                     ((5, 5), (5, 9))])
def test_resize_restatement_against_an_independent_implementation(mode, tmode, src, dst):
    """cv2 itself is not installable here (oracle/dataprep.py is pinned by OpenCV's published formulas above).  A second, independent
    implementation of the same three samplers IS in the image: torch.nn.functional.interpolate with align_corners=False and no
    anti-aliasing uses OpenCV's half-pixel centres (dx + 0.5) * scale - 0.5, index clamping at the borders (BORDER_REPLICATE), A = -0.75
    for the cubic and floor(dx * scale) for nearest -- written by other people from the same definitions.  Up- and down-scaling, integer
    and non-integer factors, one axis growing while the other shrinks (utils.py:369-381 calls cv2.resize with exactly these modes)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(src[0] * 100 + dst[1])
    a = rng.standard_normal(src + (3,))
    got = O.cv2_resize(a, (dst[1], dst[0]), mode)                       # (size_x, size_y)
    t = torch.from_numpy(a).permute(2, 0, 1)[None]
    kw = {} if tmode == 'nearest' else {'align_corners': False}
    ref = F.interpolate(t, size=dst, mode=tmode, **kw)[0].permute(1, 2, 0).numpy()
    assert got.shape == ref.shape == dst + (3,)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_inter_area_integer_shrink_against_average_pooling():
    """INTER_AREA at integer ratios = non-overlapping block means = torch's avg_pool2d (the LR fields of every configuration in
    BASELINE.json are made this way: dataloader.py:60-75 with the default interpolation)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    for (h, w), (sy, sx) in (((32, 48), (4, 4)), ((30, 20), (5, 2)), ((16, 16), (8, 8))):
        a = rng.standard_normal((h, w, 2))
        got = O.cv2_resize(a, (w // sx, h // sy), 'inter_area')
        ref = F.avg_pool2d(torch.from_numpy(a).permute(2, 0, 1)[None], (sy, sx))[0].permute(1, 2, 0).numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13)
