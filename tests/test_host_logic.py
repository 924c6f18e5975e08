"""CPU tests of the host-side mirror of the reference interface: argument validation (same exception
types as dl4ds/utils.py, training/base.py), dataloader shapes, sharding, LR schedule plumbing."""
import numpy as np
import pytest

import dl4ds_amd
from dl4ds_amd import utils as U
from dl4ds_amd.dataloader import DataGenerator, create_pair_hr_lr, resize_array, crop_array
from dl4ds_amd import parallel


def test_constants_match_reference():
    assert dl4ds_amd.BACKBONE_BLOCKS == ['convnet', 'resnet', 'densenet', 'convnext', 'unet']
    assert dl4ds_amd.UPSAMPLING_METHODS == ['spc', 'rc', 'dc', 'pin']
    assert dl4ds_amd.POSTUPSAMPLING_METHODS == ['spc', 'rc', 'dc']
    assert 'dssim_mae_mse' in dl4ds_amd.LOSS_FUNCTIONS and len(dl4ds_amd.LOSS_FUNCTIONS) == 9


def test_checkargs_raise_like_the_reference():
    with pytest.raises(TypeError):
        U.checkarg_backbone(3)
    with pytest.raises(ValueError):
        U.checkarg_backbone('vgg')
    with pytest.raises(TypeError):
        U.checkarg_upsampling(None)
    with pytest.raises(ValueError):
        U.checkarg_upsampling('bicubic')
    with pytest.raises(ValueError):
        U.check_compatibility_upsbackb('unet', 'spc', None)
    with pytest.raises(ValueError):
        U.check_compatibility_upsbackb('unet', 'pin', 4)
    assert U.check_compatibility_upsbackb('resnet', 'spc', None) == ('resnet', 'spc')
    with pytest.raises(ValueError):
        U.checkarg_loss('huber')
    with pytest.raises(TypeError):
        U.checkarg_loss(None)
    assert U.checkarg_loss('dssim_mae') == 'dssim_mae'
    assert U.checkarg_loss('msdssim_mae_mse') == 'msdssim_mae_mse'
    with pytest.raises(ValueError):
        U.checkarg_dropout_variant('bernoulli')


def test_trainer_constructor_validation():
    from dl4ds_amd.training import SupervisedTrainer, CGANTrainer
    hr = np.random.rand(6, 16, 16, 1).astype(np.float32)
    with pytest.raises(TypeError):
        SupervisedTrainer('resnet', 'spc', [1, 2, 3], hr, hr, scale=4)
    with pytest.raises(ValueError):
        SupervisedTrainer('resnet', 'spc', hr[..., 0], hr, hr, scale=4)          # 3-D data
    with pytest.raises(ValueError):
        SupervisedTrainer('resnet', 'spc', hr, hr, hr, scale=5)                  # 16 % 5 != 0
    with pytest.raises(TypeError):
        SupervisedTrainer('resnet', 'spc', hr, hr, hr, scale=4, predictors_train=hr)
    with pytest.raises(ValueError):
        SupervisedTrainer('resnet', 'spc', hr, hr, hr, scale=4, data_train_lr=np.zeros((6, 8, 8, 1)))   # wrong scale
    with pytest.raises(ValueError):
        SupervisedTrainer('resnet', 'spc', hr, hr, hr, scale=4, device='CPU')
    with pytest.raises(ValueError):
        CGANTrainer('resnet', 'spc', hr, hr, scale=4)                            # needs static vars (reference defect)
    t = SupervisedTrainer('resnet', 'spc', hr, hr, hr, scale=4, batch_size=2, learning_rate=(1e-3, 1e-4))
    assert t.lossf == 'mae' and not t.model_is_spatiotemporal and t.running_on_first_worker


def test_dataloader_shapes_and_block_mean():
    rng = np.random.default_rng(0)
    hr = rng.random((16, 16, 1))
    h, l = create_pair_hr_lr(hr, None, 'spc', 4, None)
    assert h.shape == (16, 16, 1) and l.shape == (4, 4, 1) and h.dtype == np.float32
    np.testing.assert_allclose(l[0, 0, 0], hr[:4, :4, 0].mean(), rtol=1e-6)       # INTER_AREA at integer ratio
    h, l, s = create_pair_hr_lr(hr, None, 'pin', 4, 8, static_vars=[rng.random((16, 16))], rng=rng)
    assert h.shape == (8, 8, 1) and l.shape == (8, 8, 2) and s.shape == (8, 8, 1)
    assert resize_array(rng.random((8, 8, 2)), (16, 16), 'bilinear').shape == (16, 16, 2)
    with pytest.raises(ValueError):
        resize_array(hr, (4, 4), 'spline')
    with pytest.raises(ValueError):
        crop_array(hr, 32)
    with pytest.raises(ValueError):
        DataGenerator(rng.random((8, 16, 16, 1)), None, 'resnet', 'spc', 4, patch_size=10)


def test_rank_sharding_is_a_partition():
    """Disjoint, equally long shards covering all but the n % world remainder (unequal step counts would leave ranks
    waiting in the gradient all-reduce)."""
    n, world = 37, 4
    parts = [parallel.shard_indices(n, r, world, seed=5, epoch=2) for r in range(world)]
    allidx = np.concatenate(parts)
    assert len(set(allidx.tolist())) == len(allidx) == 36 and {len(p) for p in parts} == {9}
    gens = [DataGenerator(np.zeros((n, 8, 8, 1)), None, 'resnet', 'spc', 4, batch_size=2, seed=9, rank=r, world=world)
            for r in range(world)]
    seen = np.concatenate([g.indices for g in gens])
    assert len(set(seen.tolist())) == len(seen) == 36 and {len(g) for g in gens} == {4}
    assert sorted(parallel.shard_indices(n, 0, 1, seed=5).tolist()) == list(range(n))


def test_lr_schedule_plumbing():
    from dl4ds_amd.training.engine import _lr_schedule
    assert _lr_schedule((1e-3, 1e-4), 1e5) == (1e-3, 1e-4, 1e5)
    assert _lr_schedule(2e-4, 1e5)[:2] == (2e-4, 2e-4)
    assert _lr_schedule([5e-4], 10)[:2] == (5e-4, 5e-4)


def test_spatiotemporal_sample_helpers_round_trip():
    """utils.py:20-45: windows of consecutive frames and their collapse back into the frame sequence."""
    a = np.arange(7 * 2 * 3 * 1, dtype=np.float64).reshape(7, 2, 3, 1)
    w = U.spatial_to_spatiotemporal_samples(a, 3)
    assert w.shape == (5, 3, 2, 3, 1) and np.array_equal(w[2, 1], a[3])
    np.testing.assert_array_equal(U.spatiotemporal_to_spatial_samples(w, 3), a)
    with pytest.raises(ValueError):
        U.spatiotemporal_to_spatial_samples(w, 4)
