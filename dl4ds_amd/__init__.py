"""dl4ds_amd -- MI355X-native drop-in for the conv-SR train-step path of carlos-gg/dl4ds.

Mirrors the reference's public names for that path (dl4ds/__init__.py:7-45): the constants below are
reproduced verbatim; ``models``, ``training``, ``inference``, ``losses`` and ``utils`` expose the same
builder / trainer signatures.  All arithmetic runs in libdl4ds_hip.so (hand-written gfx950 kernels);
there is no CPU fallback.
"""
__version__ = '0.1.0'

BACKBONE_BLOCKS = [
    'convnet',          # plain convolutional block w/o skip connections
    'resnet',           # residual convolutional blocks
    'densenet',         # dense convolutional blocks
    'convnext',         # convnext style residual blocks
    'unet']             # unet (encoder-decoder) backbone

UPSAMPLING_METHODS = [
    'spc',              # pixel shuffle or subpixel convolution in post-upscaling
    'rc',               # resize convolution in post-upscaling
    'dc',               # deconvolution or transposed convolution in post-upscaling
    'pin']              # pre-upsampling via (bicubic) interpolation
POSTUPSAMPLING_METHODS = ['spc', 'rc', 'dc']

INTERPOLATION_METHODS = ['inter_area', 'nearest', 'bicubic', 'bilinear', 'lanczos']

LOSS_FUNCTIONS = ['mae', 'mse', 'dssim', 'dssim_mae', 'dssim_mse', 'dssim_mae_mse',
                  'msdssim', 'msdssim_mae', 'msdssim_mae_mse']

DROPOUT_VARIANTS = ['vanilla', 'gaussian', 'spatial', 'mcdrop', 'mcgaussiandrop', 'mcspatialdrop']


def __getattr__(name):
    # lazy: importing the package must not need the GPU library (tests/test_abi.py runs on CPU)
    import importlib
    lazy = {'SupervisedTrainer': '.training', 'CGANTrainer': '.training', 'Predictor': '.inference', 'compute_metrics': '.metrics',
            'predict': '.inference', 'net_postupsampling': '.models', 'net_pin': '.models', 'unet_pin': '.models',
            'recnet_postupsampling': '.models', 'recnet_pin': '.models', 'residual_discriminator': '.models',
            'DataGenerator': '.dataloader', 'create_batch_hr_lr': '.dataloader', 'create_pair_hr_lr': '.dataloader',
            'crop_array': '.dataloader', 'resize_array': '.dataloader', 'checkarray_ndim': '.dataloader',
            'spatial_to_spatiotemporal_samples': '.utils', 'spatiotemporal_to_spatial_samples': '.utils',
            'checkarg_backbone': '.utils', 'checkarg_upsampling': '.utils', 'checkarg_loss': '.utils',
            'checkarg_dropout_variant': '.utils', 'check_compatibility_upsbackb': '.utils'}
    if name in lazy:
        return getattr(importlib.import_module(lazy[name], __name__), name)
    raise AttributeError(name)
