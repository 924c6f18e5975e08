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
