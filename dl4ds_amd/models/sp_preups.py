"""net_pin / unet_pin -- same signatures as dl4ds/models/sp_preups.py:13-28,192-209."""
from ..graph import GraphBuilder, Model, resizable
from ..utils import checkarg_backbone, checkarg_dropout_variant
from .blocks import (conv_block, subpixel_block, resize_conv_block, deconv_block, pad_concat,
                     _reject_unsupported)
from .sp_postups import backbone_section, tail_section


@resizable('hr_size')
def net_pin(backbone_block, n_channels, n_aux_channels, hr_size, n_channels_out=1, n_filters=8, n_blocks=6,
            dropout_rate=0, dropout_variant=None, normalization=None, attention=False, activation='relu',
            output_activation=None, localcon_layer=False, seed=None):
    """sp_preups.py:83-189."""
    backbone_block = checkarg_backbone(backbone_block)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    h_hr, w_hr = int(hr_size[0]), int(hr_size[1])
    g = GraphBuilder()
    x_in = g.input(h_hr, w_hr, n_channels)
    s_in = g.input(h_hr, w_hr, n_aux_channels) if n_aux_channels > 0 else None
    x, nf = backbone_section(g, x_in, backbone_block, n_filters, n_blocks, activation, normalization,
                             attention, dropout_rate, dropout_variant)
    x = tail_section(g, x, s_in, n_filters, nf, n_channels_out, activation, output_activation,
                     normalization, dropout_rate, localcon_layer, convnext=(backbone_block == 'convnext'))
    g.finalize(x, seed)
    shapes = [(h_hr, w_hr, n_channels)] + ([(h_hr, w_hr, n_aux_channels)] if s_in is not None else [])
    return Model(g, backbone_block + '_pin', shapes)


def _check_nblocks(shape, power):
    """sp_preups.py:318-324."""
    while shape[0] // 2 ** power < 2 or shape[1] // 2 ** power < 2:
        print(f'`n_blocks` is too large, cannot downsample {power} times given the input grid size. '
              f'Setting `n_blocks` to {power-1}')
        power -= 1
    return power


@resizable('hr_size')
def unet_pin(backbone_block, n_channels, n_aux_channels, n_filters, n_blocks, hr_size, n_channels_out=1,
             activation='relu', dropout_rate=0, dropout_variant=None, normalization=None, attention=False,
             decoder_upsampling='rc', rc_interpolation='bilinear', output_activation=None, width_cap=256,
             localcon_layer=False, seed=None):
    """sp_preups.py:230-315."""
    backbone_block = checkarg_backbone(backbone_block)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    _reject_unsupported(normalization, dropout_rate, dropout_variant)
    blk = dict(activation=activation, normalization=normalization, attention=attention, dropout_variant=dropout_variant)
    n_blocks = _check_nblocks(hr_size, n_blocks)
    h_hr, w_hr = int(hr_size[0]), int(hr_size[1])
    g = GraphBuilder()
    x_in = g.input(h_hr, w_hr, n_channels)
    s_in = g.input(h_hr, w_hr, n_aux_channels) if n_aux_channels > 0 else None
    init_n_filters = n_filters
    x = x_in
    skips, nfl = [], []
    for i in range(n_blocks):
        # EncoderBlock: the reference's `droprate = dropout_rate if i == n_blocks else 0` never fires (sp_preups.py:255)
        y = conv_block(g, f'EncoderBlock{i+1}/conv', x, n_filters, dropout_rate=0, **blk)
        x = g.maxpool2(y, f'EncoderBlock{i+1}/maxpool')
        skips.append(y)
        nfl.append(n_filters)
        n_filters = min(width_cap, n_filters * 2)
    x = conv_block(g, 'Bottleneck', x, n_filters, activation=activation, dropout_rate=dropout_rate,
                   dropout_variant=dropout_variant, normalization=None)             # "following Isola et al 2016"
    nfl = nfl[::-1]
    for j, skip in enumerate(reversed(skips)):
        n_filters = nfl[j]
        if decoder_upsampling == 'spc':
            x = subpixel_block(g, f'SubpixelConvolution{j+1}', x, 2, n_filters)
        elif decoder_upsampling == 'rc':
            x = resize_conv_block(g, f'ResizeConvolution{j+1}', x, 2, n_filters, rc_interpolation)
        elif decoder_upsampling == 'dc':
            x = deconv_block(g, f'Deconvolution{j+1}', x, 2, n_filters, activation)
        else:
            raise ValueError(f'decoder_upsampling must be spc, rc or dc, got {decoder_upsampling}')
        x = pad_concat(g, f'Concatenate_SkipConnection{j+1}', x, skip)
        x = conv_block(g, f'DecoderConvBlock{j+1}', x, n_filters, dropout_rate=0, **blk)
    x = g.dropout(x, dropout_rate, 'decoder_dropout', dropout_variant)                # sp_preups.py:287
    x = tail_section(g, x, s_in, init_n_filters, n_filters, n_channels_out, activation, output_activation,
                     normalization, dropout_rate, localcon_layer)
    g.finalize(x, seed)
    shapes = [(h_hr, w_hr, n_channels)] + ([(h_hr, w_hr, n_aux_channels)] if s_in is not None else [])
    return Model(g, backbone_block + '_pin', shapes)
