"""net_postupsampling -- same signature as dl4ds/models/sp_postups.py:14-32, graph per :95-217."""
import os
from ..graph import GraphBuilder, Model, resizable
from ..utils import checkarg_backbone, checkarg_upsampling, checkarg_dropout_variant
from .blocks import (conv_block, residual_block, dense_block, transition_block, localized_conv_block,
                     subpixel_block, resize_conv_block, deconv_block, convnext_block, _reject_unsupported)


def backbone_section(g, x_in, backbone_block, n_filters, n_blocks, activation, normalization, attention,
                     dropout_rate, dropout_variant):
    """Backbone shared by sp_postups.py:132-168 and sp_preups.py:115-151."""
    if backbone_block == 'unet':
        raise ValueError("backbone_block='unet' is built by unet_pin")
    _reject_unsupported(normalization, dropout_rate, dropout_variant)
    if backbone_block == 'convnext':
        # sp_postups.py:120-131 / sp_preups.py:104-114: 7x7 stem, ConvNext blocks (no dropout, no attention), then
        # TransitionBlock(stem) + blocks
        init_n_filters = n_filters
        x = b = g.conv2d(x_in, 'stem', n_filters, 7)
        for i in range(n_blocks):
            n_filters = init_n_filters * (i + 1)
            b = convnext_block(g, f'ConvNextBlock{i+1}', b, n_filters, use_1x1conv=(i != 0), activation=activation,
                               normalization=normalization)
        x = transition_block(g, 'TransitionSkip', x, n_filters, activation)
        return g.add(x, b, name='backbone_add'), n_filters
    blk = dict(activation=activation, normalization=normalization, attention=attention, dropout_rate=dropout_rate,
               dropout_variant=dropout_variant)
    init_n_filters = n_filters
    x = b = g.conv2d(x_in, 'stem', n_filters, 3)
    for i in range(n_blocks):
        n_filters = init_n_filters * (i + 1)
        if backbone_block == 'convnet':
            b = conv_block(g, f'ConvBlock{i+1}', b, n_filters, **blk)
        elif backbone_block == 'resnet':
            b = residual_block(g, f'ResidualBlock{i+1}', b, n_filters, use_1x1conv=(i != 0), **blk)
        elif backbone_block == 'densenet':
            b = dense_block(g, f'DenseBlock{i+1}', b, n_filters, **blk)
            b = transition_block(g, f'Transition{i+1}', b, b.C // 2)
    b = g.conv2d(b, 'backbone_last', n_filters, 3, activation=activation)
    b = g.dropout(b, dropout_rate, 'backbone_dropout', dropout_variant)          # sp_postups.py:155
    if backbone_block == 'convnet':
        x = b
    elif backbone_block == 'resnet':
        # x = TransitionBlock(x) ; x = Add()([x, b]) with b = act(conv(b)): the Add is fused into the 1x1 conv's
        # epilogue only when no activation separates them, so keep it explicit: b first, then skip + b.
        x = transition_block(g, 'TransitionSkip', x, n_filters, activation)
        x = g.add(x, b, name='backbone_add')
    elif backbone_block == 'densenet':
        x = g.concat([x, b], 'backbone_concat')
        x = transition_block(g, 'TransitionBackboneLast', x, n_filters, activation)
    return x, n_filters


def aux_branch(g, s_in, n_filters_aux, activation, normalization, convnext=False):
    """The HR auxiliary channels' own block -- sp_postups.py:186-198."""
    if convnext:
        return convnext_block(g, 'ConvNextBlock_aux', s_in, n_filters_aux, use_1x1conv=True, activation=activation,
                              normalization=normalization)
    return conv_block(g, 'ConvBlock_aux', s_in, n_filters_aux, activation=activation, normalization=normalization,
                      attention=False)


def tail_section(g, x, s_in, init_n_filters, n_filters_aux, n_channels_out, activation, output_activation,
                 normalization, dropout_rate, localcon_layer, convnext=False, transition_done=False):
    """sp_postups.py:184-212 / sp_preups.py:155-183 / :289-309.  ``convnext``: the auxiliary branch is a ConvNextBlock
    and the two closing ConvBlocks use 7x7 kernels (`ks`, sp_postups.py:121,193-210).  ``transition_done``: the
    auxiliary branch, the concatenation and TransitionLast are already part of `x` (composed upsampling tail)."""
    ks = 7 if convnext else 3
    if localcon_layer:
        lws = localized_conv_block(g, 'LocalizedConvBlock', x)
        x = g.concat([x, lws], 'lcb_concat')
    if s_in is not None and not transition_done:
        s = aux_branch(g, s_in, n_filters_aux, activation, normalization, convnext)
        x = g.concat([x, s], 'aux_concat')
    if not transition_done:          # else: composed with the upsampling block's last convolution (see net_postupsampling)
        x = transition_block(g, 'TransitionLast', x, init_n_filters)
    x = conv_block(g, 'ConvBlock_att', x, init_n_filters, ks_cl1=ks, ks_cl2=ks, activation=None,
                   normalization=normalization, attention=True, dropout_rate=dropout_rate)
    return conv_block(g, 'ConvBlock_out', x, n_channels_out, ks_cl1=ks, ks_cl2=ks, activation=output_activation,
                      normalization=normalization, attention=False)


@resizable('lr_size')
def net_postupsampling(backbone_block, upsampling, scale, n_channels, n_aux_channels, lr_size,
                       n_channels_out=1, n_filters=8, n_blocks=6, normalization=None, dropout_rate=0,
                       dropout_variant=None, attention=False, activation='relu', output_activation=None,
                       rc_interpolation='bilinear', localcon_layer=False, seed=None):
    backbone_block = checkarg_backbone(backbone_block)
    upsampling = checkarg_upsampling(upsampling)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    h_lr, w_lr = int(lr_size[0]), int(lr_size[1])
    h_hr, w_hr = int(h_lr * scale), int(w_lr * scale)

    g = GraphBuilder()
    x_in = g.input(h_lr, w_lr, n_channels)
    s_in = g.input(h_hr, w_hr, n_aux_channels) if n_aux_channels > 0 else None
    x, nf = backbone_section(g, x_in, backbone_block, n_filters, n_blocks, activation, normalization,
                             attention, dropout_rate, dropout_variant)
    model_name = backbone_block + '_' + upsampling
    # Without auxiliary / localized branches 'TransitionLast' (1x1, nf -> n_filters, ReLU) reads the upsampling block's
    # output directly and nothing else does: the two linear layers are evaluated as one convolution with the composed
    # filter (same variables, same gradients; csrc/graph_ops3.hip).  DL4DS_NO_FOLD=1 keeps them separate.
    # With HR auxiliary channels TransitionLast reads Concatenate([x, s]); a 1x1 convolution of a concatenation is the sum of
    # two, so the x part still composes and the s part is added by the composed convolution's epilogue (FoldedConvOp's
    # auxiliary form; needs n_filters % 4 == 0).  DL4DS_NO_FOLD_AUX=1 keeps that case unfolded.
    fold = (not localcon_layer and upsampling in ('spc', 'rc') and not os.environ.get('DL4DS_NO_FOLD') and
            (s_in is None or (n_filters % 4 == 0 and not os.environ.get('DL4DS_NO_FOLD_AUX'))))
    fold_into = ('TransitionLast', n_filters, 'relu') if fold else None
    fold_aux = None
    if fold and s_in is not None:
        fold_aux = aux_branch(g, s_in, nf, activation, normalization, convnext=(backbone_block == 'convnext'))
    if upsampling == 'spc':
        x = subpixel_block(g, 'SubpixelConvolution', x, scale, nf, fold_into=fold_into, fold_aux=fold_aux)
    elif upsampling == 'rc':
        x = resize_conv_block(g, 'ResizeConvolution', x, scale, nf, rc_interpolation, fold_into=fold_into, fold_aux=fold_aux)
    elif upsampling == 'dc':
        x = transition_block(g, 'TransitionDC', x, n_filters, activation)
        x = deconv_block(g, 'Deconvolution', x, scale, nf, activation)
    else:
        raise ValueError("net_postupsampling needs a post-upsampling method ('spc', 'rc' or 'dc')")
    x = tail_section(g, x, s_in, n_filters, nf, n_channels_out, activation, output_activation,
                     normalization, dropout_rate, localcon_layer, convnext=(backbone_block == 'convnext'),
                     transition_done=fold)
    g.finalize(x, seed)
    shapes = [(h_lr, w_lr, n_channels)] + ([(h_hr, w_hr, n_aux_channels)] if s_in is not None else [])
    return Model(g, model_name, shapes)
