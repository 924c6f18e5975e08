"""recnet_postupsampling -- same signature as dl4ds/models/spt_postups.py:12-31, graph per :96-163.
5-D tensors (B,T,H,W,C) are graph tensors with batch multiplier T (TimeDistributed = fold T into N)."""
from ..graph import GraphBuilder, Model, resizable
from ..utils import checkarg_backbone, checkarg_upsampling, checkarg_dropout_variant
from .blocks import (recurrent_conv_block, conv_block, transition_block, localized_conv_block,
                     subpixel_block, resize_conv_block, deconv_block)


def rec_backbone(g, x_in, backbone_block, n_filters, n_blocks, time_window, activation, normalization,
                 dropout_rate, dropout_variant):
    x = b = recurrent_conv_block(g, 'RecurrentConvBlock1', x_in, n_filters, time_window, activation, normalization)
    for i in range(n_blocks):
        b = recurrent_conv_block(g, f'RecurrentConvBlock{i+2}', b, n_filters, time_window, activation,
                                 normalization, dropout_rate, dropout_variant)
    b = g.dropout(b, dropout_rate, 'backbone_dropout', dropout_variant, dim=3)      # spt_postups.py:113
    if backbone_block == 'convnet':
        return b, n_filters
    if backbone_block == 'resnet':
        return g.add(x, b, name='backbone_add'), n_filters
    if backbone_block == 'densenet':
        x = g.concat([x, b], 'backbone_concat')
        return x, x.C
    raise NotImplementedError(f'backbone_block={backbone_block!r} is not available for spatio-temporal models')


def rec_tail(g, x, s_in, n_filters, n_channels_out, time_window, activation, output_activation, attention,
             normalization, dropout_rate, localcon_layer, transition_filters=None):
    """spt_postups.py:133-157 (TransitionLast -> C//2) / spt_preups.py:114-138 (TransitionLast -> n_filters)."""
    s = None
    if s_in is not None:
        s = conv_block(g, 'ConvBlock_aux', s_in, n_filters, activation=activation, attention=attention)
    co = None
    if s is not None and localcon_layer:
        co = (x.C + s.C + 2) // 2 if transition_filters is None else transition_filters
    if co is not None and g.rec_tail_supported(x.C, s.C, co):
        # Concatenate([x, repeat(s)]) -> LocalizedConvBlock -> Concatenate -> TransitionLast: one pass per direction, nothing of it
        # materialised (csrc/graph_ops4.hip; same variables).  DL4DS_NO_REC_TAIL_FUSION=1: the separate layers below.
        x = g.rec_tail(x, s, time_window, co)
    else:
        if s is not None:
            s = g.repeat_time(s, time_window, 'aux_repeat')
            x = g.concat([x, s], 'aux_concat')
        if localcon_layer:
            lws = localized_conv_block(g, 'LocalizedConvBlock', x)
            x = g.concat([x, lws], 'lcb_concat')
        x = transition_block(g, 'TransitionLast', x, x.C // 2 if transition_filters is None else transition_filters)
    x = conv_block(g, 'ConvBlock_att', x, n_filters, activation=None, normalization=normalization, attention=True,
                   dropout_rate=dropout_rate, time_window_5d=time_window)
    return conv_block(g, 'ConvBlock_out', x, n_channels_out, activation=output_activation,
                      normalization=normalization)


@resizable('lr_size')
def recnet_postupsampling(backbone_block, upsampling, scale, n_channels, n_aux_channels, lr_size, time_window,
                          n_channels_out=1, n_filters=8, n_blocks=4, dropout_rate=0, dropout_variant=None,
                          normalization=None, attention=False, activation='relu', output_activation=None,
                          rc_interpolation='bilinear', localcon_layer=False, seed=None):
    backbone_block = checkarg_backbone(backbone_block)
    upsampling = checkarg_upsampling(upsampling)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    h_lr, w_lr = int(lr_size[0]), int(lr_size[1])
    h_hr, w_hr = int(h_lr * scale), int(w_lr * scale)
    T = int(time_window)
    g = GraphBuilder()
    x_in = g.input(h_lr, w_lr, n_channels, nmul=T)
    s_in = g.input(h_hr, w_hr, n_aux_channels) if n_aux_channels > 0 else None
    x, nf_ups = rec_backbone(g, x_in, backbone_block, n_filters, n_blocks, T, activation, normalization,
                             dropout_rate, dropout_variant)
    if upsampling == 'spc':
        x = subpixel_block(g, 'upsampling_spc', x, scale, nf_ups)
    elif upsampling == 'rc':
        x = resize_conv_block(g, 'upsampling_rc', x, scale, nf_ups, rc_interpolation)
    elif upsampling == 'dc':
        x = deconv_block(g, 'upsampling_dc', x, scale, nf_ups)
    else:
        raise ValueError("recnet_postupsampling needs a post-upsampling method ('spc', 'rc' or 'dc')")
    x = rec_tail(g, x, s_in, n_filters, n_channels_out, T, activation, output_activation, attention,
                 normalization, dropout_rate, localcon_layer)
    g.finalize(x, seed)
    shapes = [(T, h_lr, w_lr, n_channels)] + ([(h_hr, w_hr, n_aux_channels)] if s_in is not None else [])
    return Model(g, 'rec' + backbone_block + '_' + upsampling, shapes)
