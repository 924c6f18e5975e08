"""Blocks of dl4ds/models/blocks.py lowered onto the gfx950 graph runtime.

Each function takes the GraphBuilder ``g``, a layer-name prefix and the input Tensor(s) and returns
the output Tensor.  Fusions are decided here: Conv2D+bias+ReLU(+residual Add) is one kernel launch,
SubpixelConvolution's depth_to_space is the conv's store pattern, LayerNormalization /
BatchNormalization carry the ReLU that follows them.
"""
from ..utils import checkarg_dropout_variant


def _check_normalization(normalization):
    if normalization is not None and normalization not in ['bn', 'ln']:
        raise ValueError(f'Normalization not supported, got {normalization}')
    return normalization


def _reject_unsupported(normalization, dropout_rate, dropout_variant=None):
    """Kept for the builders that have no normalised / dropout form in the reference either."""
    checkarg_dropout_variant(dropout_variant)
    _check_normalization(normalization)


def _conv_norm_act(g, name, norm_name, x, filters, ks, activation, normalization):
    """Conv2D(use_bias = normalization is None) -> [norm] -> activation (blocks.py:50-62,90-93)."""
    if normalization is None:
        return g.conv2d(x, name, filters, ks, activation=activation)
    y = g.conv2d(x, name, filters, ks, use_bias=False)
    return g.norm(y, norm_name, normalization, activation=activation)


def conv_block(g, name, x, filters, ks_cl1=3, ks_cl2=3, activation='relu', normalization=None,
               attention=False, dropout_rate=0, dropout_variant=None, time_window_5d=0):
    """ConvBlock.call -- blocks.py:87-103: [drop] conv1 [norm1] act [drop] conv2 [norm2] act [att]."""
    _check_normalization(normalization)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    y = g.dropout(x, dropout_rate, name + '/dropout1', dropout_variant)
    y = _conv_norm_act(g, name + '/conv1', name + '/norm1', y, filters, ks_cl1, activation, normalization)
    y = g.dropout(y, dropout_rate, name + '/dropout2', dropout_variant)
    y = _conv_norm_act(g, name + '/conv2', name + '/norm2', y, filters, ks_cl2, activation, normalization)
    if attention:
        y = g.channel_attention(y, name + '/att', filters, time_window_5d=time_window_5d)
    return y


def residual_block(g, name, x, filters, activation='relu', normalization=None, attention=False,
                   dropout_rate=0, dropout_variant=None, use_1x1conv=False):
    """ResidualBlock.call -- blocks.py:210-230:
    [drop] conv1 [norm1] act [drop] conv2 [norm2] [att] -> (+ conv1x1(X) | X) -> act."""
    _check_normalization(normalization)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    y = g.dropout(x, dropout_rate, name + '/dropout1', dropout_variant)
    y = _conv_norm_act(g, name + '/conv1', name + '/norm1', y, filters, 3, activation, normalization)
    y = g.dropout(y, dropout_rate, name + '/dropout2', dropout_variant)
    skip = g.conv2d(x, name + '/conv1x1', filters, 1) if use_1x1conv else x
    if attention or normalization is not None:
        y = g.conv2d(y, name + '/conv2', filters, 3, use_bias=normalization is None)
        if normalization is not None:
            y = g.norm(y, name + '/norm2', normalization)
        if attention:
            y = g.channel_attention(y, name + '/att', filters)
        if activation in (None, 'relu', 'linear'):
            return g.add(y, skip, relu=(activation == 'relu'), name=name + '/add')
        return g.act(g.add(y, skip, name=name + '/add'), activation, name + '/act')
    # fused: conv2 + bias + skip + activation in one epilogue
    return g.conv2d(y, name + '/conv2', filters, 3, activation=activation, add=skip)


def convnext_block(g, name, x, filters, use_1x1conv=False, activation='gelu', normalization='ln'):
    """ConvNextBlock.call -- blocks.py:175-187 as the builders instantiate it (drop_path=0, layer_scale_init_value=0:
    no DropPath, no gamma): dwconv 7x7 -> norm (LayerNormalization(epsilon=1e-6) | BatchNormalization()) -> Dense(4f)
    -> activation -> Dense(f) -> + (conv1x1(input) | input).  The reference only defines ``self.norm`` for 'bn' / 'ln'
    (:155-159) and then calls it unconditionally (:177), so normalization=None fails there with AttributeError; the
    same request is refused here."""
    if normalization not in ('bn', 'ln'):
        raise ValueError("ConvNextBlock needs normalization='ln' or 'bn' (the reference raises AttributeError for None: "
                         "blocks.py:155-159,177)")
    y = g.dwconv(x, name + '/dwconv', 7)
    y = g.norm(y, name + '/norm', normalization, epsilon=1e-6 if normalization == 'ln' else 1e-3)
    y = g.conv2d(y, name + '/pwconv1', 4 * filters, 1, activation=activation, dense=True)
    skip = g.conv2d(x, name + '/conv1x1', filters, 1) if use_1x1conv else x
    if skip.C != filters:
        raise ValueError(f'ConvNextBlock: input has {skip.C} channels but filters={filters}; use_1x1conv=True is needed')
    return g.conv2d(y, name + '/pwconv2', filters, 1, add=skip, dense=True)


def dense_block(g, name, x, filters, activation='relu', normalization=None, attention=False,
                dropout_rate=0, dropout_variant=None):
    """DenseBlock.call -- blocks.py:262-277.  conv1 consumes the RAW X (line 267): norm1 / dropout1 are
    evaluated by the reference and their result dropped, so norm1 only contributes its variables here.
    Both convolutions keep their bias (they are re-created without use_bias, lines 249-258)."""
    _check_normalization(normalization)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    if normalization is not None:
        g.norm_variables(name + '/norm1', x.C, normalization)
    y = g.conv2d(x, name + '/conv1', 4 * filters, 1, activation=None if normalization else activation)
    if normalization is not None:
        y = g.norm(y, name + '/norm2', normalization, activation=activation)
    y = g.dropout(y, dropout_rate, name + '/dropout2', dropout_variant)
    y = g.conv2d(y, name + '/conv2', filters, 3)
    if attention:
        y = g.channel_attention(y, name + '/att', filters)
    return g.concat([y, x], name + '/concat')


def transition_block(g, name, x, filters, activation='relu', normalization=None):
    """TransitionBlock.call -- blocks.py:301-309: 1x1 conv -> act, or (only for 'bn') BN -> act -> 1x1 conv."""
    if normalization == 'bn':
        y = g.norm(x, name + '/batch_norm', 'bn', activation=activation)
        return g.conv2d(y, name + '/conv', filters, 1)
    return g.conv2d(x, name + '/conv', filters, 1, activation=activation)


def localized_conv_block(g, name, x, filters=2):
    """LocalizedConvBlock -- blocks.py:312-333 (TransitionBlock(2) -> LocallyConnected2D 1x1 + bias)."""
    y = transition_block(g, name + '/transition', x, filters)
    return g.localconv(y, name + '/localconv', filters, use_bias=True)


def recurrent_conv_block(g, name, x, filters, time_window, activation='relu', normalization=None,
                         dropout_rate=0, dropout_variant=None):
    """RecurrentConvBlock.call -- blocks.py:380-398: ConvLSTM2D 5x5 -> act -> ConvLSTM2D 3x3 -> act."""
    _check_normalization(normalization)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    y = g.dropout(x, dropout_rate, name + '/dropout1', dropout_variant, dim=3)
    if normalization is None:
        y = g.convlstm(y, name + '/convlstm1', filters, 5, time_window, activation=activation)
    else:
        y = g.norm(g.convlstm(y, name + '/convlstm1', filters, 5, time_window), name + '/norm1', normalization,
                   activation=activation)
    y = g.dropout(y, dropout_rate, name + '/dropout2', dropout_variant, dim=3)
    if normalization is None:
        return g.convlstm(y, name + '/convlstm2', filters, 3, time_window, activation=activation)
    return g.norm(g.convlstm(y, name + '/convlstm2', filters, 3, time_window), name + '/norm2', normalization,
                  activation=activation)


def subpixel_block(g, name, x, scale, n_filters, fold_into=None, fold_aux=None):
    """SubpixelConvolutionBlock.call -- blocks.py:433-454.  ``conv2x`` is ONE weight set applied at
    every x2 stage; depth_to_space is fused into the conv store.  ``fold_into=(name, filters, activation)``: the 1x1
    TransitionBlock that consumes the block's output directly is composed with the last stage's filter
    (GraphBuilder.conv2d_folded) instead of being run on the n_filters-channel HR tensor."""
    seq = {2: [2], 4: [2, 2], 8: [2, 2, 2], 10: [2, 5], 20: [2, 2, 5]}.get(scale, [scale])
    for i, f in enumerate(seq):
        sub = {2: 'conv2x', 5: 'conv5x'}.get(f, 'conv')
        if fold_into is not None and i == len(seq) - 1:
            x = g.conv2d_folded(x, f'{name}/{sub}', n_filters, 3, f, fold_into[0] + '/conv', fold_into[1], fold_into[2],
                                aux=fold_aux)
        else:
            x = g.conv2d(x, f'{name}/{sub}', n_filters * f * f, 3, d2s=f)
    return x


def resize_conv_block(g, name, x, scale, n_filters, interpolation='bilinear', fold_into=None, fold_aux=None):
    """ResizeConvolutionBlock.call -- blocks.py:485-491 (``fold_into``: see subpixel_block)."""
    y = g.resize(x, int(x.H * scale), int(x.W * scale), name + '/resize', interpolation)
    if fold_into is not None:
        return g.conv2d_folded(y, name + '/conv', n_filters, 3, 0, fold_into[0] + '/conv', fold_into[1], fold_into[2],
                               aux=fold_aux)
    return g.conv2d(y, name + '/conv', n_filters, 3)


def deconv_block(g, name, x, scale, n_filters, output_activation=None):
    """DeconvolutionBlock.call -- blocks.py:522-534 (9x9 Conv2DTranspose, no bias).  The reference
    falls through for scale == 4 and applies a third (x4) deconvolution, yielding a 16x grid; that
    defect is reported instead of reproduced."""
    if scale == 4:
        raise ValueError("DeconvolutionBlock(scale=4) is broken in the reference (blocks.py:525-533 applies "
                         "x2, x2 and then x4); use scale 2 or 8, or the 'spc'/'rc' upsamplers")
    if scale == 8:
        x = g.conv2d_transpose(x, name + '/deconv_1of2_scale_x2', n_filters, 9, 2)
        x = g.conv2d_transpose(x, name + '/deconv_2of2_scale_x2', n_filters, 9, 2, activation=output_activation)
        return g.conv2d_transpose(x, name + '/deconv_2of2_scale_x2', n_filters, 9, 2, activation=output_activation)
    return g.conv2d_transpose(x, f'{name}/deconv_scale_x{scale}', n_filters, 9, scale, activation=output_activation)


def pad_concat(g, name, t1, t2):
    """PadConcat.call -- blocks.py:629-656: the smaller tensor is zero-padded at the bottom / right (odd grids lose a
    row / column in MaxPooling2D that the decoder's x2 upsampling does not bring back), then both are concatenated."""
    h, w = max(t1.H, t2.H), max(t1.W, t2.W)
    if (t1.H, t1.W) != (h, w):
        t1 = g.pad_bottom_right(t1, h, w, name + '/pad1')
    if (t2.H, t2.W) != (h, w):
        t2 = g.pad_bottom_right(t2, h, w, name + '/pad2')
    return g.concat([t1, t2], name)
