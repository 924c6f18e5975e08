"""residual_discriminator -- same signature as dl4ds/models/discriminator.py:11-22, graph per :25-80."""
from ..graph import GraphBuilder, Model
from .. import POSTUPSAMPLING_METHODS
from .blocks import residual_block, recurrent_conv_block


def residual_discriminator(n_channels, upsampling, is_spatiotemporal, scale, lr_size, n_filters=8,
                           n_res_blocks=4, normalization=None, activation='relu', attention=False,
                           hr_size=None, time_window=None, seed=None):
    """Two branches -- the conditioning (LR, or HR-interpolated for 'pin') array and the HR reference / generated
    array -- merged on the grid of the first.  Post-upsampling generators: the HR branch is brought down to the LR
    grid by two stride-2 convolutions (scale 4: 'same'; scale 5: 'valid' + a one-pixel crop) or, for any other scale,
    by bilinear resizing (discriminator.py:52-63).  Spatio-temporal: ConvLSTM stem with LayerNormalization on the
    conditioning branch, time-distributed residual blocks, 3-D global pooling (:31-33, :73-74); ``time_window`` sizes
    the static graph (the reference leaves the axis dynamic).  ``normalization='bn'``: the CGAN step evaluates the real and
    the generated batch in ONE pass over [real ; fake]; BatchNormalization then normalises the two halves separately and
    updates its moving averages half after half, which is what the reference's two calls do (csrc/graph_ops3.hip, NormOp)."""
    T = 1
    if is_spatiotemporal:
        if not time_window:
            raise ValueError('the spatio-temporal discriminator needs time_window')
        T = int(time_window)
    post = upsampling in POSTUPSAMPLING_METHODS
    if hr_size is None:
        hr_size = (int(lr_size[0] * scale), int(lr_size[1] * scale))
    h, w = int(hr_size[0]), int(hr_size[1])
    h_in, w_in = (int(lr_size[0]), int(lr_size[1])) if post else (h, w)
    g = GraphBuilder()
    x_in = g.input(h_in, w_in, n_channels, nmul=T)
    x_ref = g.input(h, w, 1, nmul=T, requires_grad=True)     # the generator's adversarial gradient flows through it
    blk = dict(normalization=normalization, attention=attention)
    if is_spatiotemporal:
        x1 = b = recurrent_conv_block(g, 'RecurrentConvBlock', x_in, n_filters, T, activation, 'ln')
    else:
        x1 = b = g.conv2d(x_in, 'branch1_in', n_filters, 3)
    for i in range(n_res_blocks):
        b = residual_block(g, f'ResidualBlock{i+1}_branch1', b, n_filters, **blk)
    x1 = g.conv2d(b, 'branch1_out', n_filters, 3, add=x1)
    x2 = c = g.conv2d(x_ref, 'branch2_in', n_filters, 3)
    for i in range(n_res_blocks):
        c = residual_block(g, f'ResidualBlock{i+1}_branch2', c, n_filters, **blk)
    if post:
        if scale == 5:
            c = g.conv2d_strided(c, 'branch2_down1', n_filters, 3, 2, 'valid')
            c = g.conv2d_strided(c, 'branch2_down2', n_filters, 3, 2, 'valid')
            x2 = g.slice2d(c, 0, 0, 1, c.H - 1, c.W - 1, 'Cropping2D')
        elif scale == 4:
            c = g.conv2d_strided(c, 'branch2_down1', n_filters, 3, 2, 'same')
            x2 = g.conv2d_strided(c, 'branch2_down2', n_filters, 3, 2, 'same')
        else:
            x2 = g.resize(c, h_in, w_in, 'InterpolationDownsampling')
        if (x2.H, x2.W) != (x1.H, x1.W):
            raise ValueError(f'discriminator branches end on different grids: {(x1.H, x1.W)} vs {(x2.H, x2.W)} '
                             f'(lr_size {tuple(lr_size)}, scale {scale})')
    else:
        x2 = g.conv2d(c, 'branch2_out', n_filters, 3, add=x2)
    x = g.concat([x1, x2], 'Concat2Branches')
    x = residual_block(g, 'ResidualBlock_merge', x, x.C, **blk)
    x = g.gap(x, 'GlobalAveragePooling', over_time=is_spatiotemporal)
    x = g.dropout(x, 0.4, 'dropout')
    x = g.dense(x, 'dense1', 32, activation='sigmoid')
    x = g.dense(x, 'dense2', 1, activation='sigmoid')
    g.finalize(x, seed)
    lead = (T,) if is_spatiotemporal else ()
    return Model(g, 'discriminator', [lead + (h_in, w_in, n_channels), lead + (h, w, 1)])
