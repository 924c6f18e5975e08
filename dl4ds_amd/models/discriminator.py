"""residual_discriminator -- same signature as dl4ds/models/discriminator.py:11-22, graph per :25-80.
Spatial 'pin' branch (both inputs on the HR grid) is what the CGAN benchmark configuration uses."""
from ..graph import GraphBuilder, Model
from .. import POSTUPSAMPLING_METHODS
from .blocks import residual_block


def residual_discriminator(n_channels, upsampling, is_spatiotemporal, scale, lr_size, n_filters=8,
                           n_res_blocks=4, normalization=None, activation='relu', attention=False,
                           hr_size=None, seed=None):
    if is_spatiotemporal:
        raise NotImplementedError('spatio-temporal discriminator is not implemented on the MI355X path yet')
    if upsampling in POSTUPSAMPLING_METHODS:
        raise NotImplementedError('post-upsampling discriminator branches (strided convs / resize) are not '
                                  'implemented on the MI355X path yet; use upsampling="pin"')
    if hr_size is None:
        hr_size = (int(lr_size[0] * scale), int(lr_size[1] * scale))
    h, w = int(hr_size[0]), int(hr_size[1])
    g = GraphBuilder()
    x_in = g.input(h, w, n_channels)
    x_ref = g.input(h, w, 1, requires_grad=True)     # the generator's adversarial gradient flows through it
    x1 = b = g.conv2d(x_in, 'branch1_in', n_filters, 3)
    for i in range(n_res_blocks):
        b = residual_block(g, f'ResidualBlock{i+1}_branch1', b, n_filters, normalization=normalization,
                           attention=attention)
    x1 = g.conv2d(b, 'branch1_out', n_filters, 3, add=x1)
    x2 = c = g.conv2d(x_ref, 'branch2_in', n_filters, 3)
    for i in range(n_res_blocks):
        c = residual_block(g, f'ResidualBlock{i+1}_branch2', c, n_filters, normalization=normalization,
                           attention=attention)
    x2 = g.conv2d(c, 'branch2_out', n_filters, 3, add=x2)
    x = g.concat([x1, x2], 'Concat2Branches')
    x = residual_block(g, 'ResidualBlock_merge', x, x.C, normalization=normalization, attention=attention)
    x = g.gap(x, 'GlobalAveragePooling')
    x = g.dropout(x, 0.4, 'dropout')
    x = g.dense(x, 'dense1', 32, activation='sigmoid')
    x = g.dense(x, 'dense2', 1, activation='sigmoid')
    g.finalize(x, seed)
    return Model(g, 'discriminator', [(h, w, n_channels), (h, w, 1)])
