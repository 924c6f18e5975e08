"""recnet_pin -- same signature as dl4ds/models/spt_preups.py:12-28, graph per :85-144."""
from ..graph import GraphBuilder, Model, resizable
from ..utils import checkarg_backbone, checkarg_dropout_variant
from .spt_postups import rec_backbone, rec_tail


@resizable('hr_size')
def recnet_pin(backbone_block, n_channels, n_aux_channels, hr_size, time_window, n_channels_out=1, n_filters=8,
               n_blocks=6, normalization=None, dropout_rate=0, dropout_variant=None, attention=False,
               activation='relu', output_activation=None, localcon_layer=False, seed=None):
    backbone_block = checkarg_backbone(backbone_block)
    dropout_variant = checkarg_dropout_variant(dropout_variant)
    h_hr, w_hr = int(hr_size[0]), int(hr_size[1])
    T = int(time_window)
    g = GraphBuilder()
    x_in = g.input(h_hr, w_hr, n_channels, nmul=T)
    s_in = g.input(h_hr, w_hr, n_aux_channels) if n_aux_channels > 0 else None
    x, _ = rec_backbone(g, x_in, backbone_block, n_filters, n_blocks, T, activation, normalization,
                        dropout_rate, dropout_variant)
    x = rec_tail(g, x, s_in, n_filters, n_channels_out, T, activation, output_activation, attention,
                 normalization, dropout_rate, localcon_layer, transition_filters=n_filters)
    g.finalize(x, seed)
    shapes = [(T, h_hr, w_hr, n_channels)] + ([(h_hr, w_hr, n_aux_channels)] if s_in is not None else [])
    return Model(g, 'rec' + backbone_block + '_pin', shapes)
