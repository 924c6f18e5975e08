"""The arena-name <-> Keras-variable map (tests/golden/keras_name_map.py) that lets a maintainer with TensorFlow
regenerate tests/golden/*.npz from the REAL reference (tests/golden/make_golden_tf.py): it must cover every variable of
every builder configuration, and its attribute chains must exist in the reference's class definitions."""
import ast
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import keras_name_map as K        # noqa: E402

from oracle import np_ops as N    # noqa: E402
from oracle import models as M    # noqa: E402

SP = [dict(backbone_block=b, upsampling=u, scale=s, n_blocks=2, **extra)
      for b in ('convnet', 'resnet', 'densenet') for u, s in (('spc', 4), ('rc', 2), ('dc', 2), ('spc', 10))
      for extra in (dict(), dict(attention=True, localcon_layer=True), dict(normalization='bn'), dict(normalization='ln'))]
BUILDS = (
    [('net_postupsampling', (1, 4, 4, 2), (1, 4 * c['scale'], 4 * c['scale'], 1) if i % 2 else None, c) for i, c in enumerate(SP)]
    + [('net_postupsampling', (1, 8, 8, 1), None, dict(backbone_block='convnext', upsampling='spc', scale=2, n_blocks=2, normalization='ln')),
       ('net_postupsampling', (1, 8, 8, 1), (1, 16, 16, 1), dict(backbone_block='convnext', upsampling='rc', scale=2, n_blocks=2, normalization='bn')),
       ('net_pin', (1, 8, 8, 2), None, dict(backbone_block='resnet', n_blocks=2)),
       ('net_pin', (1, 8, 8, 2), (1, 8, 8, 1), dict(backbone_block='densenet', n_blocks=2, attention=True, localcon_layer=True, normalization='ln')),
       ('net_pin', (1, 8, 8, 2), None, dict(backbone_block='convnet', n_blocks=2, normalization='bn')),
       ('unet_pin', (1, 16, 16, 3), (1, 16, 16, 1), dict(n_filters=4, n_blocks=2, decoder_upsampling='dc')),
       ('unet_pin', (1, 16, 16, 3), None, dict(n_filters=4, n_blocks=2, decoder_upsampling='spc', normalization='ln')),
       ('unet_pin', (1, 16, 16, 3), (1, 16, 16, 2), dict(n_filters=4, n_blocks=2, decoder_upsampling='rc', attention=True, localcon_layer=True)),
       ('recnet_postupsampling', (1, 3, 6, 6, 1), (1, 12, 12, 1), dict(backbone_block='densenet', upsampling='rc', scale=2, time_window=3, n_filters=4, n_blocks=1, attention=True, localcon_layer=True)),
       ('recnet_postupsampling', (1, 2, 6, 6, 1), None, dict(backbone_block='resnet', upsampling='spc', scale=2, time_window=2, n_filters=4, n_blocks=2, normalization='ln')),
       ('recnet_postupsampling', (1, 2, 6, 6, 1), None, dict(backbone_block='convnet', upsampling='dc', scale=2, time_window=2, n_filters=4, n_blocks=1)),
       ('recnet_pin', (1, 2, 6, 6, 2), None, dict(backbone_block='resnet', time_window=2, n_filters=4, n_blocks=1)),
       ('recnet_pin', (1, 2, 6, 6, 2), (1, 6, 6, 1), dict(backbone_block='densenet', time_window=2, n_filters=4, n_blocks=2, attention=True, localcon_layer=True)),
       ])


def _names(model, xs, ss, cfg):
    cfg = dict(cfg)
    norm = cfg.pop('normalization', None)                   # the oracle takes normalization through its Ctx
    if norm is not None:
        cfg['ctx'] = M.Ctx(training=True, normalization=norm)
    try:
        return list(M.init_params(model, xs, ss, **cfg).keys())
    except (ValueError, NotImplementedError, TypeError, AssertionError) as e:
        pytest.skip(f'oracle does not build this variant: {e}')


@pytest.mark.parametrize('idx', range(len(BUILDS)))
def test_map_covers_every_variable_of_every_builder(idx):
    model, xs, ss, cfg = BUILDS[idx]
    names = _names(model, xs, ss, cfg)
    assert names
    seen = set()
    for n in names:
        sel, chain, var = K.parse(n)                       # KeyError = a variable the map cannot place
        key = (sel, tuple(chain), var)
        assert key not in seen, f'{n} collides with another variable under the map'
        seen.add(key)


@pytest.mark.parametrize('kw,spt', [
    (dict(upsampling='pin', scale=8), False), (dict(upsampling='spc', scale=4), False),
    (dict(upsampling='spc', scale=5), False), (dict(upsampling='rc', scale=3), False),
    (dict(upsampling='pin', scale=2, attention=True, normalization='ln'), False),
    (dict(upsampling='pin', scale=2), True)])
def test_map_covers_the_discriminator(kw, spt):
    P = M.Params(create=True)
    s = kw['scale']
    lead = (1, 2) if spt else (1,)
    if kw['upsampling'] == 'pin':
        x, hr = np.zeros(lead + (16, 16, 2), np.float32), np.zeros(lead + (16, 16, 1), np.float32)
    else:
        l = {4: 4, 5: 8, 3: 5}[s]                       # scale 5: two 'valid' stride-2 convolutions + Cropping2D need lr 8
        x, hr = np.zeros(lead + (l, l, 2), np.float32), np.zeros(lead + (l * s, l * s, 1), np.float32)
        kw = dict(kw, lr_size=(l, l))
    try:
        M.residual_discriminator(N, P, x, hr, **kw)
    except (ValueError, NotImplementedError, TypeError, AssertionError) as e:
        pytest.skip(f'oracle does not build this variant: {e}')
    seen = set()
    for n in P.keys():
        key = K.parse(n, spatiotemporal_discriminator=spt)
        key = (key[0], tuple(key[1]), key[2])
        assert key not in seen, n
        seen.add(key)


def test_unknown_names_are_rejected():
    for bad in ('kernel', 'ResidualBlock1/conv9/kernel', 'ConvBlock_att/att/kernel', 'Mystery/conv/kernel',
                'TransitionLast/conv/weights', 'EncoderBlock1/conv/kernel'):
        with pytest.raises(KeyError):
            K.parse(bad)
    assert K.parse('Deconvolution2/deconv_scale_x2/kernel') == (('name', 'Deconvolution2'), ['conv2dtranspose'], 'kernel')
    assert K.parse('ConvBlock_att/att/conv2/bias') == (('auto', 'ConvBlock', -2), ['att', 'conv2'], 'bias')
    assert K._auto_index('conv_block_3', 'ConvBlock') == 3 and K._auto_index('conv_block', 'ConvBlock') == 0
    assert K._auto_index('ConvBlock_aux', 'ConvBlock') is None and K._auto_index('conv2d_12', 'Conv2D') == 12
    assert K._snake('RecurrentConvBlock') == 'recurrent_conv_block' and K._snake('TransitionBlock') == 'transition_block'


REF_BLOCKS = '/root/reference/dl4ds/models/blocks.py'


@pytest.mark.skipif(not os.path.exists(REF_BLOCKS), reason='reference checkout not present (GPU box)')
def test_attribute_chains_exist_in_the_reference_classes():
    """Every sub-layer attribute the map walks is a ``self.<attr> = <Layer>(...)`` assignment of that reference class (or
    of a base class), and every explicit layer name the map selects by is a name the reference builders assign."""
    tree = ast.parse(open(REF_BLOCKS).read())
    attrs, bases = {}, {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            bases[node.name] = [b.id for b in node.bases if isinstance(b, ast.Name)]
            got = set()
            for sub in ast.walk(node):
                if isinstance(sub, ast.Assign):
                    for t in sub.targets:
                        if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == 'self':
                            got.add(t.attr)
            attrs[node.name] = got

    def all_attrs(cls):
        out = set(attrs.get(cls, ()))
        for b in bases.get(cls, ()):
            out |= all_attrs(b)
        return out

    for cls, want in K.REF_CLASS_ATTRS.items():
        assert cls in attrs, f'reference has no class {cls}'
        missing = want - all_attrs(cls)
        assert not missing, f'{cls}: attributes {missing} are not defined by the reference class'
    src = ''.join(open(os.path.join('/root/reference/dl4ds/models', f)).read()
                  for f in ('sp_postups.py', 'sp_preups.py', 'spt_postups.py', 'spt_preups.py', 'discriminator.py', 'blocks.py'))
    for literal in ("'ResidualBlock' + str(i+1)", "'ConvBlock' + str(i+1)", "'DenseBlock' + str(i+1)", "'Transition' + str(i+1)",
                    "'ConvNextBlock' + str(i+1)", "name='TransitionLast'", "name='ConvBlock_aux'", "name='TransitionBackboneLast'",
                    "name='TransitionDC'", "name='Bottleneck'", "'DecoderConvBlock' + str(j+1)", "name='EncoderBlock' + name_suffix",
                    "name='SubpixelConvolution' + name_suffix", "name='ResizeConvolution' + name_suffix",
                    "name='Deconvolution' + name_suffix", "name='upsampling_' + upsampling", "name='localized_conv_block'",
                    "_branch1'", "_branch2'", "name='deconv_1of2_scale_x2'", "name='deconv_2of2_scale_x2'",
                    "name='deconv_scale_x' + str(self.scale)"):
        assert literal in src, f'the reference no longer assigns {literal}'


def test_tf_regeneration_script_imports_and_compares_without_tensorflow():
    """tests/golden/make_golden_tf.py cannot run here (no TensorFlow); its TensorFlow-free parts must at least import, share
    the case list with make_golden.py, and its comparison must accept a fixture against itself and flag a perturbed one."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_tf', os.path.join(HERE, 'golden', 'make_golden_tf.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert set(m.CASES) == {'cfg1_net_pin', 'cfg2_resnet_spc', 'cfg2_dssim_mae', 'cfg4_rec_dense_rc', 'cfg5_unet_dc'}
    z = np.load(os.path.join(HERE, 'golden', 'cfg2_resnet_spc.npz'))
    same = {k: z[k] for k in z.files}
    assert m.compare('self', same, z) == []
    off = dict(same, pred=same['pred'] * 1.01)
    assert any(b.startswith('pred') for b in m.compare('perturbed', off, z))
    # every case's variables resolve through the map (the script asserts the converse -- every Keras weight hit -- under TF)
    for name, c in m.CASES.items():
        P0 = M.init_params(c['model'], (1,) + c['x'][1:], None if c['s'] is None else (1,) + c['s'][1:], **c['cfg'])
        for k in P0:
            K.parse(k)
