"""Generate the golden fixtures under tests/golden/ with the ORACLE (fp64 torch-CPU restatement).

The reference itself cannot run here (TensorFlow is not installable: SURVEY.md section 8c), so these vectors
pin the oracle against regressions and give the GPU path a committed target; they are NOT outputs of the
reference ("parity unpinned").  Weights are regenerated from names by ``golden_weights`` so fixtures stay small.

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import torch_ops as T      # noqa: E402
from oracle import models as M         # noqa: E402
from oracle import train as TR         # noqa: E402


def golden_weights(name, shape):
    """Deterministic pseudo-random weights from the parameter NAME (fan-in scaled; biases small non-zero)."""
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    shape = tuple(shape)
    if len(shape) <= 1 or name.endswith('bias'):
        return (0.05 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[:-1]))
    return (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


CASES = {
    'cfg1_net_pin': dict(model='net_pin', cfg=dict(backbone_block='resnet'), x=(2, 16, 16, 2), s=None, loss='mae'),
    'cfg2_resnet_spc': dict(model='net_postupsampling', cfg=dict(backbone_block='resnet', upsampling='spc', scale=4),
                            x=(2, 8, 8, 1), s=None, loss='mae'),
    'cfg2_dssim_mae': dict(model='net_postupsampling', cfg=dict(backbone_block='resnet', upsampling='spc', scale=4,
                                                                n_blocks=2), x=(2, 8, 8, 1), s=None, loss='dssim_mae'),
    'cfg4_rec_dense_rc': dict(model='recnet_postupsampling',
                              cfg=dict(backbone_block='densenet', upsampling='rc', scale=4, time_window=3, attention=True,
                                       localcon_layer=True, n_blocks=2), x=(1, 3, 6, 6, 1), s=(1, 24, 24, 1), loss='mae'),
    'cfg5_unet_dc': dict(model='unet_pin', cfg=dict(n_filters=8, n_blocks=3, decoder_upsampling='dc'),
                         x=(2, 32, 32, 5), s=(2, 32, 32, 1), loss='mae'),
}


def run_case(name, c):
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 1)
    P0 = M.init_params(c['model'], (1,) + c['x'][1:], None if c['s'] is None else (1,) + c['s'][1:], **c['cfg'])
    P = M.Params()
    for k, v in P0.items():
        P[k] = golden_weights(k, v.shape).astype(np.float64)
    x = rng.standard_normal(c['x']).astype(np.float32)
    s = None if c['s'] is None else rng.standard_normal(c['s']).astype(np.float32)
    PT = M.convert(P, T, requires_grad=True)
    pred0 = TR.forward(c['model'], c['cfg'], PT, T.asarray(x.astype(np.float64)),
                       None if s is None else T.asarray(s.astype(np.float64)))
    y = (rng.random(tuple(pred0.shape)) ).astype(np.float32)
    lv, grads, pred = TR.supervised_step(c['model'], c['cfg'], PT, T.asarray(x.astype(np.float64)),
                                         None if s is None else T.asarray(s.astype(np.float64)),
                                         T.asarray(y.astype(np.float64)), loss=c['loss'])
    out = dict(x=x, y_true=y, pred=pred.numpy().astype(np.float32), loss=np.float64(lv))
    if s is not None:
        out['s'] = s
    names = sorted(grads.keys())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array([float(grads[k].norm()) for k in names])
    out['grad_heads'] = np.stack([np.pad(grads[k].numpy().ravel()[:4], (0, max(0, 4 - grads[k].numel()))) for k in names])
    return out


def run_cgan():
    name = 'cfg5_cgan_step'
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    gcfg = dict(n_filters=8, n_blocks=3, decoder_upsampling='dc')
    dcfg = dict(upsampling='pin', scale=8, n_filters=8, n_res_blocks=2)
    B, H = 2, 32
    PG0 = M.init_params('unet_pin', (1, H, H, 5), (1, H, H, 1), **gcfg)
    PD0 = M.Params(create=True)
    from oracle import np_ops as N
    M.residual_discriminator(N, PD0, np.zeros((1, H, H, 5), np.float32), np.zeros((1, H, H, 1), np.float32), **dcfg)
    PG, PD = M.Params(), M.Params()
    for k, v in PG0.items():
        PG[k] = golden_weights('G/' + k, v.shape).astype(np.float64)
    for k, v in PD0.items():
        PD[k] = golden_weights('D/' + k, v.shape).astype(np.float64)
    lr = rng.random((B, H, H, 5)).astype(np.float32)
    st = rng.random((B, H, H, 1)).astype(np.float32)
    hr = rng.random((B, H, H, 1)).astype(np.float32)
    mask = (rng.random((2 * B, 16)) > 0.4).astype(np.float32)
    t64 = lambda a: T.asarray(a.astype(np.float64))
    r = TR.cgan_step('unet_pin', gcfg, M.convert(PG, T, requires_grad=True), dcfg, M.convert(PD, T, requires_grad=True),
                     t64(lr), t64(hr), t64(st), dropout_masks=(t64(mask[:B]), t64(mask[B:])))
    gn = sorted(r['gradsG'])
    dn = sorted(r['gradsD'])
    return dict(lr=lr, st=st, hr=hr, mask=mask, losses=np.array([r['gen_total'], r['gen_gan'], r['gen_px'], r['disc']]),
                gen=r['gen'].numpy().astype(np.float32), d_real=r['d_real'].numpy(), d_fake=r['d_fake'].numpy(),
                g_names=np.array(gn), g_norms=np.array([float(r['gradsG'][k].norm()) for k in gn]),
                d_names=np.array(dn), d_norms=np.array([float(r['gradsD'][k].norm()) for k in dn]))


if __name__ == '__main__':
    for name, c in CASES.items():
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **run_case(name, c))
        print('wrote', name)
    np.savez_compressed(os.path.join(HERE, 'cfg5_cgan_step.npz'), **run_cgan())
    print('wrote cfg5_cgan_step')
