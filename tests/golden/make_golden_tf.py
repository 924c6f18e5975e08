"""Pin the oracle against the REAL reference: rebuild the tests/golden/*.npz cases with carlos-gg/dl4ds itself under
TensorFlow and compare (default) or rewrite (--write) the committed fixtures.

    python tests/golden/make_golden_tf.py --reference /path/to/dl4ds-checkout            # compare, exit 1 on mismatch
    python tests/golden/make_golden_tf.py --reference /path/to/dl4ds-checkout --write    # overwrite the .npz files

Needs an environment in which `import tensorflow` (2.6 - 2.15, Keras 2) and `import dl4ds` from that checkout work
(the reference's setup.py dependencies: tensorflow, opencv-python, xarray, ecubevis, ...).  Such an environment does
NOT exist in the build container of this repository (no TensorFlow wheel, no network) nor on the GPU box, so this
script has never been executed there: the committed fixtures are ORACLE outputs ("parity unpinned")
until someone runs it.  What IS checked without TensorFlow (tests/test_name_map.py): the name map used below covers
every variable of every builder configuration and walks attributes that exist in the reference classes.

For every case of tests/golden/make_golden.py this script
  1. builds the reference model with the same builder arguments (dl4ds.models.*),
  2. writes the name-derived pseudo-random weights of `golden_weights` into the Keras variables through
     tests/golden/keras_name_map.py (and checks that EVERY Keras weight was hit exactly once),
  3. runs forward + loss + tf.GradientTape gradients on the same seeded inputs,
  4. stores / compares `pred`, `loss`, per-variable gradient norms and the first four gradient entries.
The CGAN case replaces the discriminator's Dropout(0.4) by the fixture's keep-mask and evaluates generator_loss /
discriminator_loss (training/cgan.py:525-572) on the reference's own functions.
"""
import argparse
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import keras_name_map as K                                  # noqa: E402
from make_golden import CASES, golden_weights               # noqa: E402  (pure numpy part of the oracle generator)
from oracle import models as M                              # noqa: E402  (variable names + shapes only)
from oracle import np_ops as N                              # noqa: E402

TOL = 1e-3          # north_star: outputs within 1e-3 relative fp32


def build_reference(tf, dl4ds, model, cfg, x_shape, s_shape):
    tf.keras.backend.clear_session()                         # auto-named layers restart at conv2d / conv_block / ...
    mods = dl4ds.models
    n_aux = 0 if s_shape is None else s_shape[-1]
    if model == 'net_pin':
        return mods.net_pin(n_channels=x_shape[-1], n_aux_channels=n_aux, hr_size=x_shape[1:3], **cfg)
    if model == 'net_postupsampling':
        return mods.net_postupsampling(n_channels=x_shape[-1], n_aux_channels=n_aux, lr_size=x_shape[1:3], **cfg)
    if model == 'unet_pin':
        return mods.unet_pin('unet', n_channels=x_shape[-1], n_aux_channels=n_aux, hr_size=x_shape[1:3], **cfg)
    if model == 'recnet_postupsampling':
        return mods.recnet_postupsampling(n_channels=x_shape[-1], n_aux_channels=n_aux, lr_size=x_shape[2:4], **cfg)
    if model == 'recnet_pin':
        return mods.recnet_pin(n_channels=x_shape[-1], n_aux_channels=n_aux, hr_size=x_shape[2:4], **cfg)
    raise KeyError(model)


def to_keras_layout(var, value, name):
    """Arena layout -> Keras variable layout.  Identical (HWIO / HWOI / [in,out] / 1-D) except LocallyConnected2D with
    implementation=3 (blocks.py:322-328), whose kernel is the 1-D vector of the entries listed in `kernel_idxs`
    (input flat index, output flat index), while the arena holds it as (H, W, C, F)."""
    if tuple(var.shape) == tuple(value.shape):
        return value
    if name.endswith('localconv/kernel') and len(var.shape) == 1 and int(var.shape[0]) == value.size:
        raise NotImplementedError('LocallyConnected2D(implementation=3) kernel: map through layer.kernel_idxs '
                                  '(see load_weights)')
    raise ValueError(f'{name}: Keras shape {tuple(var.shape)} vs arena shape {value.shape}')


def load_weights(model, names_shapes, prefix='', spt_disc=False):
    """golden_weights(name) into every Keras variable; returns {arena name: tf.Variable}."""
    hit, out = set(), {}
    for name, shape in names_shapes.items():
        var = K.resolve(model, name, spatiotemporal_discriminator=spt_disc)
        value = golden_weights(prefix + name, shape)
        if name.endswith('localconv/kernel') and len(var.shape) == 1:
            sel, chain, _ = K.parse(name)
            layer = K.select_layer(model, sel)
            for a in chain:
                layer = getattr(layer, a)
            H, W, C, F = value.shape
            flat = np.empty(int(var.shape[0]), np.float32)
            for i, (a, b) in enumerate(sorted(layer.kernel_idxs)):     # (input flat idx, output flat idx), channels_last
                h, w, c = np.unravel_index(a, (H, W, C))
                flat[i] = value[h, w, c, b % F]
            value = flat
        var.assign(to_keras_layout(var, value, name))
        assert id(var) not in hit, f'{name}: two arena variables map onto {var.name}'
        hit.add(id(var))
        out[name] = var
    missed = [w.name for w in model.weights if id(w) not in hit]
    assert not missed, f'Keras weights without an arena variable: {missed}'
    return out


def run_case(tf, dl4ds, name, c):
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 1)
    P0 = M.init_params(c['model'], (1,) + c['x'][1:], None if c['s'] is None else (1,) + c['s'][1:], **c['cfg'])
    model = build_reference(tf, dl4ds, c['model'], c['cfg'], c['x'], c['s'])
    var_of = load_weights(model, {k: tuple(v.shape) for k, v in P0.items()})
    x = rng.standard_normal(c['x']).astype(np.float32)
    s = None if c['s'] is None else rng.standard_normal(c['s']).astype(np.float32)
    inputs = [tf.constant(x)] if s is None else [tf.constant(x), tf.constant(s)]
    pred0 = model(inputs, training=True)
    y = rng.random(tuple(pred0.shape)).astype(np.float32)
    lossf = dl4ds.utils.checkarg_loss(c['loss'])
    with tf.GradientTape() as tape:
        pred = model(inputs, training=True)
        lv = lossf(tf.constant(y), pred)
    tv = {id(v): v for v in model.trainable_variables}
    names = sorted(k for k, v in var_of.items() if id(v) in tv)
    grads = tape.gradient(lv, [var_of[k] for k in names])
    g = {k: np.asarray(t, np.float64) for k, t in zip(names, grads)}
    out = dict(x=x, y_true=y, pred=np.asarray(pred, np.float32), loss=np.float64(lv))
    if s is not None:
        out['s'] = s
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array([float(np.linalg.norm(g[k])) for k in names])
    out['grad_heads'] = np.stack([np.pad(g[k].ravel()[:4], (0, max(0, 4 - g[k].size))) for k in names])
    return out


def run_cgan(tf, dl4ds):
    name = 'cfg5_cgan_step'
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    gcfg = dict(n_filters=8, n_blocks=3, decoder_upsampling='dc')
    dcfg = dict(upsampling='pin', scale=8, n_filters=8, n_res_blocks=2)
    B, H = 2, 32
    PG0 = M.init_params('unet_pin', (1, H, H, 5), (1, H, H, 1), **gcfg)
    PD0 = M.Params(create=True)
    M.residual_discriminator(N, PD0, np.zeros((1, H, H, 5), np.float32), np.zeros((1, H, H, 1), np.float32), **dcfg)
    tf.keras.backend.clear_session()
    gen = dl4ds.models.unet_pin('unet', n_channels=5, n_aux_channels=1, hr_size=(H, H), **gcfg)
    vg = load_weights(gen, {k: tuple(v.shape) for k, v in PG0.items()}, prefix='G/')
    tf.keras.backend.clear_session()
    disc = dl4ds.models.residual_discriminator(n_channels=5, upsampling='pin', is_spatiotemporal=False, scale=8,
                                               lr_size=(H // 8, H // 8), n_filters=8, n_res_blocks=2)
    vd = load_weights(disc, {k: tuple(v.shape) for k, v in PD0.items()}, prefix='D/')
    lr = rng.random((B, H, H, 5)).astype(np.float32)
    st = rng.random((B, H, H, 1)).astype(np.float32)
    hr = rng.random((B, H, H, 1)).astype(np.float32)
    mask = (rng.random((2 * B, 16)) > 0.4).astype(np.float32)
    # Dropout(0.4) (discriminator.py:77) -> the fixture's keep-mask (inverted dropout: kept units scaled by 1/0.6)
    drop = [l for l in disc.layers if type(l).__name__ == 'Dropout']
    assert len(drop) == 1
    holder = {}
    drop[0].call = lambda inputs, training=None: inputs * holder['m'] / 0.6
    from dl4ds.training.cgan import generator_loss, discriminator_loss
    px = dl4ds.utils.checkarg_loss('mae')
    with tf.GradientTape() as gt, tf.GradientTape() as dt:
        g_out = gen([tf.constant(lr), tf.constant(st)], training=True)
        holder['m'] = tf.constant(mask[:B])
        d_real = disc([tf.constant(lr), tf.constant(hr)], training=True)
        holder['m'] = tf.constant(mask[B:])
        d_fake = disc([tf.constant(lr), g_out], training=True)
        gen_total, gen_gan, gen_px = generator_loss(d_fake, g_out, tf.constant(hr), px)
        d_loss = discriminator_loss(d_real, d_fake)
    gn = sorted(k for k, v in vg.items() if any(v is t for t in gen.trainable_variables))
    dn = sorted(k for k, v in vd.items() if any(v is t for t in disc.trainable_variables))
    gg = gt.gradient(gen_total, [vg[k] for k in gn])
    dg = dt.gradient(d_loss, [vd[k] for k in dn])
    return dict(lr=lr, st=st, hr=hr, mask=mask,
                losses=np.array([float(gen_total), float(gen_gan), float(gen_px), float(d_loss)]),
                gen=np.asarray(g_out, np.float32), d_real=np.asarray(d_real), d_fake=np.asarray(d_fake),
                g_names=np.array(gn), g_norms=np.array([float(np.linalg.norm(np.asarray(t, np.float64))) for t in gg]),
                d_names=np.array(dn), d_norms=np.array([float(np.linalg.norm(np.asarray(t, np.float64))) for t in dg]))


def compare(name, new, old):
    """Relative deviations of the reference's results from the committed (oracle) fixture."""
    bad = []
    for k in old.files:
        a, b = np.asarray(old[k]), np.asarray(new[k])
        if a.dtype.kind in 'US':
            if list(a) != list(b):
                bad.append(f'{k}: variable lists differ: {sorted(set(a) ^ set(b))[:6]}')
            continue
        if a.shape != b.shape:
            bad.append(f'{k}: shape {a.shape} (fixture) vs {b.shape} (reference)')
            continue
        scale = max(float(np.abs(a).max()), 1e-30)
        err = float(np.abs(a.astype(np.float64) - b).max()) / scale
        flag = 'ok' if err <= TOL else 'MISMATCH'
        print(f'  {name:22s} {k:12s} max |oracle - reference| / max|oracle| = {err:.3e}  {flag}')
        if err > TOL:
            bad.append(f'{k}: {err:.3e}')
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', required=True, help='path of a carlos-gg/dl4ds checkout (the directory holding dl4ds/)')
    ap.add_argument('--write', action='store_true', help='overwrite tests/golden/*.npz with the reference results')
    args = ap.parse_args()
    sys.path.insert(0, os.path.abspath(args.reference))
    import tensorflow as tf
    import dl4ds
    print('tensorflow', tf.__version__, '| dl4ds', getattr(dl4ds, '__version__', '?'))
    failures = {}
    results = {n: run_case(tf, dl4ds, n, c) for n, c in CASES.items()}
    results['cfg5_cgan_step'] = run_cgan(tf, dl4ds)
    for n, r in results.items():
        path = os.path.join(HERE, n + '.npz')
        if args.write:
            np.savez_compressed(path, **r)
            print('wrote', path)
        else:
            bad = compare(n, r, np.load(path))
            if bad:
                failures[n] = bad
    if failures:
        print('\nThe oracle disagrees with the reference beyond', TOL, ':')
        for n, bad in failures.items():
            print(' ', n, bad)
        raise SystemExit(1)
    if not args.write:
        print('\nAll committed fixtures agree with the reference within', TOL, '-- the oracle is pinned for these cases.')


if __name__ == '__main__':
    main()
