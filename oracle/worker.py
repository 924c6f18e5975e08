"""Worker process of oracle/reference.py: evaluates the oracle for a subset of the samples of one job.

    python oracle/worker.py <job.npz> <worker index> <n workers> <out.npz>

The job file holds the model kind / configuration (JSON), the weights by name, the inputs and -- for a CGAN step -- the
discriminator's weights and dropout mask.  Sample i of the batch is handled by worker i % n.  Per sample the worker runs
  * the fp64 torch oracle with no displacement ('z': THE reference gradient),
  * twice more in fp64 with every derivative discontinuity displaced by +band and by -band (oracle/torch_ops.py: KINK): their
    difference is the per-entry allowance for a unit the two sides of whose kink lie within rounding distance, and
  * once in fp32 (band 0): what an independent single-precision evaluation of the same graph yields,
and returns the sums over its samples.  Losses that are batch means and models without batch statistics only."""
import json
import os
import sys

import numpy as np


def run_job(job, meta, mine):
    """``job``: mapping name -> array (an NpzFile or a dict), ``meta``: the decoded description, ``mine``: sample indices.
    -> dict of fp64 sums ('p/..', 'm/..', 'f32/..' gradients, 'loss/<tag>', 'pred', 'pred_idx')."""
    from oracle import torch_ops as T
    from oracle import models as M
    from oracle import train as TR
    B = int(meta['B'])
    band = float(meta['band'])
    files = set(job.files) if hasattr(job, 'files') else set(job.keys())

    def params(prefix, dt):
        P = M.Params()
        for k in meta[prefix + '_names']:
            P[k] = job[prefix + '/' + k].astype(dt)
        return M.convert(P, T, requires_grad=True)

    def arr(name, sl, dt):
        return None if name not in files else T.asarray(job[name][sl].astype(dt))

    acc = {}

    def add(tag, grads, scale):
        for k, v in grads.items():
            a = np.zeros(tuple(v_shape[k]), np.float64) if v is None else v.detach().numpy().astype(np.float64) * scale
            key = tag + '/' + k
            acc[key] = a if key not in acc else acc[key] + a

    losses = {}
    preds = []
    v_shape = {}
    for i in mine:
        sl = slice(i, i + 1)
        for tag, shift, dt in (('p', +band, np.float64), ('m', -band, np.float64), ('z', 0.0, np.float64), ('f32', 0.0, np.float32)):
            with T.kink_shift(shift):
                if meta['what'] == 'supervised':
                    PT = params('w', dt)
                    v_shape.update({k: v.shape for k, v in PT.items()})
                    lv, g, pred = TR.supervised_step(meta['kind'], meta['cfg'], PT, arr('x', sl, dt), arr('s', sl, dt), arr('y', sl, dt),
                                                     loss=meta['loss'])
                    add(tag, g, 1.0 / B)
                    losses.setdefault(tag, [0.0])[0] += lv / B
                else:
                    PG, PD = params('g', dt), params('d', dt)
                    v_shape.update({k: v.shape for k, v in PG.items()})
                    v_shape.update({k: v.shape for k, v in PD.items()})
                    mask = job['mask']
                    r = TR.cgan_step(meta['kind'], meta['cfg'], PG, meta['dcfg'], PD, arr('x', sl, dt), arr('y', sl, dt), arr('s', sl, dt),
                                     dropout_masks=(T.asarray(mask[:B][sl].astype(dt)), T.asarray(mask[B:][sl].astype(dt))),
                                     px_loss=meta['loss'])
                    add(tag + 'G', r['gradsG'], 1.0 / B)
                    add(tag + 'D', r['gradsD'], 1.0 / B)
                    cur = losses.setdefault(tag, [0.0, 0.0, 0.0, 0.0])
                    for j, k in enumerate(('gen_total', 'gen_gan', 'gen_px', 'disc')):
                        cur[j] += r[k] / B
    if meta['what'] == 'supervised':
        # the reference PREDICTION is the plain fp64 forward pass (no displacement): the mid-point of the displaced passes
        # is a reference for gradients, its forward values sit ~1e-5 off
        import torch
        with torch.no_grad():
            P0 = params('w', np.float64)
            for i in mine:
                sl = slice(i, i + 1)
                preds.append(TR.forward(meta['kind'], meta['cfg'], P0, arr('x', sl, np.float64), arr('s', sl, np.float64)).numpy())
    out = dict(acc)
    for tag, v in losses.items():
        out['loss/' + tag] = np.asarray(v, np.float64)
    if preds:
        out['pred'] = np.concatenate(preds, axis=0)
        out['pred_idx'] = np.asarray(mine)
    return out


def main():
    job_path, wi, nw, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import torch
    nt = int(os.environ.get('ORACLE_WORKER_THREADS', '0'))
    if nt > 0:
        torch.set_num_threads(nt)
    job = np.load(job_path, allow_pickle=False)
    meta = json.loads(str(job['meta']))
    np.savez(out_path, **run_job(job, meta, list(range(wi, int(meta['B']), nw))))


if __name__ == '__main__':
    main()
