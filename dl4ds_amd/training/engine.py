"""Thin Python handles over the C-ABI trainers (dl4ds_trainer_* / dl4ds_cgan_*): one call = one
optimisation step executed entirely by libdl4ds_hip.so (forward, loss, backward, RCCL gradient
all-reduce, Keras-form Adam)."""
import ctypes
from collections import OrderedDict

import numpy as np

from .. import _lib
from ..ops import LOSS_KINDS


def _lr_schedule(learning_rate, lr_decay_after):
    """supervised.py:336-352: a (lr0, lr1) pair becomes PiecewiseConstantDecay([lr_decay_after], [lr0, lr1])."""
    if isinstance(learning_rate, (tuple, list)) and len(learning_rate) > 1:
        return float(learning_rate[0]), float(learning_rate[1]), float(lr_decay_after)
    if isinstance(learning_rate, (tuple, list)):
        learning_rate = learning_rate[0]
    return float(learning_rate), float(learning_rate), 1e30


class SupervisedEngine:
    """fit() inner step of SupervisedTrainer.run (supervised.py:353,396-406)."""

    def __init__(self, model, loss='mae', learning_rate=1e-3, lr_decay_after=1e5, beta_1=0.9, beta_2=0.999,
                 epsilon=1e-7):
        if loss not in LOSS_KINDS:
            raise ValueError(f'loss {loss!r} not available on the MI355X path; one of {sorted(LOSS_KINDS)}')
        self.model = model
        self._l = _lib.lib()
        lr0, lr1, boundary = _lr_schedule(learning_rate, lr_decay_after)
        h = ctypes.c_void_p()
        _lib.check(self._l.dl4ds_trainer_create(model.graph.h, LOSS_KINDS[loss], lr0, lr1, boundary, beta_1, beta_2,
                                                epsilon, ctypes.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self._l.dl4ds_trainer_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _host_args(self, inputs, y_true):
        inputs, b = self.model._prep_inputs(inputs)
        y = np.ascontiguousarray(y_true, np.float32)
        if y.shape != (b,) + self.model.output_shape:
            raise ValueError(f'y_true shape {y.shape} != {(b,) + self.model.output_shape}')
        ptrs = (ctypes.c_void_p * len(inputs))(*[a.ctypes.data for a in inputs])
        return inputs, y, ptrs, b

    def step(self, inputs, y_true):
        """One optimisation step on host arrays; returns the batch loss (synchronises)."""
        inputs, y, ptrs, b = self._host_args(inputs, y_true)
        loss = ctypes.c_float()
        _lib.check(self._l.dl4ds_trainer_step(self.h, ptrs, len(inputs), y.ctypes.data, b, 1, ctypes.byref(loss)))
        return float(loss.value)

    def step_device(self, input_ptrs, y_ptr, batch, want_loss=False):
        """One step on HBM-resident buffers (raw device pointers); asynchronous unless want_loss."""
        ptrs = (ctypes.c_void_p * len(input_ptrs))(*input_ptrs)
        loss = ctypes.c_float()
        _lib.check(self._l.dl4ds_trainer_step(self.h, ptrs, len(input_ptrs), y_ptr, int(batch), 0,
                                              ctypes.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def loss_and_grads(self, inputs, y_true):
        """Loss and parameter gradients without the Adam update (model.evaluate / test hook)."""
        inputs, y, ptrs, b = self._host_args(inputs, y_true)
        loss = ctypes.c_float()
        _lib.check(self._l.dl4ds_trainer_loss_and_grads(self.h, ptrs, len(inputs), y.ctypes.data, b, 1,
                                                        ctypes.byref(loss)))
        return float(loss.value), self.model.get_gradients()

    def evaluate(self, inputs, y_true):
        """model.evaluate: inference-mode forward + loss (no dropout, BatchNormalization on its moving statistics)."""
        inputs, y, ptrs, b = self._host_args(inputs, y_true)
        loss = ctypes.c_float()
        _lib.check(self._l.dl4ds_trainer_evaluate(self.h, ptrs, len(inputs), y.ctypes.data, b, 1, ctypes.byref(loss)))
        return float(loss.value)

    def evaluate_device(self, input_ptrs, y_ptr, batch):
        """Loss on HBM-resident buffers (no update)."""
        ptrs = (ctypes.c_void_p * len(input_ptrs))(*input_ptrs)
        loss = ctypes.c_float()
        _lib.check(self._l.dl4ds_trainer_evaluate(self.h, ptrs, len(input_ptrs), y_ptr, int(batch), 0,
                                                  ctypes.byref(loss)))
        return float(loss.value)

    def save_checkpoint(self, path):
        """Weights (by variable name) + Adam slots + optimizer.iterations in one .npz -- the native counterpart of the
        reference's tf.train.Checkpoint (cgan.py:288-292) / saved model + trained_epochs (supervised.py:322-325)."""
        m, v, step = self.optimizer_state()
        blob = {'w/' + k: a for k, a in self.model.get_weights().items()}
        blob.update({'m/' + k: a for k, a in m.items()})
        blob.update({'v/' + k: a for k, a in v.items()})
        blob['step'] = np.int64(step)
        np.savez(path, **blob)

    def load_checkpoint(self, path):
        """Restore a `save_checkpoint` file into this engine (same architecture); returns optimizer.iterations."""
        z = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
        names = list(self.model.get_weights().keys())
        missing = [k for k in names if 'w/' + k not in z]
        if missing:
            raise ValueError(f'checkpoint lacks variables {missing[:3]}...')
        self.model.set_weights({k: z['w/' + k] for k in names})
        n_arena = ctypes.c_size_t()
        n_params = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_param_count(self.model.graph.h, ctypes.byref(n_arena), ctypes.byref(n_params)))
        m = np.zeros(n_arena.value, np.float32)
        v = np.zeros(n_arena.value, np.float32)
        for name, p in self.model.graph.params.items():
            off, n = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(self._l.dl4ds_graph_param_info(self.model.graph.h, p['pid'], ctypes.byref(off), ctypes.byref(n)))
            m[off.value:off.value + n.value] = np.asarray(z['m/' + name], np.float32).ravel()
            v[off.value:off.value + n.value] = np.asarray(z['v/' + name], np.float32).ravel()
        step = int(z['step'])
        _lib.check(self._l.dl4ds_trainer_set_state(self.h, m.ctypes.data, v.ctypes.data, step))
        return step

    def last_loss(self):
        loss = ctypes.c_float()
        _lib.check(self._l.dl4ds_trainer_last_loss(self.h, ctypes.byref(loss)))
        return float(loss.value)

    def optimizer_state(self):
        n_arena = ctypes.c_size_t()
        n_params = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_param_count(self.model.graph.h, ctypes.byref(n_arena), ctypes.byref(n_params)))
        m = np.empty(n_arena.value, np.float32)
        v = np.empty(n_arena.value, np.float32)
        step = ctypes.c_long()
        _lib.check(self._l.dl4ds_trainer_get_state(self.h, m.ctypes.data, v.ctypes.data, ctypes.byref(step)))
        out_m, out_v = OrderedDict(), OrderedDict()
        for name, p in self.model.graph.params.items():
            off, n = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(self._l.dl4ds_graph_param_info(self.model.graph.h, p['pid'], ctypes.byref(off), ctypes.byref(n)))
            out_m[name] = m[off.value:off.value + n.value].reshape(p['shape'])
            out_v[name] = v[off.value:off.value + n.value].reshape(p['shape'])
        return out_m, out_v, int(step.value)


def cgan_learning_rates(learning_rates):
    """cgan.py:271-278: ``genlr, dislr = learning_rates`` for a pair; a float or a 1-tuple sets both."""
    if isinstance(learning_rates, (tuple, list)) and len(learning_rates) > 1:
        if len(learning_rates) != 2:
            raise ValueError('`learning_rates` must be a float or a (generator, discriminator) pair')
        genlr, dislr = learning_rates
    elif isinstance(learning_rates, (tuple, list)) and len(learning_rates) == 1:
        genlr = dislr = learning_rates[0]
    elif isinstance(learning_rates, (int, float, np.floating)):
        genlr = dislr = learning_rates
    else:
        raise TypeError('`learning_rates` must be a float or a tuple/list of one or two floats')
    return float(genlr), float(dislr)


class CGANEngine:
    """train_step of the CGAN trainer (cgan.py:575-639): one simultaneous generator + discriminator update."""

    def __init__(self, generator, discriminator, loss='mae', learning_rate=2e-4, beta_1=0.5, lambda_scaling_factor=100.0):
        if loss not in LOSS_KINDS:
            raise ValueError(f'loss {loss!r} not available on the MI355X path; one of {sorted(LOSS_KINDS)}')
        genlr, dislr = cgan_learning_rates(learning_rate)
        self.generator, self.discriminator = generator, discriminator
        self.learning_rates = (genlr, dislr)
        self._l = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(self._l.dl4ds_cgan_create(generator.graph.h, discriminator.graph.h, LOSS_KINDS[loss],
                                             genlr, float(beta_1), float(lambda_scaling_factor), ctypes.byref(h)))
        self.h = h
        _lib.check(self._l.dl4ds_cgan_set_learning_rates(self.h, genlr, dislr))

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self._l.dl4ds_trainer_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def step(self, gen_inputs, hr_array, dropout_keep=None, apply_update=True):
        """Returns (gen_total_loss, gen_gan_loss, gen_px_loss, disc_loss).  dropout_keep: optional (2B, C) keep-mask
        (real rows first) for the discriminator's Dropout(0.4); None -> generated on the device."""
        inputs, b = self.generator._prep_inputs(gen_inputs)
        hr = np.ascontiguousarray(hr_array, np.float32)
        if hr.shape != (b,) + self.generator.output_shape:
            raise ValueError(f'hr_array shape {hr.shape} != {(b,) + self.generator.output_shape}')
        ptrs = (ctypes.c_void_p * len(inputs))(*[a.ctypes.data for a in inputs])
        mask = None
        if dropout_keep is not None:
            mask = np.ascontiguousarray(dropout_keep, np.float32)
        out = (ctypes.c_float * 4)()
        _lib.check(self._l.dl4ds_cgan_step(self.h, ptrs, len(inputs), hr.ctypes.data, b, 1,
                                           None if mask is None else mask.ctypes.data, int(apply_update), out))
        return tuple(float(v) for v in out)

    # --- optimiser state / checkpoints (tf.train.Checkpoint of both optimisers and both models, cgan.py:288-292,370-382)
    def _arena(self, model):
        n_arena, n_params = ctypes.c_size_t(), ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_param_count(model.graph.h, ctypes.byref(n_arena), ctypes.byref(n_params)))
        spans = {}
        for name, p in model.graph.params.items():
            off, n = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(self._l.dl4ds_graph_param_info(model.graph.h, p['pid'], ctypes.byref(off), ctypes.byref(n)))
            spans[name] = (off.value, n.value, p['shape'])
        return n_arena.value, spans

    def optimizer_state(self, which):
        """(m, v, iterations) of the generator (``which='generator'``) or discriminator optimiser, by variable name."""
        model = self.generator if which == 'generator' else self.discriminator
        n, spans = self._arena(model)
        m, v, step = np.empty(n, np.float32), np.empty(n, np.float32), ctypes.c_long()
        _lib.check(self._l.dl4ds_cgan_get_state(self.h, 0 if which == 'generator' else 1, m.ctypes.data, v.ctypes.data,
                                                ctypes.byref(step)))
        return (OrderedDict((k, m[o:o + c].reshape(s)) for k, (o, c, s) in spans.items()),
                OrderedDict((k, v[o:o + c].reshape(s)) for k, (o, c, s) in spans.items()), int(step.value))

    def save_checkpoint(self, path):
        blob = {}
        for which, model in (('generator', self.generator), ('discriminator', self.discriminator)):
            m, v, step = self.optimizer_state(which)
            blob.update({f'{which}/w/{k}': a for k, a in model.get_weights().items()})
            blob.update({f'{which}/m/{k}': a for k, a in m.items()})
            blob.update({f'{which}/v/{k}': a for k, a in v.items()})
            blob[f'{which}/step'] = np.int64(step)
        np.savez(path, **blob)

    def load_checkpoint(self, path):
        """Restore a `save_checkpoint` file (same architectures) -- cgan.py:447-522 `load_checkpoint`."""
        z = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
        for idx, (which, model) in enumerate((('generator', self.generator), ('discriminator', self.discriminator))):
            names = list(model.get_weights().keys())
            missing = [k for k in names if f'{which}/w/{k}' not in z]
            if missing:
                raise ValueError(f'checkpoint lacks {which} variables {missing[:3]}...')
            model.set_weights({k: z[f'{which}/w/{k}'] for k in names})
            n, spans = self._arena(model)
            m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
            for k, (o, c, s) in spans.items():
                m[o:o + c] = np.asarray(z[f'{which}/m/{k}'], np.float32).ravel()
                v[o:o + c] = np.asarray(z[f'{which}/v/{k}'], np.float32).ravel()
            _lib.check(self._l.dl4ds_cgan_set_state(self.h, idx, m.ctypes.data, v.ctypes.data, int(z[f'{which}/step'])))

    def step_device(self, input_ptrs, hr_ptr, batch, want_losses=False):
        ptrs = (ctypes.c_void_p * len(input_ptrs))(*input_ptrs)
        out = (ctypes.c_float * 4)()
        _lib.check(self._l.dl4ds_cgan_step(self.h, ptrs, len(input_ptrs), hr_ptr, int(batch), 0, None, 1,
                                           out if want_losses else None))
        return tuple(float(v) for v in out) if want_losses else None
