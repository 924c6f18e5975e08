"""Python side of the graph runtime: lowers layer calls to the C-ABI graph (dl4ds_graph_*) and wraps
the result in a Keras-like ``Model`` (the surface the reference uses: ``model(inputs, training=)``,
``.predict``, ``.name``, ``.count_params``, ``.get_weights/.set_weights``, ``.summary`` --
dl4ds/training/supervised.py:320,396-409; cgan.py:597-600; inference.py:172-173,238)."""
import ctypes
from collections import OrderedDict

import numpy as np

from . import _lib

ACT_KINDS = {'relu': 1, 'sigmoid': 2, 'tanh': 3, 'elu': 4, 'leaky_relu': 5, 'selu': 6, 'gelu': 7}


class Tensor:
    __slots__ = ('id', 'H', 'W', 'C', 'nmul')

    def __init__(self, id, H, W, C, nmul=1):
        self.id, self.H, self.W, self.C, self.nmul = id, H, W, C, nmul

    @property
    def shape(self):
        return (self.nmul, self.H, self.W, self.C)


def _check_activation(act):
    if act is None or act == 'linear':
        return None
    if act not in ACT_KINDS:
        raise ValueError(f'activation {act!r} not supported; one of {sorted(ACT_KINDS)} or None')
    return act


class GraphBuilder:
    """Thin object API over dl4ds_graph_*; owns parameter names, shapes and initialisers."""

    def __init__(self):
        self._l = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(self._l.dl4ds_graph_create(ctypes.byref(h)))
        self.h = h
        self.params = OrderedDict()          # name -> dict(pid, shape, init)
        self.inputs = []
        self.outputs = []
        self.layers = []                     # (kind, name, out_shape) for summary()
        self.finalized = False

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self._l.dl4ds_graph_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---------------------------------------------------------------- tensors / params
    def input(self, H, W, C, nmul=1, requires_grad=False):
        tid = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_input(self.h, int(H), int(W), int(C), int(nmul), ctypes.byref(tid)))
        if requires_grad:
            _lib.check(self._l.dl4ds_graph_input_requires_grad(self.h, tid.value))
        t = Tensor(tid.value, int(H), int(W), int(C), int(nmul))
        self.inputs.append(t)
        return t

    def param(self, name, shape, init='glorot', trainable=True):
        shape = tuple(int(s) for s in shape)
        if name in self.params:                 # shared weights (e.g. spc conv2x applied twice)
            p = self.params[name]
            if p['shape'] != shape:
                raise ValueError(f'parameter {name} re-used with shape {shape} != {p["shape"]}')
            return p['pid']
        pid = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_param(self.h, int(np.prod(shape)), ctypes.byref(pid)))
        self.params[name] = dict(pid=pid.value, shape=shape, init=init, trainable=trainable)
        return pid.value

    def _out(self, tid, kind, name):
        s = (ctypes.c_int * 4)()
        _lib.check(self._l.dl4ds_graph_tensor_shape(self.h, tid, s))
        t = Tensor(tid, s[1], s[2], s[3], s[0])
        self.layers.append((kind, name, t.shape))
        return t

    # ---------------------------------------------------------------- ops
    def conv2d(self, x, name, filters, ks, use_bias=True, activation=None, add=None, d2s=0, dense=False):
        """``dense=True``: a Keras Dense layer applied to the channel axis (ConvNextBlock.pwconv1/2, blocks.py:146,149)
        -- a 1x1 convolution whose kernel variable keeps the Dense shape (Cin, units)."""
        activation = _check_activation(activation)
        w = self.param(name + '/kernel', (x.C, filters) if dense else (ks, ks, x.C, filters))
        b = self.param(name + '/bias', (filters,), 'zeros') if use_bias else -1
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_conv2d(self.h, x.id, w, b, -1 if add is None else add.id, int(ks),
                                              int(filters), int(activation == 'relu'), int(d2s), ctypes.byref(out)))
        y = self._out(out.value, 'conv2d', name)
        if activation not in (None, 'relu'):
            y = self.act(y, activation, name + '/act')
        return y

    def conv2d_folded(self, x, name, filters, ks, d2s, name2, filters2, activation=None, aux=None):
        """Conv2D(ks, d2s^2 * filters) [+ depth_to_space(d2s)] directly followed by Conv2D(1x1, filters2) + activation,
        evaluated with the composed filter (csrc/graph_ops3.hip).  Variables are those of the two Keras layers.
        ``aux``: an HR tensor that the reference concatenates to the first convolution's output before the 1x1 layer
        (Concatenate([x, s]) -> TransitionLast, sp_postups.py:184-203): the 1x1 kernel then has filters + aux.C input
        channels and the auxiliary part is added as a separate 1x1 convolution."""
        activation = _check_activation(activation)
        r2 = d2s * d2s if d2s and d2s > 1 else 1
        w1 = self.param(name + '/kernel', (ks, ks, x.C, r2 * filters))
        b1 = self.param(name + '/bias', (r2 * filters,), 'zeros')
        w2 = self.param(name2 + '/kernel', (1, 1, filters + (aux.C if aux is not None else 0), filters2))
        b2 = self.param(name2 + '/bias', (filters2,), 'zeros')
        out = ctypes.c_int()
        if aux is not None:
            _lib.check(self._l.dl4ds_graph_conv2d_folded_aux(self.h, x.id, aux.id, w1, b1, w2, b2, int(ks), int(filters),
                                                             int(filters2), int(activation == 'relu'), int(d2s or 0),
                                                             ctypes.byref(out)))
        else:
            _lib.check(self._l.dl4ds_graph_conv2d_folded(self.h, x.id, w1, b1, w2, b2, int(ks), int(filters), int(filters2),
                                                         int(activation == 'relu'), int(d2s or 0), ctypes.byref(out)))
        y = self._out(out.value, 'conv2d+conv1x1 (folded)', name + ' -> ' + name2)
        if activation not in (None, 'relu'):
            y = self.act(y, activation, name2 + '/act')
        return y

    def dwconv(self, x, name, ks=7, use_bias=True):
        """DepthwiseConv2D(kernel_size=ks, padding='same', depth_multiplier=1) -- blocks.py:143-144."""
        w = self.param(name + '/depthwise_kernel', (ks, ks, x.C, 1))
        b = self.param(name + '/bias', (x.C,), 'zeros') if use_bias else -1
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_dwconv(self.h, x.id, w, b, int(ks), ctypes.byref(out)))
        return self._out(out.value, 'depthwise_conv2d', name)

    def conv2d_transpose(self, x, name, filters, ks, stride, activation=None):
        activation = _check_activation(activation)
        w = self.param(name + '/kernel', (ks, ks, filters, x.C))
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_conv2d_transpose(self.h, x.id, w, int(ks), int(stride), int(filters),
                                                        int(activation == 'relu'), ctypes.byref(out)))
        y = self._out(out.value, 'conv2d_transpose', name)
        if activation not in (None, 'relu'):
            y = self.act(y, activation, name + '/act')
        return y

    def channel_attention(self, x, name, nf, r=4, time_window_5d=0):
        cr = int(nf / r)
        if cr < 1:
            raise ValueError(f'ChannelAttention2D needs nf >= r (got nf={nf}, r={r})')
        w1 = self.param(name + '/conv1/kernel', (1, 1, x.C, cr))
        b1 = self.param(name + '/conv1/bias', (cr,), 'zeros')
        w2 = self.param(name + '/conv2/kernel', (1, 1, cr, nf))
        b2 = self.param(name + '/conv2/bias', (nf,), 'zeros')
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_chatt(self.h, x.id, w1, b1, w2, b2, cr, int(time_window_5d), ctypes.byref(out)))
        return self._out(out.value, 'channel_attention', name)

    def concat(self, xs, name='concat'):
        ids = (ctypes.c_int * len(xs))(*[t.id for t in xs])
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_concat(self.h, ids, len(xs), ctypes.byref(out)))
        return self._out(out.value, 'concat', name)

    def add(self, a, b, relu=False, name='add'):
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_add(self.h, a.id, b.id, int(relu), ctypes.byref(out)))
        return self._out(out.value, 'add', name)

    def act(self, x, kind, name='act'):
        kind = _check_activation(kind)
        if kind is None:
            return x
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_act(self.h, x.id, ACT_KINDS[kind], ctypes.byref(out)))
        return self._out(out.value, 'activation:' + kind, name)

    def maxpool2(self, x, name='maxpool'):
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_maxpool2(self.h, x.id, ctypes.byref(out)))
        return self._out(out.value, 'maxpool2', name)

    def resize(self, x, ho, wo, name='resize', interpolation='bilinear'):
        methods = {'bilinear': 0, 'nearest': 1, 'bicubic': 2, 'lanczos3': 3, 'lanczos5': 4, 'gaussian': 5, 'mitchellcubic': 6}
        if interpolation == 'area':
            raise NotImplementedError("Resizing(interpolation='area'): TensorFlow's ResizeArea op has no gradient, the reference "
                                      "cannot train a model built with it either")
        if interpolation not in methods:
            raise ValueError(f'unknown Resizing interpolation {interpolation!r}; one of {sorted(methods)}')
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_resize_method(self.h, x.id, int(ho), int(wo), methods[interpolation], ctypes.byref(out)))
        return self._out(out.value, 'resize_' + interpolation, name)

    def localconv(self, x, name, filters=2, use_bias=True):
        w = self.param(name + '/kernel', (x.H, x.W, x.C, filters))
        b = self.param(name + '/bias', (x.H, x.W, filters), 'zeros') if use_bias else -1
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_localconv(self.h, x.id, w, b, int(filters), ctypes.byref(out)))
        return self._out(out.value, 'locally_connected', name)

    def rec_tail_supported(self, cx, cs, co):
        yes = ctypes.c_int(0)
        _lib.check(self._l.dl4ds_rec_tail_supported(int(cx), int(cs), int(co), ctypes.byref(yes)))
        return bool(yes.value)

    def rec_tail(self, x, s, T, co, lcb_name='LocalizedConvBlock', tl_name='TransitionLast'):
        """[x, repeat(s, T)] -> LocalizedConvBlock -> Concatenate -> TransitionLast as one op (csrc/graph_ops4.hip); the
        variables carry the names and shapes of the separate layers (creation order as in models/spt_postups.py: rec_tail)."""
        c24 = x.C + s.C
        wt = self.param(lcb_name + '/transition/conv/kernel', (1, 1, c24, 2))
        bt = self.param(lcb_name + '/transition/conv/bias', (2,), 'zeros')
        wl = self.param(lcb_name + '/localconv/kernel', (x.H, x.W, 2, 2))
        bl = self.param(lcb_name + '/localconv/bias', (x.H, x.W, 2), 'zeros')
        w = self.param(tl_name + '/conv/kernel', (1, 1, c24 + 2, co))
        b = self.param(tl_name + '/conv/bias', (co,), 'zeros')
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_rec_tail(self.h, x.id, s.id, wt, bt, wl, bl, w, b, int(T), int(co), ctypes.byref(out)))
        return self._out(out.value, 'rec_tail', tl_name)

    def repeat_time(self, x, T, name='repeat_time'):
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_repeat_time(self.h, x.id, int(T), ctypes.byref(out)))
        return self._out(out.value, 'repeat_time', name)

    def convlstm(self, x, name, filters, ks, T, activation=None):
        activation = _check_activation(activation)
        wk = self.param(name + '/kernel', (ks, ks, x.C, 4 * filters))
        wr = self.param(name + '/recurrent_kernel', (ks, ks, filters, 4 * filters), 'orthogonal')
        b = self.param(name + '/bias', (4 * filters,), 'lstm_bias')
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_convlstm(self.h, x.id, wk, wr, b, int(ks), int(filters), int(T),
                                                int(activation == 'relu'), ctypes.byref(out)))
        y = self._out(out.value, 'convlstm2d', name)
        if activation not in (None, 'relu'):
            y = self.act(y, activation, name + '/act')
        return y

    def gap(self, x, name='gap', over_time=False):
        out = ctypes.c_int()
        fn = self._l.dl4ds_graph_gap3d if over_time else self._l.dl4ds_graph_gap
        _lib.check(fn(self.h, x.id, ctypes.byref(out)))
        return self._out(out.value, 'global_avg_pool', name)

    def slice2d(self, x, oy, ox, step, ho, wo, name='slice'):
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_slice(self.h, x.id, int(oy), int(ox), int(step), int(ho), int(wo), ctypes.byref(out)))
        return self._out(out.value, 'slice', name)

    def pad_bottom_right(self, x, ho, wo, name='zero_padding'):
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_pad(self.h, x.id, int(ho), int(wo), ctypes.byref(out)))
        return self._out(out.value, 'zero_padding', name)

    def conv2d_strided(self, x, name, filters, ks, stride, padding='same'):
        """Conv2D(filters, ks, strides=stride, padding=...) as the stride-1 'same' convolution (MFMA path) followed by
        the sub-sampling slice; see csrc/graph_ops3.hip for the index algebra."""
        if padding not in ('same', 'valid'):
            raise ValueError(padding)
        half = ks // 2

        def geom(n):
            if padding == 'same':
                out = -(-n // stride)
                pad = max((out - 1) * stride + ks - n, 0)
                return out, half - pad // 2
            return (n - ks) // stride + 1, half
        (ho, oy), (wo, ox) = geom(x.H), geom(x.W)
        y = self.conv2d(x, name, filters, ks)
        return self.slice2d(y, oy, ox, stride, ho, wo, name + '/stride')

    def dense(self, x, name, units, activation=None):
        activation = _check_activation(activation)
        w = self.param(name + '/kernel', (x.C, units))
        b = self.param(name + '/bias', (units,), 'zeros')
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_dense(self.h, x.id, w, b, int(units), ACT_KINDS.get(activation, 0),
                                             ctypes.byref(out)))
        return self._out(out.value, 'dense', name)

    def dropout(self, x, rate, name='dropout', variant=None, dim=2):
        """get_dropout_layer -- blocks.py:679-706: identity for rate 0; 'vanilla' / 'gaussian' / 'spatial' and their
        'mc*' forms that stay active at inference.  ``dim=3``: SpatialDropout3D (mask shared over the time axis)."""
        if not rate or rate <= 0:
            return x
        kinds = {None: (0, 0), 'vanilla': (0, 0), 'gaussian': (1, 0), 'spatial': (2, 0), 'mcdrop': (0, 1),
                 'mcgaussiandrop': (1, 1), 'mcspatialdrop': (2, 1)}
        if variant not in kinds:
            raise ValueError(f'dropout variant {variant!r} not supported')
        kind, mc = kinds[variant]
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_dropout_variant(self.h, x.id, float(rate), kind, mc, int(dim), ctypes.byref(out)))
        return self._out(out.value, 'dropout' if variant is None else variant, name)

    def norm(self, x, name, kind, activation=None, epsilon=1e-3):
        """LayerNormalization() ('ln') / BatchNormalization() ('bn') with Keras variable names; a ReLU that follows is
        fused, any other activation is appended."""
        activation = _check_activation(activation)
        if kind not in ('bn', 'ln'):
            raise ValueError(f'Normalization not supported, got {kind}')
        gamma = self.param(name + '/gamma', (x.C,), 'ones')
        beta = self.param(name + '/beta', (x.C,), 'zeros')
        mm = mv = -1
        if kind == 'bn':
            mm = self.param(name + '/moving_mean', (x.C,), 'zeros', trainable=False)
            mv = self.param(name + '/moving_variance', (x.C,), 'ones', trainable=False)
        out = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_norm(self.h, x.id, gamma, beta, mm, mv, int(kind == 'bn'), float(epsilon),
                                            int(activation == 'relu'), ctypes.byref(out)))
        y = self._out(out.value, 'batch_norm' if kind == 'bn' else 'layer_norm', name)
        if activation not in (None, 'relu'):
            y = self.act(y, activation, name + '/act')
        return y

    def norm_variables(self, name, channels, kind):
        """Variables of a normalisation layer the reference builds and calls but whose output it discards
        (DenseBlock.norm1, blocks.py:263-267): present in the weight list, absent from the graph."""
        self.param(name + '/gamma', (channels,), 'ones')
        self.param(name + '/beta', (channels,), 'zeros')
        if kind == 'bn':
            self.param(name + '/moving_mean', (channels,), 'zeros', trainable=False)
            self.param(name + '/moving_variance', (channels,), 'ones', trainable=False)

    # ---------------------------------------------------------------- dropout noise (tests / reproducibility)
    def dropout_count(self):
        n = ctypes.c_int()
        _lib.check(self._l.dl4ds_graph_dropout_count(self.h, ctypes.byref(n)))
        return n.value

    def dropout_mask(self, index, batch):
        """Noise the last forward pass of dropout op ``index`` used (flat array, see include/dl4ds_hip.h)."""
        n = ctypes.c_size_t()
        _lib.check(self._l.dl4ds_graph_dropout_mask_size(self.h, int(index), int(batch), ctypes.byref(n)))
        out = np.empty(n.value, np.float32)
        _lib.check(self._l.dl4ds_graph_dropout_get_mask(self.h, int(index), int(batch), out.ctypes.data))
        return out

    def set_dropout_mask(self, index, batch, mask):
        n = ctypes.c_size_t()
        _lib.check(self._l.dl4ds_graph_dropout_mask_size(self.h, int(index), int(batch), ctypes.byref(n)))
        mask = np.ascontiguousarray(mask, np.float32).ravel()
        if mask.size != n.value:
            raise ValueError(f'dropout op {index}: mask of {n.value} entries expected, got {mask.size}')
        _lib.check(self._l.dl4ds_graph_dropout_set_mask(self.h, int(index), int(batch), mask.ctypes.data))

    # ---------------------------------------------------------------- finish
    def finalize(self, output, seed=None):
        _lib.check(self._l.dl4ds_graph_output(self.h, output.id))
        self.outputs.append(output)
        _lib.check(self._l.dl4ds_graph_finalize(self.h))
        self.finalized = True
        self.initialize(seed)

    def initialize(self, seed=None):
        """Keras default initialisers: glorot_uniform kernels, zero biases, ConvLSTM2D orthogonal
        recurrent kernel and unit forget bias."""
        rng = np.random.default_rng(seed)
        for name, p in self.params.items():
            shape, init = p['shape'], p['init']
            if init == 'zeros':
                val = np.zeros(shape, np.float32)
            elif init == 'ones':
                val = np.ones(shape, np.float32)
            elif init == 'lstm_bias':
                f = shape[0] // 4
                val = np.zeros(shape, np.float32)
                val[f:2 * f] = 1.0
            elif init == 'orthogonal':
                rows = int(np.prod(shape[:-1]))
                a = rng.standard_normal((max(rows, shape[-1]), min(rows, shape[-1])))
                q, r = np.linalg.qr(a)
                q *= np.sign(np.diag(r))
                if rows < shape[-1]:
                    q = q.T
                val = q[:rows, :shape[-1]].reshape(shape).astype(np.float32)
            else:
                if len(shape) == 4:
                    rf = shape[0] * shape[1]
                    fan_in, fan_out = rf * shape[2], rf * shape[3]
                else:
                    fan_in, fan_out = shape[0], shape[-1]
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                val = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            self.set_param(name, val)

    def set_param(self, name, value):
        p = self.params[name]
        value = np.ascontiguousarray(value, np.float32)
        if value.shape != p['shape']:
            raise ValueError(f'{name}: expected shape {p["shape"]}, got {value.shape}')
        _lib.check(self._l.dl4ds_graph_set_param(self.h, p['pid'], value.ctypes.data))

    def get_param(self, name, grad=False):
        p = self.params[name]
        out = np.empty(p['shape'], np.float32)
        fn = self._l.dl4ds_graph_get_grad if grad else self._l.dl4ds_graph_get_param
        _lib.check(fn(self.h, p['pid'], out.ctypes.data))
        return out


class Model:
    """What a dl4ds builder returns (stands in for tf.keras.Model on the hot path)."""

    def __init__(self, gb, name, input_shapes):
        self.graph = gb
        self.name = name
        self.input_shapes = input_shapes      # list of per-sample shapes, Keras style (H,W,C) / (T,H,W,C)
        self.output_shape = self._keras_shape(gb.outputs[0])

    @staticmethod
    def _keras_shape(t):
        return (t.H, t.W, t.C) if t.nmul == 1 else (t.nmul, t.H, t.W, t.C)

    # --- weights
    def count_params(self):
        return int(sum(np.prod(p['shape']) for p in self.graph.params.values()))

    @property
    def weight_names(self):
        return list(self.graph.params.keys())

    def get_weights(self):
        return OrderedDict((k, self.graph.get_param(k)) for k in self.graph.params)

    def set_weights(self, weights):
        if isinstance(weights, dict):
            for k, v in weights.items():
                self.graph.set_param(k, np.asarray(v))
        else:
            for k, v in zip(self.graph.params, weights):
                self.graph.set_param(k, np.asarray(v))

    def get_gradients(self):
        return OrderedDict((k, self.graph.get_param(k, grad=True)) for k in self.graph.params)

    @property
    def variables(self):
        return list(self.get_weights().items())

    @property
    def trainable_variables(self):
        """Everything but the BatchNormalization moving statistics (they live in the same arena; their gradient is
        identically zero, so Adam leaves them to the forward pass that maintains them)."""
        return [(k, v) for k, v in self.get_weights().items() if self.graph.params[k].get('trainable', True)]

    # --- forward
    def _prep_inputs(self, inputs):
        if isinstance(inputs, np.ndarray):
            inputs = [inputs]
        inputs = [np.ascontiguousarray(a, np.float32) for a in inputs]
        if len(inputs) != len(self.graph.inputs):
            raise ValueError(f'model {self.name} expects {len(self.graph.inputs)} inputs, got {len(inputs)}')
        b = inputs[0].shape[0]
        for a, t, ks in zip(inputs, self.graph.inputs, self.input_shapes):
            if tuple(a.shape[1:]) != tuple(ks) or a.shape[0] != b:
                raise ValueError(f'input shape {a.shape} does not match (batch,)+{tuple(ks)}')
        return inputs, b

    def __call__(self, inputs, training=False):
        inputs, b = self._prep_inputs(inputs)
        out = np.empty((b,) + self.output_shape, np.float32)
        ptrs = (ctypes.c_void_p * len(inputs))(*[a.ctypes.data for a in inputs])
        _lib.check(self.graph._l.dl4ds_graph_forward(self.graph.h, ptrs, len(inputs), b, int(training), 1,
                                                     out.ctypes.data))
        return out

    def predict(self, inputs, batch_size=32, verbose=0):
        """model.predict.  The reference builds its generators on ``Input(shape=(None, None, C))`` (sp_postups.py:112-115),
        so a trained model predicts on ANY grid; the graph here is planned for the grid it was built with, so an input of
        another size runs on a sibling graph planned for that size (cached per grid) that receives the current weights."""
        if isinstance(inputs, np.ndarray):
            inputs = [inputs]
        first = np.asarray(inputs[0])
        grid = tuple(first.shape[-3:-1])
        if grid != tuple(self.input_shapes[0][-3:-1]):
            return self.resized(grid).predict(inputs, batch_size=batch_size, verbose=verbose)
        n = first.shape[0]
        inputs = [np.ascontiguousarray(a, np.float32) for a in inputs]
        # one result array, page-locked while the batches land in it: no per-batch allocation, no concatenation, device-to-host copies
        # at the link's rate (a pageable 67 MB batch of 512^2 fields took 18 ms of the 22 ms per batch)
        out = np.empty((n,) + self.output_shape, np.float32)
        lib = self.graph._l
        pinned = n > 0 and out.nbytes >= (1 << 22) and lib.dl4ds_host_register(out.ctypes.data, out.nbytes) == 0
        try:
            for i in range(0, n, batch_size):
                part, b = self._prep_inputs([a[i:i + batch_size] for a in inputs])
                ptrs = (ctypes.c_void_p * len(part))(*[a.ctypes.data for a in part])
                _lib.check(lib.dl4ds_graph_forward(self.graph.h, ptrs, len(part), b, 0, 1, out[i:i + b].ctypes.data))
        finally:
            if pinned:
                lib.dl4ds_host_unregister(out.ctypes.data)
        return out

    def resized(self, grid):
        """The same architecture planned for inputs of spatial size ``grid`` (size of the FIRST input: the LR grid of a
        post-upsampling model, the HR grid of a 'pin' model), carrying this model's current weights."""
        rb = getattr(self, '_rebuild', None)
        if rb is None:
            raise ValueError(f'model {self.name} was not built by a dl4ds_amd.models builder: cannot re-plan it for grid {grid}')
        fn, kwargs, size_key = rb
        if kwargs.get('localcon_layer'):
            # LocallyConnected2D weights are per grid point: the reference fixes the Input shape too (sp_postups.py:106-115)
            raise ValueError('a model with localcon_layer=True is tied to the grid it was built for')
        cache = self.__dict__.setdefault('_resized_cache', OrderedDict())
        grid = (int(grid[0]), int(grid[1]))
        m = cache.get(grid)
        if m is None:
            m = fn(**dict(kwargs, **{size_key: grid}))
            cache[grid] = m
            while len(cache) > self.RESIZED_CACHE_GRIDS:       # least recently used grid goes (its graph frees its HBM)
                cache.popitem(last=False)
        else:
            cache.move_to_end(grid)
        self._copy_weights_to(m)
        return m

    RESIZED_CACHE_GRIDS = 4

    def _arena(self):
        w, g = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(self.graph._l.dl4ds_graph_arena_ptrs(self.graph.h, ctypes.byref(w), ctypes.byref(g)))
        n, k = ctypes.c_size_t(), ctypes.c_int()
        _lib.check(self.graph._l.dl4ds_graph_param_count(self.graph.h, ctypes.byref(n), ctypes.byref(k)))
        return w, n.value

    def _copy_weights_to(self, other):
        """Current weights -> a sibling graph of the same architecture: the parameter arenas have the same layout (no
        parameter of a resizable model depends on the grid), so ONE device-to-device copy replaces the per-variable
        download + upload (13.6 M parameters for the U-Net)."""
        (src, n), (dst, m) = self._arena(), other._arena()
        same = n == m and [(k, p['shape']) for k, p in self.graph.params.items()] == \
            [(k, p['shape']) for k, p in other.graph.params.items()]
        if same and n:
            _lib.check(self.graph._l.dl4ds_memcpy_d2d(dst, src, n * 4))
            _lib.check(self.graph._l.dl4ds_sync())
        else:
            other.set_weights(self.get_weights())

    def summary(self, line_length=100, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        print_fn('_' * line_length)
        for kind, name, shape in self.graph.layers:
            print_fn(f'{name[:50]:52s}{kind:24s}{str(shape)}')
        print_fn('=' * line_length)
        print_fn(f'Total params: {self.count_params():,}')


def resizable(size_key):
    """Decorator of the model builders: remembers the call so that ``Model.predict`` can re-plan the graph for another
    grid (``size_key``: the builder argument that holds the grid of the first input, 'lr_size' or 'hr_size')."""
    import functools
    import inspect

    def deco(fn):
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapper(*args, **kw):
            model = fn(*args, **kw)
            bound = sig.bind(*args, **kw)
            bound.apply_defaults()
            model._rebuild = (wrapper, dict(bound.arguments), size_key)
            return model
        return wrapper
    return deco
