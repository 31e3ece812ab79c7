"""Host-side mirror of the reference's Keras-layer call surface for the DR-NMF hot path.

Same names, constructor arguments and protocol as the reference (custom_layers.py; enhance.py:
139-317) so an enhance.py-style driver can switch imports:

    from drnmf_amd.layers import (SimpleDeepRNN, DenseNonNegW, DivideAbyAplusB,
                                  divide_A_by_AplusB, build_alt, build_unfolded_snmf)

Weights live in torch tensors on the GPU (containers only); every forward goes through the
hand-written HIP kernels in libdrnmf.so: the fused factored kernels for the configuration
build_unfolded_snmf constructs, the general dense-matrix kernel (SimpleDeepRNN.step as written) for
every other configuration of the layer.  Training likewise: the fused BPTT for the shipped
trainable set (log_D, log_alph, log_lam1, log_h0), the dense-matrix BPTT (csrc/cell_dense_bwd.hip)
for everything else the reference can train -- log_U1 / log_Uk, caller maps, free weights, other
activations -- with torch autograd carrying the matrix gradients through the caller's maps.  There
is no CPU/oracle fallback: what the kernels do not cover raises NotImplementedError (dropout_U
trains on the dense-matrix path).  The KL / beta extension of the cell trains through its own BPTT
(drnmf_cell_backward_ista).
"""
from collections import OrderedDict

import re

import numpy as np
import torch

from . import _capi, ops


# ------------------------------------------------------------------------------------------
# build_alt  (enhance.py:139-206)
# ------------------------------------------------------------------------------------------
class AltMaps(dict):
    """maps_from_alt as returned by build_alt: {'U','S','W','b'} -> per-layer callables, plus the
    per-layer parameter labels so SimpleDeepRNN can dispatch them to the fused kernels."""
    labels_per_k = None
    output_dim = None
    K_layers = None


def _xp(v):
    return torch if isinstance(v, torch.Tensor) else np


def _unit_cols(a_logD):
    xp = _xp(a_logD)
    D = xp.exp(a_logD)
    return D / xp.sqrt((D * D).sum(0, keepdims=True) if xp is np else (D * D).sum(0, keepdim=True))


def build_alt(output_dim, K_layers, params, params_untied=[]):
    """enhance.py:139-206.  params = {'W','U1','Uk','alph','lam1'} (numpy).  Returns
    (alt_params, maps_from_alt).  The maps are plain callables on a dict of numpy arrays or torch
    tensors (row-vector convention, as the reference's Theano lambdas)."""
    f32 = np.float32
    alt_params = OrderedDict()
    alt_params['log_D'] = np.log(f32(1e-7) + np.asarray(params['W'], f32))        # :147
    alt_params['log_U1'] = np.log(f32(1e-7) + np.asarray(params['U1'], f32))
    alt_params['log_Uk'] = np.log(f32(1e-7) + np.asarray(params['Uk'], f32))
    alt_params['log_alph'] = np.log(f32(1e-7) + np.asarray(params['alph'], f32))
    alt_params['log_lam1'] = np.log(f32(1e-7) + np.asarray(params['lam1'], f32))

    labels_per_k = {}
    for name in ['log_D', 'log_alph', 'log_lam1']:                                # :150-159
        if name in params_untied:
            labels_per_k[name] = [name + ('_%d' % k) for k in range(K_layers)]
            v = alt_params.pop(name)
            for k in range(K_layers):
                alt_params[name + ('_%d' % k)] = np.array(v, copy=True)
        else:
            labels_per_k[name] = [name] * K_layers

    maps = AltMaps()
    maps.labels_per_k, maps.output_dim, maps.K_layers = labels_per_k, output_dim, K_layers

    def _t(m):
        return m.T if isinstance(m, np.ndarray) else m.t()

    maps['U'] = [lambda a: _t(_xp(a['log_U1']).exp(a['log_U1']))]                # :163
    maps['U'] += [(lambda a: _t(_xp(a['log_Uk']).exp(a['log_Uk'])))] * (K_layers - 1)   # :165-167

    def make_S(k):
        lD, lA = labels_per_k['log_D'][k], labels_per_k['log_alph'][k]

        def Sk(a):                                                                # :172-181
            Dn = _unit_cols(a[lD])
            xp = _xp(Dn)
            eye = np.eye(output_dim, dtype=np.float32) if xp is np else \
                torch.eye(output_dim, dtype=Dn.dtype, device=Dn.device)
            return _t(eye - _t(Dn / xp.exp(a[lA])) @ Dn)
        return Sk

    def make_W(k):
        lD, lA = labels_per_k['log_D'][k], labels_per_k['log_alph'][k]
        return lambda a: _unit_cols(a[lD]) / _xp(a[lA]).exp(a[lA])               # :187-194

    def make_b(k):
        lA, lL = labels_per_k['log_alph'][k], labels_per_k['log_lam1'][k]

        def bk(a):                                                                # :201-203
            xp = _xp(a[lA])
            ones = np.ones((output_dim,), np.float32) if xp is np else \
                torch.ones(output_dim, dtype=a[lA].dtype, device=a[lA].device)
            return -ones * xp.exp(a[lL]) / xp.exp(a[lA])
        return bk

    maps['S'] = [make_S(k) for k in range(1, K_layers)]
    maps['W'] = [make_W(k) for k in range(K_layers)]
    maps['b'] = [make_b(k) for k in range(K_layers)]
    return alt_params, maps


# ------------------------------------------------------------------------------------------
# small layer objects (Keras protocol subset used by enhance.py)
# ------------------------------------------------------------------------------------------
class _Layer(object):
    _counters = {}

    def __init__(self, name=None, **kwargs):
        if name is None:
            # Keras' auto names: CamelCase -> snake_case + '_<n>' [K2.0.4-memory]
            base = re.sub('(.)([A-Z][a-z0-9]+)', r'\1_\2', type(self).__name__)
            base = re.sub('([a-z])([A-Z])', r'\1_\2', base).lower()
            i = _Layer._counters.get(base, 0) + 1
            _Layer._counters[base] = i
            name = '%s_%d' % (base, i)
        self.name = name
        self.built = False

    @property
    def weights(self):
        return []

    def get_weights(self):
        return [np.array(w.detach().cpu().numpy(), copy=True) for w in self.weights]

    def set_weights(self, weights):
        ws = self.weights
        if len(weights) != len(ws):
            raise ValueError('Layer %s expects %d weight arrays, got %d' %
                             (self.name, len(ws), len(weights)))
        for w, v in zip(ws, weights):
            v = np.asarray(v, dtype=np.float32)
            if v.size == 1 and w.numel() == 1:       # () and (1,) name the same scalar
                v = v.reshape(tuple(w.shape))
            if tuple(v.shape) != tuple(w.shape):
                raise ValueError('Layer %s weight shape %s != provided %s' %
                                 (self.name, tuple(w.shape), tuple(v.shape)))
            w.copy_(torch.from_numpy(np.ascontiguousarray(v)).reshape(tuple(w.shape)))
        self._weights_changed()

    def _weights_changed(self):
        pass


class InputLayer(_Layer):
    def __init__(self, input_shape, **kw):
        super(InputLayer, self).__init__(**kw)
        self.input_shape = (None,) + tuple(input_shape)


class Masking(_Layer):
    """keras.layers.Masking [K2.0.4-memory] (enhance.py:253): frames whose bins ALL equal
    mask_value are masked.  Zeroing and the validity flags are computed inside the cell kernel."""

    def __init__(self, mask_value=0., input_shape=None, **kw):
        super(Masking, self).__init__(**kw)
        self.mask_value = mask_value
        self.input_shape = input_shape


class Lambda(_Layer):
    """Placeholder for the slice / square Lambda layers (enhance.py:277-300); fused in the head."""

    def __init__(self, what, **kw):
        super(Lambda, self).__init__(**kw)
        self.what = what


class DenseNonNegW(_Layer):
    """custom_layers.py:15-29: inputs . exp(kernel), use_bias=False, no activation.
    `weights=[kernel]` initialises the (input_dim, units) log-domain kernel (enhance.py:283)."""

    def __init__(self, units, use_bias=False, weights=None, activation=None, device=None, **kw):
        super(DenseNonNegW, self).__init__(**kw)
        if use_bias or activation not in (None, 'linear'):
            raise NotImplementedError('DenseNonNegW: only use_bias=False, linear activation '
                                      '(the reference configuration, enhance.py:283,292)')
        self.units = units
        self.use_bias = use_bias
        self.device = torch.device(device if device is not None else 'cuda')
        self.kernel = None
        if weights is not None:
            k = np.asarray(weights[0], np.float32)
            if k.shape[1] != units:
                raise ValueError('kernel shape %s does not match units=%d' % (k.shape, units))
            self.kernel = torch.from_numpy(np.ascontiguousarray(k)).to(self.device)
            self.built = True

    def build(self, input_shape):
        if self.kernel is None:   # 'glorot_uniform' default of keras Dense [K2.0.4-memory]
            fan_in, fan_out = input_shape[-1], self.units
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            k = np.random.uniform(-lim, lim, (fan_in, fan_out)).astype(np.float32)
            self.kernel = torch.from_numpy(k).to(self.device)
        self.built = True

    @property
    def weights(self):
        return [] if self.kernel is None else [self.kernel]

    def __call__(self, inputs):
        """inputs [..., r] -> inputs . exp(kernel), via the head kernel with an empty noise half
        is wasteful; standalone use computes A through head_forward with want_ab."""
        if not self.built:
            self.build(tuple(inputs.shape))
        r = self.kernel.shape[0]
        pad = torch.zeros(inputs.shape[:-1] + (2 * r,), dtype=torch.float32,
                          device=inputs.device)
        pad[..., :r] = inputs
        zk = torch.full_like(self.kernel, -80.0)   # exp(-80) ~ 0: the unused second half
        _, A, _ = ops.head_forward(pad, self.kernel, zk, want_ab=True)
        return A


class TimeDistributed(_Layer):
    def __init__(self, layer, **kw):
        super(TimeDistributed, self).__init__(**kw)
        self.layer = layer

    @property
    def weights(self):
        return self.layer.weights

    def _weights_changed(self):
        self.layer._weights_changed()

    def __call__(self, inputs):
        return self.layer(inputs)


class DivideAbyAplusB(_Layer):
    """custom_layers.py:33-45: exp(log(1e-7+A) - log(1e-7+A+B)).  Stand-alone elementwise form
    (inside the model it is fused into the head kernel's epilogue)."""

    def __call__(self, inputs):
        if len(inputs) != 2:
            raise ValueError('DivideAbyAplusB takes exactly 2 inputs')
        A, B = inputs
        return ops.divide_a_by_aplusb(A, B)


def divide_A_by_AplusB(inputs, **kwargs):
    """custom_layers.py:48-56."""
    return DivideAbyAplusB(**kwargs)(inputs)


# ------------------------------------------------------------------------------------------
# SimpleDeepRNN  (custom_layers.py:104-412)
# ------------------------------------------------------------------------------------------
class SimpleDeepRNN(_Layer):
    """The unfolded-ISTA deep recurrent cell.  Constructor mirrors custom_layers.py:131-143;
    **kwargs accepts the Keras Recurrent arguments the reference uses (return_sequences,
    input_shape, stateful, name) plus `device`."""

    def __init__(self, output_dim, init='glorot_uniform', inner_init='orthogonal',
                 activation='tanh', W_regularizer=None, U_regularizer=None, b_regularizer=None,
                 dropout_W=0., dropout_U=0., K_layers=1, alt_params=None, keys_trainable=None,
                 maps_from_alt=None, flag_connect_input_to_layers=False, flag_nonnegative=False,
                 flag_return_all_hidden=False, return_sequences=False, input_shape=None,
                 stateful=False, device=None, operand_dtype='float32', divergence='ed', beta=1.5,
                 **kwargs):
        super(SimpleDeepRNN, self).__init__(**kwargs)
        # extension (not in the reference): 'kl' / 'beta' run the reference's
        # ista_kl / ista_beta iteration (enhance.py:421-456) recurrently instead of the Euclidean
        # cell -- K full ISTA steps per frame warm-started from the previous frame; trains through
        # its own BPTT (drnmf_cell_backward_ista)
        if divergence not in ops.DIVERGENCES:
            raise ValueError("divergence must be 'ed', 'kl' or 'beta'")
        self.divergence, self.beta = divergence, float(beta)
        # extension (BASELINE config 5): 'float16' rounds dictionary and activations to fp16 where
        # they enter the matrix cores (fp32 accumulation and state); training then runs this forward
        # and an fp32 BPTT (mixed precision)
        if operand_dtype not in ('float32', 'float16'):
            raise ValueError("operand_dtype must be 'float32' or 'float16'")
        self.operand_dtype = operand_dtype
        self.units = self.output_dim = int(output_dim)
        self.init, self.inner_init, self.activation = init, inner_init, activation
        self.W_regularizer, self.U_regularizer, self.b_regularizer = (W_regularizer,
                                                                      U_regularizer,
                                                                      b_regularizer)
        self.dropout_W, self.dropout_U = dropout_W, dropout_U
        self.K_layers = int(K_layers)
        self.alt_params = OrderedDict() if alt_params is None else alt_params
        self.keys_trainable = (list(self.alt_params.keys()) if keys_trainable is None
                               else list(keys_trainable))                # custom_layers.py:158-161
        self.maps_from_alt = {} if maps_from_alt is None else maps_from_alt
        self.flag_connect_input_to_layers = flag_connect_input_to_layers
        self.flag_nonnegative = flag_nonnegative
        self.flag_return_all_hidden = flag_return_all_hidden
        self.return_sequences = return_sequences
        self.input_shape = input_shape
        self.stateful = stateful
        self.consume_less = 'gpu'
        self.states = [None]
        self.device = torch.device(device if device is not None else 'cuda')
        self._params_block = None
        self._ws = {}
        # The fused factored kernels cover everything the reference's enhance.py constructs
        # (build_alt maps, relu, both flags on).  Any other configuration of the layer -- caller-
        # supplied maps, free W/U/b/S weights, another activation, flags off -- runs the general
        # dense-matrix kernels (csrc/cell_dense.hip, cell_dense_bwd.hip: SimpleDeepRNN.step as written).
        if activation not in ops.ACTIVATIONS:
            raise ValueError("activation must be one of %s (got %r)" %
                             (sorted(ops.ACTIVATIONS), activation))
        self._generic = (not isinstance(self.maps_from_alt, AltMaps) or activation != 'relu' or
                         not flag_connect_input_to_layers or not flag_nonnegative)
        self._dense_now = self._generic
        if divergence != 'ed' and (self._generic or operand_dtype != 'float32'):
            raise NotImplementedError("divergence='kl'|'beta' exists for the build_alt "
                                      "configuration with fp32 operands")
        # Dropout is the identity outside the training phase (K.in_train_phase,
        # custom_layers.py:377-395).  In training the reference applies ONLY dropout_U: the layer sets
        # consume_less = 'gpu' (custom_layers.py:169), so get_constants never builds the dropout_W
        # mask (custom_layers.py:386: B_W = 1) -- dropout_W is accepted and has no effect, as there.
        # dropout_U (custom_layers.py:361: (prev_output * B_U) U_k in every layer, one mask per
        # sequence and atom for the whole call, 377-384) trains on the dense-matrix path
        # (drnmf_dense_cell_forward_dropout / _backward_dropout): forward_train draws the mask with
        # torch's device generator (the reference's comes from Theano's RNG; `_drop_u_mask` lets a
        # caller supply one).
        # Regularizers belong to the FREE matrices only (custom_layers.py:245-269: add_weight(...,
        # regularizer=...) is reached where maps_from_alt has no map for W / U / b; S has none): with
        # build_alt's maps -- everything enhance.py constructs -- there is nothing for them to act
        # on; on free weights the model adds their penalty and its gradient (UnfoldedSNMFModel.
        # _add_regularizers).
        self._train_blockers = []
        if dropout_U and not (0.0 < float(dropout_U) < 1.0):
            raise ValueError("dropout_U must lie in [0, 1)")
        if dropout_U and (divergence != 'ed' or operand_dtype != 'float32'):
            raise NotImplementedError("dropout_U trains on the dense-matrix path: the Euclidean cell "
                                      "with fp32 operands")
        self._drop_u_mask = None     # test hook: a (B,N) mask to use instead of a fresh draw

    # -- Keras protocol --------------------------------------------------------------------
    def compute_output_shape(self, input_shape):                          # custom_layers.py:175-185
        if isinstance(input_shape, list):
            input_shape = input_shape[0]
        units = self.K_layers * self.units if self.flag_return_all_hidden else self.units
        if self.return_sequences:
            return (input_shape[0], input_shape[1], units)
        return (input_shape[0], units)

    def build(self, input_shape):                                         # custom_layers.py:187-294
        self.input_dim = int(input_shape[2])
        N, F = self.output_dim, self.input_dim
        if self.flag_nonnegative:
            # 'uniform' initializer = U(-0.05, 0.05) [K2.0.4-memory]; h0 = softplus(log_h0)
            log_h0 = np.random.uniform(-0.05, 0.05, (N,)).astype(np.float32)
            self.log_h0 = torch.from_numpy(log_h0).to(self.device)
        else:
            self.h0 = torch.zeros(N, dtype=torch.float32, device=self.device)   # :208-211
        self._alt = OrderedDict()
        for key in self.alt_params:                                       # custom_layers.py:216-228
            v = np.array(self.alt_params[key], dtype=np.float32, copy=True)   # never alias the caller's
            # (scalars -- log_alph, log_lam1 of build_alt -- keep Keras' shape (): K.variable of a 0-d array;
            # np.ascontiguousarray alone would make them (1,))
            self._alt[key] = torch.from_numpy(np.ascontiguousarray(v)).reshape(v.shape).to(self.device)
        self.trainable_keys = [k for k in self._alt if k in self.keys_trainable]
        # matrices without a map are free weights of the layer (custom_layers.py:241-281)
        self._free = OrderedDict()
        for k in range(self.K_layers):
            if 'W' not in self.maps_from_alt:
                self._free['W_%d' % k] = self._initializer(self.init, (F, N))
            if 'U' not in self.maps_from_alt:
                self._free['U_%d' % k] = self._initializer(self.inner_init, (N, N))
            if 'b' not in self.maps_from_alt:
                self._free['b_%d' % k] = self._initializer('zero', (N,))
            if k > 0 and 'S' not in self.maps_from_alt:
                self._free['S_%dto%d' % (k - 1, k)] = self._initializer(self.inner_init, (N, N))
        if not self._generic:
            lab = self.maps_from_alt.labels_per_k
            d0 = self._alt[lab['log_D'][0]]
            if tuple(d0.shape) != (F, N):
                raise ValueError('log_D has shape %s, expected (input_dim=%d, output_dim=%d)' %
                                 (tuple(d0.shape), F, N))
        self.built = True
        self._weights_changed()

    def _initializer(self, name, shape):
        """Keras initializers by name [K2.0.4-memory]: glorot_uniform U(+-sqrt(6/(fan_in+fan_out))),
        orthogonal (QR of a Gaussian, gain 1), uniform U(+-0.05), zero, one, identity."""
        name = getattr(name, '__name__', name)
        if name in ('zero', 'zeros'):
            v = np.zeros(shape, np.float32)
        elif name in ('one', 'ones'):
            v = np.ones(shape, np.float32)
        elif name == 'identity':
            v = np.eye(shape[0], shape[1], dtype=np.float32)
        elif name == 'uniform':
            v = np.random.uniform(-0.05, 0.05, shape)
        elif name == 'glorot_uniform':
            lim = np.sqrt(6.0 / (shape[0] + shape[-1]))
            v = np.random.uniform(-lim, lim, shape)
        elif name == 'orthogonal':
            q, r = np.linalg.qr(np.random.normal(0.0, 1.0, (max(shape), max(shape))))
            v = (q * np.sign(np.diag(r)))[:shape[0], :shape[1]]
        else:
            raise ValueError('unknown initializer %r' % (name,))
        return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(self.device)

    @property
    def weights(self):
        """Order: log_h0 (or h0), the alt params in build_alt's insertion order (the reference's
        order is Python-2 dict order, custom_layers.py:203-228), then the free W/U/b/S weights
        layer by layer (custom_layers.py:234-287)."""
        first = self.log_h0 if self.flag_nonnegative else self.h0
        return [first] + list(self._alt.values()) + list(self._free.values())

    @property
    def weight_names(self):
        first = '%s_log_h0' if self.flag_nonnegative else '%s_h0'
        return [first % self.name] + ['%s_%s' % (self.name, k) for k in self._alt] + \
            ['%s_%s' % (self.name, k) for k in self._free]

    def _weights_changed(self):
        self._params_block_valid = False
        self._dense_block_valid = False
        if self._generic:
            return
        a = {k: v.detach().cpu().numpy() for k, v in self._alt.items()
             if k in ('log_U1', 'log_Uk')}
        U1, Uk = np.exp(a['log_U1']), np.exp(a['log_Uk'])                 # enhance.py:163-167
        N = U1.shape[0]
        d = np.diag(U1)
        off = U1[~np.eye(N, dtype=bool)] if N > 1 else np.zeros((1,), np.float32)
        # the factored kernels rely on the rank-structured U of build_alt's initialisation
        # (diagonal + constant); once log_U1/log_Uk are trained away from it the layer runs on the
        # dense-matrix kernel.  No shipped config trains them (params_unfolded_snmf_*.yaml:10).
        self._dense_now = not (np.all(d == d[0]) and np.all(off == off[0]) and
                               np.all(Uk == Uk.flat[0]))
        self._u = (float(d[0]), float(off[0]) if N > 1 else 0.0, float(Uk.flat[0]))
        if self.divergence != 'ed':
            # The KL / beta variant is ista_kl / ista_beta run recurrently: it has NO U term at all
            # (DESIGN.md section 7), so whatever log_U1 / log_Uk hold is ignored -- the dense-matrix
            # kernels implement the Euclidean step only and must never be chosen for this cell
            # (a set_weights() that broke U's structure silently switched models).
            self._dense_now = False

    # -- general dense-matrix path -------------------------------------------------------------
    def dense_matrices(self):
        """Uk, Sk, Wk, bk of SimpleDeepRNN.build (custom_layers.py:234-287) as float32 numpy
        stacks: the maps are called on a dict of the alt params (host numpy arrays) exactly as the
        reference calls its Theano lambdas; a missing map means the free weight."""
        K = self.K_layers
        a = OrderedDict((k, v.detach().cpu().numpy()) for k, v in self._alt.items())

        def get(kind, k, free_name):
            if kind in self.maps_from_alt:
                m = self.maps_from_alt[kind]
                m = m[k] if isinstance(m, (list, tuple)) else m
                v = m(a)
                if isinstance(v, torch.Tensor):
                    v = v.detach().cpu().numpy()
                return np.asarray(v, np.float32)
            return self._free[free_name].detach().cpu().numpy()
        N, F = self.output_dim, self.input_dim
        U = np.stack([get('U', k, 'U_%d' % k).reshape(N, N) for k in range(K)])
        b = np.stack([np.broadcast_to(get('b', k, 'b_%d' % k), (N,)) for k in range(K)])
        S = np.stack([get('S', k - 1, 'S_%dto%d' % (k - 1, k)).reshape(N, N)
                      for k in range(1, K)]) if K > 1 else None
        W = np.stack([get('W', k, 'W_%d' % k).reshape(F, N) for k in range(K)]) \
            if self.flag_connect_input_to_layers else None
        return U, S, W, b

    def initial_state_vector(self):
        """h0_last of custom_layers.py:203-211 (softplus(log_h0), or the h0 weight)."""
        if self.flag_nonnegative:
            z = self.log_h0.detach().cpu().numpy().astype(np.float64)
            return np.where(z > 20, z, np.log1p(np.exp(np.minimum(z, 20)))).astype(np.float32)
        return self.h0.detach().cpu().numpy()

    def _call_dense(self, x, mask_value, out):
        B, T, F = x.shape
        desc = ops.make_dense_desc(B, T, F, self.output_dim, self.K_layers,
                                   self.flag_connect_input_to_layers, self.activation,
                                   self.flag_return_all_hidden,
                                   operand_f16=self.operand_dtype == 'float16')
        if not self._dense_block_valid:
            U, S, W, b = self.dense_matrices()
            dev = x.device
            tt = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v)).to(dev)
            self._dense_block = ops.dense_prepare_params(desc, tt(U), tt(S), tt(W), tt(b),
                                                         out=getattr(self, '_dense_block', None))
            self._dense_h0 = torch.from_numpy(self.initial_state_vector()).to(dev)
            self._dense_block_valid = True
        key = ('dense', B, T)
        if key not in self._ws:
            self._ws.clear()
            self._ws[key] = ops.dense_workspace(desc, x.device)
        init = fin = None
        if self.stateful:
            if getattr(self, 'states', None) is None or self.states[0] is None or \
                    tuple(self.states[0].shape) != (B, self.output_dim):
                self.states = [torch.zeros((B, self.output_dim), dtype=torch.float32,
                                           device=x.device)]
            init = fin = self.states[0]
        h = ops.dense_cell_forward(x, mask_value, self._dense_block, desc, self._dense_h0, out=out,
                                   workspace=self._ws[key], initial_state=init, final_state=fin)
        return h if self.return_sequences else h[:, -1]

    def get_config(self):                                                 # custom_layers.py:397-412
        return {'name': self.name, 'output_dim': self.output_dim, 'init': self.init,
                'inner_init': self.inner_init, 'activation': self.activation,
                'W_regularizer': None, 'U_regularizer': None, 'b_regularizer': None,
                'dropout_W': self.dropout_W, 'dropout_U': self.dropout_U,
                'K_layers': self.K_layers, 'return_sequences': self.return_sequences,
                'stateful': self.stateful,
                'flag_connect_input_to_layers': self.flag_connect_input_to_layers}

    def reset_states(self):                                               # custom_layers.py:296-318
        """Stateful mode: zero the carried state (the reference sets np.zeros, not h0)."""
        assert self.stateful, 'Layer must be stateful.'
        if getattr(self, 'states', None) is None or self.states[0] is None:
            raise ValueError('If a RNN is stateful, it needs to know its batch size: call the layer '
                             'once (or pass batch_input_shape) before reset_states().')
        self.states[0].zero_()

    # -- forward ---------------------------------------------------------------------------
    def _stacked(self, name):
        lab = self.maps_from_alt.labels_per_k[name]
        if lab[0] == lab[-1] and len(set(lab)) == 1:
            return self._alt[lab[0]].reshape((1,) + tuple(self._alt[lab[0]].shape)), 1
        return torch.stack([self._alt[k] for k in lab], 0), len(lab)

    def _desc(self, B, T):
        _, nD = self._stacked_meta('log_D')
        _, nA = self._stacked_meta('log_alph')
        _, nL = self._stacked_meta('log_lam1')
        lab = self.maps_from_alt.labels_per_k
        alph_len = int(self._alt[lab['log_alph'][0]].numel())
        return ops.make_desc(B, T, self.input_dim, self.output_dim, self.K_layers, nD, nA,
                             alph_len, nL, self.flag_return_all_hidden,
                             operand_f16=self.operand_dtype == 'float16',
                             divergence=self.divergence)

    def _stacked_meta(self, name):
        lab = self.maps_from_alt.labels_per_k[name]
        return lab, (1 if len(set(lab)) == 1 else len(lab))

    def prepare(self, B, T):
        """(Re)build the prepared parameter block if the weights changed."""
        desc = self._desc(B, T)
        if not self._params_block_valid:
            logD, _ = self._stacked('log_D')
            logA, _ = self._stacked('log_alph')
            logL, _ = self._stacked('log_lam1')
            self._params_block = ops.prepare_params(desc, logD, logA.reshape(logA.shape[0], -1),
                                                    logL.reshape(-1), out=self._params_block)
            self._params_block_valid = True
        return desc

    def call(self, x, mask_value=None, out=None):
        """x [B,T,F] float32 CUDA tensor -> h [B,T,N] (Masking semantics if mask_value given)."""
        if not self.built:
            self.build(tuple(x.shape))
        B, T, F = x.shape
        if F != self.input_dim:
            raise ValueError('input has %d features, layer was built for %d' % (F, self.input_dim))
        if self._dense_now:
            return self._call_dense(x, mask_value, out)
        desc = self.prepare(B, T)
        key = (B, T)
        if key not in self._ws:
            self._ws.clear()
            self._ws[key] = ops.cell_workspace(desc, x.device)
        init = fin = None
        if self.stateful:
            # Keras stateful RNN: the state left by the previous batch enters this one; the first
            # batch starts from zeros (Recurrent.reset_states / custom_layers.py:315-318)
            if getattr(self, 'states', None) is None or self.states[0] is None or \
                    tuple(self.states[0].shape) != (B, self.output_dim):
                self.states = [torch.zeros((B, self.output_dim), dtype=torch.float32,
                                           device=x.device)]
            init = fin = self.states[0]
        if self.divergence != 'ed':
            h = ops.cell_forward_ista(x, mask_value, self._params_block, desc, self.log_h0,
                                      beta=self.beta, out=out, workspace=self._ws[key],
                                      initial_state=init, final_state=fin)
        else:
            h = ops.cell_forward(x, mask_value, self._params_block, desc, self.log_h0, self._u,
                                 out=out, workspace=self._ws[key], initial_state=init,
                                 final_state=fin)
        return h if self.return_sequences else h[:, -1]

    __call__ = call

    # -- training (BPTT) -------------------------------------------------------------------
    def forward_train(self, x, mask_value=None):
        """Forward that keeps every layer's hidden state: returns hall [B,T,K*N] (the last N
        columns are the layer output) and leaves the workspace ready for `backward`."""
        if self._train_blockers:
            raise NotImplementedError('training with %s is not implemented' %
                                      ', '.join(self._train_blockers))
        if not self.built:
            self.build(tuple(x.shape))
        if self._dense_now or getattr(self, '_train_dense', False) or self.dropout_U:
            return self._forward_train_dense(x, mask_value)
        B, T, F = x.shape
        self.prepare(B, T)
        lab = self.maps_from_alt.labels_per_k
        _, nD = self._stacked_meta('log_D')
        _, nA = self._stacked_meta('log_alph')
        _, nL = self._stacked_meta('log_lam1')
        # operand_dtype='float16': the forward runs on fp16 matrix-core operands, the BPTT in fp32
        # from the stored hiddens (mixed precision; the rounding is treated as the identity)
        desc = ops.make_desc(B, T, self.input_dim, self.output_dim, self.K_layers, nD, nA,
                             int(self._alt[lab['log_alph'][0]].numel()), nL, True,
                             operand_f16=self.operand_dtype == 'float16',
                             divergence=self.divergence)
        key = ('train', B, T)
        if key not in self._ws:
            self._ws.clear()
            self._ws[key] = ops.cell_workspace(desc, x.device)
        self._train_init = None
        if self.stateful:
            if getattr(self, 'states', None) is None or self.states[0] is None or \
                    tuple(self.states[0].shape) != (B, self.output_dim):
                self.states = [torch.zeros((B, self.output_dim), dtype=torch.float32, device=x.device)]
            self._train_init = self.states[0].clone()
        if self.divergence != 'ed':
            # the KL / beta variant (an extension): ista_kl / ista_beta (enhance.py:421-456) run
            # recurrently; its BPTT is drnmf_cell_backward_ista (stateful: ..._ista_stateful, as below)
            hall = ops.cell_forward_ista(x, mask_value, self._params_block, desc, self.log_h0,
                                         beta=self.beta, workspace=self._ws[key],
                                         initial_state=self._train_init,
                                         final_state=self.states[0] if self.stateful else None)
        elif self.stateful:
            # Keras stateful RNN under fit / train_on_batch (custom_layers.py:296-318): the state the previous
            # batch left enters this one as a CONSTANT of the gradient (zeros before the first batch /
            # after reset_states); the state this batch leaves is kept for the next.  The entering state is
            # copied: the BPTT needs it after the forward has overwritten `states`.
            hall = ops.cell_forward(x, mask_value, self._params_block, desc, self.log_h0, self._u,
                                    workspace=self._ws[key], initial_state=self._train_init,
                                    final_state=self.states[0])
        else:
            hall = ops.cell_forward(x, mask_value, self._params_block, desc, self.log_h0, self._u,
                                    workspace=self._ws[key])
        self._train_ctx = (desc, key, mask_value)
        return hall

    def backward(self, x, hall, d_out, grads=None, profile=None):
        """Gradients w.r.t. the stacked log-domain parameters (see ops.cell_backward); on the
        dense-matrix path a dict {'by_name': {weight name: gradient}} (see _backward_dense)."""
        if self._train_ctx[0] == 'dense':
            return self._backward_dense(x, hall, d_out)
        desc, key, _ = self._train_ctx
        return ops.cell_backward(x, self._params_block, desc, self.log_h0, self._u, hall, d_out,
                                 self._ws[key], grads=grads, profile=profile, beta=self.beta,
                                 initial_state=getattr(self, '_train_init', None))

    # -- training on the dense-matrix path ---------------------------------------------------
    # Whatever maps_from_alt produce (build_alt's maps with a trainable log_U1 / log_Uk, a caller's
    # own maps, free W/U/b/S weights, any activation): the device computes the gradients w.r.t.
    # the matrices of the step (csrc/cell_dense_bwd.hip) and torch autograd carries them through
    # the maps to the weights -- the maps are the caller's callables, as the reference's are Theano
    # lambdas differentiated by Theano (custom_layers.py:216-287).
    def trainable_weight_items(self):
        """[(name, tensor)] of the layer's trainable weights: log_h0 / h0, the alt parameters in
        keys_trainable, every free matrix (custom_layers.py:203-287)."""
        items = [('log_h0', self.log_h0)] if self.flag_nonnegative else [('h0', self.h0)]
        items += [(k, v) for k, v in self._alt.items() if k in self.keys_trainable]
        items += list(self._free.items())
        return items

    def _dense_matrices_torch(self):
        """Uk/Sk/Wk/bk stacks and the initial state as torch tensors ON THE AUTOGRAD TAPE of fresh
        leaf copies of the trainable weights; returns (leaves, U, S, W, b, h0)."""
        K, N, F = self.K_layers, self.output_dim, self.input_dim
        leaves = OrderedDict((n, t.detach().clone().requires_grad_(True))
                             for n, t in self.trainable_weight_items())
        a = OrderedDict((k, leaves.get(k, v)) for k, v in self._alt.items())

        def get(kind, k, free_name):
            if kind in self.maps_from_alt:
                m = self.maps_from_alt[kind]
                m = m[k] if isinstance(m, (list, tuple)) else m
                v = m(a)
                if not isinstance(v, torch.Tensor):
                    v = torch.as_tensor(np.asarray(v, np.float32), device=self.device)
                return v.to(torch.float32)
            return leaves[free_name]
        U = torch.stack([get('U', k, 'U_%d' % k).reshape(N, N) for k in range(K)])
        b = torch.stack([torch.broadcast_to(get('b', k, 'b_%d' % k).reshape(-1), (N,))
                         for k in range(K)])
        S = torch.stack([get('S', k - 1, 'S_%dto%d' % (k - 1, k)).reshape(N, N)
                         for k in range(1, K)]) if K > 1 else None
        W = torch.stack([get('W', k, 'W_%d' % k).reshape(F, N) for k in range(K)]) \
            if self.flag_connect_input_to_layers else None
        h0 = torch.nn.functional.softplus(leaves['log_h0'], threshold=20.0) \
            if self.flag_nonnegative else leaves['h0']
        return leaves, U, S, W, b, h0

    def _forward_train_dense(self, x, mask_value):
        B, T, F = x.shape
        leaves, U, S, W, b, h0 = self._dense_matrices_torch()
        # (operand_dtype='float16': the forward on fp16 matrix-core operands, the BPTT in fp32 from the stored
        # hiddens -- mixed precision, as on the fused path)
        desc = ops.make_dense_desc(B, T, F, self.output_dim, self.K_layers,
                                   self.flag_connect_input_to_layers, self.activation, True,
                                   operand_f16=self.operand_dtype == 'float16')
        c = lambda v: None if v is None else v.detach().contiguous()
        block = ops.dense_prepare_params(desc, c(U), c(S), c(W), c(b),
                                         out=getattr(self, '_dense_train_block', None))
        self._dense_train_block = block
        key = ('dense', B, T)
        if key not in self._ws:
            self._ws.clear()
            self._ws[key] = ops.dense_workspace(desc, x.device)
        drop = None
        if self.dropout_U:
            # get_constants (custom_layers.py:377-384): K.dropout(ones(B, N), p) = Bernoulli(1-p) / (1-p)
            drop = self._drop_u_mask
            if drop is None:
                keep = 1.0 - float(self.dropout_U)
                drop = torch.bernoulli(torch.full((B, self.output_dim), keep, dtype=torch.float32,
                                                  device=x.device)) / keep
            else:
                drop = torch.as_tensor(drop, dtype=torch.float32, device=x.device).contiguous()
        self._train_init = None
        if self.stateful:
            # the carried state enters as a constant of the gradient (custom_layers.py:296-318), as on
            # the fused path (forward_train)
            if getattr(self, 'states', None) is None or self.states[0] is None or \
                    tuple(self.states[0].shape) != (B, self.output_dim):
                self.states = [torch.zeros((B, self.output_dim), dtype=torch.float32, device=x.device)]
            self._train_init = self.states[0].clone()
        hall = ops.dense_cell_forward(x, mask_value, block, desc, c(h0), workspace=self._ws[key],
                                      drop_u=drop, initial_state=self._train_init,
                                      final_state=self.states[0] if self.stateful else None)
        self._train_ctx = ('dense', leaves, (U, S, W, b, h0), mask_value, drop)
        return hall

    def _backward_dense(self, x, hall, d_out):
        _, leaves, (U, S, W, b, h0), mask_value, drop = self._train_ctx
        B, T, F = x.shape
        N, K = self.output_dim, self.K_layers
        all_hidden = d_out.shape[-1] == K * N and K > 1
        desc = ops.make_dense_desc(B, T, F, N, K, self.flag_connect_input_to_layers,
                                   self.activation, all_hidden)
        c = lambda v: None if v is None else v.detach().contiguous()
        g = ops.dense_cell_backward(x, mask_value, desc, c(U), c(S), c(W), c(b), c(h0), hall, d_out,
                                    workspace=getattr(self, '_dense_bwd_ws', None), drop_u=drop,
                                    initial_state=getattr(self, '_train_init', None))
        self._dense_bwd_ws = g['workspace']
        outs, grads = [], []
        for t, gt in ((U, g['dU']), (S, g['dS']), (W, g['dW']), (b, g['db']), (h0, g['dh0'])):
            if t is not None and t.requires_grad:
                outs.append(t)
                grads.append(gt)
        if outs:
            torch.autograd.backward(outs, grads)
        by_name = OrderedDict((n, (l.grad if l.grad is not None else torch.zeros_like(l)))
                              for n, l in leaves.items())
        self._train_ctx = ('dense', None, None, None, None)
        return {'by_name': by_name}

    def grad_slices(self):
        """[(weight name, stacked-gradient key, index)] mapping each alt parameter to its slice
        of the stacked gradients returned by `backward`."""
        out = []
        lab = self.maps_from_alt.labels_per_k
        for name, gkey in (('log_D', 'd_log_D'), ('log_alph', 'd_log_alph'),
                           ('log_lam1', 'd_log_lam1')):
            labels = lab[name]
            if len(set(labels)) == 1:
                out.append((labels[0], gkey, 0))
            else:
                out += [(lk, gkey, k) for k, lk in enumerate(labels)]
        return out


# ------------------------------------------------------------------------------------------
# build_unfolded_snmf  (enhance.py:209-317)
def _require_h5py(path):
    """h5py when it is importable, else the in-tree ctypes binding of the system's libhdf5
    (h5lite.py: the same File / Group / attrs idioms)."""
    try:
        import h5py
        return h5py
    except ImportError:
        pass
    from . import h5lite
    try:
        h5lite.lib()
    except ImportError as e:
        raise ImportError("reading/writing the Keras HDF5 weight file %r needs h5py or a libhdf5 "
                          ">= 1.10 for drnmf_amd.h5lite (%s); use a '.npz' path (same tree of "
                          "names)" % (path, e))
    return h5lite


# ------------------------------------------------------------------------------------------
# Report-ring slots of a device are shared by every model of the process (the ring belongs to the library
# handle): a slot is handed out again only after the DeviceLoss that last used it has been read.
_ring_next = {}
_ring_owner = {}


def _claim_report_slot(dev_index, n_slots):
    import weakref
    i = _ring_next.get(dev_index, 0)
    _ring_next[dev_index] = i + 1
    slot = i % n_slots
    prev = _ring_owner.get((dev_index, slot))
    if prev is not None:
        prev = prev()
        if prev is not None and prev._value is None:
            try:
                prev._get()              # (waits for a step issued n_slots steps ago: long finished)
            except Exception:            # noqa: BLE001 -- its fault is raised to whoever reads THAT loss
                pass
    return slot, (lambda loss: _ring_owner.__setitem__((dev_index, slot), weakref.ref(loss)))


class DeviceLoss(object):
    """The loss of one optimiser step, still on its way from the device.

    train_on_batch() only ENQUEUES work (forward, loss, BPTT, all-reduce, one fused Adam launch): the
    step's normalisation, the global-norm clip and the update itself read their factors from device
    memory, so the host never waits for the gradients.  The Adam kernel leaves [loss, fault, scale, count]
    in a slot of the handle's host-mapped report ring; float(loss) waits for the event recorded behind
    that launch and reads the slot.  Behaves like a number wherever one is needed (float(), arithmetic,
    comparisons, formatting); a step whose persistent chain timed out raises here -- on every rank of a
    data-parallel group at the same step, because the fault word is part of the all-reduced buffer."""
    __slots__ = ('_event', '_slot', '_value', '_device', '_error', '_on_fault', '__weakref__')

    def __init__(self, event, slot, device, on_fault=None):
        self._event, self._slot, self._value, self._device, self._error = event, slot, None, device, None
        self._on_fault = on_fault        # called ONCE when the report says the step was skipped

    def ready(self):
        return self._value is not None or self._event.query()

    def _get(self):
        if self._error is not None:
            raise self._error
        if self._value is None:
            if not self._event.query():
                self._event.synchronize()
            loss, fault = float(self._slot[0]), float(self._slot[1])
            self._slot = None                # (the ring slot may be reused from here on)
            if fault != 0.0:
                self._value = float('nan')
                if self._on_fault is not None:
                    self._on_fault()
                    self._on_fault = None
                self._error = _capi.DrnmfError(
                    'train_on_batch: a persistent small-shape chain of this step timed out on at least '
                    'one rank (DRNMF_ERR_TIMEOUT): the step was SKIPPED on every rank (weights and Adam '
                    'state unchanged) -- rerun it, or set DRNMF_PERSIST=0')
                raise self._error
            self._value = loss
        return self._value

    def __float__(self):
        return self._get()

    item = __float__

    def __repr__(self):
        return repr(self._get())

    def __format__(self, spec):
        return format(self._get(), spec)

    def __add__(self, o):
        return self._get() + float(o)

    __radd__ = __add__

    def __sub__(self, o):
        return self._get() - float(o)

    def __rsub__(self, o):
        return float(o) - self._get()

    def __mul__(self, o):
        return self._get() * float(o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._get() / float(o)

    def __lt__(self, o):
        return self._get() < float(o)

    def __le__(self, o):
        return self._get() <= float(o)

    def __gt__(self, o):
        return self._get() > float(o)

    def __ge__(self, o):
        return self._get() >= float(o)

    def __eq__(self, o):
        return self._get() == float(o)

    def __ne__(self, o):
        return self._get() != float(o)

    def __hash__(self):
        return hash(self._get())

    def __neg__(self):
        return -self._get()

    def __abs__(self):
        return abs(self._get())

    def __round__(self, n=None):
        return round(self._get(), n)

    def __rtruediv__(self, o):
        return float(o) / self._get()

    def __pow__(self, o):
        return self._get() ** float(o)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._get(), dtype=dtype or np.float64)


class UnfoldedSNMFModel(object):
    """Masking -> SimpleDeepRNN -> [:r]/[r:] -> DenseNonNegW x2 -> (square) -> A/(A+B).
    Exposes the slice of keras.Model that enhance.py uses: layers, get/set_weights,
    predict_on_batch; `forward` keeps everything on the device."""

    def __init__(self, layers, cell, clean, noise, mask_value, square):
        self.layers = layers
        self.cell, self.clean, self.noise = cell, clean, noise
        self.mask_value, self.square = mask_value, square

    @property
    def weights(self):
        return self.cell.weights + self.clean.weights + self.noise.weights

    def get_weights(self):
        return [np.array(w.detach().cpu().numpy(), copy=True) for w in self.weights]

    def set_weights(self, weights):
        n = len(self.cell.weights)
        self.cell.set_weights(weights[:n])
        self.clean.set_weights(weights[n:n + 1])
        self.noise.set_weights(weights[n + 1:n + 2])

    # -- weight files (enhance.py:1096, 1119-1129, 1135, 1160-1166: ModelCheckpoint(
    #    save_weights_only=True) / save_weights / load_weights on Keras HDF5 files) ------------
    def _weighted_layers(self):
        return [l for l in self.layers if l.weights]

    def weights_tree(self):
        """Keras' save_weights layout as a flat dict: 'layer_names', '<layer>/weight_names' and one
        array per '<layer>/<weight name>' (custom_layers.py:203-228 names the cell's weights
        '<layer>_log_h0', '<layer>_<alt key>'; Dense kernels are 'kernel')."""
        tree = {}
        names = []
        for l in self._weighted_layers():
            names.append(l.name)
            if hasattr(l, 'weight_names'):
                wn = l.weight_names
            elif len(l.weights) == 1:
                wn = ['kernel']
            else:
                wn = ['weight_%d' % i for i in range(len(l.weights))]
            tree[l.name + '/weight_names'] = np.array(wn)
            for n, w in zip(wn, l.get_weights()):
                tree[l.name + '/' + n] = w
        tree['layer_names'] = np.array(names)
        return tree

    def load_weights_tree(self, tree):
        """Inverse of weights_tree, with Keras' topological matching (layers that own weights, in
        order); inside the recurrent cell weights are matched BY NAME, because the reference's
        order there is Python-2 dict order (custom_layers.py:216-228)."""
        file_layers = [str(n) for n in tree['layer_names']
                       if len(tree[str(n) + '/weight_names']) > 0]
        mine = self._weighted_layers()
        if len(file_layers) != len(mine):
            raise ValueError('weight file has %d layers with weights, the model has %d'
                             % (len(file_layers), len(mine)))
        for fname, layer in zip(file_layers, mine):
            wn = [str(n) for n in tree[fname + '/weight_names']]
            vals = [np.asarray(tree[fname + '/' + n]) for n in wn]
            if hasattr(layer, 'weight_names'):
                strip = lambda n, pre: n[len(pre) + 1:] if n.startswith(pre + '_') else n
                by_key = {strip(n.split(':')[0], fname): v for n, v in zip(wn, vals)}
                want = [strip(n, layer.name) for n in layer.weight_names]
                missing = [k for k in want if k not in by_key]
                if missing:
                    raise ValueError('weight file lacks %s for layer %s' % (missing, layer.name))
                vals = [by_key[k] for k in want]
            layer.set_weights(vals)

    def save_weights(self, path):
        """'.npz': the weights_tree dict (numpy).  '.h5'/'.hdf5': the same tree in Keras 2.0.4's
        HDF5 layout (root attrs layer_names / backend / keras_version, one group per layer with a
        weight_names attr and one dataset per weight), written with h5py or, without it, through
        the system's libhdf5 (h5lite.py)."""
        tree = self.weights_tree()
        if path.endswith('.npz'):
            np.savez(path, **tree)
            return
        h5py = _require_h5py(path)
        with h5py.File(path, 'w') as f:
            f.attrs['layer_names'] = [n.encode('utf8') for n in tree['layer_names']]
            f.attrs['backend'] = b'theano'
            f.attrs['keras_version'] = b'2.0.4'
            for ln in tree['layer_names']:
                g = f.create_group(str(ln))
                wn = [str(n) for n in tree[str(ln) + '/weight_names']]
                g.attrs['weight_names'] = [n.encode('utf8') for n in wn]
                for n in wn:
                    g.create_dataset(n, data=tree[str(ln) + '/' + n])

    def load_weights(self, path):
        if path.endswith('.npz'):
            with np.load(path, allow_pickle=False) as z:
                return self.load_weights_tree({k: z[k] for k in z.files})
        h5py = _require_h5py(path)
        dec = lambda b: b.decode('utf8') if isinstance(b, bytes) else str(b)
        tree = {}
        with h5py.File(path, 'r') as f:
            g0 = f['model_weights'] if 'layer_names' not in f.attrs and 'model_weights' in f else f
            names = [dec(n) for n in g0.attrs['layer_names']]
            tree['layer_names'] = np.array(names)
            for ln in names:
                wn = [dec(n) for n in g0[ln].attrs['weight_names']]
                tree[ln + '/weight_names'] = np.array(wn)
                for n in wn:
                    tree[ln + '/' + n] = np.asarray(g0[ln][n])
        return self.load_weights_tree(tree)

    def forward(self, x, want_hidden=False):
        """Device tensors in, device tensors out, nothing waited for: a fault on the device (a persistent
        small-shape chain that timed out) is NOT raised here -- call ops.check_status(device) once you have
        synchronised, as predict_on_batch does after its copy; a later training / test step drops such a stale
        word instead of charging it to itself (_drop_stale_fault)."""
        h = self.cell.call(x, mask_value=self.mask_value)
        mask = ops.head_forward(h, self.clean.kernel, self.noise.kernel, square=self.square,
                                h_off=h.shape[-1] - self.cell.output_dim)
        return (mask, h) if want_hidden else mask

    def predict_on_batch(self, x, lengths=None):
        # (one slab of `predict`: pinned staging both ways -- a pageable device-to-host copy runs at a sixth
        # of the link rate; a chain that timed out raises after the copy has synchronised)
        return self.predict(x, batch_size=max(1, len(x)), lengths=lengths)

    @staticmethod
    def valid_lengths(x, mask_value):
        """1 + index of the last frame of every sequence that is NOT masked (Masking(mask_value), enhance.py:253:
        a frame is masked when EVERY bin equals mask_value); 0 for a sequence without a valid frame.  x [n,T,F]
        float32 numpy.  Exact, and cheap for the reference's layout (padding behind the valid prefix,
        audio_dataset.py:144-161): the first bin decides almost every frame, whole rows are compared only over the
        trailing run that bin leaves undecided."""
        n, T, _ = x.shape
        mv = np.float32(mask_value)
        first_ok = x[:, :, 0] != mv                                      # frames that are surely valid
        lens = (first_ok * np.arange(1, T + 1, dtype=np.int64)).max(axis=1) if T else np.zeros((n,), np.int64)
        for i in range(n):
            lo = int(lens[i])
            if lo < T:
                tail = x[i, lo:]
                # (plain numpy: single-threaded, no thread-pool hand-off per utterance -- 0.7 ms each through
                # torch on a 256-thread host)
                if tail.min() != mv or tail.max() != mv:                 # undecided frames: every bin
                    ok = (tail != mv).any(axis=-1)
                    lens[i] = lo + 1 + int(np.nonzero(ok)[0][-1])
        return lens

    PREDICT_T_STEP = 32     # slab lengths are rounded up to a multiple of this many frames (graph reuse)
    HOST_COPY_THREADS = 8   # worker threads of predict's host-side staging copies

    def _host_pool(self):
        pool = getattr(self, '_host_pool_obj', None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._host_pool_obj = ThreadPoolExecutor(max_workers=self.HOST_COPY_THREADS,
                                                            thread_name_prefix='drnmf-stage')
        return pool

    def predict(self, x, batch_size=250, verbose=0, lengths=None, length_aware=True):
        """keras Model.predict(x, batch_size): enhance.py's inference loop (1185-1193 validation, 1215-1223
        test: `predict_on_batch` over slabs of 250 utterances) as ONE call.  x [n,T,F] numpy -> masks
        [n,T,F] numpy.  Slab s+1 goes up and slab s-1's masks come down on two copy streams, through pinned
        staging buffers the model keeps, while slab s computes: the PCIe time of a slab hides behind the chain
        of the next.

        Length-aware (default).  The reference pads every utterance to the longest of the data set and crops
        the masks afterwards (enhance.py:1181, 1200-1203): a run at T_max for every row spends 1 - mean(len) /
        T_max of its frames on padding.  Here the utterances are sorted by valid length, slabs are formed of
        neighbours, and a slab runs (and is copied, both ways) at the longest valid length IN THE SLAB, rounded up
        to PREDICT_T_STEP frames; results return to the caller's order.  Nothing the caller can see changes:
        a frame computed is the same frame of the same sequence (rows never interact, custom_layers.py:337-338,
        and a sequence's frames depend on its earlier frames only), and a masked frame repeats the output of
        the frame before it (K.rnn), so the frames behind a slab's length are filled with the slab's last
        computed frame -- the full [n,T,F] array equals the one-length run (bit for bit wherever the head takes
        the same kernel for both row counts, i.e. from 2048 frames per slab on; tests/test_gpu_parity.py).
        `lengths` [n] (optional): 1 + the index of each sequence's last valid frame, as the caller's data set
        knows it (audio_dataset.py:159-161) -- frames at or behind it are TAKEN to be padding; without it the
        lengths are read off x (valid_lengths).  length_aware=False: every slab at T, input order."""
        x = np.asarray(x, dtype=np.float32)
        if x.ndim != 3:
            raise ValueError('predict: x must be (n, T, F), got shape %s' % (x.shape,))
        n, T, F = x.shape
        bs = int(batch_size)
        if bs <= 0:
            raise ValueError('predict: batch_size must be positive')
        dev = self.cell.device
        Fo = int(self.clean.kernel.shape[1])
        out = np.empty((n, T, Fo), dtype=np.float32)
        if n == 0 or T == 0:
            return out
        bs = min(bs, n)
        stateful = bool(getattr(self.cell, 'stateful', False))
        if length_aware and not stateful:
            if lengths is None:
                lens = self.valid_lengths(x, self.mask_value)
            else:
                lens = np.asarray(lengths, dtype=np.int64).reshape(-1)
                if lens.shape[0] != n or (lens < 0).any() or (lens > T).any():
                    raise ValueError('predict: lengths must be %d values in [0, %d]' % (n, T))
            order = np.argsort(-lens, kind='stable')          # longest first: the largest buffers come first
            step = self.PREDICT_T_STEP
            slabs = []
            for lo in range(0, n, bs):
                idx = order[lo:lo + bs]
                Ts = int(min(T, max(1, -(-int(lens[idx].max()) // step) * step)))
                slabs.append((idx, Ts))
        else:
            # (a stateful layer carries row i's state to row i of the next batch: order and length stay)
            slabs = [(np.arange(lo, min(lo + bs, n)), T) for lo in range(0, n, bs)]
        nbuf = 1 if len(slabs) == 1 else 2
        cap_T = max(Ts for _, Ts in slabs)
        pipe = getattr(self, '_predict_pipe', None)
        # (keyed on CAPACITY: the reference's loop -- 250, ..., the remainder, every epoch -- and the slabs of
        # different lengths above reuse one set of pinned buffers; they are re-made only to grow)
        if (pipe is None or pipe['F'] != (F, Fo) or pipe['cap_in'] < bs * cap_T * F or
                pipe['cap_out'] < bs * cap_T * Fo or pipe['nbuf'] < nbuf or pipe['dev'] != dev):
            self._predict_pipe = None                # (the old buffers go before the new ones come)
            cin, cout = bs * cap_T * F, bs * cap_T * Fo
            pin = lambda c: [torch.empty((c,), dtype=torch.float32, pin_memory=True) for _ in range(nbuf)]
            pipe = dict(F=(F, Fo), cap_in=cin, cap_out=cout, nbuf=nbuf, dev=dev, stage=pin(cin), back=pin(cout),
                        xd=[torch.empty((cin,), dtype=torch.float32, device=dev) for _ in range(nbuf)],
                        up=torch.cuda.Stream(dev), down=torch.cuda.Stream(dev))
            self._predict_pipe = pipe
        main = torch.cuda.current_stream(dev)
        ev = lambda: [torch.cuda.Event(), torch.cuda.Event()]
        ev_up, ev_comp, ev_down = ev(), ev(), ev()
        xs = torch.from_numpy(np.ascontiguousarray(x))
        view = lambda buf, b, Ts, w: buf[:b * Ts * w].view(b, Ts, w)

        # Host side of a slab: rows gathered into / scattered out of the pinned staging buffers by a few worker
        # threads (numpy releases the GIL around its copy loops).  One torch op per row costs ~0.5 ms of thread-pool
        # hand-off on a 256-thread host, one strided gather per slab runs at a third of memcpy speed -- either way the
        # host, not the GPU, then bounds the shipped r = 1000 model (tools/ragged_probe.py).
        x_np, out_np = xs.numpy(), out
        pool = self._host_pool()

        def rows(fn, b):
            nw = max(1, min(self.HOST_COPY_THREADS, b))
            futs = [pool.submit(fn, range(t, b, nw)) for t in range(nw)]
            for f in futs:
                f.result()

        def collect(s):                              # slab s: pinned staging -> the caller's array
            idx, Ts = slabs[s]
            ev_down[s & 1].synchronize()
            back = view(pipe['back'][s & 1], len(idx), Ts, Fo).numpy()

            def put(ks):
                for k in ks:
                    i = int(idx[k])
                    out_np[i, :Ts] = back[k]
                    if Ts < T:                       # masked frames repeat the last output (K.rnn)
                        out_np[i, Ts:] = back[k, Ts - 1]
            rows(put, len(idx))

        def send(s):                                 # slab s: the caller's array -> pinned staging -> device
            idx, Ts = slabs[s]
            j, b = s & 1, len(idx)
            if s >= 2:
                ev_up[j].synchronize()               # slab s-2 has left this staging buffer
            stage = view(pipe['stage'][j], b, Ts, F)
            stage_np = stage.numpy()

            def get(ks):
                for k in ks:
                    stage_np[k] = x_np[int(idx[k]), :Ts]
            rows(get, b)
            with torch.cuda.stream(pipe['up']):
                if s >= 2:
                    pipe['up'].wait_event(ev_comp[j])    # ... and its chain has read xd[j]
                view(pipe['xd'][j], b, Ts, F).copy_(stage, non_blocking=True)
                ev_up[j].record(pipe['up'])
        try:
            send(0)
            for s, (idx, Ts) in enumerate(slabs):
                j, b = s & 1, len(idx)
                if s + 1 < len(slabs):
                    send(s + 1)  # one slab ahead: enqueueing a slab's launches keeps this thread busy for
                                 # about as long as the device needs to run them
                main.wait_event(ev_up[j])
                mask = self.forward(view(pipe['xd'][j], b, Ts, F))
                ev_comp[j].record(main)
                with torch.cuda.stream(pipe['down']):
                    pipe['down'].wait_event(ev_comp[j])
                    view(pipe['back'][j], b, Ts, Fo).copy_(mask, non_blocking=True)
                    ev_down[j].record(pipe['down'])
                mask.record_stream(pipe['down'])
                if s >= 1:
                    collect(s - 1)
            collect(len(slabs) - 1)
        finally:
            # (an exception inside the loop -- out of memory, a raise from forward -- must not leave copies
            # in flight on the side streams into buffers the next call reuses)
            pipe['up'].synchronize()
            pipe['down'].synchronize()
            main.synchronize()
        ops.check_status(dev)
        return out

    def free_predict_buffers(self):
        """Drop the pinned staging buffers, device slabs and copy streams `predict` / `predict_on_batch` keep
        between calls (they are re-made on the next call)."""
        self._predict_pipe = None
        pool = getattr(self, '_host_pool_obj', None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._host_pool_obj = None

    __call__ = forward

    # -- training: loss 'mse_of_masked' + Adam (enhance.py:1040-1073, 1152) -----------------
    N_SCALARS = 4      # tail of the flat buffer: [sum w*mse, #frames with w != 0, #frames, fault]
    PENDING_STEPS = 8  # optimiser steps whose reports may be outstanding before the host looks at one

    def compile(self, loss='mse', optimizer='adam', lr=1e-3, clipnorm=0., decay=0., beta_1=0.9,
                beta_2=0.999, epsilon=1e-8, sample_weight_mode='temporal',
                loss_norm='masked_mean'):
        """model.compile(loss='mse', optimizer=Adam(lr, clipnorm, decay),
        sample_weight_mode='temporal') applied to output_masked = input * mask
        (enhance.py:1042, 1057, 1071-1073).

        loss_norm ([K2.0.4-memory] either way): 'masked_mean' = sum(w*mse) /
        #(w != 0), Keras' weighted loss when the Masking layer's mask does NOT reach the output;
        'keras204' = the same divided once more by p = mean(mask) -- Keras 2.0.4's
        weighted_masked_objective applies score*m/mean(m) AND score*w/mean(w != 0) when the mask
        does propagate (the sample weights are the mask itself, enhance.py:1148-1152).  The two
        differ by the per-batch factor 1/p on loss and gradients (Adam largely cancels it).
        Under torch.distributed rank 0's weights are broadcast so that all replicas start equal."""
        if loss != 'mse' or optimizer != 'adam' or sample_weight_mode != 'temporal':
            raise NotImplementedError("only loss='mse', optimizer='adam', temporal sample weights "
                                      "(the reference's training configuration)")
        if loss_norm not in ('masked_mean', 'keras204'):
            raise ValueError("loss_norm must be 'masked_mean' or 'keras204'")
        self.loss_norm = loss_norm
        self.opt = dict(lr=float(lr), clipnorm=float(clipnorm), decay=float(decay),
                        b1=float(beta_1), b2=float(beta_2), eps=float(epsilon), iterations=0)
        cell = self.cell
        if cell.divergence != 'ed' and any(k in ('log_U1', 'log_Uk') for k in cell.keys_trainable):
            raise NotImplementedError('the KL / beta variant of the cell has no U term to train')
        if cell._train_blockers:
            raise NotImplementedError('training with %s is not implemented' %
                                      ', '.join(cell._train_blockers))
        # The fused BPTT (csrc/cell_backward.hip) covers log_D, log_alph, log_lam1 and log_h0 of the
        # build_alt configuration -- the shipped params_trainable is [log_D, log_alph].  Any other
        # trainable key (log_U1 / log_Uk: U leaves its rank structure with the first update), a
        # trained dense U, caller maps or free weights train on the dense-matrix path.
        cell._train_dense = False
        if cell._generic or cell._dense_now:
            cell._train_dense = True
        else:
            covered = set(w for w, _, _ in cell.grad_slices())
            if any(k in cell.keys_trainable and k not in covered for k in cell._alt):
                cell._train_dense = True
        # (a step may move the matrices away from what the fused inference kernels assume -- unless
        # the dense path was chosen for recurrent dropout alone: those kernels carry B_U)
        cell._dense_after_step = cell._train_dense
        if cell.dropout_U:
            cell._train_dense = True
        if cell._train_dense:
            self._train_items = cell.trainable_weight_items()
        else:
            # flat-buffer order = the order of the BPTT's stacked outputs (d_log_D [n_D,F,N], d_log_h0,
            # d_log_alph, d_log_lam1), so that the kernels write their gradients straight into the flat
            # buffer (_grad_targets) instead of 2K+3 device-to-device copies per step
            sl = cell.grad_slices()
            names = [w for w, gk, _ in sl if gk == 'd_log_D'] + ['log_h0'] + \
                    [w for w, gk, _ in sl if gk != 'd_log_D']
            names += [k for k in cell._alt if k not in names]
            self._train_items = [(k, cell.log_h0 if k == 'log_h0' else cell._alt[k]) for k in names
                                 if k == 'log_h0' or k in cell.keys_trainable]
        self._train_items += [('kernel_clean', self.clean.kernel), ('kernel_noise', self.noise.kernel)]
        total = sum(int(t.numel()) for _, t in self._train_items)
        self._flat = torch.zeros(total + self.N_SCALARS, dtype=torch.float32, device=cell.device)
        # Adam moments as flat buffers parallel to the gradient (ONE fused launch per step,
        # ops.adam_step_flat); _opt_state keeps per-weight views of them
        self._mflat = torch.zeros(total, dtype=torch.float32, device=cell.device)
        self._vflat = torch.zeros(total, dtype=torch.float32, device=cell.device)
        self._gview, self._opt_state, o = {}, {}, 0
        for n, t in self._train_items:
            self._gview[n] = self._flat[o:o + t.numel()].view(t.shape)
            self._opt_state[n] = (self._mflat[o:o + t.numel()].view(t.shape),
                                  self._vflat[o:o + t.numel()].view(t.shape))
            o += t.numel()
        self._grad_targets = self._make_grad_targets()
        self._adam_table = None          # (table, n_blocks, storage pointers it was built for)
        self._sumsq = torch.zeros(256, dtype=torch.float32, device=cell.device)
        self._pending = []               # DeviceLoss of recent steps, oldest first
        self._step_no = 0
        # applied optimiser steps, ON THE DEVICE (ping-pong pair: launch n reads [n & 1], writes [(n + 1) & 1];
        # drnmf_adam_step_flat_counted): a step the fault word skipped is not counted, identically on every rank
        self._step_dev = torch.zeros((2,), dtype=torch.float32, device=cell.device)
        self.sync_replicas()
        return self

    def _make_grad_targets(self):
        """Views of the flat gradient buffer the kernels write into: (`grads` of cell.backward -- a
        stacked output whose slices are ALL trained and lie next to each other in stack order --,
        (sums, d_kernel_clean, d_kernel_noise) of the loss head).  What has no view is copied."""
        cell, gv, ns = self.cell, self._gview, self.N_SCALARS
        head = (self._flat[-ns:-ns + 2], gv['kernel_clean'], gv['kernel_noise'])
        if cell._train_dense:
            return {}, head
        off, o = {}, 0
        for n, t in self._train_items:
            off[n] = (o, int(t.numel()))
            o += int(t.numel())
        grads = {'d_log_h0': gv['log_h0'].view(-1)}
        by_key = {}
        for wname, gkey, idx in cell.grad_slices():
            by_key.setdefault(gkey, []).append((idx, wname))
        for gkey, lst in by_key.items():
            lst.sort()
            if any(w not in off for _, w in lst):
                continue
            start, size = off[lst[0][1]]
            if all(off[w] == (start + i * size, size) for i, (_, w) in enumerate(lst)):
                grads[gkey] = self._flat[start:start + len(lst) * size].view(len(lst), -1)
        return grads, head

    def _collect_grads(self, g):
        """Whatever cell.backward did not write into the flat buffer itself."""
        gv = self._gview
        if 'by_name' in g:                          # dense-matrix path: gradients per weight name
            for wname, gw in g['by_name'].items():
                gv[wname].copy_(gw.reshape(gv[wname].shape))
            return
        direct = self._grad_targets[0]
        for wname, gkey, idx in self.cell.grad_slices():
            if wname in gv and gkey not in direct:
                gv[wname].copy_(g[gkey][idx].reshape(gv[wname].shape))

    def sync_replicas(self, root=0):
        """Data parallelism: every weight of the model (trainable or not) is replaced by rank
        `root`'s, in one broadcast of a flat buffer -- replicas must not depend on every rank
        having drawn the same random initial log_h0.  No-op for a single rank."""
        from . import dp
        if dp.world_size() <= 1:
            return
        ws = self.weights
        flat = torch.cat([w.reshape(-1) for w in ws]).contiguous()
        dp.broadcast_(flat, root)
        o = 0
        for w in ws:
            w.copy_(flat[o:o + w.numel()].view(w.shape))
            o += w.numel()
        for l in (self.cell, self.clean, self.noise):
            l._weights_changed()

    def loss_and_grads(self, x, y, sample_weight, live=True):
        """Unnormalised loss/gradients of one (local) batch into the flat buffer; returns the
        flat tensor [grads..., sum w*mse, count, frames].  `self.phase_events` (bench.py only): a
        dict that receives (start, end) torch events of the cell forward, the head + loss, and the
        BPTT, recorded on the current stream without synchronising."""
        cell = self.cell
        N, K = cell.output_dim, cell.K_layers
        self._drop_stale_fault()
        pe = getattr(self, 'phase_events', None)

        def mark(name, which):
            if pe is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                pe.setdefault(name, [None, None])[which] = ev
        mark('cell_forward', 0)
        hall = cell.forward_train(x, mask_value=self.mask_value)
        mark('cell_forward', 1)
        mark('head_and_loss', 0)
        h_off = (K - 1) * N
        mask, A, Bn = ops.head_forward(hall, self.clean.kernel, self.noise.kernel,
                                       square=self.square, want_ab=True, h_off=h_off)
        grads, head_out = self._grad_targets
        _, d_hidden, _, _ = ops.loss_head_backward(
            x, hall, self.clean.kernel, self.noise.kernel, mask, A, Bn, y, sample_weight,
            square=self.square, h_off=h_off, out=head_out)
        mark('head_and_loss', 1)
        mark('cell_backward', 0)
        g = cell.backward(x, hall, d_hidden, grads=grads,
                          profile=getattr(self, 'backward_profile', None))
        mark('cell_backward', 1)
        self._collect_grads(g)
        self._set_scalars(x, live)
        return self._flat

    def _drop_stale_fault(self):
        """Stream-ordered read-and-clear of the handle's fault word into a scratch, at the START of a
        training / test / validation step: a fault raised by an EARLIER asynchronous call on this device
        (model.forward, cell.call, ops.cell_forward -- whose caller is told by ops.check_status after its own
        synchronisation, or not at all) must not be charged to this step, which would then be skipped, or
        raised, on every rank.  The count of dropped words stays readable in `self._stale_faults`."""
        sf = getattr(self, '_stale_faults', None)
        if sf is None or sf.device != torch.device(self.cell.device):
            sf = self._stale_faults = torch.zeros(1, dtype=torch.float32, device=self.cell.device)
        ops.status_take(sf)

    def _set_scalars(self, x, live=True):
        # tail of the flat buffer: [sum w*mse, count] were written by the loss head (_grad_targets), then
        # rows and the fault word.  (A step that is not live -- fit() on a rank whose shard has run out
        # replays a batch with zero weights -- contributes no rows either: 'rows' feeds p = count / rows of
        # loss_norm='keras204', which must be the GLOBAL batch's, not inflated by the replay)
        # (fill_, a kernel: `tensor[i] = python_float` is a BLOCKING host-to-device copy of a CPU scalar --
        # one hidden hipMemcpyWithStream per step in rounds 1-3, profiles/r04g_step_hip_api_delta.txt)
        self._flat[-2:-1].fill_(float(x.shape[0] * x.shape[1]) if live else 0.0)
        self._flat[-1:].fill_(0.0)
        # fault word: 1.0 if a persistent chain of THIS step's forward / BPTT gave up (stream-ordered
        # read-and-clear of the handle's flag; all-reduced with the gradients, so that every rank skips
        # the update and reports the step -- ADVICE r3)
        ops.status_take(self._flat[-1:])

    @staticmethod
    def _reg_coeffs(reg):
        """(l1, l2) of a Keras-style regularizer: an object with .l1 / .l2 (keras.regularizers.L1L2),
        a dict {'l1':, 'l2':} (its get_config) or a pair."""
        if reg is None:
            return 0.0, 0.0
        if isinstance(reg, dict):
            return float(reg.get('l1', 0.0)), float(reg.get('l2', 0.0))
        if isinstance(reg, (tuple, list)):
            return float(reg[0]), float(reg[1])
        return float(getattr(reg, 'l1', 0.0)), float(getattr(reg, 'l2', 0.0))

    def _regularized_items(self):
        """[(weight name, tensor, l1, l2)]: the cell's free W_k / U_k / b_k matrices under its
        W_ / U_ / b_regularizer (custom_layers.py:245-269); empty for build_alt's configuration."""
        cell, out = self.cell, []
        regs = {'W_': cell.W_regularizer, 'U_': cell.U_regularizer, 'b_': cell.b_regularizer}
        for n, p in self._train_items:
            for prefix, reg in regs.items():
                l1, l2 = self._reg_coeffs(reg)
                if n.startswith(prefix) and n in getattr(cell, '_free', {}) and (l1 or l2):
                    out.append((n, p, l1, l2))
        return out

    def _regularization_loss(self):
        tot = 0.0
        for _, p, l1, l2 in self._regularized_items():
            tot += l1 * float(p.abs().sum()) + l2 * float((p * p).sum())
        return tot

    def _add_regularizers(self, scale):
        """Adds d/dw [l1 |w| + l2 w^2] of every regularized weight to the flat gradient -- divided
        by `scale`, the factor the optimiser step applies to the DATA gradient sums (the penalty is
        not a sum over frames; it enters once, after the all-reduce) -- and returns the penalty
        (Keras: total_loss = loss + sum of layer.losses)."""
        tot = 0.0
        for n, p, l1, l2 in self._regularized_items():
            self._gview[n].add_((l1 * torch.sign(p) + (2.0 * l2) * p) / scale)
            tot += l1 * float(p.abs().sum()) + l2 * float((p * p).sum())
        return tot

    def _adam_blocks(self):
        """Block table of the fused Adam launch, rebuilt if a weight tensor has been re-bound."""
        ptrs = tuple(int(t.data_ptr()) for _, t in self._train_items)
        if self._adam_table is None or self._adam_table[2] != ptrs:
            table, nb = ops.adam_block_table(self._train_items, self.cell.device)
            self._adam_table = (table, nb, ptrs)
        return self._adam_table[0], self._adam_table[1]

    def apply_gradients(self, flat):
        """Adam step from the (already all-reduced) flat buffer [gradients..., sum w*mse, count, frames,
        fault]: ONE launch (csrc/train.hip adam_flat_kernel) that derives 1/count, the keras204 factor and
        the clip scale from device memory and skips the update if the fault word is set.  Nothing here
        waits for the device; returns the normalised loss as a DeviceLoss (float() it to wait)."""
        o = self.opt
        ns = self.N_SCALARS
        keras204 = getattr(self, 'loss_norm', 'masked_mean') == 'keras204'
        reg_loss = 0.0
        if self._regularized_items():
            # free-weight configurations with W_/U_/b_regularizer only (never build_alt's): the penalty's
            # gradient enters scaled by the step's data-gradient scale, which the host must then know
            sse, cnt, rows = (float(v) for v in flat[-ns:-1].tolist())
            scale = 1.0 / max(cnt, 1.0)
            if keras204:
                scale *= rows / max(cnt, 1.0)
            reg_loss = self._add_regularizers(scale)     # (before the clip: Keras clips the total gradient)
        if o['clipnorm'] > 0:                           # global-norm clip [K2.0.4-memory]
            ops.sumsq_partials(flat[:-ns], out=self._sumsq)
        table, nb = self._adam_blocks()
        ring, base = ops.host_report_ring(self.cell.device)
        slot, register = _claim_report_slot(ops._dev_of(self.cell.device), ring.shape[0])
        j = self._step_no & 1
        self._step_no += 1
        # (lr_t -- decay and bias correction -- is evaluated by the launch from the device's own count of applied
        # steps: a host-side count, corrected whenever a fault report happened to be read, made the ranks of a
        # data-parallel group pass different lr_t for a few steps after a skipped one)
        ops.adam_step_flat_counted(table, nb, flat, self._mflat, self._vflat, flat[-ns:], o['lr'], o['decay'],
                                   self._step_dev[j:j + 1], self._step_dev[1 - j:2 - j], beta1=o['b1'],
                                   beta2=o['b2'], eps=o['eps'], clipnorm=o['clipnorm'], keras204=keras204,
                                   reg_loss=reg_loss, sumsq256=self._sumsq if o['clipnorm'] > 0 else None,
                                   report=base + 16 * slot)
        ev = torch.cuda.Event()
        ev.record()
        # opt['iterations'] is the host's MIRROR of that count (get_config, logging): steps enqueued minus the
        # skipped ones it has been told about so far; nothing on the device depends on it
        def uncount(o=o):
            o['iterations'] = max(0, o['iterations'] - 1)
        loss = DeviceLoss(ev, ring[slot], self.cell.device, on_fault=uncount)
        register(loss)
        o['iterations'] += 1
        # the prepared parameter block is stale; the (u0_diag, u0_off, uk_off) scalars are derived
        # from log_U1 / log_Uk, which compile() refuses to train: re-deriving them here would cost
        # two N x N device-to-host copies and a host pass per step
        self.cell._params_block_valid = False
        if getattr(self.cell, '_dense_after_step', getattr(self.cell, '_train_dense', False)):
            # the matrices have moved: inference runs on the dense kernel from now on (no host
            # round trip to re-examine U's structure after every step)
            self.cell._dense_now = True
            self.cell._dense_block_valid = False
        # Reports are consumed at most PENDING_STEPS late: a fault is then raised from THIS call (the
        # wait is for a step that finished long ago; it never drains the queue), and a ring slot is
        # never reused before it was read.
        self._pending.append(loss)
        while len(self._pending) > self.PENDING_STEPS or (self._pending and self._pending[0].ready()):
            old = self._pending.pop(0)
            if old._error is None:       # (a fault its reader has already been told about is not raised again)
                old._get()
        return loss

    def train_on_batch(self, x, y, sample_weight=None, _live=True):
        """One optimiser step.  x, y: (B,T,F); sample_weight: (B,T) (the data mask,
        enhance.py:1148-1152).  Under torch.distributed the flat gradient + (sum, count, frames, fault)
        are all-reduced (RCCL) before the update, so every rank applies the same step.  Returns the
        loss as a DeviceLoss: the call enqueues the step and returns without waiting for it (float()
        the result, as Keras' train_on_batch would have, to wait)."""
        from . import dp
        dev = self.cell.device
        tt = lambda a: a if isinstance(a, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        x, y = tt(x), tt(y)
        if sample_weight is None:
            sample_weight = torch.ones(x.shape[:2], dtype=torch.float32, device=dev)
        flat = self.loss_and_grads(x, y, tt(sample_weight), live=_live)
        dp.allreduce_sum_(flat)
        return self.apply_gradients(flat)

    def test_on_batch(self, x, y, sample_weight=None):
        from . import dp
        dev = self.cell.device
        tt = lambda a: a if isinstance(a, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        x, y = tt(x), tt(y)
        w = tt(sample_weight) if sample_weight is not None else \
            torch.ones(x.shape[:2], dtype=torch.float32, device=dev)
        s3 = torch.empty(4, dtype=torch.float32, device=dev)
        self._drop_stale_fault()
        s3[:2].copy_(ops.loss_forward(y, w, x_raw=x, mask=self.forward(x)))
        s3[2:3].fill_(float(x.shape[0] * x.shape[1]))      # (fill_: no blocking scalar copy)
        s3[3:].fill_(0.0)
        ops.status_take(s3[3:])                 # fault word of this forward, all-reduced with the sums
        dp.allreduce_sum_(s3)
        sse, cnt, rows, fault = s3.tolist()
        if fault != 0.0:
            raise _capi.DrnmfError('test_on_batch: a persistent small-shape chain timed out on at least '
                                   'one rank (DRNMF_ERR_TIMEOUT); rerun, or set DRNMF_PERSIST=0')
        scale = 1.0 / max(cnt, 1.0)
        if getattr(self, 'loss_norm', 'masked_mean') == 'keras204':
            scale *= rows / max(cnt, 1.0)
        return sse * scale + (self._regularization_loss() if hasattr(self, '_train_items') else 0.0)

    def fit(self, x, y, sample_weight=None, batch_size=32, epochs=1, validation_data=None,
            shuffle=True, seed=7654, verbose=0, callbacks=None, resident_bytes=64 << 30):
        """Minimal keras.Model.fit: shuffled mini-batches (np.random.seed(7654), enhance.py:7),
        Keras-style callbacks (callbacks.py: on_train_begin / on_batch_end / on_epoch_end with
        logs {'loss', 'val_loss'}; a callback may set model.stop_training), returns {'loss': [...],
        'val_loss': [...]} per epoch.  Under torch.distributed every rank passes ITS shard of the
        data; batches are all-reduced per step and the ranks take the same number of steps
        (a rank with fewer batches joins the remaining reductions with zero weights).
        Host (numpy) data sets of up to `resident_bytes` (64 GiB: the reference's whole CHiME2 training
        tensors are ~6 GB, the GPU has 288 GB) are uploaded ONCE and mini-batches gathered on the
        device -- a per-batch host-to-device copy of x and y costs ~1 ms of the 10-ms step of the
        shipped r = 100 configuration; larger sets (or resident_bytes=0) are copied batch by batch as
        Keras does."""
        from . import dp
        dev = self.cell.device
        # what may stay resident: at most `resident_bytes` and at most half of the device memory that is
        # free right now (workspaces, other tenants of the GPU); a failed upload falls back to
        # per-batch copies
        try:
            free_now = int(torch.cuda.mem_get_info(dev)[0])
        except Exception:
            free_now = 0
        budget = min(int(resident_bytes), free_now // 2)

        def upload(arrays):
            """Device copies of the numpy members of `arrays` (one copy per distinct array: the
            pretraining model passes y = x) if they fit the budget, else None."""
            nonlocal budget
            uniq = {}
            for a in arrays:
                if isinstance(a, np.ndarray):
                    uniq.setdefault(id(a), a)
            need = sum(a.size * 4 for a in uniq.values())
            if not uniq or need > budget:
                return None
            try:
                dev_of = {k: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
                          for k, a in uniq.items()}
            except torch.cuda.OutOfMemoryError:
                torch.cuda.empty_cache()
                return None
            budget -= need
            return [dev_of[id(a)] if isinstance(a, np.ndarray) else a for a in arrays]
        up = upload([x, y, sample_weight])
        if up is not None:
            x, y, sample_weight = up
        if validation_data is not None:
            up = upload(list(validation_data))
            if up is not None:
                validation_data = tuple(up)
        take = lambda a, b: a.index_select(0, torch.as_tensor(b, dtype=torch.long, device=a.device)) \
            if isinstance(a, torch.Tensor) else a[b]
        n = x.shape[0]
        rng = np.random.RandomState(seed)
        hist = {'loss': [], 'val_loss': []}
        # file-writing callbacks (rank0_only: ModelCheckpoint, LossHistory) run on rank 0 only --
        # every rank holds the same weights and logs; the others (EarlyStopping) run everywhere so
        # that all ranks stop together
        callbacks = [cb for cb in (callbacks or [])
                     if dp.rank() == 0 or not getattr(cb, 'rank0_only', False)]
        self.stop_training = False
        for cb in callbacks:
            if hasattr(cb, 'set_model'):
                cb.set_model(self)
            else:
                cb.model = self
            if hasattr(cb, 'on_train_begin'):
                cb.on_train_begin({})
        # a callback that looks at every batch's loss needs the number: the step then waits for it, as
        # Keras does; without one the losses of an epoch are read when the epoch ends
        per_batch_cbs = [cb for cb in callbacks if hasattr(cb, 'on_batch_end')]
        # every train_on_batch is a collective: all ranks must take the same number of steps even
        # when their shards differ in size (dp.shard hands the remainder to the first ranks)
        steps = dp.max_over_ranks(epoch_steps(n, batch_size))
        # an empty shard has no batch to replay with zero weights; every rank must learn of it HERE
        # (one that raised alone would leave the others waiting in the first all-reduce)
        if dp.max_over_ranks(1 if n == 0 else 0):
            raise ValueError("fit(): a rank holds no sequences (%d here); give every rank at least "
                             "one (dp.shard of fewer sequences than ranks?)" % n)
        for ep in range(epochs):
            for cb in callbacks:
                if hasattr(cb, 'on_epoch_begin'):
                    cb.on_epoch_begin(ep, {})
            idx = rng.permutation(n) if shuffle else np.arange(n)
            ep_losses, cnt = [], 0
            for b, live in epoch_batches(idx, batch_size, steps):
                sw = None if sample_weight is None else take(sample_weight, b)
                if not live:
                    # this rank has run out of data: it joins the all-reduce with zero weights
                    # (zero gradient, zero count) on a batch it has already used
                    sw = np.zeros((len(b), x.shape[1]), np.float32)
                loss = self.train_on_batch(take(x, b), take(y, b), sw, _live=live)
                ep_losses.append(loss)
                for cb in per_batch_cbs:
                    cb.on_batch_end(cnt, {'batch': cnt, 'size': len(b), 'loss': float(loss)})
                cnt += 1
            logs = {'loss': sum(float(l) for l in ep_losses) / max(cnt, 1)}
            hist['loss'].append(logs['loss'])
            if validation_data is not None:
                logs['val_loss'] = self._validate(validation_data, batch_size, take)
                hist['val_loss'].append(logs['val_loss'])
            if verbose and dp.rank() == 0:
                print('epoch %d loss %.6f%s' % (ep + 1, logs['loss'],
                      (' val_loss %.6f' % logs['val_loss']) if validation_data else ''))
            for cb in callbacks:
                if hasattr(cb, 'on_epoch_end'):
                    cb.on_epoch_end(ep, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            if hasattr(cb, 'on_train_end'):
                cb.on_train_end({})
        return hist

    def _validation_sums(self, x, y, w):
        """[sum w*loss, #frames with w != 0] of one validation batch as a device tensor of 2 floats."""
        return ops.loss_forward(y, w, x_raw=x, mask=self.forward(x))

    def _validate(self, validation_data, batch_size, take):
        """Validation loss of fit(), evaluated in mini-batches of `batch_size` sequences as Keras'
        test loop does (enhance.py:1152-1157: model.fit(..., validation_data=...)); the reference's
        2 460-utterance CHiME2 validation set at N = 2000 would otherwise need a ~40 GB hidden tensor.
        The sums [sum w*loss, count, frames, fault] are accumulated on the device over the batches and
        all-reduced ONCE (every rank holds its shard), so the value is the loss of the whole set."""
        from . import dp
        dev = self.cell.device
        tt = lambda a: a if isinstance(a, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        xv, yv = validation_data[0], validation_data[1]
        wv = validation_data[2] if len(validation_data) > 2 else None
        nv, Tv = xv.shape[0], xv.shape[1]
        s4 = torch.zeros(4, dtype=torch.float32, device=dev)
        self._drop_stale_fault()
        # Length-aware, as predict(): a frame behind a sequence's last non-zero weight adds nothing to either
        # sum (the reference's sample weights are the data mask, enhance.py:1148-1152) and nothing to the
        # frames before it, so the sequences are taken in order of that length and every mini-batch runs at the
        # longest one IN IT (rounded up to PREDICT_T_STEP frames).  The same terms are summed; only their
        # grouping into mini-batches -- fp32 partial sums -- differs from the input-order pass.
        order, lens = np.arange(nv), None
        if wv is not None and not bool(getattr(self.cell, 'stateful', False)) and Tv > self.PREDICT_T_STEP:
            key = (id(wv), tuple(wv.shape))
            cached = getattr(self, '_val_lens', None)
            if cached is None or cached[0] != key:
                wt = wv if isinstance(wv, torch.Tensor) else torch.from_numpy(np.asarray(wv))
                pos = torch.arange(1, Tv + 1, device=wt.device).unsqueeze(0)
                cached = (key, ((wt != 0) * pos).amax(dim=1).cpu().numpy())
                self._val_lens = cached
            lens = cached[1]
            order = np.argsort(-lens, kind='stable')
        for lo in range(0, nv, int(batch_size)):
            b = order[lo:min(nv, lo + int(batch_size))]
            Tb = Tv
            if lens is not None:
                step = self.PREDICT_T_STEP
                Tb = int(min(Tv, max(1, -(-int(lens[b].max()) // step) * step)))
            cut = (lambda a: a) if Tb == Tv else (lambda a: a[:, :Tb].contiguous() if isinstance(a, torch.Tensor)
                                                  else np.ascontiguousarray(a[:, :Tb]))
            xb, yb = tt(cut(take(xv, b))), tt(cut(take(yv, b)))
            wb = tt(cut(take(wv, b))) if wv is not None else \
                torch.ones(xb.shape[:2], dtype=torch.float32, device=dev)
            s4[:2].add_(self._validation_sums(xb, yb, wb))
            s4[2] += float(len(b) * Tv)          # (frames of the PADDED batch: loss_norm='keras204' divides by them)
        ops.status_take(s4[3:])
        dp.allreduce_sum_(s4)
        sse, cnt, rows, fault = s4.tolist()
        if fault != 0.0:
            raise _capi.DrnmfError('fit(): a persistent small-shape chain timed out during validation on '
                                   'at least one rank (DRNMF_ERR_TIMEOUT); rerun, or set DRNMF_PERSIST=0')
        scale = 1.0 / max(cnt, 1.0)
        if getattr(self, 'loss_norm', 'masked_mean') == 'keras204':
            scale *= rows / max(cnt, 1.0)
        return sse * scale + (self._regularization_loss() if hasattr(self, '_train_items') else 0.0)


def epoch_steps(n_local, batch_size):
    """Mini-batches one epoch of fit() takes over n_local sequences."""
    return (int(n_local) + int(batch_size) - 1) // int(batch_size)


def epoch_batches(idx, batch_size, steps):
    """The (index array, live) pairs of one epoch: the rank's own mini-batches of `idx`, then --
    when the data-parallel group takes more steps than this rank has batches -- its first batch
    again, flagged not live (fit() gives it zero sample weights)."""
    n = len(idx)
    own = epoch_steps(n, batch_size)
    for s in range(steps):
        if s < own:
            yield idx[s * batch_size:(s + 1) * batch_size], True
        else:
            yield idx[:batch_size], False


def l1_of_output(y_true=None, y_pred=None):
    """Marker for the second pretraining loss, `K.mean(K.abs(y_pred), axis=-1)` (enhance.py:1027)."""
    raise NotImplementedError("l1_of_output is evaluated on the device inside "
                              "SNMFCostPretrainModel; pass it to compile(loss=[...]) only")


class SNMFCostPretrainModel(UnfoldedSNMFModel):
    """enhance.py:1023-1035: `model_pretrain = Model(inputs=model.input, outputs=[x_recon,
    h_estimated])` with x_recon = clean_est + noise_est and h_estimated the cell output, compiled
    with loss ['mse', l1_of_output] and loss_weights [0.5, lam1*2r/input_dim], fitted on
    (x, [x, x]) with sample_weight [mask, mask] (enhance.py:1110-1115).  Shares every weight tensor
    with the model it was made from, so after pretraining `model` already holds the weights
    (the reference reloads them from the checkpoint, enhance.py:1118-1119)."""

    def __init__(self, model):
        if model.square:
            raise NotImplementedError("pretrain_with_snmf_cost is defined by layer indices that "
                                      "assume no transform_before_irm (enhance.py:1024-1025)")
        UnfoldedSNMFModel.__init__(self, model.layers, model.cell, model.clean, model.noise,
                                   model.mask_value, False)
        self.loss_weights = None

    def compile(self, loss=('mse', l1_of_output), loss_weights=None, **kw):
        loss = list(loss)
        if len(loss) != 2 or loss[0] != 'mse' or loss[1] is not l1_of_output:
            raise NotImplementedError("pretraining loss must be ['mse', l1_of_output]")
        if loss_weights is None or len(loss_weights) != 2:
            raise ValueError("loss_weights=[0.5, lam1*2r/input_dim] is required (enhance.py:1035)")
        if abs(float(loss_weights[0]) - 0.5) > 1e-12:
            raise NotImplementedError("the reconstruction loss weight is 0.5 (enhance.py:1035)")
        self.loss_weights = [float(loss_weights[0]), float(loss_weights[1])]
        return UnfoldedSNMFModel.compile(self, loss='mse', **kw)

    @staticmethod
    def _first(a):
        return a[0] if isinstance(a, (list, tuple)) else a

    def forward(self, x, want_hidden=False):
        """-> [x_recon, h_estimated]"""
        h = self.cell.call(x, mask_value=self.mask_value)
        _, A, Bn = ops.head_forward(h, self.clean.kernel, self.noise.kernel, want_ab=True)
        return [ops.add(A, Bn), h]

    def predict_on_batch(self, x):
        xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.cell.device)
        out = [o.cpu().numpy() for o in self.forward(xt)]
        ops.check_status(self.cell.device)      # (the copies synchronised: a chain that timed out raises HERE)
        return out

    def predict(self, x, batch_size=250, verbose=0):
        """Two outputs per slab (pretraining only: the plain loop, no copy streams)."""
        bs = max(1, int(batch_size))
        parts = [self.predict_on_batch(x[s:s + bs]) for s in range(0, max(len(x), 1), bs)]
        return [np.concatenate([p[i] for p in parts]) for i in range(2)]

    def loss_and_grads(self, x, y, sample_weight, live=True):
        cell = self.cell
        N, K = cell.output_dim, cell.K_layers
        self._drop_stale_fault()
        hall = cell.forward_train(x, mask_value=self.mask_value)
        h_off = (K - 1) * N
        _, A, Bn = ops.head_forward(hall, self.clean.kernel, self.noise.kernel, want_ab=True,
                                    h_off=h_off)
        # the targets are the input itself (enhance.py:1110); y is accepted for API symmetry
        grads, head_out = self._grad_targets
        _, d_hidden, _, _ = ops.snmf_cost_head_backward(
            self._first(y), hall, self.clean.kernel, self.noise.kernel, A, Bn,
            self._first(sample_weight), self.loss_weights[1], h_off=h_off, out=head_out)
        g = cell.backward(x, hall, d_hidden, grads=grads)
        self._collect_grads(g)
        self._set_scalars(x, live)
        return self._flat

    def train_on_batch(self, x, y=None, sample_weight=None, _live=True):
        y = x if y is None else self._first(y)
        return UnfoldedSNMFModel.train_on_batch(self, x, y, self._first(sample_weight), _live=_live)

    def test_on_batch(self, x, y=None, sample_weight=None):
        from . import dp
        dev = self.cell.device
        tt = lambda a: a if isinstance(a, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        x = tt(x)
        y = x if y is None else tt(self._first(y))
        sw = self._first(sample_weight)
        w = tt(sw) if sw is not None else torch.ones(x.shape[:2], dtype=torch.float32, device=dev)
        s3 = torch.zeros(3, dtype=torch.float32, device=dev)
        self._drop_stale_fault()
        s3[:2].copy_(self._validation_sums(x, y, w))
        ops.status_take(s3[2:])
        dp.allreduce_sum_(s3)
        sse, cnt, fault = s3.tolist()
        if fault != 0.0:
            raise _capi.DrnmfError('test_on_batch: a persistent small-shape chain timed out on at least '
                                   'one rank (DRNMF_ERR_TIMEOUT); rerun, or set DRNMF_PERSIST=0')
        return sse / max(cnt, 1.0)

    def _validation_sums(self, x, y, w):
        h = self.cell.call(x, mask_value=self.mask_value)
        _, A, Bn = ops.head_forward(h, self.clean.kernel, self.noise.kernel, want_ab=True)
        return ops.loss_forward(y, w, A=A, Bn=Bn, hidden=h, l1_weight=self.loss_weights[1])

    def fit(self, x, y=None, sample_weight=None, validation_data=None, **kw):
        y = x if y is None else self._first(y)
        if validation_data is not None:
            xv, yv, wv = validation_data
            validation_data = (xv, self._first(yv), self._first(wv))
        return UnfoldedSNMFModel.fit(self, x, y, sample_weight=self._first(sample_weight),
                                     validation_data=validation_data, **kw)


def make_pretrain_model(model):
    """The `model_pretrain` of enhance.py:1023-1026 for `model = build_unfolded_snmf(...)`."""
    return SNMFCostPretrainModel(model)


def build_unfolded_snmf(params_unfolded_snmf, device=None):
    """enhance.py:209-317 with the same parameter dictionary keys."""
    p = params_unfolded_snmf
    input_dim, hidden_dim, output_dim = p['input_dim'], p['hidden_dim'], p['output_dim']
    mask_value, maxseq, K_layers = p['mask_value'], p['maxseq'], p['K_layers']
    W_noisy = np.asarray(p['W'], np.float32)
    if W_noisy.shape != (input_dim, hidden_dim):
        raise ValueError("params['W'] has shape %s, expected (input_dim, hidden_dim) = (%d, %d)"
                         % (W_noisy.shape, input_dim, hidden_dim))
    if output_dim != input_dim:
        raise ValueError('output_dim must equal input_dim (the mask multiplies the input)')

    params_const = {'W': W_noisy,                                         # enhance.py:219-223
                    'U1': np.eye(hidden_dim).astype(np.float32),
                    'Uk': np.zeros((hidden_dim, hidden_dim)).astype(np.float32),
                    'alph': np.float32(p['alph']), 'lam1': np.float32(p['lam1'])}
    if p.get('untie_alph'):                                               # enhance.py:225-226
        params_const['alph'] = params_const['alph'] * np.ones((hidden_dim,), np.float32)
    params_untied = p.get('params_untied', [])
    alt_params, maps_from_alt = build_alt(hidden_dim, K_layers, params_const,
                                          params_untied=params_untied)
    if 'params_trainable' not in p:
        # the reference hits a NameError here (enhance.py:239-248 -> 263); be explicit instead
        raise ValueError("params_unfolded_snmf must contain 'params_trainable'")
    keys_trainable = []
    for name in p['params_trainable']:                                    # enhance.py:241-248
        if name in params_untied:
            keys_trainable += [name + ('_%d' % k) for k in range(K_layers)]
        else:
            keys_trainable.append(name)

    transform = p.get('transform_before_irm')
    if transform not in (None, 'square'):
        raise ValueError("Unknown 'transform_before_irm' of '%s'" % transform)

    inp = InputLayer((maxseq, input_dim), name='masking_1_input')
    masking = Masking(mask_value=mask_value, input_shape=(maxseq, input_dim))
    cell = SimpleDeepRNN(hidden_dim, input_shape=(maxseq, input_dim), return_sequences=True,
                         activation='relu', K_layers=K_layers, alt_params=alt_params,
                         keys_trainable=keys_trainable, maps_from_alt=maps_from_alt,
                         flag_connect_input_to_layers=True, flag_nonnegative=True, device=device,
                         operand_dtype=p.get('operand_dtype', 'float32'),
                         divergence=p.get('divergence', 'ed'), beta=p.get('beta', 1.5))
    cell.build((None, maxseq, input_dim))
    r = hidden_dim // 2
    log_W_clean = np.log(np.float32(1e-7) + W_noisy[:, :r])               # enhance.py:282
    log_W_noise = np.log(np.float32(1e-7) + W_noisy[:, r:])               # enhance.py:291
    clean = DenseNonNegW(output_dim, use_bias=False, weights=[log_W_clean.T], device=device)
    noise = DenseNonNegW(output_dim, use_bias=False, weights=[log_W_noise.T], device=device)
    layers = [inp, masking, cell, Lambda('h[:, :, :r]', name='H_clean'),
              Lambda('h[:, :, r:]', name='H_noise'), TimeDistributed(clean, name='clean_est'),
              TimeDistributed(noise, name='noise_est')]
    if transform == 'square':
        layers += [Lambda('square', name='clean_est_xformed'),
                   Lambda('square', name='noise_est_xformed')]
    layers.append(DivideAbyAplusB())
    return UnfoldedSNMFModel(layers, cell, clean, noise, mask_value, transform == 'square')
