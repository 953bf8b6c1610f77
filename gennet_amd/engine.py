"""Keras-style model engine over the HIP kernel library: Layer / Input / Sequential / Model, compile(),
train_on_batch(), predict(), fit(), weight persistence.

Mirrors the Keras 2.2.4 surface that BBH_version/bbhMahoGANy.py uses:
  * Sequential().add(layer | model), functional Input(shape) -> Layer(...)(tensor) -> Model(inputs=, outputs=[...], name=)
    (bbhMahoGANy.py:221-295, :357-404, :432-498, :516-518, :536-538)
  * model.trainable / layer.trainable assignment with COLLECT-AT-COMPILE semantics (:797-809, :1104-1115)
  * compile(loss=, optimizer=Adam(lr=, beta_1=), metrics=['accuracy']) (:1101-1119); one optimizer state per compiled model
  * train_on_batch(x, y) -> [loss, *metrics] python floats, keras ordering for multi-output models (:1165, :1292, :1296)
  * predict(x) -> ndarray | [ndarray] (default batch_size 32) (:1185, :1248, :1343); learning phase 1 in train_on_batch for
    the WHOLE graph, 0 in predict
  * save / save_weights / load_weights / load_model (:1135-1142, :1173, :1373-1375)

Not a tracing compiler: a model is a flat list of nodes executed eagerly, one or two HIP kernels per node, with peephole
fusions decided once per graph (conv+activation epilogue, BN+activation+dropout single pass).  Weights that a compiled
model trains live in ONE flat fp32 buffer (plus one flat gradient buffer), so the optimizer is a single fused kernel
and data-parallel training needs a single all-reduce per step.
"""
import re

import numpy as np
import torch

from . import ops

_DEVICE = None


def device():
    """The HIP device of this process: cuda:LOCAL_RANK (one process per GPU)."""
    global _DEVICE
    if _DEVICE is None:
        import os
        if not torch.cuda.is_available():
            from ._lib import GennetHipError
            raise GennetHipError('gennet_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU execution path')
        idx = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(idx)
        _DEVICE = torch.device('cuda', idx)
        # conv arithmetic (ops.set_conv_math): transform-domain fp32 for the unit-stride 5-tap layers by default; GENNET_CONV_MATH=fp32 keeps every
        # launch on the direct kernels, =bf16x3 is the opt-in operand-split experiment
        ops.set_conv_math(ops.default_conv_math(), float(os.environ.get('GENNET_CONV_WS_GB', '7')), _DEVICE)
    return _DEVICE


def to_device(a, dtype=torch.float32):
    if isinstance(a, torch.Tensor):
        t = a.to(device=device(), dtype=dtype)
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(a)), dtype=dtype).to(device())
    return t.contiguous()


_INIT_RNG = np.random.RandomState(12345)


def set_init_seed(seed):
    """Seed of the numpy RNG that draws initial weights (glorot_uniform)."""
    global _INIT_RNG
    _INIT_RNG = np.random.RandomState(seed)


def glorot_uniform(shape):
    """keras VarianceScaling(scale=1, mode='fan_avg', distribution='uniform')."""
    if len(shape) == 2:
        fi, fo = shape
    else:
        rec = int(np.prod(shape[:-2]))
        fi, fo = rec * shape[-2], rec * shape[-1]
    lim = np.sqrt(6.0 / (fi + fo))
    return _INIT_RNG.uniform(-lim, lim, size=shape).astype(np.float32)


# --------------------------------------------------------------------------------------------------------------
# parameters and flat groups
# --------------------------------------------------------------------------------------------------------------
class Param(object):
    """A weight tensor.  The initial value stays on the host until the first device access, so graphs can be built,
    planned and inspected (summary, count_params, get_weights) on a machine without a GPU; all arithmetic needs one."""

    def __init__(self, name, value, trainable=True):
        self.name = name
        self.shape = tuple(value.shape)
        self._host = np.ascontiguousarray(value, np.float32)
        self._data = None
        self.grad = None
        self.trainable = trainable
        self.group = None
        self.offset = 0

    @property
    def data(self):
        if self._data is None:
            self._data = to_device(self._host)
            self._host = None
        return self._data

    @data.setter
    def data(self, t):
        self._data = t
        self._host = None

    def numpy(self):
        return self._host.copy() if self._data is None else self._data.detach().cpu().numpy().copy()

    def assign(self, w):
        w = np.ascontiguousarray(w, np.float32)
        assert tuple(w.shape) == self.shape, '%s: %s vs %s' % (self.name, w.shape, self.shape)
        if self._data is None:
            self._host = w
        else:
            self._data.copy_(to_device(w))

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1


class ParamGroup(object):
    """Contiguous storage for a set of parameters: data, grad; every parameter starts on a 256-byte boundary."""
    ALIGN = 64

    def __init__(self, params):
        off = 0
        for p in params:
            p.offset = off
            off += -(-p.size // self.ALIGN) * self.ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device())
        self.grad = torch.zeros(off, dtype=torch.float32, device=device())
        for p in params:
            self.data[p.offset:p.offset + p.size].copy_(p.data.reshape(-1))
            p.data = self.data[p.offset:p.offset + p.size].view(p.shape)
            p.grad = self.grad[p.offset:p.offset + p.size].view(p.shape)
            p.group = self
        self.params = list(params)


def group_params(params):
    fresh = [p for p in params if p.group is None]
    if fresh:
        ParamGroup(fresh)


def segments(params):
    """Maximal runs of adjacent parameters inside their groups: [(group, start, stop)]."""
    spans = sorted(((id(p.group), p.offset, p) for p in params), key=lambda t: (t[0], t[1]))
    segs = []
    for _, off, p in spans:
        end = off + -(-p.size // ParamGroup.ALIGN) * ParamGroup.ALIGN
        if segs and segs[-1][0] is p.group and segs[-1][2] == off:
            segs[-1][2] = end
        else:
            segs.append([p.group, off, end])
    return [tuple(s) for s in segs]


# --------------------------------------------------------------------------------------------------------------
# run context
# --------------------------------------------------------------------------------------------------------------
class PhiloxStream(object):
    """Counter allocator for the device RNG: every consumer takes a disjoint counter range of one (seed) stream."""

    def __init__(self, seed=0):
        self.seed = int(seed)
        self.offset = 0

    def take(self, n_values):
        off = self.offset
        self.offset += (int(n_values) + 3) // 4
        return self.seed, off

    def take_rows(self, row_len, blocks, global_rows):
        """Counter ranges for (a rank's part of) a GLOBAL tensor of global_rows rows of row_len values each, the batch axis leading (SURVEY 8e:
        "latent z and noise likewise" -- N ranks x B / N rows draw exactly what one process draws for B rows, same seed everywhere).
        blocks = [(global_row_start, n_rows), ...]: the global rows this caller holds, in its local row order.  Returns (seed, [counter offset
        of every block]); the stream advances by the counters of the WHOLE global tensor on every rank, whatever part a rank takes.
        One block (0, global_rows) is take(global_rows * row_len)."""
        row_len, global_rows = int(row_len), int(global_rows)
        base = self.offset
        self.offset += (global_rows * row_len + 3) // 4
        offs = []
        for g0, n in blocks:
            if (int(g0) * row_len) % 4:
                raise ValueError('PhiloxStream.take_rows: a block starting at global row %d of %d-value rows does not start on a Philox counter '
                                 '(4 values each)' % (g0, row_len))
            offs.append(base + int(g0) * row_len // 4)
        return self.seed, offs


_RNG = PhiloxStream(0)


# --------------------------------------------------------------------------------------------------------------
# hipGraph capture of a whole train step
# --------------------------------------------------------------------------------------------------------------
_CAPTURE = None


def capturing():
    """The StepGraph being captured, or None."""
    return _CAPTURE


class StepGraph(object):
    """One train step captured as a hipGraph on the launch stream and replayed (the loops of bbhMahoGANy.py at the script's own batch
    size 8 are launch-bound: ~250 launches of a few microseconds each per iteration).  The library never allocates or synchronises, torch's
    allocator serves the step's temporaries from the graph's private pool, so the captured launches are exactly the eager ones.  What changes
    from step to step cannot be a by-value kernel argument (it would be frozen): it lives in ONE small parameter block in device memory
    that the host refreshes (pinned copy, stream-ordered) before every replay --
      * the Philox stream position: every draw inside the graph adds the block's rng base (gn_set_rng_base) = stream position at replay
        - stream position at capture, so a replay draws exactly what the eager step would have drawn at that point of the stream;
      * Adam's lr_t, BatchNormalization's local_step, the CNN loop's noise sigma: slot(fmt, provider) -- the provider runs once per replay,
        advances the host-side state (iteration counters) and returns the value.
    Batch indices and other per-step inputs are static device tensors the caller overwrites before replay()."""
    SLOTS = 64

    def __init__(self):
        import struct
        self._struct = struct
        self.host = torch.zeros(self.SLOTS * 8, dtype=torch.uint8).pin_memory()
        self.hostbuf = self.host.numpy()
        self.dev = torch.zeros(self.SLOTS * 8, dtype=torch.uint8, device=device())
        self.providers = []
        self.graph = None
        self.rng_start = 0
        self.rng_taken = 0
        self.copied = torch.cuda.Event()
        self._copied_once = False
        self.outputs = None
        self.scratch = []                  # every ops.workspace() buffer handed out during capture: the graph holds their addresses

    def slot(self, fmt, provider):
        """Reserve one 8-byte slot holding a scalar of struct format `fmt` ('f', 'i', 'Q'); provider() -> value is called before every replay."""
        i = len(self.providers)
        if i >= self.SLOTS:
            raise RuntimeError('StepGraph: more than %d step-varying scalars' % self.SLOTS)
        self.providers.append((i, fmt, provider))
        return ops.DevScalar(self.dev.data_ptr() + 8 * i)

    def capture(self, fn):
        """Record fn() (kernel launches on the capture stream; nothing executes now).  Host-side counters that fn advances per step are
        advanced by the providers at replay time instead; the Philox stream is rewound to where it stood."""
        global _CAPTURE
        if _CAPTURE is not None:
            raise RuntimeError('StepGraph.capture: already capturing')
        rng = device_rng()
        self.rng_start = rng.offset

        def keep(buf):
            if not any(b is buf for b in self.scratch):
                self.scratch.append(buf)

        def rng_base():
            base = device_rng().offset - self.rng_start
            device_rng().offset += self.rng_taken
            return base
        base = self.slot('Q', rng_base)
        self.graph = torch.cuda.CUDAGraph()
        _CAPTURE = self
        ops.set_rng_base(base.ptr)
        ops._ws_on_use = keep
        prof = ops.prof_enabled()          # the launch-stream HIP events of the profiling hooks must not be recorded into the graph
        if prof:
            ops.prof_enable(False)
        try:
            with torch.cuda.graph(self.graph):
                self.outputs = fn()
        finally:
            ops.set_rng_base(None)
            ops._ws_on_use = None
            _CAPTURE = None
            if prof:
                ops.prof_enable(True)
            self.rng_taken = rng.offset - self.rng_start
            rng.offset = self.rng_start    # also when fn raised: the draws of a failed capture never executed
        return self.outputs

    def wait_inputs_consumed(self):
        """Block until the previous replay's host -> device copies are done, so pinned staging buffers (this block, the caller's batch
        indices) may be overwritten.  The host stays at most one step ahead of the device."""
        if self._copied_once:
            self.copied.synchronize()

    def replay(self):
        """Refresh the parameter block, then launch the graph on the current stream.  The caller has already waited
        (wait_inputs_consumed) before touching its own staging buffers and enqueued their copies on the current stream."""
        for i, fmt, provider in self.providers:
            self._struct.pack_into('<' + fmt, self.hostbuf, 8 * i, provider())
        self.dev.copy_(self.host, non_blocking=True)
        self.copied.record()
        self._copied_once = True
        self.graph.replay()
        return self.outputs


def set_device_seed(seed):
    global _RNG
    _RNG = PhiloxStream(seed)


def device_rng():
    return _RNG


class RunContext(object):
    def __init__(self, training, dp=None, dropout_masks=None, train_params=None, site=None, row_map=None):
        self.training = training
        # which rows of the GLOBAL batch this step's local rows are: [(global_row_start, n_rows), ...] in local order, and the global row count.
        # Every per-row random draw inside the step (dropout masks) takes the counters of its global rows (PhiloxStream.take_rows), so N ranks
        # draw what one process would.  None: the rows are the whole batch.
        self.row_map = row_map
        self.site = site                   # name of the model whose train_on_batch runs (BatchNormalization keeps per-site state)
        self.capture = None                # testing hook: dict that receives {layer name: output tensor} of every executed node
        self.dp = dp
        self.dropout_masks = dropout_masks or {}
        self.train_ids = None if train_params is None else set(id(p) for p in train_params)
        self.tape = {}
        self.epi = {}                      # node index -> (y, act, act_param, mask, rate): epilogue a consumer's dgrad can differentiate
        self.pre_applied = set()           # producers whose activation gradient has already been applied to their dy
        self.skip = set()                  # inference phase: BatchNormalization nodes already folded into the producing conv
        self.bn_sums = {}                  # training phase: BN node index -> fp64 (sum x, sum x^2) produced by the conv epilogue

    def wants_grad(self, layer):
        if self.train_ids is None:
            return False
        return any(id(p) in self.train_ids for p in layer.trainable_params())


# --------------------------------------------------------------------------------------------------------------
# layers: base class + symbolic tensors
# --------------------------------------------------------------------------------------------------------------
class SymTensor(object):
    def __init__(self, shape, layer=None, inbound=(), name=None):
        self.shape = tuple(shape)          # without the batch axis
        self.layer = layer
        self.inbound = tuple(inbound)
        self.name = name                   # only graph inputs carry a name (keras InputLayer)


_NAME_COUNTS = {}


def to_snake_case(name):
    """keras.engine.base_layer._to_snake_case: Conv1D -> conv1d, BatchNormalization -> batch_normalization,
    LeakyReLU -> leaky_re_lu, UpSampling1D -> up_sampling1d (the names keras writes into .h5 files)."""
    intermediate = re.sub('(.)([A-Z][a-z0-9]+)', r'\1_\2', name)
    insecure = re.sub('([a-z])([A-Z])', r'\1_\2', intermediate).lower()
    return 'private' + insecure if insecure[0] == '_' else insecure


def _unique_name(base):
    _NAME_COUNTS[base] = _NAME_COUNTS.get(base, 0) + 1
    return '%s_%d' % (base, _NAME_COUNTS[base])


def Input(shape=None, name=None, batch_shape=None, dtype=None):
    if shape is None and batch_shape is not None:
        shape = tuple(batch_shape[1:])
    return SymTensor(shape, name=name or _unique_name('input'))


class Layer(object):
    """Base layer.  Subclasses implement build(input_shape), compute_output_shape(input_shape),
    forward(ctx, node, x) and backward(ctx, node, dy, need_dx, need_dw)."""

    def __init__(self, input_shape=None, name=None, trainable=True, **kwargs):
        if name is None:
            name = _unique_name(to_snake_case(self.__class__.__name__))
        self.name = name
        self.trainable = trainable
        self.built = False
        self.input_shape_arg = tuple(input_shape) if input_shape is not None else None
        self.params = []
        self.buffers = []

    # -- keras-style weight bookkeeping
    def add_weight(self, name, value, trainable=True):
        p = Param('%s/%s' % (self.name, name), value, trainable)
        (self.params if trainable else self.buffers).append(p)
        return p

    def trainable_params(self):
        return self.params

    @property
    def trainable_weights(self):
        return list(self.params) if self.trainable else []

    @property
    def non_trainable_weights(self):
        return list(self.buffers) + ([] if self.trainable else list(self.params))

    @property
    def weights(self):
        return list(self.params) + list(self.buffers)

    def get_weights(self):
        return [p.numpy() for p in self.weights]

    def set_weights(self, ws):
        assert len(ws) == len(self.weights), '%s expects %d arrays' % (self.name, len(self.weights))
        for p, w in zip(self.weights, ws):
            p.assign(w)

    def count_params(self):
        return sum(p.size for p in self.weights)

    # -- shape protocol
    def build(self, input_shape):
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape

    # A subclass that defines keras' `call(self, x)` (and not this engine's forward/backward) is a USER-DEFINED layer in keras'
    # conventions (bbhMahoGANy.py:164-188): its build / compute_output_shape see shapes WITH the batch axis, and its call is traced
    # once on a symbolic operand and lowered to a HIP kernel (gennet_amd/keras/backend.py).
    def _is_user_layer(self):
        return callable(getattr(type(self), 'call', None)) and type(self).forward is Layer.forward

    def _ensure_built(self, input_shape):
        if not self.built:
            self.build((None,) + tuple(input_shape) if self._is_user_layer() else tuple(input_shape))
            self.built = True

    def _shape_after(self, input_shape):
        """Output shape without the batch axis for an input shape without the batch axis."""
        if self._is_user_layer():
            out = tuple(self.compute_output_shape((None,) + tuple(input_shape)))[1:]
            self._lower(tuple(input_shape), out)
            return out
        return tuple(self.compute_output_shape(tuple(input_shape)))

    def _lower(self, input_shape, out_shape):
        if getattr(self, '_lowered', None) is None or self._lowered_for != input_shape:
            from .keras import backend as K
            self._lowered = K.lower_layer_call(self, input_shape)
            self._lowered_for = input_shape
            if tuple(out_shape) != (input_shape[0], 2, 1):
                raise ValueError('%s.compute_output_shape gives %r; its call produces %r' % (type(self).__name__, (None,) + tuple(out_shape), (None, input_shape[0], 2, 1)))
        return self._lowered

    # -- functional API
    def __call__(self, x):
        self._ensure_built(x.shape)
        return SymTensor(self._shape_after(x.shape), self, (x,))

    # -- execution
    fusable_act = False      # can take an activation epilogue
    fusable_drop = False     # can take a dropout epilogue
    act_spec = None          # (kind, param) if this layer IS an activation
    drop_rate = None         # rate if this layer IS a dropout

    def forward(self, ctx, node, x):
        if not self._is_user_layer():
            raise NotImplementedError
        low = self._lower(tuple(x.shape[1:]), tuple(node.out_shape))
        b0, b1 = low.betas()
        return ops.affine_stack_fwd(x.contiguous(), low.a0, b0, low.a1, b1)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        if not self._is_user_layer():
            raise NotImplementedError
        return ops.affine_stack_bwd(dy.contiguous(), self._lowered.a0, self._lowered.a1)


class Node(object):
    __slots__ = ('layer', 'inbound', 'fused_act', 'fused_drop', 'absorbed', 'index', 'out_shape', 'owners', 'fuse_prev', 'infer_bn', 'fold_up', 'lazy_bn')

    def __init__(self, layer, inbound, out_shape, owners=()):
        self.layer = layer
        self.owners = tuple(owners)        # nested models this node was spliced in from (their .trainable also gates it)
        self.inbound = list(inbound)       # indices: >= 0 node, < 0 graph input (-1 - k)
        self.fused_act = None              # (kind, param) applied in this node's epilogue
        self.fused_drop = None             # (rate, dropout layer) applied in this node's epilogue
        self.absorbed = False              # this node's work happens inside its producer
        self.index = -1
        self.out_shape = out_shape
        self.fuse_prev = -1                # producer node whose [activation -> dropout] backward this node's dgrad absorbs


# --------------------------------------------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------------------------------------------
class Adam(object):
    """keras.optimizers.Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0.0, **kwargs):
        # float32 variables in Keras (K.variable): the values that take part -- and that Keras writes into training_config -- are the
        # float32-rounded ones (the reference's own d_model.hdf5 records beta_2 = 0.9990000128746033); epsilon stays a python float
        self.lr, self.beta_1, self.beta_2 = (float(np.float32(x)) for x in (lr, beta_1, beta_2))
        self.epsilon = 1e-7 if epsilon is None else float(epsilon)
        if decay:
            raise NotImplementedError('Adam(decay != 0) is not on the BBH hot path')
        self.iterations = 0
        self.state = None                  # [(group, start, stop, m, v)]

    def bind(self, params):
        self.state = []
        for grp, a, b in segments(params):
            n = b - a
            self.state.append((grp, a, b, torch.zeros(n, dtype=torch.float32, device=device()), torch.zeros(n, dtype=torch.float32, device=device())))

    def _next_lr_t(self):
        self.iterations += 1
        t = self.iterations
        return self.lr * np.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def step(self):
        cap = capturing()
        # inside a captured step graph the iteration count advances once per REPLAY and lr_t reaches the kernel through device memory
        lr_t = self._next_lr_t() if cap is None else cap.slot('f', self._next_lr_t)
        for grp, a, b, m, v in self.state:
            ops.adam_step(grp.data[a:b], grp.grad[a:b], m, v, lr_t, self.beta_1, self.beta_2, self.epsilon)

    def _moment_views(self, p):
        for grp, a, b, m, v in self.state:
            if grp is p.group and a <= p.offset and p.offset + p.size <= b:
                return m[p.offset - a:p.offset - a + p.size], v[p.offset - a:p.offset - a + p.size]
        raise KeyError('%s is not trained by this optimizer' % p.name)

    def param_moments(self, params):
        """[(m, v)] numpy arrays in the parameters' shapes (keras optimizer_weights order is decided by the caller)."""
        out = []
        for p in params:
            m, v = self._moment_views(p)
            out.append((m.cpu().numpy().reshape(p.shape), v.cpu().numpy().reshape(p.shape)))
        return out

    def set_param_moments(self, params, moments):
        for p, (mn, vn) in zip(params, moments):
            m, v = self._moment_views(p)
            m.copy_(to_device(np.asarray(mn, np.float32).reshape(-1))); v.copy_(to_device(np.asarray(vn, np.float32).reshape(-1)))

    def get_state(self):
        return {'iterations': self.iterations, 'mv': [(m.cpu().numpy(), v.cpu().numpy()) for _, _, _, m, v in (self.state or [])],
                'config': (self.lr, self.beta_1, self.beta_2, self.epsilon)}

    def set_state(self, st):
        self.iterations = st['iterations']
        for (_, _, _, m, v), (mn, vn) in zip(self.state, st['mv']):
            m.copy_(to_device(mn)); v.copy_(to_device(vn))


def _get_optimizer(opt):
    if isinstance(opt, str):
        if opt.lower() == 'adam':
            return Adam()
        raise NotImplementedError('optimizer %r' % opt)
    return opt


LOSSES = ('binary_crossentropy', 'mean_squared_error')


# --------------------------------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------------------------------
class Model(Layer):
    """Functional model: Model(inputs=Input(...), outputs=[t1, t2], name=...).  Also the base of Sequential."""

    def __init__(self, inputs=None, outputs=None, name=None, **kwargs):
        Layer.__init__(self, name=name)
        self.nodes = []
        self.input_shapes = []
        self.output_ids = []
        self._planned = False
        self.optimizer = None
        self.loss = None
        self.metrics = []
        self._train_params = None
        self.data_parallel = None
        if inputs is not None:
            self._from_symbolic(inputs, outputs)

    # -- graph construction
    def _from_symbolic(self, inputs, outputs):
        ins = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        outs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        self.input_shapes = [t.shape for t in ins]
        self._sym_inputs, self._sym_outputs = ins, outs        # kept for model_config / keras layer order (keras_io.py)
        index = {}
        for k, t in enumerate(ins):
            index[id(t)] = -1 - k

        def visit(t):
            if id(t) in index:
                return index[id(t)]
            if t.layer is None:
                raise ValueError('graph reaches an Input that is not listed in inputs=')
            inb = [visit(s) for s in t.inbound]
            idx = self._append(t.layer, inb, t.shape)
            index[id(t)] = idx
            return idx

        self.output_ids = [visit(t) for t in outs]
        self.built = True

    def _append(self, layer, inbound, out_shape, owners=()):
        """Append a leaf layer, or splice in a nested model's nodes (weights are shared with the nested model)."""
        if isinstance(layer, Model):
            assert len(layer.input_shapes) == 1 and len(layer.output_ids) == 1 and len(inbound) == 1, 'nested models must be 1-in / 1-out'
            remap = {}
            for n in layer.nodes:
                inb = [inbound[0] if i < 0 else remap[i] for i in n.inbound]
                remap[n.index] = self._append(n.layer, inb, n.out_shape, tuple(owners) + (layer,) + n.owners)
            return remap[layer.output_ids[0]]
        node = Node(layer, inbound, out_shape, owners)
        node.index = len(self.nodes)
        self.nodes.append(node)
        self._planned = False
        return node.index

    @property
    def layers(self):
        seen, out = set(), []
        for n in self.nodes:
            if id(n.layer) not in seen:
                seen.add(id(n.layer))
                out.append(n.layer)
        return out

    @property
    def output_shape(self):
        shapes = [(None,) + tuple(self.nodes[i].out_shape) for i in self.output_ids]
        return shapes[0] if len(shapes) == 1 else shapes

    def compute_output_shape(self, input_shape):
        return self.nodes[self.output_ids[0]].out_shape

    # -- trainability (keras: model.trainable = False freezes; what counts is the value at compile time)
    def trainable_params(self):
        out = []
        for l in self.layers:
            out.extend(l.trainable_params())
        return out

    @property
    def trainable_weights(self):
        if not self.trainable:
            return []
        out = []
        for l in self.layers:
            out.extend(l.trainable_weights)
        return out

    @property
    def non_trainable_weights(self):
        out = []
        for l in self.layers:
            out.extend(l.non_trainable_weights if self.trainable else l.weights)
        return out

    @property
    def weights(self):
        out = []
        for l in self.layers:
            out.extend(l.weights)
        return out

    def get_weights(self):
        return [w for l in self.layers for w in l.get_weights()]

    def set_weights(self, ws):
        ws = list(ws)
        for l in self.layers:
            n = len(l.weights)
            l.set_weights(ws[:n])
            ws = ws[n:]
        assert not ws, 'too many weight arrays'

    # -- planning: consumers, peephole fusions
    def _plan(self):
        for n in self.nodes:
            n.fused_act, n.fused_drop, n.absorbed = None, None, False
        consumers = {n.index: [] for n in self.nodes}
        for n in self.nodes:
            for i in n.inbound:
                if i >= 0:
                    consumers[i].append(n.index)
        outs = set(self.output_ids)

        def sole_consumer(i):
            if i in outs or len(consumers[i]) != 1:
                return None
            return self.nodes[consumers[i][0]]

        for n in self.nodes:
            if n.absorbed:
                continue
            tail = n
            if n.layer.fusable_act:
                c = sole_consumer(tail.index)
                if c is not None and c.layer.act_spec is not None:
                    n.fused_act = c.layer.act_spec
                    c.absorbed = True
                    tail = c
            if n.layer.fusable_drop:
                c = sole_consumer(tail.index)
                if c is not None and c.layer.drop_rate is not None:
                    n.fused_drop = (c.layer.drop_rate, c.layer)
                    c.absorbed = True
        # UpSampling1D(2) whose only consumer is a 5-tap 'same' Conv1D: the pair runs as a 3-tap conv on the un-upsampled tensor with
        # folded weights (layers.Conv1D._geometry); the upsample node passes its input through and nothing is materialised
        for n in self.nodes:
            n.fold_up = None
        for n in self.nodes:
            if n.absorbed or not getattr(n.layer, 'is_upsample2', False) or len(n.inbound) != 1 or n.inbound[0] < 0:
                continue
            c = sole_consumer(n.index)
            if c is not None and not c.absorbed and hasattr(c.layer, 'can_fold_upsample') and c.layer.can_fold_upsample():
                c.fold_up = n
                n.absorbed = True
        # inference-phase fusion: a BatchNormalization that is the sole consumer of a linear conv folds into that conv's weights
        # (predict only: the training phase normalises with batch statistics)
        for n in self.nodes:
            n.infer_bn = None
            if n.absorbed or not getattr(n.layer, 'can_fold_bn', False) or n.fused_act is not None or n.fused_drop is not None:
                continue
            c = sole_consumer(n.index)
            if c is not None and getattr(c.layer, 'is_batchnorm', False):
                n.infer_bn = c
        # backward fusion: node n's data gradient can carry the producer p's activation/dropout derivative in its epilogue when p's
        # output reaches n through shape-only nodes (absorbed activations, Flatten, Reshape) and nothing else consumes it
        for n in self.nodes:
            n.fuse_prev = -1
            if n.absorbed or len(n.inbound) != 1 or not getattr(n.layer, 'can_absorb_prev_act_bwd', False):
                continue
            cur, ok = n.inbound[0], True
            while cur >= 0 and (self.nodes[cur].absorbed or getattr(self.nodes[cur].layer, 'shape_only', False)):
                if len(consumers[cur]) != 1 or cur in outs:
                    ok = False
                    break
                cur = self.nodes[cur].inbound[0]
            if ok and cur >= 0 and len(consumers[cur]) == 1 and cur not in outs and getattr(self.nodes[cur].layer, 'offers_act_bwd', False):
                n.fuse_prev = cur
        # a 1-filter stride-1 Conv1D whose input comes from a BatchNormalization through absorbed nodes only (the generator's output
        # conv): its data gradient is not materialised, the BatchNormalization's backward passes form it on the fly (ops.ConvGrad1)
        for n in self.nodes:
            n.lazy_bn = -1
            if n.absorbed or len(n.inbound) != 1 or n.fold_up is not None or not hasattr(n.layer, 'can_defer_dgrad'):
                continue
            cur = n.inbound[0]
            while cur >= 0 and self.nodes[cur].absorbed and len(consumers[cur]) == 1 and cur not in outs:
                cur = self.nodes[cur].inbound[0]
            if cur < 0 or len(consumers[cur]) != 1 or cur in outs or not getattr(self.nodes[cur].layer, 'is_batchnorm', False) or self.nodes[cur].absorbed:
                continue
            shp = self.nodes[cur].out_shape
            if len(shp) == 2 and n.layer.can_defer_dgrad(shp[1]):
                n.lazy_bn = cur
        self._planned = True

    # -- execution
    def _forward(self, inputs, ctx):
        if not self._planned:
            self._plan()
        vals = {}
        for k, x in enumerate(inputs):
            vals[-1 - k] = x
        for n in self.nodes:
            xs = [vals[i] for i in n.inbound]
            if n.absorbed:
                vals[n.index] = xs[0]
                continue
            vals[n.index] = n.layer.forward(ctx, n, xs[0] if len(xs) == 1 else xs)
            if ctx.capture is not None:
                ctx.capture[n.layer.name] = vals[n.index]
        return [vals[i] for i in self.output_ids]

    def _backward(self, out_grads, ctx):
        grads = {}
        for i, g in zip(self.output_ids, out_grads):
            grads[i] = g
        for n in reversed(self.nodes):
            dy = grads.pop(n.index, None)
            if dy is None:
                continue
            if n.absorbed:
                dx = dy
            else:
                need_dx = any(i >= 0 for i in n.inbound)
                need_dw = ctx.wants_grad(n.layer)
                if not need_dx and not need_dw:
                    continue
                prev = ctx.epi.get(n.fuse_prev) if (n.fuse_prev >= 0 and need_dx) else None
                dx = n.layer.backward(ctx, n, dy, need_dx, need_dw, prev) if prev is not None else n.layer.backward(ctx, n, dy, need_dx, need_dw)
            if dx is None:
                continue
            for i in n.inbound:
                if i < 0:
                    continue
                if i in grads:
                    grads[i] = ops.axpy(grads[i], dx, 1.0)
                else:
                    grads[i] = dx

    # Model used as a layer inside another graph is expanded by _append, so forward/backward are never called on it.

    # -- keras training surface
    def compile(self, loss=None, optimizer=None, metrics=None, data_parallel=None, **kwargs):
        """Collects the trainable weights NOW (keras semantics): later changes of .trainable do not affect this model."""
        self.loss = loss
        n_out = len(self.output_ids)
        losses = list(loss) if isinstance(loss, (list, tuple)) else [loss] * n_out
        self._loss_scales = []
        for k, l in enumerate(losses):
            scale = 1.0
            if callable(l):                # e.g. the script's chisquare_Loss (bbhMahoGANy.py:146-162): traced once and lowered
                from .keras import backend as K
                losses[k], scale = K.lower_loss(l)
            elif l not in LOSSES:
                raise NotImplementedError('loss %r: only %s (or a callable squared-error loss) run on the HIP path' % (l, LOSSES))
            self._loss_scales.append(scale)
        self._losses = losses
        self.metrics = list(metrics or [])
        for m in self.metrics:
            if m not in ('accuracy', 'acc', 'binary_accuracy'):
                raise NotImplementedError('metric %r' % (m,))
        self.optimizer = _get_optimizer(optimizer)
        params, seen = [], set()
        if self.trainable:
            for n in self.nodes:
                if n.layer.trainable and all(o.trainable for o in n.owners):
                    for p in n.layer.trainable_params():
                        if id(p) not in seen:
                            seen.add(id(p)); params.append(p)
        self._train_params = params
        self._bound = False                # flat grouping + optimizer state are created at the first device step
        if data_parallel is not None:
            self.data_parallel = data_parallel
        return self

    def _ensure_bound(self):
        if not self._bound:
            group_params(self._train_params)
            self.optimizer.bind(self._train_params)
            self._bound = True
            pend = getattr(self, '_pending_optimizer_weights', None)
            if pend is not None:           # optimizer_weights of a loaded .h5: [iterations, m..., v..., (vhat stubs...)]
                self._pending_optimizer_weights = None
                order = self._keras_train_order()
                n = len(order)
                if len(pend) >= 1 + 2 * n:
                    self.optimizer.iterations = int(np.asarray(pend[0]).reshape(-1)[0])
                    self.optimizer.set_param_moments(order, list(zip(pend[1:1 + n], pend[1 + n:1 + 2 * n])))

    def _keras_train_order(self):
        """The compiled trainable weights in keras' model.trainable_weights order (the order of optimizer_weights in .h5 files)."""
        from . import keras_io
        ids = set(id(p) for p in self._train_params)
        order = [p for p in keras_io._tw(self) if id(p) in ids]
        seen = set(id(p) for p in order)
        return order + [p for p in self._train_params if id(p) not in seen]

    @property
    def metrics_names(self):
        names = ['loss']
        n_out = len(self.output_ids)
        if n_out > 1:
            names += ['out%d_loss' % k for k in range(n_out)]
        for k in range(n_out):
            if self.metrics:
                names.append('acc' if n_out == 1 else 'out%d_acc' % k)
        return names

    def _prep_inputs(self, x):
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        assert len(xs) == len(self.input_shapes), 'model expects %d input arrays' % len(self.input_shapes)
        out = []
        for a, shp in zip(xs, self.input_shapes):
            t = to_device(a)
            if tuple(t.shape[1:]) != tuple(shp):
                t = t.reshape((t.shape[0],) + tuple(shp))
            out.append(t)
        return out

    def _prep_targets(self, y, B):
        n_out = len(self.output_ids)
        ys = list(y) if (isinstance(y, (list, tuple)) and n_out > 1) else [y]
        assert len(ys) == n_out, 'model has %d outputs' % n_out
        return [to_device(a).reshape(B, 1) for a in ys]

    def train_on_batch(self, x, y, dropout_masks=None, capture=None, row_map=None):
        """One optimizer step.  Returns [loss, (per-output losses,) (accuracies)] as python floats, keras order.
        `dropout_masks` ({dropout layer name: uint8 keep mask}) is a testing hook that replaces the Philox draws; `capture` (a dict) is
        another: it receives {layer name: output tensor} of every executed layer (outputs include the fused activation / dropout).
        row_map (data parallelism): ([(global_row_start, n_rows), ...], global_rows) when the local rows are not the rank's contiguous slice
        [rank * B, (rank + 1) * B) of the global batch (the default) -- the discriminator batch of bbh.gan_train_step."""
        xs = self._prep_inputs(x)
        B = xs[0].shape[0]
        ys = self._prep_targets(y, B)
        stats = self.train_on_batch_device(xs, ys, dropout_masks, capture, row_map)
        return self.train_result(stats, B)

    def train_on_batch_device(self, xs, ys, dropout_masks=None, capture=None, row_map=None):
        """train_on_batch on device tensors (inputs as the graph takes them, targets (B, 1)) without the final device -> host read: returns
        the (n_outputs, 2) device tensor [summed loss term, metric hits] that train_result turns into keras' list.  No host synchronisation
        anywhere in it, so the whole step can be captured into a hipGraph (engine.StepGraph)."""
        if self.optimizer is None:
            raise RuntimeError('compile() the model before train_on_batch')
        self._ensure_bound()
        B = xs[0].shape[0]
        dp = self.data_parallel
        world = dp.world_size if dp is not None else 1
        masks = {k: to_device(v, torch.uint8) for k, v in (dropout_masks or {}).items()}
        if row_map is None and dp is not None and world > 1:
            row_map = ([(dp.rank * B, B)], B * world)
        ctx = RunContext(True, dp, masks, self._train_params, self.name, row_map)
        ctx.capture = capture
        outs = self._forward(xs, ctx)
        dps, stats = [], []
        for p, t, kind, scale in zip(outs, ys, self._losses, self._loss_scales):
            d, o = ops.loss(kind, p.reshape(B, 1), t, B * world)
            if scale != 1.0:
                ops.axpy(d, d.clone(), scale - 1.0)
            dps.append(d.reshape(p.shape)); stats.append(o)
        for grp, a, b in segments(self._train_params):
            grp.grad[a:b].zero_()
        self._backward(dps, ctx)
        stats = torch.stack(stats)
        if dp is not None:
            for grp, a, b in segments(self._train_params):
                dp.all_reduce_sum(grp.grad[a:b])
            dp.all_reduce_sum(stats)
        self.optimizer.step()
        return stats

    def train_result(self, stats, B):
        """[loss, (per-output losses,) (accuracies)] as python floats, keras order, from train_on_batch_device's statistics (one device -> host read)."""
        world = self.data_parallel.world_size if self.data_parallel is not None else 1
        st = stats.cpu().numpy().astype(np.float64)
        losses = [float(v) * sc for v, sc in zip(st[:, 0], self._loss_scales)]
        res = [float(sum(losses))]
        if len(losses) > 1:
            res += losses
        if self.metrics:
            res += [float(h) / (B * world) for h in st[:, 1]]
        return res

    def predict_device(self, x, batch_size=32):
        """predict() that keeps inputs and outputs in HBM (torch tensors): inference phase, chunks of batch_size."""
        xs = self._prep_inputs(x)
        B = xs[0].shape[0]
        if B == 0:                         # nothing to run: empty outputs of the right shapes (the kernels are never launched)
            outs = [torch.empty((0,) + tuple(self.nodes[i].out_shape), dtype=torch.float32, device=device()) for i in self.output_ids]
            return outs if len(outs) > 1 else outs[0]
        chunks = []
        for s in range(0, B, batch_size):
            ctx = RunContext(False)
            chunks.append(self._forward([t[s:s + batch_size] for t in xs], ctx))
        outs = [torch.cat([c[k] for c in chunks]) if len(chunks) > 1 else chunks[0][k] for k in range(len(self.output_ids))]
        return outs if len(outs) > 1 else outs[0]

    def predict(self, x, batch_size=32, verbose=0):
        out = self.predict_device(x, batch_size)
        if isinstance(out, list):
            return [o.cpu().numpy() for o in out]
        return out.cpu().numpy()

    def fit(self, x, y, batch_size=32, epochs=1, verbose=0, shuffle=True, **kwargs):
        """Minimal keras fit: epochs of shuffled mini-batches through train_on_batch; returns {'loss': [...]}."""
        x = np.asarray(x)
        n = x.shape[0]
        ys = list(y) if (isinstance(y, (list, tuple)) and len(self.output_ids) > 1) else [y]
        ys = [np.asarray(a) for a in ys]
        hist = {'loss': []}
        rng = np.random.RandomState(0)
        for ep in range(epochs):
            order = rng.permutation(n) if shuffle else np.arange(n)
            tot, cnt = 0.0, 0
            for s in range(0, n, batch_size):
                idx = order[s:s + batch_size]
                yy = [a[idx] for a in ys]
                r = self.train_on_batch(x[idx], yy if len(yy) > 1 else yy[0])
                tot += r[0] * len(idx); cnt += len(idx)
            hist['loss'].append(tot / max(cnt, 1))
            if verbose:
                print('Epoch %d/%d - loss: %.6f' % (ep + 1, epochs, hist['loss'][-1]))
        return hist

    def summary(self, print_fn=None):
        lines = ['_' * 65, '%-29s%-26s%s' % ('Layer (type)', 'Output Shape', 'Param #'), '=' * 65]
        tot = 0
        for n in self.nodes:
            cnt = n.layer.count_params()
            tot += cnt
            lines.append('%-29s%-26s%d' % ('%s (%s)' % (n.layer.name, n.layer.__class__.__name__), str((None,) + tuple(n.out_shape)), cnt))
        tr = sum(p.size for p in self.trainable_weights)
        lines += ['=' * 65, 'Total params: {:,}'.format(tot), 'Trainable params: {:,}'.format(tr), 'Non-trainable params: {:,}'.format(tot - tr), '_' * 65]
        s = '\n'.join(lines)
        (print_fn or print)(s)

    # -- persistence: Keras' HDF5 layout through gennet_amd/keras_io.py + h5lite.py (bbhMahoGANy.py:1135-1142, :1171-1173, :1372-1375)
    def save_weights(self, filepath, overwrite=True, writer=None):
        """`writer` (a hostio.BackgroundWriter, not a keras argument): the weights are copied to the host now, the file is written by the
        writer's thread."""
        import os
        from . import keras_io
        if os.path.exists(filepath) and not overwrite:
            raise IOError('%s exists' % filepath)
        keras_io.save_weights(self, filepath, writer)

    def load_weights(self, filepath, by_name=False):
        from . import keras_io
        if by_name:
            raise NotImplementedError('load_weights(by_name=True)')
        keras_io.load_weights(self, filepath)      # raises h5lite.H5Error on anything that is not an HDF5 file (never unpickles)

    def save(self, filepath, overwrite=True, include_optimizer=True, writer=None):
        import os
        from . import keras_io
        if os.path.exists(filepath) and not overwrite:
            raise IOError('%s exists' % filepath)
        keras_io.save_model(self, filepath, include_optimizer, writer)

    def get_config(self):
        from . import keras_io
        return keras_io.model_config(self)['config']

    def to_json(self):
        import json
        from . import keras_io
        return json.dumps(keras_io.model_config(self))


class Sequential(Model):
    def __init__(self, layers=None, name=None):
        Model.__init__(self, name=name)
        self._top = []                     # what was add()-ed, nested models unflattened: keras' `model.layers`
        for l in (layers or []):
            self.add(l)

    def add(self, layer):
        if not self.nodes:
            shp = layer.input_shape_arg if not isinstance(layer, Model) else (layer.input_shapes[0] if layer.input_shapes else None)
            if shp is None:
                raise ValueError('the first layer of a Sequential needs input_shape=')
            self.input_shapes = [tuple(shp)]
            prev, prev_shape = -1, tuple(shp)
        else:
            prev = self.output_ids[0]
            prev_shape = self.nodes[prev].out_shape
        if isinstance(layer, Model):
            out_shape = layer.nodes[layer.output_ids[0]].out_shape
        else:
            layer._ensure_built(prev_shape)
            out_shape = layer._shape_after(prev_shape)
        idx = self._append(layer, [prev], out_shape)
        self._top.append(layer)
        self.output_ids = [idx]
        self.built = True
        return self


def load_model(filepath, custom_objects=None, compile=True):
    """keras.models.load_model: rebuilds the model from the file's model_config, loads model_weights, and (compile=True)
    re-creates the optimizer with its iteration count and Adam moments.  Custom layers (the script's MyLayer,
    bbhMahoGANy.py:164-188, needs its constant) come through custom_objects={'MyLayer': layer_instance | class | factory(config)}."""
    from . import keras_io
    return keras_io.load_model(filepath, custom_objects, compile)


def model_from_json(text, custom_objects=None):
    import json
    from . import keras_io
    return keras_io.model_from_config(json.loads(text), custom_objects)
