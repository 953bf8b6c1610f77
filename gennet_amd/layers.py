"""Keras-style layers used by BBH_version/bbhMahoGANy.py, executing on the HIP kernel library (gennet_amd.ops).

Layer list and defaults follow bbhMahoGANy.py:33-40 and SURVEY Appendix B: channels_last, glorot_uniform kernels, zero
biases, BatchNormalization(axis=-1, epsilon=1e-3, gamma=1, beta=0, moving_mean=0, moving_variance=1).
Layers that the reference imports but never places on the hot path are not provided; asking for an unsupported
configuration raises instead of silently running something else.
"""
import numpy as np
import torch

from . import ops
from .engine import Layer, Model, capturing, device, device_rng, glorot_uniform

import os as _os
_NO_DROPGEN = bool(_os.environ.get('GN_NO_DROPGEN'))      # A/B switch: separate dropout-mask kernel instead of drawing it inside bn_apply
_NO_LAZYGRAD = bool(_os.environ.get('GN_NO_LAZYGRAD'))    # A/B switch: materialise the 1-filter conv's data gradient in front of a BatchNormalization
_NO_UPFOLD = bool(_os.environ.get('GN_NO_UPFOLD'))        # A/B switch: materialise UpSampling1D instead of folding it into the conv
_NO_CONVSTATS = bool(_os.environ.get('GN_NO_CONVSTATS'))  # A/B switch: separate BatchNorm statistics pass instead of the conv epilogue

_ACT_NAMES = {'relu': ('relu', 0.0), 'tanh': ('tanh', 0.0), 'sigmoid': ('sigmoid', 0.0), 'linear': ('linear', 0.0), None: ('linear', 0.0)}


def _check_init(kernel_initializer):
    if kernel_initializer not in (None, 'glorot_uniform'):
        raise NotImplementedError('kernel_initializer %r (only glorot_uniform is used on the hot path)' % (kernel_initializer,))


def _conv_fwd(node, ctx, x, w, b, stride, pl, Lout, act, out_shape_for_mask):
    """conv + activation epilogue, plus the following Dropout when the planner fused one (training phase only)."""
    if node.fused_drop is not None and ctx.training and node.fused_drop[0] > 0.0:
        rate, drop_layer = node.fused_drop
        mask = drop_layer.make_mask(ctx, out_shape_for_mask)
        return ops.conv1d_fwd_dropout(x, w, b, mask, stride, pl, Lout, act[0], act[1], rate), mask, rate
    return ops.conv1d_fwd(x, w, b, stride, pl, Lout, act[0], act[1]), None, 0.0


def _conv_bwd_epilogue(dy, y, act, mask, rate, ctx=None, node=None):
    """gradient through [activation -> dropout] expressed through the layer output, in place, one pass -- unless the consumer's
    data-gradient kernel already applied it in its epilogue (engine backward fusion)."""
    if ctx is not None and node.index in ctx.pre_applied:
        return dy
    if mask is not None:
        return ops.act_dropout_bwd(dy, y, mask, act[0], act[1], rate, inplace=True)
    if act[0] != 'linear':
        return ops.act_bwd(dy, y, act[0], act[1], inplace=True)
    return dy


class Dense(Layer):
    """bbhMahoGANy.py:234 (100 -> 256*n_pix/2, MFMA GEMM), :377,:399,:494 (flatten -> 1 heads, streaming dot product)."""
    fusable_act = True
    offers_act_bwd = True

    @property
    def can_absorb_prev_act_bwd(self):
        return self.units <= 4

    def __init__(self, units, activation=None, kernel_initializer='glorot_uniform', use_bias=True, **kw):
        Layer.__init__(self, **kw)
        _check_init(kernel_initializer)
        if not use_bias:
            raise NotImplementedError('Dense(use_bias=False)')
        self.units = int(units)
        self.activation = _ACT_NAMES[activation]

    def build(self, input_shape):
        assert len(input_shape) == 1, 'Dense expects (batch, features); Flatten first'
        self.kernel = self.add_weight('kernel', glorot_uniform((input_shape[0], self.units)))
        self.bias = self.add_weight('bias', np.zeros(self.units, np.float32))

    def compute_output_shape(self, input_shape):
        return (self.units,)

    def _act(self, node):
        a = node.fused_act or self.activation
        if node.fused_act is not None and self.activation[0] != 'linear':
            raise NotImplementedError('Dense(activation=...) followed by another activation layer')
        return a

    def forward(self, ctx, node, x):
        a = self._act(node)
        y = ops.dense_fwd(x, self.kernel.data, self.bias.data, a[0], a[1])
        ctx.tape[node.index] = (x, y, a)
        if ctx.training and a[0] != 'linear':
            ctx.epi[node.index] = (y, a[0], a[1], None, 0.0)
        return y

    def backward(self, ctx, node, dy, need_dx, need_dw, prev=None):
        x, y, a = ctx.tape.pop(node.index)
        dy = _conv_bwd_epilogue(dy.contiguous(), y, a, None, 0.0, ctx, node)
        if prev is not None:
            ctx.pre_applied.add(node.fuse_prev)
        if need_dw:
            dx, _, _ = ops.dense_bwd(x, self.kernel.data, dy, need_dx, self.kernel.grad, self.bias.grad, prev=prev)
            return dx
        # frozen layer: data gradient only (scratch weight-gradient buffers are not needed on the small-output path)
        dx, _, _ = ops.dense_bwd(x, self.kernel.data, dy, True, prev=prev)
        return dx


class Conv1D(Layer):
    """bbhMahoGANy.py:250-292, :362-394."""
    fusable_act = True
    fusable_drop = True
    offers_act_bwd = True
    can_absorb_prev_act_bwd = True

    def __init__(self, filters, kernel_size, strides=1, padding='valid', activation=None, kernel_initializer='glorot_uniform', use_bias=True, **kw):
        Layer.__init__(self, **kw)
        _check_init(kernel_initializer)
        if not use_bias:
            raise NotImplementedError('Conv1D(use_bias=False)')
        self.filters = int(filters)
        self.k = int(kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size)
        self.stride = int(strides[0] if isinstance(strides, (tuple, list)) else strides)
        if padding not in ('same', 'valid'):
            raise NotImplementedError('padding %r' % (padding,))
        self.padding = padding
        self.activation = _ACT_NAMES[activation]

    def build(self, input_shape):
        L, Cin = input_shape
        if self.stride < 1 or (Cin > 4 and self.stride > 2):
            raise NotImplementedError('Conv1D(strides=%d) on %d input channels: the matrix-core kernels implement strides 1 and 2 '
                                      '(any stride >= 1 runs for <= 4 input channels)' % (self.stride, Cin))
        if not 1 <= self.k <= 40:
            raise NotImplementedError('Conv1D(kernel_size=%d): 1..40 taps (bbhMahoGANy.py:228 names 5 and 10)' % self.k)
        self.kernel = self.add_weight('kernel', glorot_uniform((self.k, Cin, self.filters)))
        self.bias = self.add_weight('bias', np.zeros(self.filters, np.float32))

    def compute_output_shape(self, input_shape):
        return (ops.conv_geometry(input_shape[0], self.k, self.stride, self.padding)[0], self.filters)

    def _tap_fold(self, x, w):
        """More than 5 taps (`filtsize = 5 # 10 is best`, bbhMahoGANy.py:228) run as G = ceil(k/5) groups of h = ceil(k/G) taps over the input with its
        shifted copies as further channel groups (csrc/tap_fold.hip): the same <= 5-tap matrix-core kernels, the tap groups accumulating in their K loop.
        -> (x2, w2, (L, pad_left))"""
        _, pl = ops.conv_geometry(x.shape[1], self.k, self.stride, self.padding)
        return ops.conv1d_tapfold_x(x, self.k, pl), ops.conv1d_tapfold_w(w), (x.shape[1], pl)

    @property
    def can_fold_bn(self):
        return self.activation[0] == 'linear' and self.filters % 4 == 0 and self.filters > 4

    def can_defer_dgrad(self, Cin):
        """1 filter, stride 1: a BatchNormalization producer can form this layer's data gradient inside its own backward passes."""
        return self.filters == 1 and self.stride == 1 and self.k <= 5 and Cin % 4 == 0 and not _NO_LAZYGRAD

    def can_fold_upsample(self):
        """UpSampling1D(2) in front folds into the weights (ops.conv1d_up2_fold): the engine's planner asks."""
        return self.k == 5 and self.padding == 'same' and self.stride in (1, 2) and not _NO_UPFOLD

    def _geometry(self, node, x, w, b):
        """(w, b, k, stride, pad_left, Lout, Cout) of the conv that actually runs on x: the layer's own, or -- with the upsample in front
        folded (node.fold_up) -- the 3-tap stride-1 conv on the un-upsampled input; for stride 1 its (L, 2*filters) output is the layer's
        (2L, filters) output in memory."""
        if getattr(node, 'fold_up', None) is None:
            Lout, pl = ops.conv_geometry(x.shape[1], self.k, self.stride, self.padding)
            if self.k > 5:                  # w is the tap-folded kernel, x will be (forward: _tap_fold)
                return w, b, ops.tap_groups(self.k)[1], self.stride, 0, Lout, self.filters
            return w, b, self.k, self.stride, pl, Lout, self.filters
        wf, bf = ops.conv1d_up2_fold(w, b, self.stride)
        return wf, bf, 3, 1, 1, x.shape[1], wf.shape[2]

    def forward(self, ctx, node, x):
        a = node.fused_act or self.activation
        fold = getattr(node, 'fold_up', None) is not None
        B = x.shape[0]
        bn_node = getattr(node, 'infer_bn', None)
        if not ctx.training and bn_node is not None and x.shape[2] > 4:
            # inference phase: the following BatchNormalization (moving statistics) folds into the weights, its activation into the
            # epilogue: one kernel, and the pre-BN tensor is never written (generator.predict, bbhMahoGANy.py:1248)
            bn = bn_node.layer
            scale, shift = ops.bn_infer_coeffs(bn.gamma.data, bn.beta.data, bn.moving_mean.data, bn.moving_variance.data, bn.epsilon)
            w2, b2 = ops.conv_fold_bn(self.kernel.data, self.bias.data, scale, shift)
            xin = x
            if self.k > 5:
                x, w2, _ = self._tap_fold(x, w2)
            w2, b2, _, stride, pl, Lout, _ = self._geometry(node, xin, w2, b2)
            act = bn_node.fused_act or ('linear', 0.0)
            ctx.skip.add(bn_node.index)
            return ops.conv1d_fwd(x, w2, b2, stride, pl, Lout, act[0], act[1]).view(B, -1, self.filters)
        tf = None
        if self.k > 5:
            xin = x
            x, w2, tf = self._tap_fold(x, self.kernel.data)
            w, b, k, stride, pl, Lout, Ce = self._geometry(node, xin, w2, self.bias.data)
            tf = tf + (w2,)
        else:
            w, b, k, stride, pl, Lout, Ce = self._geometry(node, x, self.kernel.data, self.bias.data)
        fused_drop = node.fused_drop is not None and self.filters > 4
        if node.fused_drop is not None and not fused_drop:
            raise NotImplementedError('Dropout directly after a Conv1D with <= 4 filters')
        if ctx.training and bn_node is not None and x.shape[2] > 4 and Ce == self.filters and not _NO_CONVSTATS:
            # training phase, linear conv whose only consumer is a BatchNormalization: its batch statistics come out of the conv kernel's
            # epilogue (no separate pass over the output); the BN node picks them up from ctx.bn_sums.  (Not for the two-phase folded
            # form, whose columns are (phase, channel): that BN layer runs its own statistics pass.)
            y, sums = ops.conv1d_fwd_stats(x, w, b, stride, pl, Lout)
            ctx.bn_sums[bn_node.index] = sums
            ctx.tape[node.index] = (x, y, a, pl, None, 0.0, w if fold else None, tf)
            return y
        y, mask, rate = _conv_fwd(node, ctx, x, w, b, stride, pl, Lout, a, (B, Lout, Ce))
        ctx.tape[node.index] = (x, y, a, pl, mask, rate, w if fold else None, tf)   # x, y, mask in the shape of the conv that ran
        y = y.view(B, -1, self.filters)
        if ctx.training and (a[0] != 'linear' or mask is not None):
            ctx.epi[node.index] = (y, a[0], a[1], None if mask is None else mask.view(y.shape), rate)
        return y

    def backward(self, ctx, node, dy, need_dx, need_dw, prev=None):
        x, y, a, pl, mask, rate, wf, tf = ctx.tape.pop(node.index)
        dy = _conv_bwd_epilogue(dy.contiguous().view(y.shape), y, a, mask, rate, ctx, node)
        if tf is not None:
            # more than 5 taps: the gradients of the h-tap conv over (x, shifted x, ...) that ran, unfolded (csrc/tap_fold.hip)
            L0, pl0, w2 = tf
            if need_dw:
                dw2, _ = ops.conv1d_wgrad(x, dy, w2.shape[0], self.stride, 0, None, self.bias.grad)
                ops.conv1d_tapunfold_dw(dw2, self.k, self.kernel.grad)
            if need_dx:
                dx2 = ops.conv1d_dgrad(dy, ops.conv1d_transpose_w(w2), x.shape[1], self.stride, 0, None)
                return ops.conv1d_tapunfold_dx(dx2, L0, self.k, pl0)
            return None
        if need_dw:
            if wf is None:
                ops.conv1d_wgrad(x, dy, self.k, self.stride, pl, self.kernel.grad, self.bias.grad)
            else:
                dwf, dbf = ops.conv1d_wgrad(x, dy, 3, 1, pl)
                ops.conv1d_up2_unfold_grad(dwf, dbf, self.filters, self.stride, self.kernel.grad, self.bias.grad)
        if need_dx:
            if prev is not None and ops.can_fuse_dgrad(x.shape[2], y.shape[2]):
                ctx.pre_applied.add(node.fuse_prev)
            else:
                prev = None
            if getattr(node, 'lazy_bn', -1) >= 0:
                # 1 filter, stride 1, and the input comes straight from a BatchNormalization: that layer's backward passes form this
                # data gradient on the fly (ops.ConvGrad1); the (B, L, Cin) tensor is neither written here nor read there
                return ops.ConvGrad1(dy, self.kernel.data, x.shape[1], pl)
            if wf is None:
                return ops.conv1d_dgrad(dy, ops.conv1d_transpose_w(self.kernel.data), x.shape[1], self.stride, pl, prev)
            return ops.conv1d_dgrad(dy, ops.conv1d_transpose_w(wf), x.shape[1], 1, pl, prev)
        return None


class Conv2D(Layer):
    """bbhMahoGANy.py:439,:447: Conv2D(C, (5,5), strides=(2,1), padding='same') on a width-2 image (n, 2, Cin).
    Executed as the exactly equivalent Conv1D over H with (w,c)-interleaved channels; the dead
    width taps kw in {0,4} receive zero gradient, as they do in the reference."""
    fusable_act = True
    fusable_drop = True
    offers_act_bwd = True
    can_absorb_prev_act_bwd = True

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', activation=None, kernel_initializer='glorot_uniform', use_bias=True, **kw):
        Layer.__init__(self, **kw)
        _check_init(kernel_initializer)
        self.filters = int(filters)
        self.kh, self.kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.sh, self.sw = (strides, strides) if isinstance(strides, int) else tuple(strides)
        self.padding = padding
        self.activation = _ACT_NAMES[activation]
        if self.kw != 5 or self.sw != 1 or padding != 'same' or not use_bias:
            raise NotImplementedError('Conv2D is implemented for the hot-path case only: kernel (kh,5), strides (sh,1), padding "same", width-2 input')

    def build(self, input_shape):
        H, W, Cin = input_shape
        if W != 2:
            raise NotImplementedError('Conv2D is implemented for width-2 images (got width %d)' % W)
        self.kernel = self.add_weight('kernel', glorot_uniform((self.kh, self.kw, Cin, self.filters)))
        self.bias = self.add_weight('bias', np.zeros(self.filters, np.float32))

    def compute_output_shape(self, input_shape):
        return (ops.conv_geometry(input_shape[0], self.kh, self.sh, 'same')[0], 2, self.filters)

    def forward(self, ctx, node, x):
        a = node.fused_act or self.activation
        B, H, W, Cin = x.shape
        Lout, pl = ops.conv_geometry(H, self.kh, self.sh, 'same')
        wf, bf = ops.conv2d_w2_fold(self.kernel.data, self.bias.data)
        xf = x.reshape(B, H, 2 * Cin)
        y, mask, rate = _conv_fwd(node, ctx, xf, wf, bf, self.sh, pl, Lout, a, (B, Lout, 2, self.filters))
        ctx.tape[node.index] = (xf, y, a, pl, wf, Cin, mask, rate)
        if ctx.training and (a[0] != 'linear' or mask is not None):
            ctx.epi[node.index] = (y, a[0], a[1], mask, rate)
        return y.reshape(B, Lout, 2, self.filters)

    def backward(self, ctx, node, dy, need_dx, need_dw, prev=None):
        xf, y, a, pl, wf, Cin, mask, rate = ctx.tape.pop(node.index)
        dy = _conv_bwd_epilogue(dy.contiguous().reshape(y.shape), y, a, mask, rate, ctx, node)
        if need_dw:
            dwf, dbf = ops.conv1d_wgrad(xf, dy, self.kh, self.sh, pl)
            ops.conv2d_w2_unfold_grad(dwf, dbf, Cin, self.filters, self.kernel.grad, self.bias.grad)
        if need_dx:
            if prev is not None and ops.can_fuse_dgrad(2 * Cin, 2 * self.filters):
                ctx.pre_applied.add(node.fuse_prev)
            else:
                prev = None
            dx = ops.conv1d_dgrad(dy, ops.conv1d_transpose_w(wf), xf.shape[1], self.sh, pl, prev)
            return dx.reshape(xf.shape[0], xf.shape[1], 2, Cin)
        return None


BN_MOVING_AVERAGE = 'tf_zero_debias'       # default form of the moving-statistics update (BatchNormalization docstring)


class BatchNormalization(Layer):
    """bbhMahoGANy.py:235,:251,:260,:268,:276,:284 (momentum=0.99).  Train phase: batch mean / biased variance (fp64
    accumulation), moving statistics updated with keras' n/(n-(1+eps)) variance correction; inference: moving statistics.
    A following Activation and Dropout run in the same pass (one read, one write).  Under data parallelism the
    statistics are all-reduced (SyncBN) so that N ranks x B/N rows reproduce a single-device batch of B.

    moving_average = 'tf_zero_debias' (default) reproduces keras 2.2.4 on its TF 1.12 backend: K.moving_average_update calls
    tf moving_averages.assign_moving_average(x, value, momentum, zero_debias=True), which keeps a zero-initialised `biased` shadow
    accumulator and a `local_step` counter and sets moving = biased / (1 - momentum^local_step): the moving statistics forget their
    0 / 1 initial values at the first update (what generator.predict sees early in training, bbhMahoGANy.py:1248).  TF creates one
    such pair PER CALL SITE of the layer (the generator's layers are called again when the generator is added to another
    Sequential, :517, :537), and only the call site inside the model being trained is updated: the state is therefore kept per
    (layer, training model).  The pairs are not keras weights: real keras never writes them to .h5 files and starts them from zero
    after load_weights; keras_io keeps them in a private section of files written here.
    moving_average = 'ema': the plain exponential average (zero_debias=False; what tf.keras and later keras versions do)."""
    fusable_act = True
    fusable_drop = True

    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, moving_average=None, **kw):
        Layer.__init__(self, **kw)
        if axis != -1:
            raise NotImplementedError('BatchNormalization(axis=%r)' % (axis,))
        self.momentum, self.epsilon = float(momentum), float(epsilon)
        self.moving_average = moving_average or BN_MOVING_AVERAGE
        if self.moving_average not in ('tf_zero_debias', 'ema'):
            raise ValueError("BatchNormalization(moving_average=%r): 'tf_zero_debias' or 'ema'" % (moving_average,))
        self.zero_debias = {}              # training-model name -> [biased_mean, biased_var (device fp32), local_step (int)]

    def zero_debias_state(self, site):
        st = self.zero_debias.get(site)
        if st is None:
            C = self.gamma.shape[0]
            st = [torch.zeros(C, dtype=torch.float32, device=device()), torch.zeros(C, dtype=torch.float32, device=device()), 0]
            self.zero_debias[site] = st
        return st

    def build(self, input_shape):
        C = input_shape[-1]
        self.gamma = self.add_weight('gamma', np.ones(C, np.float32))
        self.beta = self.add_weight('beta', np.zeros(C, np.float32))
        self.moving_mean = self.add_weight('moving_mean', np.zeros(C, np.float32), trainable=False)
        self.moving_variance = self.add_weight('moving_variance', np.ones(C, np.float32), trainable=False)

    is_batchnorm = True

    def forward(self, ctx, node, x):
        if node.index in ctx.skip:         # inference phase: already folded into the producing convolution
            return x
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        act = node.fused_act or ('linear', 0.0)
        if not ctx.training:
            scale, shift = ops.bn_infer_coeffs(self.gamma.data, self.beta.data, self.moving_mean.data, self.moving_variance.data, self.epsilon)
            return ops.bn_apply(x2, scale, shift, None, act[0], act[1]).reshape(x.shape)
        sums = ctx.bn_sums.pop(node.index, None)          # handed over by the producing convolution's epilogue, if it had one
        if sums is None:
            sums = ops.bn_stats(x2)
        count = x2.shape[0]
        if ctx.dp is not None:
            ctx.dp.all_reduce_sum(sums)
            count *= ctx.dp.world_size
        zd = None
        if self.moving_average == 'tf_zero_debias':
            st = self.zero_debias_state(ctx.site)
            cap = capturing()
            if cap is None:
                st[2] += 1
                zd = (st[0], st[1], st[2])
            else:                          # captured step graph: local_step advances once per REPLAY and reaches the kernel through device memory

                def next_step(st=st):
                    st[2] += 1
                    return st[2]
                zd = (st[0], st[1], cap.slot('i', next_step))
        scale, shift, smean, sinv = ops.bn_finalize(sums, count, self.gamma.data, self.beta.data, self.epsilon, self.momentum,
                                                    self.moving_mean.data, self.moving_variance.data, zd)
        mask, rate = None, 0.0
        if node.fused_drop is not None and node.fused_drop[0] > 0.0:
            rate, drop_layer = node.fused_drop
            if ctx.dropout_masks.get(drop_layer.name) is None and C % 4 == 0 and not _NO_DROPGEN and (ctx.row_map is None or len(ctx.row_map[0]) == 1):
                # no injected mask: draw it inside the apply pass (same Philox stream as Dropout.make_mask would take; under data parallelism the
                # counters of this rank's rows of the global tensor)
                if ctx.row_map is None:
                    seed, off = device_rng().take(x2.numel())
                else:
                    (g0, nrows), grows = ctx.row_map[0][0], ctx.row_map[1]
                    seed, offs = device_rng().take_rows(x2.numel() // nrows, [(g0, nrows)], grows)
                    off = offs[0]
                y, mask = ops.bn_apply_dropgen(x2, scale, shift, act[0], act[1], rate, seed, off)
                ctx.tape[node.index] = (x2, mask, smean, sinv, count, act, rate, scale, shift)
                return y.reshape(x.shape)
            mask = drop_layer.make_mask(ctx, x2.shape)
        y = ops.bn_apply(x2, scale, shift, mask, act[0], act[1], rate)
        # the backward pass recomputes the activation output from x2 with this scale / shift (bit-identical to y before the
        # dropout scale), so the layer output is not kept for it: 4 of 13 / 17 bytes per element less in its two passes
        ctx.tape[node.index] = (x2, mask, smean, sinv, count, act, rate, scale, shift)
        return y.reshape(x.shape)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        x2, mask, smean, sinv, count, act, rate, scale, shift = ctx.tape.pop(node.index)
        y = None
        if x2.shape[1] % 4:            # the scalar fallback kernels read the stored output; not reached by the BBH nets
            raise NotImplementedError('BatchNormalization backward over %d channels (not a multiple of 4)' % x2.shape[1])
        lazy = isinstance(dy, ops.ConvGrad1)
        if lazy:
            local = ops.bn_bwd_stats_conv1(dy, x2, mask, smean, sinv, act[0], act[1], rate, scale, shift)
        else:
            dy2 = dy.contiguous().reshape(x2.shape)
            local = ops.bn_bwd_stats(dy2, y, x2, mask, smean, sinv, act[0], act[1], rate, scale, shift)
        glob = local
        if ctx.dp is not None:
            glob = local.clone()
            ctx.dp.all_reduce_sum(glob)
        if need_dw:
            dgamma, dbeta = self.gamma.grad, self.beta.grad
        else:
            dgamma = torch.empty_like(self.gamma.data); dbeta = torch.empty_like(self.beta.data)
        if lazy:
            dx = ops.bn_bwd_apply_conv1(dy, x2, mask, self.gamma.data, smean, sinv, glob, count, local, dgamma, dbeta, act[0], act[1], rate, scale, shift)
        else:
            dx = ops.bn_bwd_apply(dy2, y, x2, mask, self.gamma.data, smean, sinv, glob, count, local, dgamma, dbeta, act[0], act[1], rate, scale, shift)
        return dx.reshape(dy.shape)


class Activation(Layer):
    def __init__(self, activation, **kw):
        Layer.__init__(self, **kw)
        if activation not in _ACT_NAMES:
            raise NotImplementedError('Activation(%r)' % (activation,))
        self.act_spec = _ACT_NAMES[activation]

    def forward(self, ctx, node, x):
        y = x if self.act_spec[0] == 'linear' else ops.act_fwd(x.contiguous(), self.act_spec[0], self.act_spec[1])
        ctx.tape[node.index] = y
        return y

    def backward(self, ctx, node, dy, need_dx, need_dw):
        y = ctx.tape.pop(node.index)
        if self.act_spec[0] == 'linear':
            return dy
        return ops.act_bwd(dy.contiguous(), y, self.act_spec[0], self.act_spec[1])


class LeakyReLU(Activation):
    def __init__(self, alpha=0.3, **kw):
        Layer.__init__(self, **kw)
        self.alpha = float(np.float32(alpha))       # K.cast_to_floatx(alpha): the reference's Keras files record 0.20000000298023224 for 0.2
        self.act_spec = ('leaky', self.alpha)


class ReLU(Activation):
    """keras.layers.ReLU(max_value=...) (bbhMahoGANy.py:400: ReLU(max_value=1.0))."""

    def __init__(self, max_value=None, negative_slope=0.0, threshold=0.0, **kw):
        Layer.__init__(self, **kw)
        if negative_slope or threshold:
            raise NotImplementedError('ReLU(negative_slope/threshold)')
        self.act_spec = ('relu', 0.0) if max_value is None else ('relu_max', float(max_value))


class PReLU(Layer):
    """keras.layers.PReLU() (bbhMahoGANy.py:39; the act = 'prelu' branches of generator_model, :237-286): one learnable slope per
    feature of a sample (no shared axes), initialised to zero; y = x > 0 ? x : alpha * x."""

    def __init__(self, alpha_initializer='zeros', shared_axes=None, **kw):
        Layer.__init__(self, **kw)
        if alpha_initializer not in ('zeros', None) and not (isinstance(alpha_initializer, dict) and alpha_initializer.get('class_name') == 'Zeros'):
            raise NotImplementedError('PReLU(alpha_initializer=%r)' % (alpha_initializer,))
        if shared_axes:
            raise NotImplementedError('PReLU(shared_axes=%r)' % (shared_axes,))

    def build(self, input_shape):
        if int(np.prod(input_shape)) % 4:
            raise NotImplementedError('PReLU on %r: the feature count must be a multiple of 4' % (tuple(input_shape),))
        self.alpha = self.add_weight('alpha', np.zeros(tuple(input_shape), np.float32))

    def forward(self, ctx, node, x):
        x = x.contiguous()
        ctx.tape[node.index] = x
        return ops.prelu_fwd(x, self.alpha.data)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        x = ctx.tape.pop(node.index)
        dx, _ = ops.prelu_bwd(dy.contiguous(), x, self.alpha.data, need_dx or not need_dw, self.alpha.grad if need_dw else None)
        return dx


class Dropout(Layer):
    """Inverted dropout, active in the training phase only (which keras applies to the WHOLE graph in train_on_batch,
    including layers of frozen sub-models: bbhMahoGANy.py:1296 keeps D's Dropout(0.4) active during the G step)."""

    def __init__(self, rate, **kw):
        Layer.__init__(self, **kw)
        self.rate = float(rate)
        self.drop_rate = self.rate

    def make_mask(self, ctx, shape):
        inj = ctx.dropout_masks.get(self.name)
        if inj is not None:
            return inj.reshape(shape).contiguous()
        n = int(np.prod(shape))
        if ctx.row_map is None:
            seed, off = device_rng().take(n)
            return ops.dropout_mask(shape, self.rate, seed, off, device())
        # data parallelism: the local rows are blocks of the global batch; every block draws the counters of its global rows
        blocks, grows = ctx.row_map
        if sum(nr for _, nr in blocks) != shape[0]:
            raise ValueError('Dropout: the row map covers %d rows, the tensor has %d' % (sum(nr for _, nr in blocks), shape[0]))
        row_len = n // shape[0]
        seed, offs = device_rng().take_rows(row_len, blocks, grows)
        parts = [ops.dropout_mask((nr,) + tuple(shape[1:]), self.rate, seed, off, device()) for (_, nr), off in zip(blocks, offs)]
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def forward(self, ctx, node, x):
        if not ctx.training or self.rate == 0.0:
            ctx.tape[node.index] = None
            return x
        mask = self.make_mask(ctx, x.shape)
        ctx.tape[node.index] = mask
        return ops.dropout_apply(x.contiguous(), mask, self.rate)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        mask = ctx.tape.pop(node.index)
        if mask is None:
            return dy
        return ops.dropout_apply(dy.contiguous(), mask, self.rate)


class Reshape(Layer):
    shape_only = True

    def __init__(self, target_shape, **kw):
        Layer.__init__(self, **kw)
        self.target_shape_arg = tuple(int(v) for v in target_shape)      # may hold one -1, as keras allows
        self.target_shape = self.target_shape_arg
        if list(self.target_shape_arg).count(-1) > 1:
            raise ValueError('Reshape: at most one unknown dimension')

    def compute_output_shape(self, input_shape):
        n = int(np.prod(input_shape))
        if -1 in self.target_shape_arg:
            known = -int(np.prod(self.target_shape_arg))
            assert known > 0 and n % known == 0, 'Reshape%r does not fit an input of %d elements' % (self.target_shape_arg, n)
            self.target_shape = tuple(n // known if v == -1 else v for v in self.target_shape_arg)
        assert n == int(np.prod(self.target_shape)), 'Reshape%r does not fit an input of %d elements' % (self.target_shape_arg, n)
        return self.target_shape

    def forward(self, ctx, node, x):
        ctx.tape[node.index] = x.shape
        return x.reshape((x.shape[0],) + self.target_shape)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        return dy.reshape(ctx.tape.pop(node.index))


class Flatten(Layer):
    shape_only = True
    """Row-major flatten of channels-last activations: feature index = t*C + c (keras order, SURVEY Appendix B.7)."""

    def compute_output_shape(self, input_shape):
        return (int(np.prod(input_shape)),)

    def forward(self, ctx, node, x):
        ctx.tape[node.index] = x.shape
        return x.reshape(x.shape[0], -1)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        return dy.reshape(ctx.tape.pop(node.index))


class UpSampling1D(Layer):
    """bbhMahoGANy.py:249,:258.  In front of a 5-tap 'same' Conv1D (both uses in the generator) the planner folds it into that conv's
    weights and this layer never runs; the kernels below serve every other placement."""
    is_upsample2 = True

    def __init__(self, size=2, **kw):
        Layer.__init__(self, **kw)
        if size != 2:
            raise NotImplementedError('UpSampling1D(size=%r)' % (size,))

    def compute_output_shape(self, input_shape):
        return (2 * input_shape[0], input_shape[1])

    def forward(self, ctx, node, x):
        return ops.upsample2_fwd(x.contiguous())

    def backward(self, ctx, node, dy, need_dx, need_dw):
        return ops.upsample2_bwd(dy.contiguous())


class MaxPooling2D(Layer):
    """bbhMahoGANy.py:444, :453, :462, ... (`maxpool = True`): MaxPooling2D(pool_size=(2,1)) -- pairs of rows along H; strides = pool_size, 'valid'."""

    def __init__(self, pool_size=(2, 2), strides=None, padding='valid', **kw):
        Layer.__init__(self, **kw)
        pool_size = (pool_size, pool_size) if isinstance(pool_size, int) else tuple(pool_size)
        strides = pool_size if strides is None else ((strides, strides) if isinstance(strides, int) else tuple(strides))
        if pool_size != (2, 1) or strides != (2, 1) or padding != 'valid':
            raise NotImplementedError('MaxPooling2D is implemented for the reference\'s case only: pool_size (2,1), strides (2,1), padding "valid"')

    def compute_output_shape(self, input_shape):
        if input_shape[0] < 2:
            raise ValueError('MaxPooling2D(pool_size=(2,1)) on %d rows' % input_shape[0])
        return (input_shape[0] // 2,) + tuple(input_shape[1:])

    def forward(self, ctx, node, x):
        x = x.contiguous()
        if ctx.training:
            ctx.tape[node.index] = x
        return ops.maxpool_h2_fwd(x)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        return ops.maxpool_h2_bwd(dy.contiguous(), ctx.tape.pop(node.index))


class MyLayer(Layer):
    """bbhMahoGANy.py:164-188: stack([x, const - x], axis=2): (B, n_pix, 1) -> (B, n_pix, 2, 1), const = measured data h(t)."""

    def __init__(self, const, **kw):
        Layer.__init__(self, **kw)
        self._const_host = np.asarray(const, np.float32).reshape(-1)
        self._const = None

    @property
    def const(self):
        if self._const is None:
            from .engine import to_device
            self._const = to_device(self._const_host)
        return self._const

    def compute_output_shape(self, input_shape):
        return (input_shape[0], 2, 1)

    def forward(self, ctx, node, x):
        assert x.shape[1] == self.const.numel()
        return ops.subtract_stack_fwd(x.contiguous(), self.const)

    def backward(self, ctx, node, dy, need_dx, need_dw):
        return ops.subtract_stack_bwd(dy.contiguous())
