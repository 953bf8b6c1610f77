"""CPU oracle: the three BBH networks and their train steps, restated on top of keras_ref.py.

TEST INFRASTRUCTURE ONLY (see keras_ref.py header).  PARITY UNPINNED against real Keras/TF.

Follows /root/reference/BBH_version/bbhMahoGANy.py:
  generator_model              :212-295
  signal_pe_model (2-branch)   :356-404
  signal_discriminator_model   :408-498 (active branch num_lays=2: :439-453, :491-495)
  MyLayer / compositions       :164-188, :500-539
  compile / frozen-D semantics :797-809, :1100-1119
  CNN train step               :1153-1168
  GAN iteration                :1241-1299

Layer specs are tuples; parameters live in flat python lists of numpy arrays so a test can copy the
very same arrays into the HIP-backed model.
"""
import numpy as np
from . import keras_ref as K

LEAKY_ALPHA = float(np.float32(0.2))     # LeakyReLU(alpha=0.2) as Keras holds it (K.cast_to_floatx; 0.20000000298023224 in the reference's own Keras files)


# ----------------------------------------------------------------------------------------------
# layer specs
# ----------------------------------------------------------------------------------------------
def generator_spec(n_pix, filtsize=5):
    L = [('dense', 100, 256 * (n_pix // 2)), ('bn', 256 * (n_pix // 2)), ('act', 'tanh', 0.0), ('drop', 0.2),
         ('reshape', (n_pix // 2, 256)),
         ('up', 2), ('conv1d', 256, 64, filtsize, 2, 'same'), ('bn', 64), ('act', 'tanh', 0.0), ('drop', 0.2),
         ('up', 2), ('conv1d', 64, 128, filtsize, 1, 'same'), ('bn', 128), ('act', 'tanh', 0.0), ('drop', 0.2)]
    for cin, cout in ((128, 256), (256, 512), (512, 1024)):
        L += [('conv1d', cin, cout, filtsize, 1, 'same'), ('bn', cout), ('act', 'tanh', 0.0), ('drop', 0.2)]
    L += [('conv1d', 1024, 1, filtsize, 1, 'same'), ('act', 'linear', 0.0)]
    return L


def discriminator_spec(n_pix, num_lays=2, batchnorm=False, maxpool=False):
    """bbhMahoGANy.py:408-498; defaults = the active configuration (:424-426).  Layer 2 normalises after its activation (:449-450), layers 3-6 before."""
    L, H, cin = [], n_pix, 1
    for i, (cout, sh) in enumerate(((256, 2), (512, 2), (256, 1), (512, 1), (1024, 1), (1024, 1))[:num_lays]):
        L.append(('conv2d', cin, cout, (5, 5), (sh, 1), 'same'))
        H = K.conv_out_len(H, 5, sh, 'same')
        if batchnorm and i >= 2:
            L.append(('bn', cout))
        L.append(('act', 'leaky', LEAKY_ALPHA))
        if batchnorm and i == 1:
            L.append(('bn', cout))
        L.append(('drop', 0.4))
        if maxpool:
            L.append(('maxpool',)); H //= 2
        cin = cout
    return L + [('flatten',), ('dense', H * 2 * cin, 1), ('act', 'sigmoid', 0.0)]


def pe_branch_specs(n_pix):
    mc = [('conv1d', 1, 64, 5, 2, 'same'), ('act', 'relu', 0.0)]
    L = K.conv_out_len(n_pix, 5, 2, 'same')
    for cin, cout in ((64, 128), (128, 256), (256, 512)):
        mc += [('conv1d', cin, cout, 5, 2, 'valid'), ('act', 'relu', 0.0)]
        L = K.conv_out_len(L, 5, 2, 'valid')
    mc += [('flatten',), ('dense', L * 512, 1), ('act', 'relu', 0.0)]
    q = [('conv1d', 1, 64, 5, 1, 'same'), ('act', 'relu', 0.0)]
    L = n_pix
    for cin, cout, s in ((64, 128, 1), (128, 256, 1), (256, 512, 2), (512, 1024, 2)):
        q += [('conv1d', cin, cout, 5, s, 'valid'), ('act', 'relu', 0.0)]
        L = K.conv_out_len(L, 5, s, 'valid')
    q += [('flatten',), ('dense', L * 1024, 1), ('act', 'relu_max', 1.0)]
    return mc, q


# ----------------------------------------------------------------------------------------------
# a sequential stack with explicit forward/backward
# ----------------------------------------------------------------------------------------------
class Stack(object):
    """Parameters: self.params = list of arrays in keras weight order per layer
    (conv/dense: kernel, bias; bn: gamma, beta [+ moving_mean, moving_var in self.state])."""

    def __init__(self, spec, rng=None, dtype=np.float64, moving_average='tf_zero_debias'):
        self.spec = spec
        self.dtype = dtype
        self.moving_average = moving_average      # 'tf_zero_debias' (keras 2.2.4 TF backend) | 'ema'; K.bn_moving_update*
        self.zd = {}              # layer index -> [biased_mean, biased_var, local_step] (zero-debias shadow variables)
        self.params = []          # trainable, in order
        self.pidx = []            # per layer: indices into params
        self.state = {}           # layer index -> [moving_mean, moving_var]
        rng = rng or np.random.RandomState(0)
        for li, s in enumerate(spec):
            idx = []
            if s[0] == 'dense':
                idx = self._add(K.glorot_uniform(rng, (s[1], s[2]), dtype), np.zeros(s[2], dtype))
            elif s[0] == 'conv1d':
                idx = self._add(K.glorot_uniform(rng, (s[3], s[1], s[2]), dtype), np.zeros(s[2], dtype))
            elif s[0] == 'conv2d':
                idx = self._add(K.glorot_uniform(rng, (s[3][0], s[3][1], s[1], s[2]), dtype), np.zeros(s[2], dtype))
            elif s[0] == 'bn':
                idx = self._add(np.ones(s[1], dtype), np.zeros(s[1], dtype))
                self.state[li] = [np.zeros(s[1], dtype), np.ones(s[1], dtype)]
                self.zd[li] = [np.zeros(s[1], dtype), np.zeros(s[1], dtype), 0]
            self.pidx.append(idx)

    def _add(self, *arrs):
        i0 = len(self.params)
        self.params.extend(arrs)
        return list(range(i0, i0 + len(arrs)))

    DECISION_BAND = 1e-5      # |pre-activation| <= band * max|pre-activation|: the piecewise-linear branch is "within rounding of the kink"

    def forward(self, x, training, masks=None, momentum=0.99, update_moving=True, decisions=None):
        """masks: dict layer-index -> keep mask for dropout layers (training only).
        decisions: dict act-layer-index -> the fp32 implementation's OUTPUT of that relu / relu_max / LeakyReLU layer (zeros from a
        following dropout allowed).  ReLU-type derivatives are discontinuous: an element whose pre-activation is within fp32 rounding
        of zero can legitimately take the other branch in fp32, and with millions of activations at full size a few do -- each flips a
        whole gradient value.  Like dropout masks, those branch decisions are therefore INJECTED: inside the band the oracle takes the
        implementation's branch; outside the band its own, and a disagreement there is counted in self.decision_stats[li] =
        (elements in band, flipped in band, disagreements outside band) -- the tests require the last to be 0."""
        self.tape = []
        self.decision_stats = {}
        self.decision_worst = {}
        for li, s in enumerate(self.spec):
            p = [self.params[i] for i in self.pidx[li]]
            kind = s[0]
            if kind == 'dense':
                self.tape.append(x); x = K.dense_fwd(x, p[0], p[1])
            elif kind == 'conv1d':
                self.tape.append(x); x = K.conv1d_fwd(x, p[0], p[1], s[4], s[5])
            elif kind == 'conv2d':
                self.tape.append(x); x = K.conv2d_fwd(x, p[0], p[1], s[4], s[5])
            elif kind == 'bn':
                if training:
                    y, cache, mean, var = K.bn_train_fwd(x, p[0], p[1])
                    if update_moving:
                        n = x.size // x.shape[-1]
                        if self.moving_average == 'ema':
                            self.state[li] = list(K.bn_moving_update(self.state[li][0], self.state[li][1], mean, var, n, momentum))
                        else:
                            mm, mv, self.zd[li] = K.bn_moving_update_zero_debias(self.state[li][0], self.state[li][1], self.zd[li], mean, var, n, momentum)
                            self.state[li] = [mm, mv]
                    self.tape.append(cache); x = y
                else:
                    self.tape.append(None)
                    x = K.bn_infer_fwd(x, p[0], p[1], self.state[li][0], self.state[li][1])
            elif kind == 'act':
                pre = x
                x = K.act_fwd(x, s[1], s[2])
                if decisions is not None and li in decisions and s[1] in ('relu', 'relu_max', 'leaky'):
                    gy = np.asarray(decisions[li]).reshape(pre.shape)
                    known = (gy != 0) if s[1] == 'leaky' else np.ones(pre.shape, bool)       # leaky: 0 = dropped afterwards, no information ...
                    if s[1] == 'leaky' and masks is not None and li + 1 < len(self.spec) and self.spec[li + 1][0] == 'drop' and (li + 1) in masks:
                        # ... unless the dropout KEPT the element: then an output of exactly 0 is the implementation's pre-activation being exactly 0 (its
                        # conv sum cancelled its bias to the last bit: ~1e-8 per element, i.e. about once per ten 100-iteration trajectories), and its
                        # backward takes the not-positive branch (act'(y) from y > 0).  Round 5: one such element sent a whole output channel's
                        # gradient 7 % off at iteration 50 of one data seed of tests/test_trajectory_gpu.py.
                        known = known | (np.asarray(masks[li + 1]).reshape(pre.shape) != 0)
                    theirs = gy > 0
                    band = np.abs(pre) <= self.DECISION_BAND * np.abs(pre).max()
                    mism = known & ((pre > 0) != theirs)
                    flip = mism & band
                    self.decision_stats[li] = (int(band.sum()), int(flip.sum()), int((mism & ~band).sum()), int(pre.size))
                    self.decision_worst[li] = float((np.abs(pre)[mism]).max() / np.abs(pre).max()) if mism.any() else 0.0      # how far from the kink the farthest disagreement sits
                    if flip.any():
                        x = x.copy()
                        x[flip] = np.where(theirs[flip], 1e-300, 0.0 if s[1] != 'leaky' else -1e-300)
                self.tape.append(x)
            elif kind == 'drop':
                if training:
                    m = masks[li]
                    self.tape.append(m); x = K.dropout_fwd(x, m, s[1])
                else:
                    self.tape.append(None)
            elif kind == 'reshape':
                self.tape.append(x.shape); x = x.reshape((x.shape[0],) + tuple(s[1]))
            elif kind == 'flatten':
                self.tape.append(x.shape); x = x.reshape(x.shape[0], -1)
            elif kind == 'up':
                self.tape.append(None); x = K.upsample1d_fwd(x, s[1])
            elif kind == 'maxpool':
                H = x.shape[1]; y, second = K.maxpool_h2_fwd(x)
                if decisions is not None and li in decisions:
                    # the routing of a pair whose two values agree to within fp32 rounding is a branch decision like the ReLU kink: decisions[li] is the
                    # implementation's INPUT of this layer; inside the band its routing is taken, outside a disagreement is counted (tests: must be 0)
                    Ho = H // 2
                    xi = np.asarray(decisions[li]).reshape(x.shape)
                    a, c = x[:, 0:2 * Ho:2], x[:, 1:2 * Ho:2]
                    theirs = xi[:, 1:2 * Ho:2] > xi[:, 0:2 * Ho:2]
                    band = np.abs(c - a) <= self.DECISION_BAND * np.abs(x).max()
                    mism = theirs != second
                    flip = mism & band
                    self.decision_stats[li] = (int(band.sum()), int(flip.sum()), int((mism & ~band).sum()), int(second.size))
                    if flip.any():
                        second = np.where(flip, theirs, second); y = np.where(second, c, a)
                self.tape.append((second, H)); x = y
            else:
                raise ValueError(kind)
        return x

    def backward(self, dy):
        """Returns (dx, grads) with grads aligned to self.params (None where untouched)."""
        grads = [None] * len(self.params)
        for li in range(len(self.spec) - 1, -1, -1):
            s = self.spec[li]; t = self.tape[li]
            p = [self.params[i] for i in self.pidx[li]]
            kind = s[0]
            if kind == 'dense':
                dy, dW, db = K.dense_bwd(t, p[0], dy)
                grads[self.pidx[li][0]], grads[self.pidx[li][1]] = dW, db
            elif kind == 'conv1d':
                dy, dW, db = K.conv1d_bwd(t, p[0], dy, s[4], s[5])
                grads[self.pidx[li][0]], grads[self.pidx[li][1]] = dW, db
            elif kind == 'conv2d':
                dy, dW, db = K.conv2d_bwd(t, p[0], dy, s[4], s[5])
                grads[self.pidx[li][0]], grads[self.pidx[li][1]] = dW, db
            elif kind == 'bn':
                dy, dg, dbt = K.bn_train_bwd(dy, t, p[0])
                grads[self.pidx[li][0]], grads[self.pidx[li][1]] = dg, dbt
            elif kind == 'act':
                dy = K.act_bwd(dy, t, s[1], s[2])
            elif kind == 'drop':
                if t is not None:
                    dy = dy * t / (1.0 - s[1])
            elif kind in ('reshape', 'flatten'):
                dy = dy.reshape(t)
            elif kind == 'up':
                dy = K.upsample1d_bwd(dy, s[1])
            elif kind == 'maxpool':
                dy = K.maxpool_h2_bwd(dy, t[0], t[1])
        return dy, grads


class AdamState(object):
    """One keras optimizer instance: own t, m, v for the parameter list it was compiled with."""

    def __init__(self, params, lr=9e-5, b1=0.5):
        self.lr, self.b1, self.t = lr, b1, 0
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]

    def step(self, params, grads):
        self.t += 1
        for i, (p, g) in enumerate(zip(params, grads)):
            p_new, self.m[i], self.v[i] = K.adam_step(p, g, self.m[i], self.v[i], self.t, self.lr, self.b1)
            params[i][...] = p_new


# ----------------------------------------------------------------------------------------------
# CNN point-estimator train step  (bbhMahoGANy.py:1165: train_on_batch(x, [mc, q]))
# ----------------------------------------------------------------------------------------------
class PENet(object):
    def __init__(self, n_pix, rng=None, dtype=np.float64):
        rng = rng or np.random.RandomState(1)
        mc, q = pe_branch_specs(n_pix)
        self.mc, self.q = Stack(mc, rng, dtype), Stack(q, rng, dtype)
        self.opt = AdamState(self.mc.params + self.q.params)

    def predict(self, x):
        return [self.mc.forward(x, False), self.q.forward(x, False)]

    def train_on_batch(self, x, y_mc, y_q, decisions=(None, None)):
        """Returns [total, mc_loss, q_loss, mc_acc, q_acc] (keras multi-output order).  decisions: (mc, q) dicts for Stack.forward."""
        y_mc = np.asarray(y_mc, x.dtype).reshape(-1, 1); y_q = np.asarray(y_q, x.dtype).reshape(-1, 1)
        pm = self.mc.forward(x, True, decisions=decisions[0]); pq = self.q.forward(x, True, decisions=decisions[1])
        lm, dm = K.mse_loss(pm, y_mc); lq, dq = K.mse_loss(pq, y_q)
        out = [lm + lq, lm, lq, K.binary_accuracy(pm, y_mc), K.binary_accuracy(pq, y_q)]
        _, gm = self.mc.backward(dm); _, gq = self.q.backward(dq)
        self.last_grads = gm + gq
        self.opt.step(self.mc.params + self.q.params, gm + gq)
        return out


# ----------------------------------------------------------------------------------------------
# GAN: generator G, discriminator D, combined G -> MyLayer(event) -> D(frozen)
# ----------------------------------------------------------------------------------------------
class GAN(object):
    def __init__(self, n_pix, event, rng=None, dtype=np.float64, moving_average='tf_zero_debias', filtsize=5, d_config=None):
        rng = rng or np.random.RandomState(2)
        self.n_pix = n_pix
        self.G = Stack(generator_spec(n_pix, filtsize), rng, dtype, moving_average)
        self.D = Stack(discriminator_spec(n_pix, **(d_config or {})), rng, dtype, moving_average)
        self.event = np.asarray(event, dtype).reshape(n_pix, 1)
        self.opt_g = AdamState(self.G.params)      # signal_discriminator_on_generator (:1107), D frozen
        # `batchnorm = True` in the discriminator: its BatchNormalization layers are called once per graph they are part of (their own model and the combined
        # one), and every call creates its own zero-debias shadow accumulators (tf moving_averages.assign_moving_average(zero_debias=True) makes `biased` and
        # `local_step` per call site) that both write the one moving_mean / moving_variance: a second set for the G step
        self.D_zd_combined = {li: [np.zeros_like(v[0]), np.zeros_like(v[1]), 0] for li, v in self.D.zd.items()}
        self.opt_d = AdamState(self.D.params)      # signal_discriminator (:1115)

    def generate(self, z):
        """generator.predict: inference phase (moving-stat BN, no dropout)."""
        return self.G.forward(z, False)

    def d_train_on_batch(self, sX, sy, masks, decisions=None, p_impl=None):
        """p_impl: the implementation's own fp32 probabilities, for K.bce_loss's saturated-sample evaluation (see there); self.bce_sat records its check."""
        sy = np.asarray(sy, sX.dtype).reshape(-1, 1)
        p = self.D.forward(sX, True, masks, decisions=decisions)
        loss, dp = K.bce_loss(p, sy, p_impl)
        self.bce_sat = K.bce_loss.last
        _, g = self.D.backward(dp)
        self.last_d_grads = g
        self.opt_d.step(self.D.params, g)
        return [loss, K.binary_accuracy(p, sy)]

    def g_train_on_batch(self, z, sy, g_masks, d_masks, d_decisions=None, p_impl=None):
        """combined model: learning phase 1 for the whole graph (G batch-stat BN + dropout, D dropout active),
        gradients only into G (D collected as frozen at compile time)."""
        sy = np.asarray(sy, z.dtype).reshape(-1, 1)
        x = self.G.forward(z, True, g_masks)
        img = K.mylayer_fwd(x, self.event)
        own, self.D.zd = self.D.zd, self.D_zd_combined
        try:
            p = self.D.forward(img, True, d_masks, decisions=d_decisions)
        finally:
            self.D.zd = own
        loss, dp = K.bce_loss(p, sy, p_impl)
        self.bce_sat = K.bce_loss.last
        dimg, _ = self.D.backward(dp)
        dx = K.mylayer_bwd(dimg)
        _, g = self.G.backward(dx)
        self.last_g_grads = g
        self.opt_g.step(self.G.params, g)
        return [loss, K.binary_accuracy(p, sy)]

    def assemble_d_batch(self, real, noise, fake):
        """bbhMahoGANy.py:1268-1289.  real (B,n), noise (B,n,1) ~ N(0,1), fake (B,n,1) = G.predict(z).
        Fake half is in REVERSED sample order (np.append prepend at :1271)."""
        B = real.shape[0]
        resid = self.event[None] - fake
        fake2 = np.concatenate([fake, resid], axis=2)[::-1]
        real2 = np.concatenate([real.reshape(B, self.n_pix, 1), noise], axis=2)
        sX = np.concatenate([real2, fake2]).reshape(2 * B, self.n_pix, 2, 1)
        sy = [1.0] * B + [0.0] * B
        return sX, sy
