"""CPU baseline ("port"): the CNN train step and the GAN iteration of bbhMahoGANy.py on torch-CPU (oneDNN, all host cores).

TEST / BENCH INFRASTRUCTURE ONLY (see keras_ref.py header): bench.py times this as `cpu_baseline` on the GPU box's host
cores; tests use it as a second, independent fp32 implementation.  It stands in for the reference's Keras-2.2.4/TF-1.12
CPU path, which cannot run here; same layer stack, same Keras semantics (TF SAME padding,
BatchNorm eps 1e-3 / momentum 0.99 with the n/(n-(1+eps)) moving-variance factor, keras-form Adam, phase handling of
bbhMahoGANy.py:1248 / :1292 / :1296).  Layout is converted to channels-first internally because that is what oneDNN is
fastest at -- a generous baseline.
"""
import math

import torch
import torch.nn.functional as F

from . import keras_ref as K


def _glorot(shape, gen):
    if len(shape) == 2:
        fi, fo = shape
    else:
        rec = 1
        for s in shape[:-2]:
            rec *= s
        fi, fo = rec * shape[-2], rec * shape[-1]
    lim = math.sqrt(6.0 / (fi + fo))
    return (torch.rand(shape, generator=gen) * 2 - 1) * lim


class KerasAdam(object):
    def __init__(self, params, lr=9e-5, b1=0.5, b2=0.999, eps=1e-7):
        self.params = params
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for p, m, v in zip(self.params, self.m, self.v):
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-lr_t)
            p.grad = None


def _conv1d(x, w, b, stride, padding):
    """x (B, C, L) channels-first; w keras layout (k, Cin, Cout)."""
    k = w.shape[0]
    if padding == 'same':
        _, pl, pr = K.same_pad(x.shape[2], k, stride)
        x = F.pad(x, (pl, pr))
    return F.conv1d(x, w.permute(2, 1, 0), b, stride=stride)


class _BN(object):
    """moving_average: 'ema' = plain exponential average (tf.keras, TF's zero_debias=False); 'tf_zero_debias' = Keras 2.2.4 on TF 1.12
    (keras_ref.bn_moving_update_zero_debias: zero-initialised shadow accumulator, moving = biased / (1 - m^t))."""

    def __init__(self, C, moving_average='ema'):
        self.gamma = torch.ones(C, requires_grad=True)
        self.beta = torch.zeros(C, requires_grad=True)
        self.mm = torch.zeros(C)
        self.mv = torch.ones(C)
        self.moving_average = moving_average
        self.bm, self.bv, self.t = torch.zeros(C), torch.zeros(C), 0

    def __call__(self, x, training, momentum=0.99):
        """x (B, C, L) or (B, C)."""
        if not training:
            return F.batch_norm(x, self.mm, self.mv, self.gamma, self.beta, False, 0.0, K.BN_EPS)
        dims = [0] + list(range(2, x.ndim))
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            mean = x.mean(dims)
            var = x.var(dims, unbiased=False) * (n / (n - (1.0 + K.BN_EPS)))
            if self.moving_average == 'ema':
                self.mm.mul_(momentum).add_(mean, alpha=1 - momentum)
                self.mv.mul_(momentum).add_(var, alpha=1 - momentum)
            else:
                self.bm.mul_(momentum).add_(mean, alpha=1 - momentum)
                self.bv.mul_(momentum).add_(var, alpha=1 - momentum)
                self.t += 1
                corr = 1.0 - momentum ** self.t
                self.mm.copy_(self.bm / corr)
                self.mv.copy_(self.bv / corr)
        return F.batch_norm(x, None, None, self.gamma, self.beta, True, 0.0, K.BN_EPS)


class PENet(object):
    """signal_pe_model (bbhMahoGANy.py:356-404) + compile(mse, Adam(9e-5, 0.5)) (:1119)."""

    def __init__(self, n_pix, seed=1):
        g = torch.Generator().manual_seed(seed)
        self.mc, self.q = [], []
        L = K.conv_out_len(n_pix, 5, 2, 'same')
        cin = 1
        for cout, s, pad in ((64, 2, 'same'), (128, 2, 'valid'), (256, 2, 'valid'), (512, 2, 'valid')):
            self.mc.append([_glorot((5, cin, cout), g).requires_grad_(), torch.zeros(cout, requires_grad=True), s, pad])
            if pad == 'valid':
                L = K.conv_out_len(L, 5, s, 'valid')
            cin = cout
        self.mc_head = [_glorot((L * 512, 1), g).requires_grad_(), torch.zeros(1, requires_grad=True)]
        L, cin = n_pix, 1
        for cout, s, pad in ((64, 1, 'same'), (128, 1, 'valid'), (256, 1, 'valid'), (512, 2, 'valid'), (1024, 2, 'valid')):
            self.q.append([_glorot((5, cin, cout), g).requires_grad_(), torch.zeros(cout, requires_grad=True), s, pad])
            if pad == 'valid':
                L = K.conv_out_len(L, 5, s, 'valid')
            cin = cout
        self.q_head = [_glorot((L * 1024, 1), g).requires_grad_(), torch.zeros(1, requires_grad=True)]
        self.params = [t for l in self.mc for t in l[:2]] + self.mc_head + [t for l in self.q for t in l[:2]] + self.q_head
        self.opt = KerasAdam(self.params)

    def forward(self, x):
        """x (B, n_pix, 1) -> [mc (B,1), q (B,1)]"""
        h0 = x.permute(0, 2, 1)
        h = h0
        for w, b, s, pad in self.mc:
            h = torch.relu(_conv1d(h, w, b, s, pad))
        mc = torch.relu(h.permute(0, 2, 1).reshape(h.shape[0], -1) @ self.mc_head[0] + self.mc_head[1])
        h = h0
        for w, b, s, pad in self.q:
            h = torch.relu(_conv1d(h, w, b, s, pad))
        q = torch.clamp(h.permute(0, 2, 1).reshape(h.shape[0], -1) @ self.q_head[0] + self.q_head[1], 0, 1)
        return mc, q

    def train_on_batch(self, x, y_mc, y_q):
        mc, q = self.forward(x)
        lm = F.mse_loss(mc, y_mc.reshape(-1, 1)); lq = F.mse_loss(q, y_q.reshape(-1, 1))
        (lm + lq).backward()
        self.opt.step()
        return [float(lm.detach() + lq.detach()), float(lm.detach()), float(lq.detach())]


class GAN(object):
    """generator_model (:212-295), signal_discriminator_model (:408-498), MyLayer (:164-188), compile wiring (:1100-1119)."""

    def __init__(self, n_pix, event, seed=2, lr=9e-5, moving_average='ema', bce_grad='clip'):
        """bce_grad: 'clip' = Keras 2.2.4's binary cross-entropy from probabilities (clip to [1e-7, 1 - 1e-7]; the clip's gradient is zero
        outside the interval); 'noclip' = DIAGNOSTIC ONLY (tests/tools/gan_dynamics_cpu.py): the same loss value, gradient taken through the
        logit as if the clip were not there, i.e. never zeroed."""
        g = torch.Generator().manual_seed(seed)
        self.n_pix = n_pix
        self.bce_grad = bce_grad
        self.event = torch.as_tensor(event, dtype=torch.float32).reshape(1, n_pix, 1)
        U = 256 * (n_pix // 2)
        self.g_dense = [_glorot((100, U), g).requires_grad_(), torch.zeros(U, requires_grad=True)]
        self.g_bn0 = _BN(U, moving_average)
        self.g_convs = []
        cin = 256
        for cout, s in ((64, 2), (128, 1), (256, 1), (512, 1), (1024, 1)):
            self.g_convs.append([_glorot((5, cin, cout), g).requires_grad_(), torch.zeros(cout, requires_grad=True), s, _BN(cout, moving_average)])
            cin = cout
        self.g_out = [_glorot((5, 1024, 1), g).requires_grad_(), torch.zeros(1, requires_grad=True)]
        self.d_convs = []
        cin = 1
        for cout in (256, 512):
            self.d_convs.append([_glorot((5, 5, cin, cout), g).requires_grad_(), torch.zeros(cout, requires_grad=True)])
            cin = cout
        self.d_dense = [_glorot(((n_pix // 4) * 2 * 512, 1), g).requires_grad_(), torch.zeros(1, requires_grad=True)]
        self.g_params = self.g_dense + [self.g_bn0.gamma, self.g_bn0.beta]
        for w, b, s, bn in self.g_convs:
            self.g_params += [w, b, bn.gamma, bn.beta]
        self.g_params += self.g_out
        self.d_params = [t for l in self.d_convs for t in l] + self.d_dense
        self.opt_g = KerasAdam(self.g_params, lr=lr)
        self.opt_d = KerasAdam(self.d_params, lr=lr)

    def G(self, z, training):
        h = z @ self.g_dense[0] + self.g_dense[1]
        h = F.dropout(torch.tanh(self.g_bn0(h, training)), 0.2, training)
        h = h.reshape(-1, self.n_pix // 2, 256).permute(0, 2, 1)
        for i, (w, b, s, bn) in enumerate(self.g_convs):
            if i < 2:
                h = h.repeat_interleave(2, dim=2)
            h = F.dropout(torch.tanh(bn(_conv1d(h, w, b, s, 'same'), training)), 0.2, training)
        return _conv1d(h, self.g_out[0], self.g_out[1], 1, 'same').permute(0, 2, 1)          # (B, n_pix, 1)

    def D_logit(self, img, training):
        """img (B, n_pix, 2, 1) channels-last -> pre-sigmoid (B, 1)."""
        h = img.permute(0, 3, 1, 2)
        for w, b in self.d_convs:
            _, pt, pb = K.same_pad(h.shape[2], 5, 2)
            h = F.conv2d(F.pad(h, (2, 2, pt, pb)), w.permute(3, 2, 0, 1), b, stride=(2, 1))
            h = F.dropout(F.leaky_relu(h, 0.2), 0.4, training)
        h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
        return h @ self.d_dense[0] + self.d_dense[1]

    def D(self, img, training):
        return torch.sigmoid(self.D_logit(img, training))

    def bce(self, p, y, logit=None):
        lo, hi = K.CLIP_LO, K.CLIP_HI
        if self.bce_grad == 'noclip':
            # diagnostic: same VALUE as the clipped loss wherever the clip is inactive, gradient sigmoid(logit) - y everywhere
            return F.binary_cross_entropy_with_logits(logit, y)
        return F.binary_cross_entropy(torch.clamp(p, lo, hi), y)

    def iteration(self, real, B):
        """bbhMahoGANy.py:1243-1299 for one batch of B real templates (B, n_pix)."""
        with torch.no_grad():
            fake = self.G(torch.rand(B, 100) * 2 - 1, False)
        resid = self.event - fake
        fake2 = torch.cat([fake, resid], dim=2).flip(0)
        real2 = torch.cat([real.reshape(B, self.n_pix, 1), torch.randn(B, self.n_pix, 1)], dim=2)
        sX = torch.cat([real2, fake2]).reshape(2 * B, self.n_pix, 2, 1)
        sy = torch.cat([torch.ones(B, 1), torch.zeros(B, 1)])
        lgt = self.D_logit(sX, True)
        pd = torch.sigmoid(lgt)
        ld = self.bce(pd, sy, lgt)
        ld.backward()
        self.opt_d.step()
        for p in self.g_params:
            p.grad = None
        x = self.G(torch.rand(B, 100) * 2 - 1, True)
        img = torch.stack([x, self.event - x], dim=2)
        lgt = self.D_logit(img, True)
        pg = torch.sigmoid(lgt)
        lg = self.bce(pg, torch.ones(B, 1), lgt)
        lg.backward()
        for p in self.d_params:
            p.grad = None                                  # D frozen in the combined model
        self.opt_g.step()
        self.last_acc = [float((pg.detach().round() == 1).float().mean()), float((pd.detach().round() == sy).float().mean())]
        return [float(lg.detach()), float(ld.detach())]
