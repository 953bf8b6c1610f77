"""Pins the oracle's hand-written forward/backward passes against torch-CPU autograd in fp64.

Keras/TF are absent (parity unpinned, see oracle/keras_ref.py); torch's conv/BN autograd is the independent
implementation the restatement is checked against.  Tolerance: 1e-10 relative in fp64.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import keras_ref as K
from oracle import nets_ref as N

RTOL = 1e-10


def t64(a):
    return torch.tensor(np.asarray(a, np.float64), dtype=torch.float64, requires_grad=True)


def close(a, b, rtol=RTOL, atol=1e-14):
    a = np.asarray(a); b = np.asarray(b)
    scale = max(np.abs(b).max(), 1e-30)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= rtol * scale + atol, (np.abs(a - b).max(), scale)


@pytest.mark.parametrize("L,k,s,padding", [(16, 5, 1, 'same'), (16, 5, 2, 'same'), (17, 5, 2, 'same'),
                                           (16, 5, 1, 'valid'), (21, 5, 2, 'valid'), (22, 5, 2, 'valid')])
def test_conv1d_matches_torch(L, k, s, padding):
    rng = np.random.RandomState(0)
    x = rng.randn(3, L, 4); W = rng.randn(k, 4, 6); b = rng.randn(6)
    y = K.conv1d_fwd(x, W, b, s, padding)
    dy = rng.randn(*y.shape)
    dx, dW, db = K.conv1d_bwd(x, W, dy, s, padding)
    xt, Wt, bt = t64(x), t64(W), t64(b)
    if padding == 'same':
        out, pl, pr = K.same_pad(L, k, s)
    else:
        pl = pr = 0
    xin = F.pad(xt.permute(0, 2, 1), (pl, pr))
    yt = F.conv1d(xin, Wt.permute(2, 1, 0), bt, stride=s).permute(0, 2, 1)
    close(y, yt.detach().numpy())
    yt.backward(torch.tensor(dy))
    close(dx, xt.grad.numpy()); close(dW, Wt.grad.numpy()); close(db, bt.grad.numpy())


def test_same_pad_tf_rule():
    # s=1 -> (2,2); s=2 on even L -> (1,2)
    assert K.same_pad(2048, 5, 1) == (2048, 2, 2)
    assert K.same_pad(2048, 5, 2) == (1024, 1, 2)
    assert K.same_pad(1024, 5, 2) == (512, 1, 2)


def test_conv2d_matches_torch_and_fold():
    rng = np.random.RandomState(1)
    x = rng.randn(2, 12, 2, 3); W = rng.randn(5, 5, 3, 4); b = rng.randn(4)
    y = K.conv2d_fwd(x, W, b, (2, 1), 'same')
    assert y.shape == (2, 6, 2, 4)
    dy = rng.randn(*y.shape)
    dx, dW, db = K.conv2d_bwd(x, W, dy, (2, 1), 'same')
    xt, Wt, bt = t64(x), t64(W), t64(b)
    _, pt, pb = K.same_pad(12, 5, 2)
    xin = F.pad(xt.permute(0, 3, 1, 2), (2, 2, pt, pb))
    yt = F.conv2d(xin, Wt.permute(3, 2, 0, 1), bt, stride=(2, 1)).permute(0, 2, 3, 1)
    close(y, yt.detach().numpy())
    yt.backward(torch.tensor(dy))
    close(dx, xt.grad.numpy()); close(dW, Wt.grad.numpy()); close(db, bt.grad.numpy())
    # dead width taps never see data
    assert np.all(dW[:, 0] == 0) and np.all(dW[:, 4] == 0)
    # width-2 fold: Conv2D == Conv1D over H with (w,c) channels
    Wf = K.fold_conv2d_w2(W)
    xf = x.reshape(2, 12, 6)
    yf = K.conv1d_fwd(xf, Wf, np.tile(b, 2), 2, 'same')
    close(yf.reshape(y.shape), y)
    dxf, dWf, dbf = K.conv1d_bwd(xf, Wf, dy.reshape(2, 6, 8), 2, 'same')
    close(dxf.reshape(x.shape), dx)
    close(K.unfold_conv2d_w2_grad(dWf, 3, 4), dW)
    close(dbf.reshape(2, 4).sum(0), db)


def test_bn_train_bwd_matches_torch():
    rng = np.random.RandomState(2)
    x = rng.randn(5, 7, 3); g = rng.rand(3) + 0.5; bta = rng.randn(3)
    y, cache, mean, var = K.bn_train_fwd(x, g, bta)
    dy = rng.randn(*y.shape)
    dx, dg, db = K.bn_train_bwd(dy, cache, g)
    xt, gt, bt = t64(x), t64(g), t64(bta)
    yt = F.batch_norm(xt.permute(0, 2, 1), None, None, gt, bt, True, 0.0, K.BN_EPS).permute(0, 2, 1)
    close(y, yt.detach().numpy())
    yt.backward(torch.tensor(dy))
    close(dx, xt.grad.numpy()); close(dg, gt.grad.numpy()); close(db, bt.grad.numpy())
    # feature-BN (2-D input: reduce over batch only)
    x2 = rng.randn(6, 4); y2, c2, m2, v2 = K.bn_train_fwd(x2, np.ones(4), np.zeros(4))
    close(m2, x2.mean(0)); close(v2, x2.var(0))


def test_bn_moving_update_keras_correction():
    mm, mv = K.bn_moving_update(np.zeros(2), np.ones(2), np.array([1.0, 2.0]), np.array([4.0, 9.0]), 10, 0.99)
    close(mm, [0.01, 0.02])
    close(mv, 0.99 + 0.01 * np.array([4.0, 9.0]) * 10 / (10 - 1.001))


@pytest.mark.parametrize("kind,param", [('relu', 0), ('relu_max', 1.0), ('leaky', 0.2), ('tanh', 0), ('sigmoid', 0)])
def test_activations(kind, param):
    rng = np.random.RandomState(3)
    x = rng.randn(50) * 2
    xt = t64(x)
    ref = {'relu': torch.relu, 'relu_max': lambda v: torch.clamp(v, 0, param), 'leaky': lambda v: F.leaky_relu(v, param),
           'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[kind](xt)
    y = K.act_fwd(x, kind, param)
    close(y, ref.detach().numpy())
    dy = rng.randn(50)
    ref.backward(torch.tensor(dy))
    close(K.act_bwd(dy, y, kind, param), xt.grad.numpy())


def test_losses_match_torch():
    rng = np.random.RandomState(4)
    p = rng.rand(9, 1) * 0.98 + 0.01; y = (rng.rand(9, 1) > 0.5).astype(np.float64)
    l, dp = K.bce_loss(p, y)
    pt = t64(p)
    lt = F.binary_cross_entropy(pt, torch.tensor(y))
    close(l, lt.item()); lt.backward(); close(dp, pt.grad.numpy(), 1e-9)
    # clip region: probability saturated -> loss finite, gradient zero (TF clip_by_value)
    l2, dp2 = K.bce_loss(np.array([[1.0], [0.0]]), np.array([[0.0], [1.0]]))
    assert np.isfinite(l2) and np.all(dp2 == 0)
    close(l2, 0.5 * (-np.log(1 - K.CLIP_HI) - np.log(K.CLIP_LO)), 1e-9)
    l, d = K.mse_loss(p, y)
    pt = t64(p); lt = F.mse_loss(pt, torch.tensor(y)); lt.backward()
    close(l, lt.item()); close(d, pt.grad.numpy())
    assert K.binary_accuracy(np.array([[0.6], [0.4], [0.5]]), np.array([[1.0], [1.0], [0.0]])) == pytest.approx(2 / 3)


def test_adam_keras_form():
    p = np.array([1.0, -2.0]); g = np.array([0.5, -0.25]); m = np.zeros(2); v = np.zeros(2)
    p1, m1, v1 = K.adam_step(p, g, m, v, 1)
    lr32, b2 = float(np.float32(9e-5)), float(np.float32(0.999))
    lr_t = lr32 * np.sqrt(1 - b2) / (1 - 0.5)
    close(p1, p - lr_t * (0.5 * g) / (np.sqrt((1 - b2) * g * g) + 1e-7))


def _torch_stack_forward(stack, x, masks):
    """Re-run a Stack spec with torch autograd (training mode), sharing the numpy parameters."""
    ps = [t64(p) for p in stack.params]
    h = t64(x)
    x_in = h
    for li, s in enumerate(stack.spec):
        p = [ps[i] for i in stack.pidx[li]]
        if s[0] == 'dense':
            h = h @ p[0] + p[1]
        elif s[0] == 'conv1d':
            L = h.shape[1]
            pl, pr = (K.same_pad(L, s[3], s[4])[1:] if s[5] == 'same' else (0, 0))
            h = F.conv1d(F.pad(h.permute(0, 2, 1), (pl, pr)), p[0].permute(2, 1, 0), p[1], stride=s[4]).permute(0, 2, 1)
        elif s[0] == 'conv2d':
            _, pt, pb = K.same_pad(h.shape[1], 5, 2)
            h = F.conv2d(F.pad(h.permute(0, 3, 1, 2), (2, 2, pt, pb)), p[0].permute(3, 2, 0, 1), p[1], stride=s[4]).permute(0, 2, 3, 1)
        elif s[0] == 'bn':
            if h.ndim == 2:
                h = F.batch_norm(h, None, None, p[0], p[1], True, 0.0, K.BN_EPS)
            else:
                h = F.batch_norm(h.permute(0, 2, 1), None, None, p[0], p[1], True, 0.0, K.BN_EPS).permute(0, 2, 1)
        elif s[0] == 'act':
            h = {'linear': lambda v: v, 'relu': torch.relu, 'relu_max': lambda v: torch.clamp(v, 0, s[2]),
                 'leaky': lambda v: F.leaky_relu(v, s[2]), 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[s[1]](h)
        elif s[0] == 'drop':
            h = h * torch.tensor(masks[li]) / (1 - s[1])
        elif s[0] == 'reshape':
            h = h.reshape((h.shape[0],) + tuple(s[1]))
        elif s[0] == 'flatten':
            h = h.reshape(h.shape[0], -1)
        elif s[0] == 'up':
            h = h.repeat_interleave(s[1], dim=1)
    return x_in, ps, h


def _masks(stack, x, rng):
    """Walk the stack layer by layer (inference phase) to learn shapes; draw keep masks for dropout layers."""
    masks = {}
    h = x
    for li, s in enumerate(stack.spec):
        one = N.Stack.__new__(N.Stack)
        one.spec = [s]; one.params = [stack.params[i] for i in stack.pidx[li]]; one.pidx = [list(range(len(one.params)))]
        one.state = {0: stack.state[li]} if li in stack.state else {}
        if s[0] == 'drop':
            masks[li] = (rng.rand(*h.shape) >= s[1]).astype(np.float64)
        h = one.forward(h, False)
    return masks


def test_generator_and_discriminator_grads_match_torch():
    n_pix = 32
    rng = np.random.RandomState(5)
    event = rng.randn(n_pix, 1)
    gan = N.GAN(n_pix, event, rng)
    for p in gan.G.params + gan.D.params:          # non-trivial biases / gammas
        if p.ndim == 1:
            p += 0.1 * rng.randn(*p.shape)
    B = 3
    z = rng.uniform(-1, 1, (B, 100))
    g_masks = _masks(gan.G, z, rng)
    fake = gan.G.forward(z, True, g_masks, update_moving=False)
    img = K.mylayer_fwd(fake, gan.event)
    d_masks = _masks(gan.D, img, rng)
    p = gan.D.forward(img, True, d_masks)
    loss, dp = K.bce_loss(p, np.ones((B, 1)))
    dimg, dgrads = gan.D.backward(dp)
    _, ggrads = gan.G.backward(K.mylayer_bwd(dimg))
    # torch replica
    zt, gps, ft = _torch_stack_forward(gan.G, z, g_masks)
    imgt = torch.stack([ft, torch.tensor(gan.event) - ft], dim=2)
    _, dps, pt = _torch_stack_forward(gan.D, imgt.detach().numpy(), d_masks)
    # chain manually: D on torch image requires one graph -> rebuild D on imgt directly
    h = imgt
    dps = [t64(q) for q in gan.D.params]
    for li, s in enumerate(gan.D.spec):
        q = [dps[i] for i in gan.D.pidx[li]]
        if s[0] == 'conv2d':
            _, ptp, pbp = K.same_pad(h.shape[1], 5, 2)
            h = F.conv2d(F.pad(h.permute(0, 3, 1, 2), (2, 2, ptp, pbp)), q[0].permute(3, 2, 0, 1), q[1], stride=s[4]).permute(0, 2, 3, 1)
        elif s[0] == 'act':
            h = F.leaky_relu(h, s[2]) if s[1] == 'leaky' else torch.sigmoid(h)
        elif s[0] == 'drop':
            h = h * torch.tensor(d_masks[li]) / (1 - s[1])
        elif s[0] == 'flatten':
            h = h.reshape(h.shape[0], -1)
        elif s[0] == 'dense':
            h = h @ q[0] + q[1]
    close(p, h.detach().numpy(), 1e-9)
    lt = F.binary_cross_entropy(h, torch.ones(B, 1, dtype=torch.float64))
    close(loss, lt.item(), 1e-9)
    lt.backward()
    for g, q in zip(dgrads, dps):
        close(g, q.grad.numpy(), 1e-8)
    for g, q in zip(ggrads, gps):
        close(g, q.grad.numpy(), 1e-7)


def test_pe_train_step_matches_torch():
    n_pix = 64
    rng = np.random.RandomState(6)
    pe = N.PENet(n_pix, rng)
    x = rng.randn(4, n_pix, 1)
    y_mc = rng.uniform(20, 35, 4); y_q = rng.uniform(0.5, 1, 4)
    # bias the dense heads so that the relu / relu_max outputs are active
    pe.mc.params[-1][...] = 1.0; pe.q.params[-1][...] = 0.5
    p0 = [p.copy() for p in pe.mc.params + pe.q.params]
    out = pe.train_on_batch(x, y_mc, y_q)
    tot = 0
    gts = []
    for st, y in ((pe.mc, y_mc), (pe.q, y_q)):
        saved = [p.copy() for p in st.params]
        for p, q in zip(st.params, p0[:len(st.params)] if st is pe.mc else p0[len(pe.mc.params):]):
            p[...] = q
        _, ps, h = _torch_stack_forward(st, x, {})
        l = F.mse_loss(h, torch.tensor(y.reshape(-1, 1)))
        l.backward(); tot += l.item()
        gts += [q.grad.numpy() for q in ps]
        for p, q in zip(st.params, saved):
            p[...] = q
    close(out[0], tot, 1e-9)
    for g, gt in zip(pe.last_grads, gts):
        close(g, gt, 1e-8)
    # one keras-form Adam step from zero state: |dp| = lr_t * 0.5 g / (sqrt(0.001 g^2) + eps)
    # (lr, beta_2 are float32 variables in Keras: the values that take part are the float32-rounded ones)
    lr32, b2 = float(np.float32(9e-5)), float(np.float32(0.999))
    lr_t = lr32 * np.sqrt(1 - b2) / (1 - 0.5)
    for p_new, p_old, g in zip(pe.mc.params + pe.q.params, p0, gts):
        close(p_new, p_old - lr_t * 0.5 * g / (np.sqrt((1 - b2) * g * g) + 1e-7), 1e-9)


def test_bn_zero_debias_is_the_debiased_average_of_the_batch_values():
    """oracle/keras_ref.bn_moving_update_zero_debias restates TF 1.12's assign_moving_average(zero_debias=True): after t updates the
    moving value equals (1-m) * sum_k m^(t-k) v_k / (1 - m^t) whatever it was initialised to; the plain form keeps m^t of its start."""
    from oracle import keras_ref as K
    rng = np.random.RandomState(0)
    C, n, m = 5, 40, 0.99
    mm, mv = np.zeros(C), np.ones(C)
    zd = [np.zeros(C), np.zeros(C), 0]
    em, ev = np.zeros(C), np.ones(C)
    means, vars_ = [], []
    for t in range(1, 31):
        mean, var = rng.randn(C), rng.rand(C) + 0.5
        means.append(mean); vars_.append(var * n / (n - (1 + K.BN_EPS)))
        mm, mv, zd = K.bn_moving_update_zero_debias(mm, mv, zd, mean, var, n, m)
        em, ev = K.bn_moving_update(em, ev, mean, var, n, m)
        w = np.array([(1 - m) * m ** (t - k) for k in range(1, t + 1)])
        assert np.allclose(mm, (w[:, None] * np.array(means)).sum(0) / (1 - m ** t), rtol=1e-12, atol=1e-14)
        assert np.allclose(mv, (w[:, None] * np.array(vars_)).sum(0) / (1 - m ** t), rtol=1e-12)
        assert np.allclose(ev, m ** t * 1.0 + (w[:, None] * np.array(vars_)).sum(0), rtol=1e-12)
        if t == 1:
            assert np.allclose(mv, vars_[0], rtol=1e-13)
    assert zd[2] == 30
