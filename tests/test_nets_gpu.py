"""GPU parity of the three BBH networks against the fp64 oracle (oracle/nets_ref.py), through the Keras-style surface.

Identical initial weights (oracle draws them, rounded to fp32), identical inputs, injected dropout masks.  Tolerances
(stated per assert): loss 1e-5 relative for one train_on_batch; gradients / weights after Adam steps 1e-4 relative to the
largest entry of each tensor (SURVEY section 7 suggested bounds).
"""
import numpy as np
import pytest
import torch

from oracle import keras_ref as K
from oracle import nets_ref as N

pytestmark = pytest.mark.gpu


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def load_stack_into_layers(stack, layers):
    """Copy oracle Stack parameters (and BN moving statistics) into the trainable layers of a gennet_amd model."""
    from gennet_amd.engine import to_device
    with_params = [l for l in layers if l.weights]
    specs = [(li, s) for li, s in enumerate(stack.spec) if s[0] in ('dense', 'conv1d', 'conv2d', 'bn')]
    assert len(with_params) == len(specs)
    for l, (li, s) in zip(with_params, specs):
        ps = [stack.params[i] for i in stack.pidx[li]]
        for p, v in zip(l.params, ps):
            p.data.copy_(to_device(v.astype(np.float32)))
        if s[0] == 'bn':
            l.moving_mean.data.copy_(to_device(stack.state[li][0].astype(np.float32)))
            l.moving_variance.data.copy_(to_device(stack.state[li][1].astype(np.float32)))


def decisions_for(stack, layers, capture):
    """{oracle act-layer index: the GPU's output of that relu / relu_max / LeakyReLU layer} (Stack.forward `decisions`): the branch each
    activation took on the GPU, injected into the oracle for the few elements whose pre-activation is within fp32 rounding of the kink
    (the same role as the injected dropout masks; everywhere else the oracle's own branch must agree, which the tests assert)."""
    with_params = [l for l in layers if l.weights]
    specs = [li for li, s in enumerate(stack.spec) if s[0] in ('dense', 'conv1d', 'conv2d', 'bn')]
    out = {}
    for l, li in zip(with_params, specs):
        if li + 1 < len(stack.spec) and stack.spec[li + 1][0] == 'act' and stack.spec[li + 1][1] in ('relu', 'relu_max', 'leaky'):
            out[li + 1] = capture[l.name].detach().cpu().numpy()
    # MaxPooling2D (`maxpool = True`): the implementation's INPUT of the layer = the output of the nearest executed layer in front of it (a Dropout fused
    # into its producer does not run as a node of its own); the oracle takes its routing for pairs that agree to within the band
    pools = [l for l in layers if l.__class__.__name__ == 'MaxPooling2D']
    for l, li in zip(pools, [li for li, s in enumerate(stack.spec) if s[0] == 'maxpool']):
        j = layers.index(l) - 1
        while layers[j].name not in capture:
            j -= 1
        out[li] = capture[layers[j].name].detach().cpu().numpy()
    return out


def assert_decisions_consistent(*stacks):
    """No activation took a different branch on the GPU than in the oracle OUTSIDE the rounding band, and INSIDE it only as many as fp32
    rounding explains: the band (|pre| <= 1e-5 max|pre|) holds a few 1e-5 of a layer's elements and rounding errors of ~1e-7..1e-6 of
    max|pre| flip a few per cent of those, so a layer may flip at most 2 + 5e-6 of its elements and at most 2 + a quarter of its in-band
    elements -- a systematic error near zero (every in-band element of one sign taken the other way) would flip about half the band.
    Returns the number of in-band flips."""
    flips = 0
    for st in stacks:
        for li, (n_band, n_flip, n_outside, n_total) in st.decision_stats.items():
            assert n_outside == 0, (li, st.spec[li - 1], n_band, n_flip, n_outside)
            assert n_flip <= 2 + 5e-6 * n_total and n_flip <= 2 + 0.25 * n_band, (li, st.spec[li - 1], n_band, n_flip, n_total)
            flips += n_flip
    return flips


def round_stack(stack):
    for p in stack.params:
        p[...] = f32(p)


def spec_out_shape(s, shp):
    """Output shape of one Stack spec entry (batch axis included)."""
    kind = s[0]
    if kind == 'dense':
        return (shp[0], s[2])
    if kind == 'conv1d':
        return (shp[0], K.conv_out_len(shp[1], s[3], s[4], s[5]), s[2])
    if kind == 'conv2d':
        return (shp[0], K.conv_out_len(shp[1], s[3][0], s[4][0], s[5]), K.conv_out_len(shp[2], s[3][1], s[4][1], s[5]), s[2])
    if kind in ('bn', 'act', 'drop'):
        return tuple(shp)
    if kind == 'reshape':
        return (shp[0],) + tuple(s[1])
    if kind == 'flatten':
        return (shp[0], int(np.prod(shp[1:])))
    if kind == 'up':
        return (shp[0], shp[1] * s[1], shp[2])
    if kind == 'maxpool':
        return (shp[0], shp[1] // 2) + tuple(shp[2:])
    raise ValueError(kind)


def stack_masks(stack, x, rng):
    """Keep masks for the dropout layers of a Stack; x = the stack's input or just its shape (the shapes are walked, nothing is computed)."""
    masks = {}
    shp = tuple(x) if isinstance(x, (tuple, list)) else tuple(x.shape)
    for li, s in enumerate(stack.spec):
        if s[0] == 'drop':
            masks[li] = (rng.rand(*shp) >= s[1]).astype(np.float64)
        shp = spec_out_shape(s, shp)
    return masks


def masks_by_name(stack, masks, layers):
    from gennet_amd.layers import Dropout
    drops = [l for l in layers if isinstance(l, Dropout)]
    idx = [li for li, s in enumerate(stack.spec) if s[0] == 'drop']
    assert len(drops) == len(idx)
    return {l.name: masks[li].astype(np.uint8) for l, li in zip(drops, idx)}


@pytest.mark.parametrize("n_pix,B,steps", [(128, 6, 3), (256, 5, 3),
                                          (1024, 4, 2),        # the reference script's own default n_pix (bbhMahoGANy.py:84)
                                          (2048, 4, 2),        # BASELINE configs 1-4 size: the 64 000- and 519 168-input Dense heads, 2048-row convs
                                          (4096, 2, 1),        # BASELINE config 5 size (129 536- and 1 043 456-input heads)
                                          (2048, 64, 1)])      # round 5: a 64-row chunk at BASELINE size directly against the oracle -- the chunk
                                                               # tests/test_bench_sizes_gpu.py compares the 256- / 512- / 1024-row launches of bench.py with
def test_pe_train_on_batch_matches_oracle(n_pix, B, steps):
    from gennet_amd import bbh
    from gennet_amd.engine import Adam
    rng = np.random.RandomState(n_pix)
    ref = N.PENet(n_pix, rng)
    round_stack(ref.mc); round_stack(ref.q)
    ref.mc.params[-1][...] = 25.0; ref.q.params[-1][...] = 0.6      # heads start inside the active range of relu / relu(max 1)
    model = bbh.signal_pe_model(n_pix)
    layers = model.layers
    n_mc = len([s for s in ref.mc.spec if s[0] in ('dense', 'conv1d')])
    with_params = [l for l in layers if l.weights]
    load_stack_into_layers(ref.mc, with_params[:n_mc])
    load_stack_into_layers(ref.q, with_params[n_mc:])
    model.compile(loss='mean_squared_error', optimizer=Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])

    x = f32(rng.randn(B, n_pix, 1)); y_mc = f32(rng.uniform(20, 35, B)); y_q = f32(rng.uniform(0.5, 1, B))
    p_ref = ref.predict(x)
    p = model.predict(x)
    assert rel(p[0], p_ref[0]) < 2e-5 and rel(p[1], p_ref[1]) < 2e-5
    for step in range(steps):
        cap = {}
        out = model.train_on_batch(x, [y_mc, y_q], capture=cap)
        out_ref = ref.train_on_batch(x, y_mc, y_q, decisions=(decisions_for(ref.mc, with_params[:n_mc], cap), decisions_for(ref.q, with_params[n_mc:], cap)))
        del cap
        flips = assert_decisions_consistent(ref.mc, ref.q)
        print('n_pix %d step %d: %d relu branch decisions injected (pre-activation within 1e-5 of zero and taken the other way in fp32)' % (n_pix, step, flips))
        assert len(out) == 5                                          # [total, mc_loss, q_loss, mc_acc, q_acc]
        for a, b in zip(out[:3], out_ref[:3]):
            assert abs(a - b) <= 1e-5 * abs(b) + 1e-7, (step, out, out_ref)
        assert out[3:] == pytest.approx(out_ref[3:])
        if step == 0:
            grads = [p_.grad.cpu().numpy() for l in with_params for p_ in l.params]
            rels = [rel(gq, gr) for gq, gr in zip(grads, ref.last_grads)]
            assert max(rels) < 1e-4, ['%.1e' % r for r in rels]
    # Adam normalises every element's step to ~lr, so an element whose gradient is tiny relative to its tensor's maximum carries
    # the gradient's ABSOLUTE error at full weight: bound = 1e-4 relative + 1 % of the step budget (3 steps x lr)
    ws = [p_.data.cpu().numpy() for l in with_params for p_ in l.params]
    for w, wr in zip(ws, ref.mc.params + ref.q.params):
        assert np.abs(w - wr).max() <= 1e-4 * np.abs(wr).max() + 0.01 * 3 * 9e-5
    # predict after the updates: the whole graph again, with the updated weights
    p_ref = ref.predict(x)
    p = model.predict(x)
    assert rel(p[0], p_ref[0]) < 5e-5 and rel(p[1], p_ref[1]) < 5e-5


def _build_gan(n_pix, rng, filtsize=5, d_config=None):
    from gennet_amd import bbh
    event = f32(rng.randn(n_pix, 1))
    ref = N.GAN(n_pix, event, rng, filtsize=filtsize, d_config=d_config)
    round_stack(ref.G); round_stack(ref.D)
    for st in (ref.G, ref.D):
        for p in st.params:
            if p.ndim == 1:
                p[...] = f32(p + 0.05 * rng.randn(*p.shape))
    nets = bbh.build_and_compile(event, n_pix, do_pe=False, filtsize=filtsize, d_config=d_config)
    load_stack_into_layers(ref.G, nets.generator.layers)
    load_stack_into_layers(ref.D, nets.signal_discriminator.layers)
    return ref, nets, event


D4 = dict(num_lays=4, batchnorm=True, maxpool=True)       # the discriminator's edit-the-file knobs (bbhMahoGANy.py:424-426): Conv2D strides (1,1), BatchNormalization
D6 = dict(num_lays=6, batchnorm=True, maxpool=False)      # after (layer 2) and before (layers 3-6) the LeakyReLU, MaxPooling2D((2,1)) behind every Dropout
D3 = dict(num_lays=3, batchnorm=False, maxpool=True)


@pytest.mark.parametrize("n_pix,B,iters,filtsize,d_config", [(64, 4, 2, 5, None),
                                          (64, 4, 2, 5, D4), (64, 3, 1, 5, D6), (128, 3, 1, 5, D3),
                                          (64, 4, 2, 10, None),  # `filtsize = 5 # 10 is best` (bbhMahoGANy.py:228): every generator conv as 5 taps over (x, x shifted by 5)
                                          (256, 3, 1, 7, None),  # an odd filter size: 4 taps + a zero-padded one
                                          (1024, 3, 1, 5, None), # the reference script's own default n_pix; odd batch
                                          (4096, 2, 1, 5, None), # BASELINE config 5 size: Dense(100 -> 524 288), 4096-row convs, 1 048 576-input head
                                          (2048, 4, 1, 5, None), # BASELINE size: Dense(100 -> 262 144) + feature-BN over B, the 524 288-input head,
                                                               # channel-BN over 2048*B rows, fused dgrad epilogues, fold_bn predict inside the graph
                                          (2048, 64, 1, 5, None)])   # round 5: D step on 2 x 64 rows, G step on 64 rows at BASELINE size against
                                                               # the oracle: closes the chain oracle <-> 64-row chunk <-> bench.py's batch sizes
def test_gan_iteration_matches_oracle(n_pix, B, iters, filtsize, d_config):
    """Full GAN iterations (bbhMahoGANy.py:1241-1299) with injected masks: D step on [real | fake], then G step through the
    frozen D; at the small size a second iteration exercises the moving statistics and both Adam states."""
    from gennet_amd import bbh
    from gennet_amd.engine import to_device
    rng = np.random.RandomState(3)
    ref, nets, event = _build_gan(n_pix, rng, filtsize, d_config)
    ev_dev = to_device(event.reshape(-1))
    G, D, DG = nets.generator, nets.signal_discriminator, nets.signal_discriminator_on_generator
    # the combined model trains exactly the generator's weights; the discriminator model its own
    assert set(id(p) for p in DG._train_params) == set(id(p) for l in G.layers for p in l.params)
    assert set(id(p) for p in D._train_params) == set(id(p) for l in D.layers for p in l.params)
    for it in range(iters):
        z = f32(rng.uniform(-1, 1, (B, 100)))
        fake_ref = ref.generate(z)
        fake = G.predict(z)
        assert rel(fake, fake_ref) < 5e-5
        real = f32(rng.randn(B, n_pix)); noise = f32(rng.randn(B, n_pix, 1))
        sX_ref, sy = ref.assemble_d_batch(real, noise, fake_ref)
        sX, syd = bbh.assemble_discriminator_batch(to_device(real), to_device(noise), to_device(fake_ref), ev_dev)
        assert rel(sX.cpu().numpy(), sX_ref) < 1e-6
        assert syd.cpu().numpy().tolist() == sy
        d_masks = stack_masks(ref.D, sX_ref, rng)
        cap = {}
        out = D.train_on_batch(sX_ref, sy, dropout_masks=masks_by_name(ref.D, d_masks, D.layers), capture=cap)
        out_ref = ref.d_train_on_batch(sX_ref, sy, d_masks, decisions_for(ref.D, D.layers, cap))
        del cap
        assert_decisions_consistent(ref.D)
        assert abs(out[0] - out_ref[0]) <= 2e-5 * abs(out_ref[0]) and out[1] == pytest.approx(out_ref[1])
        dgr = [p.grad.cpu().numpy() for l in D.layers for p in l.params]
        dmax = max(np.abs(gr).max() for gr in ref.last_d_grads)
        for gq, gr in zip(dgr, ref.last_d_grads):
            # (`batchnorm = True`: a bias feeding a BatchNormalization has an exactly-zero gradient in exact arithmetic: absolute floor as in the G step below)
            assert np.abs(gq - gr).max() <= 2e-4 * np.abs(gr).max() + (1e-6 * dmax if d_config and d_config['batchnorm'] else 0.0), rel(gq, gr)
        assert np.all(dgr[0][:, 0] == 0) and np.all(dgr[0][:, 4] == 0)      # dead width taps of the 5x5 kernel

        z2 = f32(rng.uniform(-1, 1, (B, 100)))
        g_masks = stack_masks(ref.G, z2, rng)
        d_masks2 = stack_masks(ref.D, (B, n_pix, 2, 1), rng)                      # the combined model's D sees MyLayer(G(z2)): (B, n_pix, 2, 1)
        names = dict(masks_by_name(ref.G, g_masks, G.layers)); names.update(masks_by_name(ref.D, d_masks2, D.layers))
        d_before = [p.data.clone() for l in D.layers for p in l.params]
        cap = {}
        out = DG.train_on_batch(z2, [1] * B, dropout_masks=names, capture=cap)
        out_ref = ref.g_train_on_batch(z2, [1] * B, g_masks, d_masks2, decisions_for(ref.D, D.layers, cap))
        del cap
        print('n_pix %d G step: %d LeakyReLU branch decisions injected' % (n_pix, assert_decisions_consistent(ref.D)))
        assert abs(out[0] - out_ref[0]) <= 2e-5 * abs(out_ref[0]) and out[1] == pytest.approx(out_ref[1])
        ggr = [p.grad.cpu().numpy() for l in G.layers for p in l.params]
        gmax = max(np.abs(gr).max() for gr in ref.last_g_grads)
        for k, (gq, gr) in enumerate(zip(ggr, ref.last_g_grads)):
            # biases feeding a BatchNorm have an exactly-zero gradient in exact arithmetic: absolute floor 1e-6 of the largest gradient
            assert np.abs(gq - gr).max() <= 3e-4 * np.abs(gr).max() + 1e-6 * gmax, (k, rel(gq, gr))
        for a, b in zip(d_before, [p.data for l in D.layers for p in l.params]):
            assert torch.equal(a, b)                                          # D frozen as of compile time
    # weights and BN moving statistics after two iterations
    for st, model in ((ref.G, G), (ref.D, D)):
        ws = [p.data.cpu().numpy() for l in model.layers for p in l.params]
        for w, wr in zip(ws, st.params):
            bound = 2e-4 * np.abs(wr).max() + 0.02 * 2 * 9e-5                               # see the Adam note in the PE test
            if d_config is None:
                assert np.abs(w - wr).max() <= bound
            else:
                # behind a deep normalised discriminator the generator's gradients are small enough that for some elements the ABSOLUTE fp32 error exceeds
                # Adam's epsilon (1e-7): the first step of such an element is lr * g / (|g| + eps) with g of either sign -- up to 2 lr apart; a few per tensor
                bad = np.abs(w - wr) > bound
                assert bad.mean() <= 1e-3 and np.abs(w - wr).max() <= 2 * iters * 9e-5, (float(bad.mean()), float(np.abs(w - wr).max()))
    # generator.predict after the update(s): moving statistics through fold_bn inside the full graph
    z3 = f32(rng.uniform(-1, 1, (B, 100)))
    assert rel(G.predict(z3), ref.generate(z3)) < 1e-4
    for st, model in ((ref.G, G), (ref.D, D)):
        # (the discriminator's BatchNormalization layers, `batchnorm = True`, see the batch in both steps: its own and -- frozen, but in the training phase --
        # the generator's, and update their moving statistics in both, as the train functions Keras builds at the first train_on_batch do)
        bns = [l for l in model.layers if hasattr(l, 'moving_mean')]
        bn_idx = [li for li, s in enumerate(st.spec) if s[0] == 'bn']
        assert len(bns) == len(bn_idx)
        for l, li in zip(bns, bn_idx):
            assert rel(l.moving_mean.data.cpu().numpy(), st.state[li][0]) < 1e-4 or np.abs(st.state[li][0]).max() < 1e-6
            assert rel(l.moving_variance.data.cpu().numpy(), st.state[li][1]) < 1e-4


def test_unfused_layers_match_fused_epilogues():
    """Standalone Activation / Dropout kernels (graph where fusion is impossible) equal the fused epilogue path."""
    from gennet_amd.engine import Adam, Input, Model, Sequential
    from gennet_amd.layers import Activation, Conv1D, Dense, Dropout, Flatten
    rng = np.random.RandomState(21)
    x = f32(rng.randn(3, 40, 1))
    inp = Input(shape=(40, 1))
    c = Conv1D(16, 5, padding='same', name='c_shared')
    h = c(inp)
    a = Activation('tanh')(h)          # h has two consumers below -> the activation cannot be fused into the conv
    f1 = Flatten()(a); f2 = Flatten()(h)
    o1 = Dense(1)(f1); o2 = Dense(1)(f2)
    m = Model(inputs=inp, outputs=[o1, o2])
    m._plan()
    assert all(n.fused_act is None for n in m.nodes if n.layer is c)
    w, b = c.get_weights()
    y1, y2 = m.predict(x)
    d1, d2 = [l for l in m.layers if isinstance(l, Dense)]
    hh = K.conv1d_fwd(x, f32(w), f32(b), 1, 'same')
    r1 = K.dense_fwd(np.tanh(hh).reshape(3, -1), f32(d1.get_weights()[0]), f32(d1.get_weights()[1]))
    r2 = K.dense_fwd(hh.reshape(3, -1), f32(d2.get_weights()[0]), f32(d2.get_weights()[1]))
    assert rel(y1, r1) < 2e-5 and rel(y2, r2) < 2e-5
    m.compile(loss='mean_squared_error', optimizer=Adam(lr=1e-3), metrics=['accuracy'])
    out = m.train_on_batch(x, [np.zeros(3), np.zeros(3)])
    assert np.isfinite(out).all()


@pytest.mark.parametrize("stride", [1, 2])
def test_folded_upsample_trains_like_the_materialised_one(stride, monkeypatch):
    """The planner's UpSampling1D -> Conv1D fold (weights folded per step, gradient unfolded) against the same graph with the upsampled
    tensor materialised: predict, losses and the weights after two Adam steps, with the fused tanh + dropout epilogue on the folded conv
    and an UpSampling1D that cannot fold (3-tap 'valid' consumer) in the same graph."""
    from gennet_amd import engine, layers
    from gennet_amd.engine import Adam, Sequential
    from gennet_amd.layers import Activation, Conv1D, Dense, Dropout, Flatten, UpSampling1D
    rng = np.random.RandomState(5 + stride)
    B, L = 4, 20
    x = f32(rng.randn(B, L, 4)); y = f32(rng.randn(B))
    Lc = 2 * L // stride
    mask = (rng.rand(B, Lc, 16) >= 0.2).astype(np.uint8)

    def run(fold):
        monkeypatch.setattr(layers, '_NO_UPFOLD', not fold)
        engine.set_init_seed(9)
        m = Sequential()
        m.add(Conv1D(8, 5, padding='same', input_shape=(L, 4)))
        m.add(UpSampling1D(size=2))
        m.add(Conv1D(16, 5, strides=stride, padding='same'))
        m.add(Activation('tanh'))
        m.add(Dropout(0.2, name='drop_fold'))
        m.add(UpSampling1D(size=2))
        m.add(Conv1D(8, 3, padding='valid'))
        m.add(Flatten())
        m.add(Dense(1))
        m.compile(loss='mean_squared_error', optimizer=Adam(lr=1e-2))
        m._plan()
        assert [n.fold_up is not None for n in m.nodes if isinstance(n.layer, Conv1D)] == [False, fold, False]
        p0 = m.predict(x)
        losses = [m.train_on_batch(x, y, dropout_masks={'drop_fold': mask}) for _ in range(2)]
        return p0, losses, m.get_weights()

    pa, la, wa = run(True)
    pb, lb, wb = run(False)
    assert rel(pa, pb) < 2e-5
    assert np.allclose(la, lb, rtol=1e-4)
    for a, b in zip(wa, wb):
        assert rel(a, b) < 1e-4


def test_deferred_output_conv_gradient_trains_like_the_materialised_one(monkeypatch):
    """The planner's deferral of the 1-filter output conv's data gradient into the BatchNormalization backward (ops.ConvGrad1) against
    the same graph with the gradient tensor written and read: losses and weights after two Adam steps."""
    from gennet_amd import engine, layers
    from gennet_amd.engine import Adam, Sequential
    from gennet_amd.layers import Activation, BatchNormalization, Conv1D, Dense, Dropout, Flatten
    rng = np.random.RandomState(17)
    B, L = 4, 24
    x = f32(rng.randn(B, L, 4)); y = f32(rng.randn(B))
    mask = (rng.rand(B, L, 16) >= 0.2).astype(np.uint8)

    def run(lazy):
        monkeypatch.setattr(layers, '_NO_LAZYGRAD', not lazy)
        engine.set_init_seed(4)
        m = Sequential()
        m.add(Conv1D(16, 5, padding='same', input_shape=(L, 4)))
        m.add(BatchNormalization(momentum=0.99))
        m.add(Activation('tanh'))
        m.add(Dropout(0.2, name='drop_last'))
        m.add(Conv1D(1, 5, padding='same'))
        m.add(Activation('linear'))
        m.add(Flatten())
        m.add(Dense(1))
        m.compile(loss='mean_squared_error', optimizer=Adam(lr=1e-2))
        m._plan()
        assert [n.lazy_bn >= 0 for n in m.nodes if isinstance(n.layer, Conv1D)] == [False, lazy]
        losses = [m.train_on_batch(x, y, dropout_masks={'drop_last': mask}) for _ in range(2)]
        return losses, m.get_weights()

    la, wa = run(True)
    lb, wb = run(False)
    assert np.allclose(la, lb, rtol=1e-5)
    for a, b in zip(wa, wb):
        assert rel(a, b) < 1e-4


def test_save_load_roundtrip(tmp_path):
    from gennet_amd import bbh
    from gennet_amd.engine import Adam, load_model
    rng = np.random.RandomState(5)
    m = bbh.signal_pe_model(128)
    m.compile(loss='mean_squared_error', optimizer=Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])
    x = f32(rng.randn(4, 128, 1))
    m.train_on_batch(x, [np.full(4, 30.0), np.full(4, 0.8)])
    path = str(tmp_path / 'signal_pe.h5')
    m.save(path, True)
    m2 = load_model(path)
    for a, b in zip(m.predict(x), m2.predict(x)):
        assert np.array_equal(a, b)
    r1 = m.train_on_batch(x, [np.full(4, 30.0), np.full(4, 0.8)])
    r2 = m2.train_on_batch(x, [np.full(4, 30.0), np.full(4, 0.8)])
    assert r1 == r2                                                       # optimizer state restored bit-for-bit
    g = bbh.generator_model(64)
    wpath = str(tmp_path / 'generator.h5')
    g.save_weights(wpath, True)
    g2 = bbh.generator_model(64)
    g2.load_weights(wpath)
    z = f32(rng.uniform(-1, 1, (3, 100)))
    assert np.array_equal(g.predict(z), g2.predict(z))
