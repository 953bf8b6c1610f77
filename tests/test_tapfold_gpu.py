"""Conv1D with more than 5 taps -- the reference's own edit-the-file knob `filtsize = 5 # 10 is best` (bbhMahoGANy.py:228; the Conv1D(.., filtsize, ..) layers of
:250-292; 16 taps in the reference's saved Keras models) -- on the <= 5-tap matrix-core kernels (csrc/tap_fold.hip: every further tap group as a
further channel group over the shifted input), against the fp64 oracle at the kernels' own tolerance: forward with the activation epilogue, data gradient, weight and bias gradient, both strides,
both paddings, the fold kernels themselves bit for bit, and a Conv1D layer through the Keras-style surface (predict + two Adam steps).  The generator with
filtsize 10 / 7 against the oracle: tests/test_nets_gpu.py::test_gan_iteration_matches_oracle."""
import numpy as np
import pytest
import torch

from oracle import keras_ref as K

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def g(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=torch.device('cuda:0'))


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


def rel(t, ref):
    a = t.detach().cpu().numpy().astype(np.float64) if isinstance(t, torch.Tensor) else np.asarray(t, np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


CASES = [
    # B, L, Cin, Cout, k, stride, padding
    (2, 64, 64, 64, 10, 1, 'same'),       # pad_left 4: taps 0-4 on x shifted left by 4, taps 5-9 on x shifted right by 1; 5 taps x 128 channels: transform-domain kernel
    (3, 133, 64, 128, 10, 1, 'valid'),
    (2, 150, 128, 64, 10, 2, 'same'),     # stride 2: the even/odd-row kernels on the folded input
    (3, 133, 64, 128, 10, 2, 'valid'),
    (2, 77, 32, 64, 7, 1, 'same'),        # odd k: 4 taps, the last one half zero
    (2, 77, 64, 64, 7, 2, 'valid'),
    (2, 90, 16, 32, 6, 1, 'same'),        # 3 taps
    (4, 40, 8, 4, 9, 1, 'same'),          # few channels: the small kernels
    (2, 64, 128, 1, 10, 1, 'same'),       # the generator's output conv with filtsize 10 (1 filter)
    (1, 2048, 256, 512, 10, 1, 'same'),   # generator layer 4 at BASELINE length, one element
    (2, 100, 16, 32, 16, 1, 'same'),      # 4 groups of 4 taps (kernel_size 16: the Conv1D layers of the reference's saved d_model.hdf5 family)
    (2, 100, 16, 32, 13, 2, 'valid'),     # 3 groups of 5 taps, two zero taps
    (2, 64, 1, 16, 16, 1, 'same'),        # one input channel, 16 taps (the first Conv1D of the reference's saved models): 4 taps over 4 channels, the small-channel kernels
    (3, 50, 2, 8, 8, 2, 'valid'),
]


@pytest.mark.parametrize("B,L,Cin,Cout,k,stride,padding", CASES)
def test_conv_6_to_10_taps_against_oracle(B, L, Cin, Cout, k, stride, padding):
    from gennet_amd import ops
    rng = np.random.RandomState(13 * L + k + stride)
    x = f32(rng.randn(B, L, Cin)); lim = np.sqrt(6.0 / (k * (Cin + Cout)))
    w = f32(rng.uniform(-lim, lim, (k, Cin, Cout))); b = f32(rng.randn(Cout) * 0.1)
    Lout, pl = ops.conv_geometry(L, k, stride, padding)
    G, h = ops.tap_groups(k)
    y_ref = np.tanh(K.conv1d_fwd(x, w, b, stride, padding))
    dy = f32(rng.randn(B, Lout, Cout))
    dx_ref, dw_ref, db_ref = K.conv1d_bwd(x, w, dy, stride, padding)
    # the fold kernels: pure data movement, bit for bit against numpy
    x2 = ops.conv1d_tapfold_x(g(x), k, pl)
    xp = np.zeros((B, L + pl + G * h, Cin)); xp[:, pl:pl + L] = x
    assert np.array_equal(x2.cpu().numpy(), np.concatenate([xp[:, gi * h:gi * h + L + pl] for gi in range(G)], axis=2).astype(np.float32))
    w2 = ops.conv1d_tapfold_w(g(w))
    wp = np.zeros((G * h, Cin, Cout)); wp[:k] = w
    assert np.array_equal(w2.cpu().numpy(), np.concatenate([wp[gi * h:(gi + 1) * h] for gi in range(G)], axis=1).astype(np.float32))
    assert np.array_equal(ops.conv1d_tapunfold_dw(w2, k).cpu().numpy(), w.astype(np.float32))
    # forward, data gradient, weight gradient through the h-tap kernels
    y = ops.conv1d_fwd(x2, w2, g(b), stride, 0, Lout, 'tanh')
    assert rel(y, y_ref) <= RTOL
    dx2 = ops.conv1d_dgrad(g(dy), ops.conv1d_transpose_w(w2), L + pl, stride, 0)
    assert rel(ops.conv1d_tapunfold_dx(dx2, L, k, pl), dx_ref) <= RTOL
    dw2, db = ops.conv1d_wgrad(x2, g(dy), h, stride, 0)
    assert rel(ops.conv1d_tapunfold_dw(dw2, k), dw_ref) <= RTOL and rel(db, db_ref) <= 1e-6


def test_unit_stride_10_taps_take_the_transform_domain_kernels():
    """k = 10, unit stride: the folded layer is a 5-tap layer over 2*Cin channels, so forward, data gradient and weight gradient run on the F(2,5) kernels."""
    from gennet_amd import ops
    dev = torch.device('cuda:0')
    x = torch.randn(2, 256, 64, device=dev); w = torch.randn(10, 64, 128, device=dev) * 0.05; dy = torch.randn(2, 256, 128, device=dev)
    ops.prof_enable(True); ops.prof_reset()
    try:
        with ops.conv_math('wino'):
            x2, w2 = ops.conv1d_tapfold_x(x, 10, 4), ops.conv1d_tapfold_w(w)
            ops.conv1d_fwd(x2, w2, None, 1, 0, 256)
            ops.conv1d_dgrad(dy, ops.conv1d_transpose_w(w2), 260, 1, 0)
            ops.conv1d_wgrad(x2, dy, 5, 1, 0)
            assert ops.prof_collect(5)['launches'] == 2 and ops.prof_collect(6)['launches'] == 1 and ops.prof_collect(0)['launches'] == 0
    finally:
        ops.prof_enable(False)


@pytest.mark.parametrize("k,stride", [(10, 1), (7, 2), (10, 2), (6, 1), (16, 1)])
def test_conv1d_layer_with_a_long_filter_trains_like_the_oracle(k, stride):
    """Conv1D(k > 5) in a graph: [Conv1D(16, k, strides, 'same') -> tanh -> Dropout -> Conv1D(8, k, 'valid') -> relu -> Flatten -> Dense(1)], predict and two Adam steps
    against an fp64 restatement built from oracle/keras_ref.py's layer functions (fused activation / dropout epilogues on the folded conv; its data gradient
    unfused)."""
    from gennet_amd import engine
    from gennet_amd.engine import Adam, Sequential
    from gennet_amd.layers import Activation, Conv1D, Dense, Dropout, Flatten
    rng = np.random.RandomState(31 + k)
    B, L, Cin = 4, 48, 8
    x = f32(rng.randn(B, L, Cin)); t = f32(rng.randn(B))
    engine.set_init_seed(3)
    m = Sequential()
    m.add(Conv1D(16, k, strides=stride, padding='same', input_shape=(L, Cin)))
    m.add(Activation('tanh'))
    m.add(Dropout(0.25, name='drop_a'))
    m.add(Conv1D(8, k, padding='valid'))
    m.add(Activation('relu'))
    m.add(Flatten())
    m.add(Dense(1))
    lr = 1e-3
    m.compile(loss='mean_squared_error', optimizer=Adam(lr=lr, beta_1=0.5))
    W = [f32(a) for a in m.get_weights()]
    L1 = K.conv_out_len(L, k, stride, 'same')
    mask = (rng.rand(B, L1, 16) >= 0.25)
    scale = 1.0 / float(np.float32(1.0 - 0.25))

    def fwd(W, train):
        a1 = np.tanh(K.conv1d_fwd(x, W[0], W[1], stride, 'same'))
        d1 = a1 * mask * scale if train else a1
        z2 = K.conv1d_fwd(d1, W[2], W[3], 1, 'valid')
        a2 = np.maximum(z2, 0)
        return a1, d1, z2, a2, a2.reshape(B, -1) @ W[4] + W[5]

    assert rel(m.predict(x), fwd(W, False)[4]) < 2e-5
    mom = [np.zeros_like(a) for a in W]; vel = [np.zeros_like(a) for a in W]
    for step in range(1, 3):
        loss = float(np.ravel(m.train_on_batch(x, t, dropout_masks={'drop_a': mask.astype(np.uint8)}))[0])
        a1, d1, z2, a2, out = fwd(W, True)
        assert abs(loss - np.mean((out[:, 0] - t) ** 2)) <= 1e-5 * abs(loss)
        dout = (2.0 / B) * (out - t[:, None])
        gW4, gb5 = a2.reshape(B, -1).T @ dout, dout.sum(0)
        dz2 = (dout @ W[4].T).reshape(a2.shape) * (z2 > 0)
        dd1, gW2, gb3 = K.conv1d_bwd(d1, W[2], dz2, 1, 'valid')
        dz1 = dd1 * mask * scale * (1 - a1 ** 2)
        _, gW0, gb1 = K.conv1d_bwd(x, W[0], dz1, stride, 'same')
        grads = [gW0, gb1, gW2, gb3, gW4, gb5]
        if step == 1:
            got = [p.grad.cpu().numpy() for l in m.layers for p in l.params]
            for a, b in zip(got, grads):
                assert rel(a, b.reshape(a.shape)) < 1e-4
        for i, gr in enumerate(grads):
            W[i], mom[i], vel[i] = K.adam_step(W[i], gr.reshape(W[i].shape), mom[i], vel[i], step, lr, 0.5)
    for a, b in zip(m.get_weights(), W):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 0.02 * 2 * lr
