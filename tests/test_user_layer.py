"""User-defined keras layers and callable losses ("custom Layer subclass with build/call/compute_output_shape", "loss ... a
callable"): a Layer subclass written the way the reference script writes its own MyLayer (bbhMahoGANy.py:164-188: K.constant in
__init__, build calling super().build, call = K.stack([x, const - x], axis=2), compute_output_shape WITH the batch axis) and a loss
written the way it writes chisquare_Loss (:146-162) are traced once and lowered to HIP kernels (gennet_amd/keras/backend.py).
CPU: tracing / lowering / shape conventions / refusals.  GPU: the user-defined layer trains bit-identically to layers.MyLayer."""
import numpy as np
import pytest

from gennet_amd.keras import backend as K
from gennet_amd.keras.engine.topology import Layer
from gennet_amd.keras.models import Sequential

N = 32


class SubtractFromData(Layer):
    """User code in keras' conventions (same structure as the script's class, not its text): keeps the measured series as a constant
    and returns [x | data - x] stacked on a new axis 2."""

    def __init__(self, data, **kwargs):
        self.data = K.constant(data)
        self.output_dim = (N, 2, 1)
        super(SubtractFromData, self).__init__(**kwargs)

    def build(self, input_shape):
        assert input_shape[0] is None and tuple(input_shape[1:]) == (N, 1)      # keras passes the shape WITH the batch axis
        super(SubtractFromData, self).build(input_shape)

    def call(self, x):
        residual = self.data - x
        return K.stack([x, residual], axis=2)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], N, 2, 1)


def chisq(n_sig):
    def loss(yTrue, yPred):
        return K.sum(K.square(yTrue - yPred) / (n_sig ** 2), axis=-1)
    return loss


def test_user_layer_is_traced_and_lowered_to_the_affine_stack_kernel():
    data = np.linspace(-1, 1, N).reshape(N, 1)
    m = Sequential()
    m.add(SubtractFromData(data, input_shape=(N, 1)))
    assert m.output_shape == (None, N, 2, 1)
    low = m.layers[0]._lowered
    assert (low.a0, low.a1) == (1.0, -1.0) and low.b0_host is None and np.array_equal(low.b1_host, data.reshape(-1).astype(np.float32))
    assert np.array_equal(np.asarray(m.layers[0].data), data)                   # K.constant gives its values back
    cfg = m.get_config()                                                        # a user class serialises by name (load needs custom_objects)
    assert any('SubtractFromData' in str(e) for e in (cfg if isinstance(cfg, list) else cfg.get('layers', cfg)))


def test_other_affine_forms_and_refusals():
    x = K.Sym('input', name='x')
    c = K.constant(np.arange(4.0))
    assert K.affine_in(x, x) == (1.0, 0.0)
    a, b = K.affine_in(2.0 * (c - x) / 4.0 + 1.0, x)
    assert a == -0.5 and np.allclose(b, np.arange(4.0) / 2 + 1)
    a, b = K.affine_in(-(x - c), x)
    assert a == -1.0 and np.allclose(b, np.arange(4.0))
    with pytest.raises(NotImplementedError):
        K.affine_in(x * x, x)
    with pytest.raises(NotImplementedError):
        K.affine_in(K.square(x), x)

    class Bad(Layer):
        def call(self, x):
            return K.stack([x, K.square(x)], axis=2)

        def compute_output_shape(self, s):
            return (s[0], N, 2, 1)

    with pytest.raises(NotImplementedError):
        Sequential().add(Bad(input_shape=(N, 1)))

    class WrongShape(SubtractFromData):
        def compute_output_shape(self, s):
            return (s[0], N, 3, 1)

    with pytest.raises(ValueError):
        Sequential().add(WrongShape(np.zeros((N, 1)), input_shape=(N, 1)))


def test_callable_loss_lowering():
    assert K.lower_loss(chisq(2.0)) == ('mean_squared_error', 0.25)
    assert K.lower_loss(lambda t, p: K.mean(K.square(p - t), axis=-1)) == ('mean_squared_error', 1.0)
    assert K.lower_loss(lambda t, p: K.sum(3.0 * K.square(t - p), axis=-1)) == ('mean_squared_error', 3.0)
    with pytest.raises(NotImplementedError):
        K.lower_loss(lambda t, p: K.sum(t - p, axis=-1))
    with pytest.raises(NotImplementedError):
        K.lower_loss(lambda t, p: K.square(t - p))


@pytest.mark.gpu
def test_user_layer_trains_identically_to_the_builtin_mylayer():
    """generator -> user-defined subtract/stack layer -> discriminator, against the same graph with layers.MyLayer: identical losses and
    identical generator weights after two G steps (same kernels' arithmetic: a0*x + b0 with a0 = 1, b0 = 0 and -1*x + c)."""
    import torch
    from gennet_amd import bbh, engine
    from gennet_amd.layers import MyLayer
    global N
    n_pix, B = 64, 4
    N = n_pix
    rng = np.random.RandomState(0)
    event = rng.randn(n_pix, 1).astype(np.float32)
    z = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    results = []
    for make in (lambda: MyLayer(event, input_shape=(n_pix, 1)), lambda: SubtractFromData(event, input_shape=(n_pix, 1))):
        engine.set_init_seed(7); engine.set_device_seed(9)
        G = bbh.generator_model(n_pix); D = bbh.signal_discriminator_model(n_pix)
        sub = Sequential(); sub.add(make())
        GS = bbh.generator_after_subtracting_noise(G, sub)
        DG = bbh.generator_containing_signal_discriminator(GS, D)
        bbh.set_trainable(D, False)
        DG.compile(loss='binary_crossentropy', optimizer=engine.Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])
        img = GS.predict(z)
        out = [DG.train_on_batch(z, np.ones(B, np.float32)) for _ in range(2)]
        results.append((img, out, [p.numpy() for p in G.weights]))
    (img0, out0, w0), (img1, out1, w1) = results
    assert img0.shape == (B, n_pix, 2, 1) and np.array_equal(img0, img1)
    assert out0 == out1
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_callable_chisquare_loss_equals_scaled_mse():
    from gennet_amd import bbh, engine
    rng = np.random.RandomState(1)
    x = rng.randn(6, 128, 1).astype(np.float32); y = [rng.uniform(20, 35, 6).astype(np.float32), rng.uniform(0.5, 1, 6).astype(np.float32)]
    outs = []
    for loss in ('mean_squared_error', chisq(2.0)):
        engine.set_init_seed(3)
        pe = bbh.signal_pe_model(128)
        pe.compile(loss=loss, optimizer=engine.Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])
        outs.append(pe.train_on_batch(x, y))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert abs(b - 0.25 * a) <= 1e-6 * abs(a)


@pytest.mark.gpu
def test_build_and_compile_with_chi_loss_trains_the_generator_on_the_scripts_own_loss():
    """chi_loss = True (bbhMahoGANy.py:97, :1106-1109): the combined model is compiled with chisquare_Loss; its reported loss is then the
    scaled squared error of D(G(z)) against the labels, and only the generator's weights move."""
    from gennet_amd import bbh, engine
    n_pix, B = 64, 4
    rng = np.random.RandomState(6)
    event = rng.randn(n_pix, 1).astype(np.float32)
    z = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    engine.set_init_seed(8); engine.set_device_seed(1)
    nets = bbh.build_and_compile(event, n_pix, chi_loss=True, n_sig=2.0)
    assert nets.signal_discriminator_on_generator._loss_scales == [0.25]
    d_before = [w.copy() for w in nets.signal_discriminator.get_weights()]
    g_before = [w.copy() for w in nets.generator.get_weights()]
    out = nets.signal_discriminator_on_generator.train_on_batch(z, np.ones(B, np.float32))
    assert np.isfinite(out).all() and 0.0 <= out[0] <= 0.25 + 1e-6                 # (1 - p)^2 / 4 with p in [0, 1]
    assert all(np.array_equal(a, b) for a, b in zip(d_before, nets.signal_discriminator.get_weights()))
    assert any(not np.array_equal(a, b) for a, b in zip(g_before, nets.generator.get_weights()))
