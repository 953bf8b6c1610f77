"""PReLU (bbhMahoGANy.py:39 and the act = 'prelu' branches): oracle vs torch autograd (CPU), HIP kernels vs oracle, the layer
inside a train step, its Keras .h5 round trip, and the reference's own import lines against the facade."""
import numpy as np
import pytest
import torch

from oracle import keras_ref as K
from oracle import prelu_ref as P


def test_oracle_prelu_matches_torch_autograd():
    rng = np.random.RandomState(0)
    x = rng.randn(6, 5, 4); alpha = rng.randn(5, 4) * 0.3; dy = rng.randn(6, 5, 4)
    xt = torch.tensor(x, requires_grad=True); at = torch.tensor(alpha, requires_grad=True)
    yt = torch.clamp(xt, min=0) - at * torch.clamp(-xt, min=0)
    yt.backward(torch.tensor(dy))
    assert np.abs(P.prelu_fwd(x, alpha) - yt.detach().numpy()).max() < 1e-14
    dx, da = P.prelu_bwd(dy, x, alpha)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-14 and np.abs(da - at.grad.numpy()).max() < 1e-13
    z = np.zeros((2, 4)); one = np.ones((2, 4))
    dx0, da0 = P.prelu_bwd(one, z, np.full(4, 0.5))
    assert (dx0 == 0).all() and (da0 == 0).all()                       # the kink: zero gradient on both branches


def test_facade_covers_every_keras_name_the_script_imports():
    """The names of bbhMahoGANy.py:32-43, importable from the facade; the unused ones are placeholders that refuse construction."""
    import importlib
    wanted = {'models': ['Sequential', 'Model'],
              'layers': ['Dense', 'Input', 'GlobalAveragePooling1D', 'Reshape', 'AlphaDropout', 'Dropout', 'GaussianDropout', 'GaussianNoise'],
              'layers.core': ['Activation', 'Flatten'], 'layers.normalization': ['BatchNormalization'],
              'layers.convolutional': ['UpSampling2D', 'UpSampling1D', 'Conv2DTranspose', 'Conv2D', 'MaxPooling2D', 'Conv1D', 'AveragePooling1D', 'MaxPooling1D'],
              'layers.advanced_activations': ['LeakyReLU', 'PReLU', 'ThresholdedReLU', 'ReLU'],
              'engine.topology': ['Layer'], 'optimizers': ['Adam', 'RMSprop', 'Adagrad', 'Adadelta', 'Adamax', 'Nadam']}
    for mod, names in wanted.items():
        m = importlib.import_module('gennet_amd.keras.' + mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    from gennet_amd.keras import backend as Kb
    assert Kb.set_session(None) is None
    from gennet_amd.keras.layers.convolutional import MaxPooling1D
    from gennet_amd.keras.optimizers import Nadam
    for cls in (MaxPooling1D, Nadam):
        with pytest.raises(NotImplementedError):
            cls()


def test_prelu_layer_config_and_h5_round_trip(tmp_path):
    from gennet_amd import h5lite
    from gennet_amd.keras.layers import Dense, PReLU
    from gennet_amd.keras.models import Sequential, load_model
    m = Sequential()
    m.add(Dense(8, input_shape=(12,)))
    m.add(PReLU())
    m.add(Dense(1))
    pl = m._top[1]
    assert pl.name.startswith('p_re_lu_') and pl.alpha.shape == (8,) and float(np.abs(pl.alpha.numpy()).max()) == 0.0
    pl.alpha.assign(np.linspace(-0.5, 0.5, 8).astype(np.float32))
    path = str(tmp_path / 'm.h5')
    m.save(path, True)
    f = h5lite.File(path)
    assert [n.decode() for n in f['model_weights'][pl.name].attrs['weight_names'].tolist()] == [pl.name + '/alpha:0']
    m2 = load_model(path)
    assert np.array_equal(m2._top[1].alpha.numpy(), pl.alpha.numpy()) and m2._top[1].__class__.__name__ == 'PReLU'


@pytest.mark.gpu
@pytest.mark.parametrize('B,shape', [(5, (8,)), (3, (40, 16)), (64, (2048, 64)), (1, (4,))])
def test_prelu_kernels_match_oracle(B, shape):
    from gennet_amd import ops
    rng = np.random.RandomState(B + len(shape))
    x = rng.randn(B, *shape).astype(np.float32); alpha = (rng.randn(*shape) * 0.4).astype(np.float32); dy = rng.randn(B, *shape).astype(np.float32)
    x.reshape(-1)[::7] = 0.0                                            # exact zeros: the kink
    dev = torch.device('cuda:0')
    xt, at, gt = (torch.tensor(v).to(dev) for v in (x, alpha, dy))
    y = ops.prelu_fwd(xt, at).cpu().numpy()
    assert np.array_equal(y, P.prelu_fwd(x, alpha).astype(np.float32))   # one multiply: bit-exact
    dx, da = ops.prelu_bwd(gt, xt, at)
    rdx, rda = P.prelu_bwd(dy.astype(np.float64), x.astype(np.float64), alpha.astype(np.float64))
    assert np.array_equal(dx.cpu().numpy(), rdx.astype(np.float32))
    assert np.abs(da.cpu().numpy() - rda).max() <= 1e-5 * max(np.abs(rda).max(), 1.0)
    dx2, _ = ops.prelu_bwd(gt, xt, at, need_dx=False)
    assert dx2 is None


@pytest.mark.gpu
def test_prelu_inside_a_train_step_matches_oracle():
    from gennet_amd.engine import Adam
    from gennet_amd.keras.layers import Dense, PReLU
    from gennet_amd.keras.models import Sequential
    rng = np.random.RandomState(4)
    m = Sequential()
    m.add(Dense(8, input_shape=(12,))); m.add(PReLU()); m.add(Dense(1))
    d1, pl, d2 = m._top
    a0 = (rng.randn(8) * 0.3).astype(np.float32)
    pl.alpha.assign(a0)
    m.compile(loss='mean_squared_error', optimizer=Adam(lr=1e-2, beta_1=0.5))
    W1, b1, W2, b2 = (p.numpy().astype(np.float64) for p in (d1.kernel, d1.bias, d2.kernel, d2.bias))
    x = rng.randn(16, 12).astype(np.float32); t = rng.randn(16).astype(np.float32)
    out = m.train_on_batch(x, t)
    h = K.dense_fwd(x.astype(np.float64), W1, b1)
    a = P.prelu_fwd(h, a0.astype(np.float64))
    p = K.dense_fwd(a, W2, b2)
    loss, dp = K.mse_loss(p, t.astype(np.float64).reshape(-1, 1))
    da, dW2, db2 = K.dense_bwd(a, W2, dp)
    dh, dalpha = P.prelu_bwd(da, h, a0.astype(np.float64))
    _, dW1, _ = K.dense_bwd(x.astype(np.float64), W1, dh)
    assert abs(out[0] - loss) <= 1e-5 * abs(loss)
    for param, g, p0 in ((pl.alpha, dalpha, a0.astype(np.float64)), (d1.kernel, dW1, W1), (d2.kernel, dW2, W2)):
        want, _, _ = K.adam_step(p0, g, np.zeros_like(g), np.zeros_like(g), 1, lr=1e-2)
        assert np.abs(param.numpy() - want).max() <= 2e-4 * 1e-2 + 1e-6 * np.abs(want).max(), param.name
