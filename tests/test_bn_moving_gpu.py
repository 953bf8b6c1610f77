"""BatchNormalization moving statistics: keras 2.2.4's TF-backend form (assign_moving_average(zero_debias=True), the default) and the
plain exponential average, GPU (gn_bn_finalize_zero_debias / gn_bn_finalize through layers.BatchNormalization) against the fp64 oracle
(oracle/keras_ref.bn_moving_update*), including what `predict` returns after 1, 10 and 500 updates (bbhMahoGANy.py:1248 consumes the
moving statistics from the first iteration on).  Tolerances: fp32 state against fp64 oracle, 2e-5 relative after 500 updates."""
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


@pytest.mark.parametrize("form", ['tf_zero_debias', 'ema'])
def test_moving_statistics_and_predict_after_1_10_500_updates(form):
    from gennet_amd.engine import Adam, Sequential
    from gennet_amd.layers import Activation, BatchNormalization, Conv1D, Dense, Flatten
    rng = np.random.RandomState(5)
    L, Cin, C, B = 16, 2, 8, 6
    spec = [('conv1d', Cin, C, 5, 1, 'same'), ('bn', C), ('act', 'tanh', 0.0), ('flatten',), ('dense', L * C, 1)]
    ref = N.Stack(spec, rng, moving_average=form)
    for p in ref.params:
        p[...] = f32(p)
    ref.params[2][...] = f32(rng.rand(C) + 0.5); ref.params[3][...] = f32(rng.randn(C) * 0.1)          # gamma, beta
    model = Sequential()
    model.add(Conv1D(C, 5, padding='same', input_shape=(L, Cin)))
    bn = BatchNormalization(momentum=0.99, moving_average=form)
    model.add(bn); model.add(Activation('tanh')); model.add(Flatten()); model.add(Dense(1))
    for l, idx in zip([l for l in model.layers if l.weights], ([0, 1], [2, 3], [4, 5])):
        l.set_weights([ref.params[i] for i in idx] + ([np.zeros(C), np.ones(C)] if len(l.weights) == 4 else []))
    model.compile(loss='mean_squared_error', optimizer=Adam(lr=0.0), metrics=[])        # lr 0: weights stay put, statistics move
    xp = f32(rng.randn(5, L, Cin))
    assert rel(model.predict(xp), ref.forward(xp, False)) < 2e-5                          # before any update: moving mean 0, variance 1
    done = 0
    for upto in (1, 10, 500):
        while done < upto:
            x = f32(rng.randn(B, L, Cin) * (1.0 + 0.5 * np.sin(done / 7.0)) + 0.3 * np.cos(done / 11.0))      # drifting batch statistics
            ref.forward(x, True)
            model.train_on_batch(x, np.zeros(B, np.float32))
            done += 1
        mm, mv = bn.moving_mean.numpy(), bn.moving_variance.numpy()
        assert rel(mm, ref.state[1][0]) < 2e-5 and rel(mv, ref.state[1][1]) < 2e-5, (form, upto)
        assert rel(model.predict(xp), ref.forward(xp, False)) < 2e-5, (form, upto)
        if form == 'tf_zero_debias':
            st = bn.zero_debias[model.name]
            assert st[2] == upto == ref.zd[1][2]
            assert rel(st[0].cpu().numpy(), ref.zd[1][0]) < 2e-5 and rel(st[1].cpu().numpy(), ref.zd[1][1]) < 2e-5
    if form == 'tf_zero_debias':
        assert rel(bn.moving_variance.numpy(), np.ones(C)) > 1e-2                         # the initial value 1 is gone ...
    else:
        assert bn.zero_debias == {}


def test_zero_debias_forgets_the_initial_value_at_the_first_update_and_ema_does_not():
    from gennet_amd import ops
    rng = np.random.RandomState(6)
    rows, C = 64, 12
    x = f32(rng.randn(rows, C) * 3.0 + 2.0)
    dev = torch.device('cuda:0')
    xd = torch.tensor(x, dtype=torch.float32, device=dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    mean, var = x.mean(0), x.var(0) * (rows / (rows - (1.0 + K.BN_EPS)))
    for form in ('zd', 'ema'):
        mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
        zd = (torch.zeros(C, device=dev), torch.zeros(C, device=dev), 1) if form == 'zd' else None
        ops.bn_finalize(ops.bn_stats(xd), rows, gamma, beta, K.BN_EPS, 0.99, mm, mv, zd)
        if form == 'zd':
            assert rel(mm.cpu().numpy(), mean) < 1e-5 and rel(mv.cpu().numpy(), var) < 1e-5
            assert rel(zd[0].cpu().numpy(), 0.01 * mean) < 1e-5
        else:
            assert rel(mm.cpu().numpy(), 0.01 * mean) < 1e-5 and rel(mv.cpu().numpy(), 0.99 + 0.01 * var) < 1e-5
    with pytest.raises(Exception):
        ops.bn_finalize(ops.bn_stats(xd), rows, gamma, beta, K.BN_EPS, 0.99, mm, mv, (mm, mv, 0))       # local_step must be >= 1


def test_zero_debias_state_survives_save_and_load_and_is_reset_by_keras_files(tmp_path):
    """Files written here carry the shadow variables in a private group: a resumed run continues the same moving averages.  A file
    without that group (what real keras writes) restarts them at zero: the first update after loading replaces the statistics."""
    from gennet_amd import bbh, h5lite, keras_io
    from gennet_amd.engine import to_device
    rng = np.random.RandomState(7)
    n_pix, B = 64, 4
    event = f32(rng.randn(n_pix, 1))
    nets = bbh.build_and_compile(event, n_pix, do_pe=False)
    DG = nets.signal_discriminator_on_generator
    z = f32(rng.uniform(-1, 1, (B, 100)))
    for _ in range(3):
        DG.train_on_batch(z, np.ones(B, np.float32))
    bns = [l for l in nets.generator.layers if getattr(l, 'is_batchnorm', False)]
    assert all(list(l.zero_debias) == [DG.name] and l.zero_debias[DG.name][2] == 3 for l in bns)      # one call site trains G (:1296)
    path = str(tmp_path / 'signal_dis_on_gen.h5')
    DG.save_weights(path, True)
    before = [(l.zero_debias[DG.name][0].clone(), l.zero_debias[DG.name][1].clone(), l.moving_variance.numpy()) for l in bns]
    for l in bns:
        l.zero_debias = {}
        l.moving_variance.assign(np.ones(l.moving_variance.shape, np.float32))
    DG.load_weights(path)
    for l, (bm, bv, mv) in zip(bns, before):
        st = l.zero_debias[DG.name]
        assert st[2] == 3 and torch.equal(st[0], bm) and torch.equal(st[1], bv) and np.array_equal(l.moving_variance.numpy(), mv)
    # strip the private group: this is what a keras-written file looks like
    f = h5lite.File(path)
    assert keras_io.STATE_GROUP in f and keras_io.STATE_GROUP not in [n for n in np.asarray(f.attrs['layer_names']).astype(str)]
    w = h5lite.Writer()
    keras_io.save_weights_to_group(w.root, keras_io.top_layers(DG))
    plain = str(tmp_path / 'keras_like.h5')
    w.save(plain)
    DG.load_weights(plain)
    assert all(l.zero_debias == {} for l in bns)
    DG.train_on_batch(z, np.ones(B, np.float32))
    assert all(l.zero_debias[DG.name][2] == 1 for l in bns)
