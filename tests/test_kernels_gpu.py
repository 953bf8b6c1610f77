"""GPU parity: every C-ABI kernel against the fp64 numpy oracle on seeded inputs.

Tolerances (fp32 kernels vs fp64 oracle, stated per check): `tol * sum|a||b|`-style bounds are expressed as a relative
error against the max magnitude of the oracle output; 2e-5 covers K <= 2560-term fp32 fmaf chains
(v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain, ~1e-7 * sum|a b|).
"""
import os

import numpy as np
import pytest
import torch

from oracle import keras_ref as K

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RTOL = 2e-5


def dev():
    return torch.device('cuda:0')


def g(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev())


def close(t, ref, rtol=RTOL, atol=0.0):
    a = t.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(a - ref).max()
    assert err <= rtol * scale + atol, 'max err %.3e vs scale %.3e (rel %.3e)' % (err, scale, err / scale)


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


CONV_CASES = [
    # B, L, Cin, Cout, k, stride, padding
    (2, 64, 16, 128, 5, 1, 'same'),       # one tile, exact channels
    (3, 133, 64, 128, 5, 1, 'valid'),     # ragged M tile
    (2, 150, 64, 128, 5, 2, 'valid'),     # stride 2 (de-interleaved slab)
    (2, 77, 32, 64, 5, 2, 'same'),        # narrow-N tile config
    (1, 300, 48, 192, 5, 1, 'same'),      # Cout not a multiple of the N tile, Cin = 3 chunks
    (2, 40, 20, 72, 5, 2, 'same'),        # Cin not a multiple of KC
    (2, 96, 1, 64, 5, 2, 'same'),         # small-Cin (point-estimator first layer)
    (2, 96, 1, 64, 5, 1, 'same'),
    (3, 50, 2, 512, 5, 2, 'same'),        # folded discriminator first layer
    (2, 64, 256, 1, 5, 1, 'same'),        # small-Cout (generator output conv)
    (2, 64, 128, 2, 5, 2, 'same'),
    (3, 70, 512, 1, 5, 1, 'valid'),       # Cout = 1 row-run kernel, ragged run, two channel passes
    (2, 61, 128, 3, 5, 1, 'same'),        # small-Cout weight gradient looped over input rows
    (2, 50, 16, 64, 3, 1, 'same'),        # 3 taps (the folded UpSampling1D -> Conv1D layers), matrix-core weight gradient
    (2, 45, 64, 128, 2, 1, 'valid'),      # 2 taps
    (3, 66, 32, 64, 4, 2, 'same'),        # 4 taps, stride 2
    (9, 2100, 16, 512, 5, 1, 'same'),     # 648 blocks of the tall tile: 512 in XCD patch order (8 slabs x 8 column tiles) + a plain-order tail
    (20, 1100, 16, 512, 3, 1, 'same'),    # 720 blocks of the square tile, 4 column tiles: 16 x 4 patches + tail
    # round 3: the narrow-wave forms for launches with fewer than two wave tiles per SIMD (conv_pipe_try), every block height
    (8, 512, 16, 1024, 5, 1, 'same'),     # forward: 256 blocks of 8 narrow waves (256 x 64); data gradient (Cout 16) on the plain path
    (8, 250, 16, 1024, 5, 1, 'same'),     # forward: 128 x 64 blocks of 4 narrow waves, ragged rows (250 = 128 + 122)
    (8, 1000, 16, 512, 5, 2, 'valid'),    # stride-2 forward on narrow waves (M = 498)
    (8, 513, 1024, 16, 5, 2, 'same'),     # data gradient merged over both phases, odd length (257 output pairs, the last without its odd row),
                                          # 256-row blocks of 8 narrow waves with two accumulator sets; 'same' padding: even taps -> odd rows
    (7, 700, 1024, 16, 5, 2, 'valid'),    # ... 128-row blocks, 'valid' padding: even taps -> even rows
    (2, 131, 128, 24, 5, 2, 'same'),      # ... 64-row blocks, ragged rows
    (3, 700, 64, 128, 5, 1, 'same'),      # weight gradient K-splits INSIDE batch elements (14 ranges of 5 chunks over 3 x 22 chunks, ragged last chunk)
    (5, 333, 64, 64, 5, 2, 'valid'),      # ... stride 2
]


@pytest.mark.parametrize("B,L,Cin,Cout,k,s,padding", CONV_CASES)
def test_conv1d_fwd_dgrad_wgrad(B, L, Cin, Cout, k, s, padding):
    from gennet_amd import ops
    rng = np.random.RandomState(B * 1000 + L + Cin + Cout)
    x = f32(rng.randn(B, L, Cin)); w = f32(rng.randn(k, Cin, Cout) / np.sqrt(k * Cin)); b = f32(rng.randn(Cout))
    Lout, pl = ops.conv_geometry(L, k, s, padding)
    y_ref = K.conv1d_fwd(x, w, b, s, padding)
    assert y_ref.shape[1] == Lout
    for act, p in (('linear', 0.0), ('relu', 0.0), ('leaky', 0.2), ('tanh', 0.0)):
        y = ops.conv1d_fwd(g(x), g(w), g(b), s, pl, Lout, act, p)
        close(y, K.act_fwd(y_ref, act, p))
    dy = f32(rng.randn(B, Lout, Cout))
    dx_ref, dw_ref, db_ref = K.conv1d_bwd(x, w, dy, s, padding)
    wt = ops.conv1d_transpose_w(g(w))
    close(wt, np.transpose(w, (0, 2, 1)), 0.0)
    dx = ops.conv1d_dgrad(g(dy), wt, L, s, pl)
    close(dx, dx_ref)
    dw, db = ops.conv1d_wgrad(g(x), g(dy), k, s, pl)
    close(dw, dw_ref, 5e-5)
    close(db, db_ref, 5e-5)


@pytest.mark.parametrize("B,L,Cin,Cout,s,padding", [(3, 300, 64, 128, 1, 'same'),      # pipelined kernel, square tile, ragged last M tile
                                                    (2, 700, 32, 64, 1, 'same'),       # tall tile (256 rows), 700 = 2*256 + 188
                                                    (2, 512, 64, 64, 2, 'same'),       # stride 2 (de-interleaved slab)
                                                    (2, 90, 20, 72, 1, 'valid'),       # ragged channels: plain kernel + separate statistics pass
                                                    (2, 64, 2, 16, 1, 'same')])        # small-Cin kernel + separate pass
def test_conv1d_fwd_with_batchnorm_statistics(B, L, Cin, Cout, s, padding):
    """gn_conv1d_fwd_stats: y as gn_conv1d_fwd, sums = (sum y, sum y^2) over (B, Lout) in fp64 -- from the conv epilogue on the pipelined
    kernel (rows of the ragged last tile excluded), from a separate pass elsewhere."""
    from gennet_amd import ops
    rng = np.random.RandomState(L + Cin)
    x = f32(rng.randn(B, L, Cin)); w = f32(rng.randn(5, Cin, Cout) / np.sqrt(5 * Cin)); b = f32(rng.randn(Cout) * 0.3 + 0.5)
    Lout, pl = ops.conv_geometry(L, 5, s, padding)
    y_ref = K.conv1d_fwd(x, w, b, s, padding)
    y, sums = ops.conv1d_fwd_stats(g(x), g(w), g(b), s, pl, Lout)
    close(y, y_ref)
    assert torch.equal(y, ops.conv1d_fwd(g(x), g(w), g(b), s, pl, Lout))
    yd = y.double().reshape(-1, Cout)
    ref = torch.cat([yd.sum(0), (yd * yd).sum(0)]).cpu().numpy()
    got = sums.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()                       # fp64 sums of the same fp32 values: order only
    close(ops.bn_stats(y.reshape(-1, Cout)), got, 1e-12)


@pytest.mark.parametrize("B,L,Cin,Cout", [(8, 100, 64, 64), (8, 65, 64, 64), (8, 192, 128, 64), (3, 100, 64, 128)])
def test_conv1d_fwd_stats_stays_inside_an_exact_size_workspace(B, L, Cin, Cout):
    """the narrow-wave launches of conv_pipe.hip use 64-row blocks, so the fused statistics write B * ceil(Lout / 64) partial rows;
    gn_conv1d_fwd_stats_workspace sized them for 128-row tiles.  An exact-size C-API caller (no slack, unlike ops.workspace) must be safe: the
    call runs in a buffer of EXACTLY the advertised size followed by a guard region that has to come back untouched."""
    from gennet_amd import _lib, ops
    rng = np.random.RandomState(B * L + Cout)
    x = f32(rng.randn(B, L, Cin)); w = f32(rng.randn(5, Cin, Cout) / np.sqrt(5 * Cin)); b = f32(rng.randn(Cout) * 0.3)
    Lout, pl = ops.conv_geometry(L, 5, 1, 'same')
    nb = _lib.size('gn_conv1d_fwd_stats_workspace', B, Lout, Cout)
    assert nb >= B * -(-Lout // 64) * 2 * Cout * 8
    guard = 1 << 16
    buf = torch.full((nb + guard,), 0xA5, dtype=torch.uint8, device='cuda')
    xd, wd, bd = g(x), g(w), g(b)
    y = torch.empty((B, Lout, Cout), dtype=torch.float32, device='cuda')
    sums = torch.empty((2 * Cout,), dtype=torch.float64, device='cuda')
    _lib.call('gn_conv1d_fwd_stats', ops._p(xd), ops._p(wd), ops._p(bd), ops._p(y), ops._p(sums), ops._p(buf), nb, B, L, Cin, Cout, 5, 1, pl, Lout, ops._stream())
    torch.cuda.synchronize()
    assert bool((buf[nb:] == 0xA5).all()), 'statistics partials written past the advertised workspace size'
    close(y, K.conv1d_fwd(x, w, b, 1, 'same'))
    yd = y.double().reshape(-1, Cout)
    ref = torch.cat([yd.sum(0), (yd * yd).sum(0)]).cpu().numpy()
    assert np.abs(sums.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()


def test_conv1d_mfma_exact_integers():
    """A = small integers, B asymmetric integers: fp32 MFMA must be bit-exact (catches row/col swaps, tap mix-ups).  This is a statement about the
    DIRECT kernels (a k-ordered fmaf chain of integers is exact); the transform-domain kernel multiplies by G's sixths and fifteenths and is checked
    against the oracle in tests/test_wino_gpu.py, so the unit-stride case runs with the direct kernels switched in."""
    from gennet_amd import ops
    with ops.conv_math('fp32'):
        _exact_integers(ops)


def _exact_integers(ops):
    rng = np.random.RandomState(7)
    B, L, Cin, Cout, k = 2, 140, 32, 128, 5
    x = rng.randint(-3, 4, (B, L, Cin)).astype(np.float64)
    w = rng.randint(-2, 3, (k, Cin, Cout)).astype(np.float64) + np.arange(Cout)[None, None, :] % 3
    for s, padding in ((1, 'same'), (2, 'valid')):
        Lout, pl = ops.conv_geometry(L, k, s, padding)
        y = ops.conv1d_fwd(g(x), g(w), None, s, pl, Lout)
        assert np.array_equal(y.cpu().numpy().astype(np.float64), K.conv1d_fwd(x, w, None, s, padding))
        dy = rng.randint(-2, 3, (B, Lout, Cout)).astype(np.float64)
        dx_ref, dw_ref, _ = K.conv1d_bwd(x, w, dy, s, padding)
        dx = ops.conv1d_dgrad(g(dy), ops.conv1d_transpose_w(g(w)), L, s, pl)
        assert np.array_equal(dx.cpu().numpy().astype(np.float64), dx_ref)
        dw, _ = ops.conv1d_wgrad(g(x), g(dy), k, s, pl)
        assert np.array_equal(dw.cpu().numpy().astype(np.float64), dw_ref)


def test_conv2d_width2_fold_matches_conv2d():
    from gennet_amd import ops
    rng = np.random.RandomState(11)
    B, H, Cin, Cout = 2, 48, 16, 32
    x = f32(rng.randn(B, H, 2, Cin)); w = f32(rng.randn(5, 5, Cin, Cout) * 0.1); b = f32(rng.randn(Cout))
    y_ref = K.conv2d_fwd(x, w, b, (2, 1), 'same')
    wf, bf = ops.conv2d_w2_fold(g(w), g(b))
    close(wf, K.fold_conv2d_w2(w), 0.0)
    Lout, pl = ops.conv_geometry(H, 5, 2, 'same')
    y = ops.conv1d_fwd(g(x.reshape(B, H, 2 * Cin)), wf, bf, 2, pl, Lout)
    close(y.reshape(B, Lout, 2, Cout), y_ref)
    dy = f32(rng.randn(*y_ref.shape))
    dx_ref, dw_ref, db_ref = K.conv2d_bwd(x, w, dy, (2, 1), 'same')
    dyf = g(dy.reshape(B, Lout, 2 * Cout))
    dwf, dbf = ops.conv1d_wgrad(g(x.reshape(B, H, 2 * Cin)), dyf, 5, 2, pl)
    dw, db = ops.conv2d_w2_unfold_grad(dwf, dbf, Cin, Cout)
    close(dw, dw_ref, 5e-5); close(db, db_ref, 5e-5)
    dx = ops.conv1d_dgrad(dyf, ops.conv1d_transpose_w(wf), H, 2, pl)
    close(dx.reshape(B, H, 2, Cin), dx_ref)


@pytest.mark.parametrize("B,L,Cin,Cout,s", [(2, 48, 16, 64, 2), (2, 48, 16, 64, 1), (3, 37, 64, 128, 1), (3, 37, 256, 64, 2), (2, 9, 8, 8, 1)])
def test_upsample_conv_fold_matches_upsample_then_conv(B, L, Cin, Cout, s):
    """UpSampling1D(2) -> Conv1D(5, 'same', s) == a 3-tap stride-1 conv on the un-upsampled input with folded weights
    (bbhMahoGANy.py:249-250 s=2, :258-259 s=1): forward, data gradient, weight / bias gradient against the materialised definition."""
    from gennet_amd import ops
    rng = np.random.RandomState(L + Cin + s)
    x = f32(rng.randn(B, L, Cin)); w = f32(rng.randn(5, Cin, Cout) * 0.1); b = f32(rng.randn(Cout))
    u = K.upsample1d_fwd(x)
    y_ref = K.conv1d_fwd(u, w, b, s, 'same')
    wf, bf = ops.conv1d_up2_fold(g(w), g(b), s)
    assert tuple(wf.shape) == (3, Cin, Cout * (2 if s == 1 else 1))
    y = ops.conv1d_fwd(g(x), wf, bf, 1, 1, L)
    close(y.reshape(y_ref.shape), y_ref)
    dy = f32(rng.randn(*y_ref.shape))
    du_ref, dw_ref, db_ref = K.conv1d_bwd(u, w, dy, s, 'same')
    dx_ref = K.upsample1d_bwd(du_ref)
    dyf = g(dy).reshape(y.shape)
    dwf, dbf = ops.conv1d_wgrad(g(x), dyf, 3, 1, 1)
    dw, db = ops.conv1d_up2_unfold_grad(dwf, dbf, Cout, s)
    close(dw, dw_ref, 5e-5); close(db, db_ref, 5e-5)
    dx = ops.conv1d_dgrad(dyf, ops.conv1d_transpose_w(wf), L, 1, 1)
    close(dx, dx_ref, 5e-5)


@pytest.mark.parametrize("B,n_in,n_out", [(5, 100, 512), (130, 100, 1024), (4, 64, 260)])
def test_dense_large_out(B, n_in, n_out):
    from gennet_amd import ops
    rng = np.random.RandomState(B + n_out)
    x = f32(rng.uniform(-1, 1, (B, n_in))); w = f32(rng.randn(n_in, n_out) * 0.1); b = f32(rng.randn(n_out))
    close(ops.dense_fwd(g(x), g(w), g(b)), K.dense_fwd(x, w, b))
    dy = f32(rng.randn(B, n_out))
    dx_ref, dw_ref, db_ref = K.dense_bwd(x, w, dy)
    dx, dw, db = ops.dense_bwd(g(x), g(w), g(dy), need_dx=True)
    close(dx, dx_ref); close(dw, dw_ref, 5e-5); close(db, db_ref, 5e-5)
    dx2, dw2, _ = ops.dense_bwd(g(x), g(w), g(dy), need_dx=False)
    assert dx2 is None
    close(dw2, dw_ref, 5e-5)


@pytest.mark.parametrize("B,n_in,act,p", [(3, 4096, 'relu', 0.0), (7, 5000, 'relu_max', 1.0), (2, 1024, 'sigmoid', 0.0)])
def test_dense_head(B, n_in, act, p):
    from gennet_amd import ops
    rng = np.random.RandomState(n_in)
    x = f32(rng.randn(B, n_in)); w = f32(rng.randn(n_in, 1) / np.sqrt(n_in)); b = f32([0.3])
    y_ref = K.act_fwd(K.dense_fwd(x, w, b), act, p)
    close(ops.dense_fwd(g(x), g(w), g(b), act, p), y_ref)
    dy = f32(rng.randn(B, 1))
    dx_ref, dw_ref, db_ref = K.dense_bwd(x, w, dy)
    dx, dw, db = ops.dense_bwd(g(x), g(w), g(dy))
    close(dx, dx_ref); close(dw, dw_ref); close(db, db_ref)


@pytest.mark.parametrize("act,p", [('relu', 0.0), ('relu_max', 1.0), ('leaky', 0.2), ('tanh', 0.0), ('sigmoid', 0.0), ('linear', 0.0)])
def test_activations(act, p):
    from gennet_amd import ops
    rng = np.random.RandomState(3)
    x = f32(rng.randn(1003) * 2)
    y = ops.act_fwd(g(x), act, p)
    y_ref = K.act_fwd(x, act, p)
    close(y, y_ref, 1e-6)
    dy = f32(rng.randn(1003))
    close(ops.act_bwd(g(dy), y, act, p), K.act_bwd(dy, y.cpu().numpy().astype(np.float64), act, p), 1e-6)


def test_tanh_absolute_error_over_the_whole_range():
    """The library's one tanh (gn_tanhf, common.h): absolute error <= 2e-7 from 1e-8 to saturation, through the switch point at 0.35,
    odd symmetry exact, +-inf -> +-1, NaN -> NaN, tiny arguments returned unchanged."""
    from gennet_amd import ops
    x = np.concatenate([np.linspace(-12, 12, 200001), np.linspace(0.34, 0.36, 20001), np.logspace(-8, -1, 5001), [0.0, 50.0, 100.0, 1e30]]).astype(np.float32)
    y = ops.act_fwd(g(x), 'tanh', 0.0).cpu().numpy().astype(np.float64)
    ref = np.tanh(x.astype(np.float64))
    assert np.abs(y - ref).max() <= 2e-7
    small = np.abs(x) < 1e-3
    assert (np.abs(y[small] - ref[small]) <= 1.2e-7 * np.abs(ref[small])).all()          # relative, where absolute says nothing
    ym = ops.act_fwd(g(-x), 'tanh', 0.0).cpu().numpy().astype(np.float64)
    assert np.array_equal(ym, -y)
    sp = ops.act_fwd(g(np.array([np.inf, -np.inf, np.nan], np.float32)), 'tanh', 0.0).cpu().numpy()
    assert sp[0] == 1.0 and sp[1] == -1.0 and np.isnan(sp[2])


def test_dropout_mask_and_apply():
    from gennet_amd import ops
    n = 1 << 20
    m = ops.dropout_mask((n,), 0.4, 1234, 0, dev())
    keep = m.float().mean().item()
    assert abs(keep - 0.6) < 3e-3                      # 1M Bernoulli draws: sigma = 5e-4
    m2 = ops.dropout_mask((n,), 0.4, 1234, 0, dev())
    assert torch.equal(m, m2)                          # counter-based: reproducible
    m3 = ops.dropout_mask((n,), 0.4, 1234, n // 4, dev())
    assert not torch.equal(m, m3)
    x = torch.randn(n, device=dev())
    y = ops.dropout_apply(x, m, 0.4)
    close(y, x.cpu().numpy().astype(np.float64) * m.cpu().numpy() / 0.6, 1e-6)


def test_upsample_and_subtract_stack():
    from gennet_amd import ops
    rng = np.random.RandomState(5)
    x = f32(rng.randn(3, 10, 8))
    close(ops.upsample2_fwd(g(x)), K.upsample1d_fwd(x), 0.0)
    dy = f32(rng.randn(3, 20, 8))
    close(ops.upsample2_bwd(g(dy)), K.upsample1d_bwd(dy), 1e-6)
    xs = f32(rng.randn(4, 33, 1)); ev = f32(rng.randn(33, 1))
    close(ops.subtract_stack_fwd(g(xs), g(ev)), K.mylayer_fwd(xs, ev), 1e-7)
    dimg = f32(rng.randn(4, 33, 2, 1))
    close(ops.subtract_stack_bwd(g(dimg)), K.mylayer_bwd(dimg), 1e-6)


@pytest.mark.parametrize("shape", [(6, 50, 64), (37, 4096), (512, 8, 128)])
@pytest.mark.parametrize("with_drop", [False, True])
def test_batchnorm_train_fwd_bwd(shape, with_drop):
    """BN + tanh (+ dropout) fused apply against the oracle chain; channel-BN (3-D) and feature-BN (2-D)."""
    from gennet_amd import ops
    rng = np.random.RandomState(shape[0])
    Cc = shape[-1]
    x = f32(rng.randn(*shape) * 1.5 + 0.7); gamma = f32(rng.rand(Cc) + 0.5); beta = f32(rng.randn(Cc) * 0.1)
    rows = x.size // Cc
    rate = 0.2 if with_drop else 0.0
    mask = (rng.rand(*shape) >= rate).astype(np.uint8) if with_drop else None
    y_bn, cache, mean, var = K.bn_train_fwd(x, gamma, beta)
    y_act = np.tanh(y_bn)
    y_ref = K.dropout_fwd(y_act, mask, rate) if with_drop else y_act
    mm0, mv0 = f32(rng.randn(Cc) * 0.1), f32(rng.rand(Cc) + 0.5)
    mm_ref, mv_ref = K.bn_moving_update(mm0, mv0, mean, var, rows, 0.99)

    x2 = g(x).reshape(rows, Cc)
    sums = ops.bn_stats(x2)
    close(sums[:Cc], x.reshape(rows, Cc).sum(0), 1e-6, 1e-3)
    mm, mv = g(mm0), g(mv0)
    scale, shift, smean, sinv = ops.bn_finalize(sums, rows, g(gamma), g(beta), K.BN_EPS, 0.99, mm, mv)
    close(smean, mean, 1e-5, 1e-6); close(mm, mm_ref, 1e-5); close(mv, mv_ref, 1e-5)
    mt = torch.tensor(mask.reshape(rows, Cc), device=dev()) if with_drop else None
    y = ops.bn_apply(x2, scale, shift, mt, 'tanh', 0.0, rate)
    close(y, y_ref.reshape(rows, Cc), 2e-5)

    dy = f32(rng.randn(*shape))
    d_act = dy * mask / (1 - rate) if with_drop else dy
    d_bn = K.act_bwd(d_act, y_act, 'tanh')
    dx_ref, dg_ref, db_ref = K.bn_train_bwd(d_bn, cache, gamma)
    dy2 = g(dy).reshape(rows, Cc)
    dsums = ops.bn_bwd_stats(dy2, y, x2, mt, smean, sinv, 'tanh', 0.0, rate)
    dgamma = torch.empty(Cc, device=dev()); dbeta = torch.empty(Cc, device=dev())
    dx = ops.bn_bwd_apply(dy2, y, x2, mt, g(gamma), smean, sinv, dsums, rows, dsums, dgamma, dbeta, 'tanh', 0.0, rate)
    close(dgamma, dg_ref, 1e-4); close(dbeta, db_ref, 1e-4)
    close(dx, dx_ref.reshape(rows, Cc), 1e-4)
    # the variant the layer uses: the activation output recomputed from the pre-BN tensor (scale / shift), the stored output not read
    dsums2 = ops.bn_bwd_stats(dy2, None, x2, mt, smean, sinv, 'tanh', 0.0, rate, scale, shift)
    close(dsums2, dsums.cpu().numpy(), 1e-6)
    dx2 = ops.bn_bwd_apply(dy2, None, x2, mt, g(gamma), smean, sinv, dsums2, rows, dsums2, dgamma, dbeta, 'tanh', 0.0, rate, scale, shift)
    close(dgamma, dg_ref, 1e-4); close(dbeta, db_ref, 1e-4)
    close(dx2, dx_ref.reshape(rows, Cc), 1e-4)


@pytest.mark.parametrize("B,L,Cc,k,padding", [(3, 40, 64, 5, 'same'), (2, 33, 1024, 5, 'same'), (4, 21, 16, 3, 'valid'), (2, 7, 8, 5, 'same')])
def test_batchnorm_backward_with_the_1_filter_conv_gradient_formed_on_the_fly(B, L, Cc, k, padding):
    """gn_bn_bwd_stats_conv1 / gn_bn_bwd_apply_conv1: BatchNormalization -> tanh -> Dropout -> Conv1D(1, k) (bbhMahoGANy.py:284-292) with the
    conv's data gradient formed inside the BN backward passes, against the oracle chain on the materialised gradient."""
    from gennet_amd import ops
    rng = np.random.RandomState(L + Cc)
    x = f32(rng.randn(B, L, Cc) * 1.5 + 0.3); gamma = f32(rng.rand(Cc) + 0.5); beta = f32(rng.randn(Cc) * 0.1)
    rate = 0.2
    mask = (rng.rand(B, L, Cc) >= rate).astype(np.uint8)
    w = f32(rng.randn(k, Cc, 1) * 0.2)
    y_bn, cache, mean, var = K.bn_train_fwd(x, gamma, beta)
    y_act = np.tanh(y_bn)
    y = K.dropout_fwd(y_act, mask, rate)
    z = K.conv1d_fwd(y, w, np.zeros(1), 1, padding)
    gz = f32(rng.randn(*z.shape))
    dz_ref, _, _ = K.conv1d_bwd(y, w, gz, 1, padding)
    d_bn = K.act_bwd(dz_ref * mask / (1 - rate), y_act, 'tanh')
    dx_ref, dg_ref, db_ref = K.bn_train_bwd(d_bn, cache, gamma)

    rows = B * L
    x2 = g(x).reshape(rows, Cc)
    sums = ops.bn_stats(x2)
    mm, mv = g(np.zeros(Cc)), g(np.ones(Cc))
    scale, shift, smean, sinv = ops.bn_finalize(sums, rows, g(gamma), g(beta), K.BN_EPS, 0.99, mm, mv)
    mt = torch.tensor(mask.reshape(rows, Cc), device=dev())
    Lout, pl = ops.conv_geometry(L, k, 1, padding)
    cg = ops.ConvGrad1(g(gz), g(w), L, pl)
    assert cg.shape == (B, L, Cc) and cg.Lout == Lout
    dsums = ops.bn_bwd_stats_conv1(cg, x2, mt, smean, sinv, 'tanh', 0.0, rate, scale, shift)
    dgamma = torch.empty(Cc, device=dev()); dbeta = torch.empty(Cc, device=dev())
    dx = ops.bn_bwd_apply_conv1(cg, x2, mt, g(gamma), smean, sinv, dsums, rows, dsums, dgamma, dbeta, 'tanh', 0.0, rate, scale, shift)
    close(dgamma, dg_ref, 1e-4); close(dbeta, db_ref, 1e-4)
    close(dx, dx_ref.reshape(rows, Cc), 1e-4)
    # and against the same kernels fed the materialised gradient
    dz = ops.conv1d_dgrad(g(gz), ops.conv1d_transpose_w(g(w)), L, 1, pl).reshape(rows, Cc)
    close(dz, dz_ref.reshape(rows, Cc), 5e-5)
    dsums_m = ops.bn_bwd_stats(dz, None, x2, mt, smean, sinv, 'tanh', 0.0, rate, scale, shift)
    close(dsums, dsums_m.cpu().numpy(), 1e-5, 1e-6)


def test_batchnorm_infer():
    from gennet_amd import ops
    rng = np.random.RandomState(9)
    x = f32(rng.randn(4, 30, 64)); gamma = f32(rng.rand(64) + 0.5); beta = f32(rng.randn(64)); mm = f32(rng.randn(64)); mv = f32(rng.rand(64) + 0.1)
    scale, shift = ops.bn_infer_coeffs(g(gamma), g(beta), g(mm), g(mv), K.BN_EPS)
    y = ops.bn_apply(g(x).reshape(-1, 64), scale, shift, None, 'tanh')
    close(y, np.tanh(K.bn_infer_fwd(x, gamma, beta, mm, mv)).reshape(-1, 64))


def test_losses():
    from gennet_amd import ops
    rng = np.random.RandomState(13)
    B = 300
    p = f32(rng.rand(B, 1)); p[:3, 0] = [0.0, 1.0, 0.5]; y = (rng.rand(B, 1) > 0.5).astype(np.float64)
    l_ref, dp_ref = K.bce_loss(p, y)
    dp, out = ops.loss('binary_crossentropy', g(p), g(y))
    o = out.cpu().numpy()
    assert abs(o[0] - l_ref) <= 5e-6 * abs(l_ref)
    assert o[1] == np.sum(np.round(p) == y)
    close(dp, dp_ref, 2e-5)
    pm = f32(rng.randn(B, 1) * 3 + 25); ym = f32(rng.uniform(20, 35, (B, 1)))
    l_ref, dp_ref = K.mse_loss(pm, ym)
    dp, out = ops.loss('mean_squared_error', g(pm), g(ym))
    assert abs(out.cpu().numpy()[0] - l_ref) <= 2e-6 * abs(l_ref)
    close(dp, dp_ref, 1e-6)
    # data-parallel form: local rows, global mean
    dp2, out2 = ops.loss('mean_squared_error', g(pm[:100]), g(ym[:100]), Bglobal=B)
    close(dp2, dp_ref[:100], 1e-6)


def test_adam_keras_form():
    from gennet_amd import ops
    rng = np.random.RandomState(17)
    n = 100003
    p = f32(rng.randn(n)); m = np.zeros(n); v = np.zeros(n)
    pt, mt, vt = g(p), g(m), g(v)
    pr = p.copy()
    for t in (1, 2, 3):
        gr = f32(rng.randn(n) * 0.01)
        pr, m, v = K.adam_step(pr, gr, m, v, t)
        lr32, b2 = float(np.float32(9e-5)), float(np.float32(0.999))
        lr_t = lr32 * np.sqrt(1 - b2 ** t) / (1 - 0.5 ** t)
        ops.adam_step(pt, g(gr), mt, vt, lr_t, 0.5, 0.999, 1e-7)
    # beta_2 is the float32 variable 0.999f on both sides (1 - 0.999f = 0.00099998713), so v agrees to fp32 rounding
    close(pt, pr, 1e-6); close(mt, m, 1e-6); close(vt, v, 2e-6)


def test_rng_fills_and_gather():
    from gennet_amd import ops
    u = ops.fill_uniform((1 << 20,), -1.0, 1.0, 42, 0, dev())
    assert u.min().item() >= -1.0 and u.max().item() < 1.0
    assert abs(u.mean().item()) < 3e-3 and abs(u.var().item() - 1 / 3) < 3e-3
    z = ops.fill_normal((1 << 20,), 0.0, 1.0, 42, 0, dev())
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1.0) < 5e-3
    assert torch.isfinite(z).all()
    src = torch.arange(50 * 7, dtype=torch.float32, device=dev()).reshape(50, 7)
    idx = torch.tensor([3, 49, 0, 3], dtype=torch.int64, device=dev())
    assert torch.equal(ops.gather_rows(src, idx), src[idx])


def test_bad_arguments_raise_not_crash():
    from gennet_amd import ops, _lib
    x = torch.zeros((1, 8, 6), device=dev()); w = torch.zeros((5, 6, 10), device=dev())     # Cin % 4 != 0
    with pytest.raises(_lib.GennetHipError):
        ops.conv1d_fwd(x, w, None, 1, 2, 8)
    with pytest.raises(_lib.GennetHipError):
        ops.conv1d_fwd(torch.zeros((1, 8, 16)), torch.zeros((5, 16, 16)), None, 1, 2, 8)       # CPU tensors: no fallback


FULL_SIZE_LAYERS = [
    # the BASELINE-size layer shapes of the three networks at n_pix = 2048: (L, Cin, Cout, stride, padding)
    (2048, 512, 1024, 1, 'same'),      # generator conv5 (74 % of G)
    (1018, 512, 1024, 2, 'valid'),     # point-estimator q-branch conv5
    (2040, 256, 512, 2, 'valid'),      # q-branch conv4
    (1024, 512, 1024, 2, 'same'),      # discriminator conv2 after the width-2 fold
    (2048, 1024, 1, 1, 'same'),        # generator output conv
    (2048, 2, 512, 2, 'same'),         # discriminator conv1 after the fold
]


@pytest.mark.parametrize("L,Cin,Cout,s,padding", FULL_SIZE_LAYERS)
def test_full_size_layers_spot_checked_against_the_definition(L, Cin, Cout, s, padding):
    """Full BASELINE layer shapes: the oracle cannot run them whole in seconds, so 300 randomly chosen output / gradient elements
    are recomputed in fp64 straight from the definition y[b,t,co] = bias + sum_{k,ci} x[b, s t + k - pl, ci] w[k,ci,co]
    (and its two adjoints).  Tolerance 2e-5 of the output scale (K up to 2560-term fp32 chains)."""
    from gennet_amd import ops
    rng = np.random.RandomState(L + Cin)
    B, k = 3, 5
    x = rng.randn(B, L, Cin).astype(np.float32); w = (rng.randn(k, Cin, Cout) / np.sqrt(k * Cin)).astype(np.float32); b = rng.randn(Cout).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, k, s, padding)
    xd, wd = g(x), g(w)
    y = ops.conv1d_fwd(xd, wd, g(b), s, pl, Lout).cpu().numpy()
    dy = rng.randn(B, Lout, Cout).astype(np.float32)
    dyd = g(dy)
    dx = ops.conv1d_dgrad(dyd, ops.conv1d_transpose_w(wd), L, s, pl).cpu().numpy()
    dw, db = ops.conv1d_wgrad(xd, dyd, k, s, pl)
    dw = dw.cpu().numpy(); db = db.cpu().numpy()
    x64, w64, dy64 = x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
    ys = np.abs(y).max(); dxs = np.abs(dx).max(); dws = np.abs(dw).max()
    for _ in range(300):
        bb, t, co = rng.randint(B), rng.randint(Lout), rng.randint(Cout)
        acc = float(b[co])
        for kk in range(k):
            tt = s * t + kk - pl
            if 0 <= tt < L:
                acc += x64[bb, tt] @ w64[kk, :, co]
        assert abs(y[bb, t, co] - acc) <= 2e-5 * ys
        tau, ci = rng.randint(L), rng.randint(Cin)
        acc = 0.0
        for kk in range(k):
            num = tau + pl - kk
            if num % s == 0 and 0 <= num // s < Lout:
                acc += dy64[bb, num // s] @ w64[kk, ci]
        assert abs(dx[bb, tau, ci] - acc) <= 2e-5 * dxs
    for _ in range(40):
        kk, ci, co = rng.randint(k), rng.randint(Cin), rng.randint(Cout)
        ts_ = np.arange(Lout); src = s * ts_ + kk - pl
        ok = (src >= 0) & (src < L)
        acc = sum(x64[bb, src[ok], ci] @ dy64[bb, ts_[ok], co] for bb in range(B))
        assert abs(dw[kk, ci, co] - acc) <= 5e-5 * dws
    assert np.abs(db - dy64.sum(axis=(0, 1))).max() <= 5e-5 * np.abs(db).max()




@pytest.mark.parametrize("B,H,W,C", [(3, 32, 2, 256), (2, 7, 2, 5), (1, 2, 1, 1), (4, 1024, 2, 64)])
def test_maxpool_h2_bit_exact(B, H, W, C):
    """MaxPooling2D(pool_size=(2,1)) (bbhMahoGANy.py:444-490, `maxpool = True`): comparisons and copies only, so bit-identical to the oracle -- with the
    Dropout zeros in front of it (exact ties: gradient to the first row of the pair) and an odd last row (dropped, zero gradient)."""
    from gennet_amd import ops
    rng = np.random.RandomState(H + C)
    x = (rng.randn(B, H, W, C) * (rng.rand(B, H, W, C) >= 0.4)).astype(np.float32)
    x[0, :2] = x[0, 0]                                           # a non-zero tie
    y_ref, second = K.maxpool_h2_fwd(x)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    y = ops.maxpool_h2_fwd(g(x))
    assert y.shape == y_ref.shape and np.array_equal(y.cpu().numpy(), y_ref)
    dx = ops.maxpool_h2_bwd(g(dy), g(x))
    assert np.array_equal(dx.cpu().numpy(), K.maxpool_h2_bwd(dy, second, H))
