"""Fused passes against their unfused definitions, on the GPU through the C ABI:
  * gn_bn_apply_dropgen == gn_dropout_mask followed by gn_bn_apply, bit for bit (same Philox draw, same arithmetic);
  * gn_conv_fold_bn: conv with the folded weights == BN_infer(conv) within fp32 rounding, and the engine's predict path (which
    folds) equals the same model evaluated layer by layer without the fold."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('rows,C,act,rate', [(37, 8, 'tanh', 0.2), (1000, 64, 'tanh', 0.2), (64, 1024, 'linear', 0.4), (5, 4, 'relu', 0.0)])
def test_bn_apply_dropgen_is_mask_then_apply(rows, C, act, rate):
    from gennet_amd import ops
    g = torch.Generator(device='cpu'); g.manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(dev()); scale = (torch.rand(C, generator=g) + 0.5).to(dev()); shift = torch.randn(C, generator=g).to(dev())
    seed, off = 1234, 77
    mask_ref = ops.dropout_mask((rows, C), rate, seed, off, dev())
    y_ref = ops.bn_apply(x, scale, shift, mask_ref, act, 0.0, rate)
    y, mask = ops.bn_apply_dropgen(x, scale, shift, act, 0.0, rate, seed, off)
    assert torch.equal(mask, mask_ref)
    assert torch.equal(y, y_ref)
    if rate > 0 and rows * C > 10000:
        assert abs(float(mask.float().mean()) - (1 - rate)) < 0.01


def test_conv_fold_bn_matches_bn_of_conv():
    from gennet_amd import ops
    rng = np.random.RandomState(0)
    B, L, Cin, Cout, k = 3, 70, 32, 64, 5
    x = torch.tensor(rng.randn(B, L, Cin).astype(np.float32)).to(dev())
    w = torch.tensor((rng.randn(k, Cin, Cout) / np.sqrt(k * Cin)).astype(np.float32)).to(dev())
    b = torch.tensor(rng.randn(Cout).astype(np.float32)).to(dev())
    gamma = torch.tensor((rng.rand(Cout) + 0.5).astype(np.float32)).to(dev()); beta = torch.tensor(rng.randn(Cout).astype(np.float32)).to(dev())
    mm = torch.tensor(rng.randn(Cout).astype(np.float32)).to(dev()); mv = torch.tensor((rng.rand(Cout) + 0.2).astype(np.float32)).to(dev())
    Lout, pl = ops.conv_geometry(L, k, 1, 'same')
    scale, shift = ops.bn_infer_coeffs(gamma, beta, mm, mv, 1e-3)
    pre = ops.conv1d_fwd(x, w, b, 1, pl, Lout)
    ref = ops.bn_apply(pre.reshape(-1, Cout), scale, shift, None, 'tanh', 0.0).reshape(B, Lout, Cout)
    w2, b2 = ops.conv_fold_bn(w, b, scale, shift)
    got = ops.conv1d_fwd(x, w2, b2, 1, pl, Lout, 'tanh')
    assert float((got - ref).abs().max()) <= 1e-5                      # |pre-activation| up to ~10: a few fp32 roundings of it (measured 3e-6)


def test_predict_with_fold_equals_layerwise_inference():
    """generator.predict folds every Conv1D -> BatchNormalization -> tanh into one kernel; the same weights evaluated through the
    ops one layer at a time (conv, then bn_infer + tanh) must agree."""
    from gennet_amd import bbh, ops
    rng = np.random.RandomState(1)
    n_pix = 64
    G = bbh.generator_model(n_pix)
    for l in G.layers:                                                 # non-trivial moving statistics
        if l.__class__.__name__ == 'BatchNormalization':
            C = l.gamma.shape[0]
            l.moving_mean.assign((rng.randn(C) * 0.1).astype(np.float32)); l.moving_variance.assign((rng.rand(C) + 0.5).astype(np.float32))
            l.gamma.assign((rng.rand(C) + 0.5).astype(np.float32)); l.beta.assign((rng.randn(C) * 0.1).astype(np.float32))
    z = rng.uniform(-1, 1, (5, 100)).astype(np.float32)
    got = G.predict(z)
    h = torch.tensor(z).to(dev())
    layers = list(G._top)
    i = 0
    while i < len(layers):
        l = layers[i]; name = l.__class__.__name__
        if name == 'Dense':
            h = ops.dense_fwd(h, l.kernel.data, l.bias.data, 'linear')
        elif name == 'Conv1D':
            Lout, pl = ops.conv_geometry(h.shape[1], l.k, l.stride, l.padding)
            h = ops.conv1d_fwd(h, l.kernel.data, l.bias.data, l.stride, pl, Lout)
        elif name == 'BatchNormalization':
            scale, shift = ops.bn_infer_coeffs(l.gamma.data, l.beta.data, l.moving_mean.data, l.moving_variance.data, l.epsilon)
            act = 'linear'
            if i + 1 < len(layers) and layers[i + 1].__class__.__name__ == 'Activation':
                act = layers[i + 1].act_spec[0]; i += 1
            h = ops.bn_apply(h.reshape(-1, h.shape[-1]), scale, shift, None, act, 0.0).reshape(h.shape)
        elif name == 'Reshape':
            h = h.reshape((h.shape[0],) + tuple(l.target_shape))
        elif name == 'UpSampling1D':
            h = ops.upsample2_fwd(h.contiguous())
        elif name in ('Dropout', 'Activation'):
            pass                                                       # inference phase / linear
        else:
            raise AssertionError(name)
        i += 1
    ref = h.cpu().numpy()
    assert got.shape == ref.shape == (5, n_pix, 1)
    assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3)
