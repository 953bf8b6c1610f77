"""Correctness at the benchmark's OWN batch sizes (BASELINE configs[1] / [2]: n_pix 2048, CNN batch 256, GAN batch 512), where the fp64 oracle
is too slow to follow: size-independent identities of the three train_on_batch calls the loop makes (bbhMahoGANy.py:1165, :1292, :1296).

D and the CNN have no BatchNormalization, so with the loss a mean over rows and the dropout masks injected
    loss(batch) = mean over chunks of loss(chunk),      flat gradient(batch) = mean over chunks of flat gradient(chunk)
for chunks of 64 rows run through the SAME kernels (small launches of which the oracle parity tests cover, tests/test_nets_gpu.py) -- 1e-5 on
the loss, 1e-4 of each tensor's largest entry on the gradient (split-K / batch-order summation differs between the two runs).  generator.predict
is row-independent in the inference phase: 512 rows at once equal 8 x 64 rows bit for bit.  G's training-phase BatchNormalization statistics at
batch 512 are checked against fp64 sums of the captured pre-BN tensors.  Weights are held still by Adam(lr = 0).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_PIX = 2048
CHUNK = 64


def flat_grads(model):
    from gennet_amd.engine import segments
    return [grp.grad[a:b].clone() for grp, a, b in segments(model._train_params)]


def per_param_rel(model, ga, gb, gabs):
    """max |ga - gb| per trainable tensor, relative to the larger of max |gb| and max of gabs = the mean over chunks of |chunk gradient|: the
    size of the terms that were summed (a one-element bias gradient is a sum of 1024 values of both signs that nearly cancels at
    initialisation; its rounding error scales with the terms, not with the small sum)."""
    from gennet_amd.engine import segments
    segs = segments(model._train_params)
    out = {}
    for p in model._train_params:
        for (grp, a, b), fa, fb, fs in zip(segs, ga, gb, gabs):
            if grp is p.group and a <= p.offset and p.offset + p.size <= b:
                sl = slice(p.offset - a, p.offset - a + p.size)
                scale = torch.maximum(fb[sl].abs().max(), fs[sl].max()).clamp_min(1e-30)
                out[p.name + str(tuple(p.shape))] = float((fa[sl] - fb[sl]).abs().max() / scale)
    return out


def test_discriminator_step_on_2x512_rows_equals_its_64_row_chunks():
    from gennet_amd import bbh, ops
    from gennet_amd.engine import Adam, device, set_init_seed
    from gennet_amd.layers import Dropout
    B = 512
    set_init_seed(7)
    D = bbh.signal_discriminator_model(N_PIX)
    D.compile(loss='binary_crossentropy', optimizer=Adam(lr=0.0, beta_1=0.5), metrics=['accuracy'])
    sX = ops.fill_normal((2 * B, N_PIX, 2, 1), 0.0, 1.0, 11, 0, device())
    sy = torch.cat([torch.ones(B, device=device()), torch.zeros(B, device=device())])
    drops = [l for l in D.layers if isinstance(l, Dropout)]
    shapes = [(2 * B, N_PIX // 2, 2, 256), (2 * B, N_PIX // 4, 2, 512)]
    masks = {l.name: ops.dropout_mask(shp, l.rate, 99, 1 << 28 if k else 0, device()) for k, (l, shp) in enumerate(zip(drops, shapes))}
    assert all(0.55 < float(m.float().mean()) < 0.65 for m in masks.values())                      # keep probability 0.6
    full = D.train_on_batch(sX, sy, dropout_masks=masks)
    g_full = flat_grads(D)
    assert np.isfinite(full).all() and 0.2 < full[0] < 3.0
    acc = gabs = None
    losses, hits = [], []
    for s in range(0, 2 * B, CHUNK):
        r = D.train_on_batch(sX[s:s + CHUNK], sy[s:s + CHUNK], dropout_masks={k: m[s:s + CHUNK] for k, m in masks.items()})
        losses.append(r[0]); hits.append(r[1])
        g = flat_grads(D)
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        gabs = [x.abs() for x in g] if gabs is None else [a + b.abs() for a, b in zip(gabs, g)]
    n_chunks = 2 * B // CHUNK
    acc = [a / n_chunks for a in acc]; gabs = [a / n_chunks for a in gabs]
    assert abs(full[0] - np.mean(losses)) <= 1e-5 * abs(np.mean(losses)), (full, np.mean(losses))
    assert abs(full[1] - np.mean(hits)) < 1e-9                                                      # binary accuracy is a count
    rels = per_param_rel(D, g_full, acc, gabs)
    assert max(rels.values()) < 1e-4, rels
    w0 = [p.data.clone() for p in D._train_params]
    D.train_on_batch(sX[:CHUNK], sy[:CHUNK])
    assert all(torch.equal(a, p.data) for a, p in zip(w0, D._train_params))                         # lr = 0: the weights never moved


def test_pe_step_on_256_rows_equals_its_64_row_chunks():
    from gennet_amd import bbh, ops
    from gennet_amd.engine import Adam, device, set_init_seed
    B = 256
    set_init_seed(8)
    pe = bbh.signal_pe_model(N_PIX)
    for l in pe.layers:                                   # heads inside the active range of relu / relu(max 1), as a trained net's are
        if l.__class__.__name__ == 'Dense':
            l.set_weights([l.get_weights()[0], np.array([25.0 if l.get_weights()[0].shape[0] == 64000 else 0.6], np.float32)])
    pe.compile(loss='mean_squared_error', optimizer=Adam(lr=0.0, beta_1=0.5), metrics=['accuracy'])
    x = ops.fill_normal((B, N_PIX, 1), 0.0, 1.0, 12, 0, device())
    ymc = ops.fill_uniform((B,), 20.0, 35.0, 13, 0, device()); yq = ops.fill_uniform((B,), 0.5, 1.0, 14, 0, device())
    full = pe.train_on_batch(x, [ymc, yq])
    g_full = flat_grads(pe)
    assert len(full) == 5 and np.isfinite(full).all()
    acc, gabs, rows = None, None, []
    for s in range(0, B, CHUNK):
        rows.append(pe.train_on_batch(x[s:s + CHUNK], [ymc[s:s + CHUNK], yq[s:s + CHUNK]]))
        g = flat_grads(pe)
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        gabs = [v.abs() for v in g] if gabs is None else [a + b.abs() for a, b in zip(gabs, g)]
    acc = [a / (B // CHUNK) for a in acc]; gabs = [a / (B // CHUNK) for a in gabs]
    mean = np.mean(np.array(rows), axis=0)
    for k in range(3):                                     # total, mc loss, q loss
        assert abs(full[k] - mean[k]) <= 1e-5 * abs(mean[k]) + 1e-9, (k, full, mean)
    rels = per_param_rel(pe, g_full, acc, gabs)
    assert max(rels.values()) < 1e-4, rels
    # predict: 256 rows at once == 4 x 64 rows, bit for bit (row-independent kernels, fixed reduction order per output)
    p_full = pe.predict_device(x, batch_size=B)
    p_chunks = pe.predict_device(x, batch_size=CHUNK)
    assert torch.equal(p_full[0], p_chunks[0]) and torch.equal(p_full[1], p_chunks[1])


def test_generator_at_batch_512_predict_chunks_and_batchnorm_statistics():
    from gennet_amd import bbh, ops
    from gennet_amd.engine import device, set_device_seed, set_init_seed
    B = 512
    set_init_seed(9); set_device_seed(21)
    event = np.random.RandomState(5).randn(N_PIX, 1).astype(np.float32)
    nets = bbh.build_and_compile(event, N_PIX, do_pe=False)
    G = nets.generator
    z = ops.fill_uniform((B, 100), -1.0, 1.0, 15, 0, device())
    a = G.predict_device(z, batch_size=B)
    b = G.predict_device(z, batch_size=CHUNK)
    assert a.shape == (B, N_PIX, 1) and torch.isfinite(a).all() and torch.equal(a, b)
    # one G step through the frozen D at batch 512 (bbhMahoGANy.py:1296): every BatchNormalization's batch statistics against fp64 sums of
    # the captured pre-BN tensor.  From the fresh state ONE zero-debiased update leaves moving_mean = batch mean and
    # moving_variance = batch variance * n / (n - (1 + eps)), so the moving statistics ARE the batch statistics.
    d_before = [p.data.clone() for p in nets.signal_discriminator._train_params]
    cap = {}
    r = nets.signal_discriminator_on_generator.train_on_batch(z, torch.ones(B, device=device()), capture=cap)
    assert np.isfinite(r).all()
    tops = G._top
    bns = [(i, l) for i, l in enumerate(tops) if l.__class__.__name__ == 'BatchNormalization']
    assert len(bns) == 6
    for i, bn in bns:
        prod = tops[i - 1]
        assert prod.__class__.__name__ in ('Dense', 'Conv1D')
        pre = cap[prod.name].double().reshape(-1, bn.gamma.shape[0])
        n = pre.shape[0]
        mean = pre.mean(0); var = pre.var(0, unbiased=False)
        mm = bn.moving_mean.data.double(); mv = bn.moving_variance.data.double()
        assert float((mm - mean).abs().max()) <= 2e-6 * float(var.sqrt().max()) + 1e-7, (bn.name, n)
        want = var * n / (n - (1.0 + bn.epsilon))
        # TF's update is variable -= variable - biased / (1 - m^t) in fp32: the difference from the OLD value 1.0 is rounded at 1.0's ulp
        # (6e-8), an absolute error that the first layer's small variances (~2e-4) see as 1e-4 relative -- the reference arithmetic's own
        assert float(((mv - want).abs() - 1.2e-7).clamp_min(0).div(want).max()) <= 2e-5, (bn.name, n)
    del cap
    assert all(torch.equal(a_, p.data) for a_, p in zip(d_before, nets.signal_discriminator._train_params))      # D frozen in the G step


def test_largest_conv_launches_of_the_bench_against_their_chunks():
    """The generator's 512 -> 1024 convolution at the bench's B = 512, L = 2048: 2^30 output elements, 65 536 blocks in the XCD patch order
    (the oracle-parity cases of test_kernels_gpu stop at 720 blocks).  Rows of a convolution do not know their batch: the forward (with the
    BatchNormalization statistics out of its epilogue) and the data gradient at B = 512 must equal the SAME kernels run on 8 chunks of 64
    rows -- whose grids are small, differently ordered, and covered by the oracle tests -- bit for bit; the weight gradient, a sum over
    (b, m), equals the sum of the chunks' to 1e-5 of its largest entry; the epilogue statistics equal fp64 sums of the output."""
    from gennet_amd import ops
    from gennet_amd.engine import device
    B, L, Cin, Cout, k = 512, 2048, 512, 1024, 5
    x = ops.fill_normal((B, L, Cin), 0.0, 1.0, 31, 0, device())
    w = ops.fill_uniform((k, Cin, Cout), -0.02, 0.02, 32, 0, device())
    b = ops.fill_uniform((Cout,), -0.1, 0.1, 33, 0, device())
    y, sums = ops.conv1d_fwd_stats(x, w, b, 1, 2, L)
    assert y.shape == (B, L, Cout) and y.numel() == 1 << 30
    for s in range(0, B, CHUNK):
        yc = ops.conv1d_fwd(x[s:s + CHUNK], w, b, 1, 2, L)
        assert torch.equal(y[s:s + CHUNK], yc), s
    del yc
    n = B * L
    s1 = torch.zeros(Cout, dtype=torch.float64, device=device()); s2 = torch.zeros(Cout, dtype=torch.float64, device=device())
    for s in range(0, B, CHUNK):                                  # fp64 sums of the output, chunked to bound the fp64 temporary
        yd = y[s:s + CHUNK].double().reshape(-1, Cout)
        s1 += yd.sum(0); s2 += (yd * yd).sum(0)
    del yd
    assert float(((sums[:Cout] - s1).abs() / (s2.sqrt() * np.sqrt(n))).max()) < 1e-9          # sum y against sqrt(n sum y^2): its natural scale
    assert float(((sums[Cout:] - s2).abs() / s2).max()) < 1e-9
    # data gradient (transposed weights): dy := y
    wt = ops.conv1d_transpose_w(w)
    dx = ops.conv1d_dgrad(y, wt, L, 1, 2)
    for s in range(0, B, CHUNK):
        assert torch.equal(dx[s:s + CHUNK], ops.conv1d_dgrad(y[s:s + CHUNK], wt, L, 1, 2)), s
    del dx
    # weight gradient: sum over the chunks
    dw, db = ops.conv1d_wgrad(x, y, k, 1, 2)
    dw_acc = torch.zeros_like(dw, dtype=torch.float64); db_acc = torch.zeros_like(db, dtype=torch.float64)
    for s in range(0, B, CHUNK):
        dwc, dbc = ops.conv1d_wgrad(x[s:s + CHUNK], y[s:s + CHUNK], k, 1, 2)
        dw_acc += dwc.double(); db_acc += dbc.double()
    assert float((dw.double() - dw_acc).abs().max() / dw_acc.abs().max()) < 1e-5
    assert float((db.double() - db_acc).abs().max() / db_acc.abs().max()) < 1e-5


def test_stride2_discriminator_launch_of_the_bench_against_its_chunks():
    """The discriminator's folded second convolution at the bench's 2B = 1024 rows (512 -> 1024 channels, 5 taps, stride 2, L 1024 -> 512) with
    LeakyReLU + injected dropout in the epilogue, and its two-phase data gradient through the producer's LeakyReLU + dropout: bit-identical to
    64-row chunks."""
    from gennet_amd import ops
    from gennet_amd.engine import device
    B, L, Cin, Cout, k = 1024, 1024, 512, 1024, 5
    Lout, pad = 512, 1                                            # 'same' with stride 2: pad_total = 3, pad_left = 1
    x = ops.fill_normal((B, L, Cin), 0.0, 1.0, 41, 0, device())
    w = ops.fill_uniform((k, Cin, Cout), -0.02, 0.02, 42, 0, device())
    b = ops.fill_uniform((Cout,), -0.1, 0.1, 43, 0, device())
    mask = ops.dropout_mask((B, Lout, Cout), 0.4, 44, 0, device())
    y = ops.conv1d_fwd_dropout(x, w, b, mask, 2, pad, Lout, 'leaky', 0.2, 0.4)
    for s in range(0, B, CHUNK):
        assert torch.equal(y[s:s + CHUNK], ops.conv1d_fwd_dropout(x[s:s + CHUNK], w, b, mask[s:s + CHUNK], 2, pad, Lout, 'leaky', 0.2, 0.4)), s
    assert 0.55 < float((y != 0).float().mean()) < 0.65
    # data gradient of that layer fused with the PRODUCER's LeakyReLU + dropout derivative (prev = (its output, act, param, its mask, rate))
    wt = ops.conv1d_transpose_w(w)
    pmask = ops.dropout_mask((B, L, Cin), 0.4, 45, 0, device())
    xprev = torch.where(pmask.bool(), x, torch.zeros_like(x))    # a producer output consistent with its mask
    dx = ops.conv1d_dgrad(y, wt, L, 2, pad, prev=(xprev, 'leaky', 0.2, pmask, 0.4))
    for s in range(0, B, CHUNK):
        dxc = ops.conv1d_dgrad(y[s:s + CHUNK], wt, L, 2, pad, prev=(xprev[s:s + CHUNK], 'leaky', 0.2, pmask[s:s + CHUNK], 0.4))
        assert torch.equal(dx[s:s + CHUNK], dxc), s
