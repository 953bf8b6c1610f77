"""The EXPERIMENTAL opt-in convolution on the bf16 matrix cores (csrc/conv_bf16x3.hip: every fp32 operand split into three bf16
pieces, six MFMA products, fp32 accumulation).  It is not on the default path; what is checked here is the claim that makes it
worth keeping: its results are fp32-grade -- within the tolerance of the exact-fp32 MFMA path's own parity test, and not
further from the fp64 oracle than that path is."""
import os

import numpy as np
import pytest
import torch

from oracle import keras_ref as K

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # B, L, Cin, Cout, k, padding
    (2, 300, 64, 128, 5, 'same'),        # zero guard rows on both sides, ragged M tile
    (3, 277, 48, 256, 5, 'valid'),       # 3 K chunks, 4 N tiles
    (2, 200, 32, 64, 3, 'valid'),        # 3 taps (padding DMA instructions), one N tile
    (1, 128, 16, 64, 1, 'valid'),        # one tap, one chunk (Dense-like)
    (2, 100, 32, 64, 2, 'valid'),        # 2 taps: 30 DMA instructions over 4 waves -> two padding instructions
    (2, 131, 48, 128, 4, 'same'),        # 4 taps: 42 -> two padding instructions; asymmetric SAME padding
    (1, 256, 512, 1024, 5, 'same'),      # the dominant generator layer's channels
    (2, 400, 32, 64, 4, 'same'),         # round 4's 256-row blocks of 64 x 64 wave tiles: 4 taps, ragged second tile, two chunks
    (3, 700, 80, 192, 5, 'valid'),       # ... 5 taps, odd chunk count (the fragment double buffer's parity alternates per chunk), 3 N tiles, 3 M tiles
    (1, 192, 16, 64, 5, 'same'),         # ... one chunk only
    (2, 500, 80, 128, 3, 'same'),        # the three-stage form of the short-tap launches: 3 taps, 5 chunks (odd), ragged second block
    (1, 700, 64, 64, 2, 'valid'),        # ... 2 taps, 4 chunks: every chunk's DMAs go out in its first tap
    (1, 256, 1024, 512, 3, 'same'),      # ... 64 chunks: the stage ring wraps many times
]


@pytest.mark.parametrize('B,L,Cin,Cout,k,padding', CASES)
def test_bf16x3_conv_is_fp32_grade(B, L, Cin, Cout, k, padding):
    from gennet_amd import ops
    rng = np.random.RandomState(B * 131 + L + Cin)
    # wide dynamic range inside every dot product: log-normal magnitudes
    x = (rng.randn(B, L, Cin) * np.exp(rng.randn(B, L, Cin))).astype(np.float32)
    w = (rng.randn(k, Cin, Cout) / np.sqrt(k * Cin)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, k, 1, padding)
    ref = K.conv1d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 1, padding)
    ref = np.maximum(ref, 0.0)
    dev = torch.device('cuda:0')
    xt, wt, bt = (torch.tensor(v).to(dev) for v in (x, w, b))
    y32 = ops.conv1d_fwd(xt, wt, bt, 1, pl, Lout, 'relu').cpu().numpy().astype(np.float64)
    y3 = ops.conv1d_fwd_bf16x3(xt, wt, bt, 1, pl, Lout, 'relu').cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max()
    e32, e3 = np.abs(y32 - ref).max() / scale, np.abs(y3 - ref).max() / scale
    r32, r3 = np.sqrt(np.mean((y32 - ref) ** 2)) / scale, np.sqrt(np.mean((y3 - ref) ** 2)) / scale
    assert e3 <= 2e-5, e3                                   # the tolerance of test_kernels_gpu.test_conv1d_fwd_dgrad_wgrad
    assert e3 <= 2.0 * e32 + 1e-7, (e3, e32)                # not further from fp64 than the exact-fp32 MFMA path (measured: 0.75-1.1 x)
    assert r3 <= 1.5 * r32 + 1e-8, (r3, r32)


def test_bf16x3_is_non_finite_exactly_where_fp32_is():
    """A documented difference: an infinite input comes out as NaN, not inf (its hi piece meets the zero lo piece of the other
    operand: inf * 0).  What is kept: outputs are non-finite exactly where the fp32 path's are, and overflow of the sum is inf."""
    from gennet_amd import ops
    dev = torch.device('cuda:0')
    x = torch.zeros(1, 64, 16, device=dev); w = torch.ones(1, 16, 64, device=dev); b = torch.zeros(64, device=dev)
    x[0, 3, 2] = float('inf'); x[0, 9, 5] = float('nan'); x[0, 20, 1] = 3.0e38; x[0, 20, 2] = 3.0e38
    y32 = ops.conv1d_fwd(x, w, b, 1, 0, 64).cpu().numpy()
    y3 = ops.conv1d_fwd_bf16x3(x, w, b, 1, 0, 64).cpu().numpy()
    assert np.array_equal(np.isfinite(y3), np.isfinite(y32))
    assert np.isposinf(y32[0, 3]).all() and np.isnan(y3[0, 3]).all()
    assert np.isnan(y3[0, 9]).all() and np.isnan(y32[0, 9]).all()
    assert np.isposinf(y3[0, 20]).all() and np.isposinf(y32[0, 20]).all()          # overflow of the fp32 sum
    assert (y3[0, 30] == 0).all()


def test_bf16x3_rejects_unsupported_shapes():
    from gennet_amd import _lib, ops
    dev = torch.device('cuda:0')
    x = torch.zeros(1, 64, 20, device=dev); w = torch.ones(5, 20, 64, device=dev); b = torch.zeros(64, device=dev)
    with pytest.raises(_lib.GennetHipError):
        ops.conv1d_fwd_bf16x3(x, w, b, 1, 2, 64)                                  # Cin % 16 != 0
    x = torch.zeros(1, 64, 32, device=dev); w = torch.ones(5, 32, 64, device=dev)
    with pytest.raises(_lib.GennetHipError):
        ops.conv1d_fwd_bf16x3(x, w, b, 2, 1, 32)                                  # stride 2 is not implemented


_S2_DGRAD_CASES = [(2, 800, 256, 256, 'same'),        # pad_left 1: taps (1, 2) and (3, 4) of the merged launch read the same rows
                   (1, 1030, 256, 512, 'valid'),     # pad_left 0: taps (0, 1) and (2, 3); 515 rows per phase
                   (2, 801, 256, 256, 'same')]       # odd length (pad_left 2): the odd phase is one row shorter than the even one


def _s2_dgrad(B, L, Cin, Cout, padding):
    """(result under the opt-in math, split launches it took, exact-kernel result, fp64 definition)"""
    from gennet_amd import ops
    rng = np.random.RandomState(L + Cin)
    w = (rng.randn(5, Cin, Cout) / np.sqrt(5 * Cin)).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, 5, 2, padding)
    dy = (rng.randn(B, Lout, Cout) * np.exp(rng.randn(B, Lout, Cout))).astype(np.float32)
    ref = K.conv1d_bwd(np.zeros((B, L, Cin)), w.astype(np.float64), dy.astype(np.float64), 2, padding)[0]
    dev = torch.device('cuda:0')
    dyt, wt = torch.tensor(dy).to(dev), ops.conv1d_transpose_w(torch.tensor(w).to(dev))
    exact = ops.conv1d_dgrad(dyt, wt, L, 2, pl).cpu().numpy()
    ops.prof_enable(True); ops.prof_reset()
    ops.set_conv_math('bf16x3', workspace_gb=0.25)
    try:
        got = ops.conv1d_dgrad(dyt, wt, L, 2, pl).cpu().numpy()
        used = ops.prof_collect(2)['launches']
    finally:
        ops.set_conv_math()
        ops.prof_enable(False)
    return got, used, exact, ref


@pytest.mark.parametrize('B,L,Cin,Cout,padding', _S2_DGRAD_CASES)
def test_stride2_data_gradient_under_the_opt_in_split(B, L, Cin, Cout, padding):
    """The data gradient of a stride-2 convolution under the opt-in math: ONE launch of the 64 x 64-wave-tile kernel with two accumulator sets (five taps in
    kernel order, even taps -> one output phase, odd taps -> the other; tap pairs that read the same rows share their x fragments).  Checked against the
    fp64 definition at the exact path's tolerance."""
    got, used, exact, ref = _s2_dgrad(B, L, Cin, Cout, padding)
    assert used == (2 if os.environ.get('GN_BF16X3_NO_MERGE') else 1), used
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-6 * scale, np.abs(got - ref).max() / scale
    assert np.abs(got - ref).max() <= 2.0 * np.abs(exact - ref).max() + 1e-7 * scale


def test_stride2_data_gradient_as_two_phase_launches(tmp_path):
    """GN_BF16X3_NO_MERGE=1 (A/B switch, read once per process: a child process runs this leg): the two output phases as launches of 3 and 2 taps on the
    three-stage kernel, the second phase reusing the first one's split planes.  Same bounds."""
    import subprocess
    import sys
    out = str(tmp_path / 'phases.npz')
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch; import test_bf16x3_gpu as T; "
            "r = [T._s2_dgrad(*c) for c in T._S2_DGRAD_CASES]; assert all(x[1] == 2 for x in r), [x[1] for x in r]; "
            "np.savez(%r, *[x[0] for x in r])") % (ROOT, os.path.join(ROOT, 'tests'), out)
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, GN_BF16X3_NO_MERGE='1'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    phases = np.load(out)
    for k, case in enumerate(_S2_DGRAD_CASES):
        got, used, exact, ref = _s2_dgrad(*case)
        scale = np.abs(ref).max()
        two = phases['arr_%d' % k]
        assert np.abs(two - ref).max() <= 2e-6 * scale, (case, np.abs(two - ref).max() / scale)
        assert np.abs(two - got).max() <= 2e-6 * scale                      # the merged launch and the two phases agree to rounding


@pytest.mark.parametrize('B,L,Cin,Cout,padding,stride', [
    (2, 300, 256, 256, 'same', 1),       # 10 chunks per batch element, the last one ragged (M = 300)
    (1, 100, 256, 320, 'valid', 1),      # M = 96: whole chunks; 5 column tiles
    (3, 77, 384, 256, 'same', 1),        # 3 row tiles of 128 ci, M = 77
    (4, 20, 256, 256, 'same', 1),        # a batch element shorter than one chunk
    (2, 1024, 512, 1024, 'same', 1),     # the dominant generator layer's channels; K split inside a batch element
    (2, 600, 512, 256, 'same', 2),       # stride 2 (even / odd input rows as two classes; from 512 input channels on): pad_left 1 -> the odd class leads
    (1, 601, 512, 320, 'same', 2),       # ... pad_left 2
    (3, 155, 640, 256, 'valid', 2),      # ... pad_left 0, M = 76
    (2, 1024, 512, 1024, 'same', 2)])    # ... the discriminator's / point-estimator's channel counts
def test_weight_gradient_under_the_opt_in_split(B, L, Cin, Cout, padding, stride):
    """The weight gradient on the bf16 matrix cores (csrc/wgrad_bf16x3.hip: operands split AND transposed, the tap shift done in registers): against the
    fp64 definition at the exact kernel's tolerance and not further from it than the exact kernel is; the bias gradient still comes out."""
    from gennet_amd import ops
    rng = np.random.RandomState(B + L + Cin)
    x = (rng.randn(B, L, Cin) * np.exp(rng.randn(B, L, Cin))).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, 5, stride, padding)
    dy = (rng.randn(B, Lout, Cout) * np.exp(rng.randn(B, Lout, Cout))).astype(np.float32)
    _, ref, refb = K.conv1d_bwd(x.astype(np.float64), np.zeros((5, Cin, Cout)), dy.astype(np.float64), stride, padding)
    dev = torch.device('cuda:0')
    xt, dyt = torch.tensor(x).to(dev), torch.tensor(dy).to(dev)
    exact = ops.conv1d_wgrad(xt, dyt, 5, stride, pl)[0].cpu().numpy()
    ops.prof_enable(True); ops.prof_reset()
    ops.set_conv_math('bf16x3', workspace_gb=0.25)
    try:
        dw, db = ops.conv1d_wgrad(xt, dyt, 5, stride, pl)
        dw, db = dw.cpu().numpy(), db.cpu().numpy()
        used = ops.prof_collect(2)['launches']
    finally:
        ops.set_conv_math()
        ops.prof_enable(False)
    assert used == 1, used
    scale = np.abs(ref).max()
    e3, e32 = np.abs(dw - ref).max() / scale, np.abs(exact - ref).max() / scale
    assert e3 <= 2e-5, e3
    assert e3 <= 2.0 * e32 + 1e-7, (e3, e32)
    assert np.sqrt(np.mean((dw - ref) ** 2)) <= 1.5 * np.sqrt(np.mean((exact - ref) ** 2)) + 1e-8 * scale
    assert np.abs(db - refb).max() <= 1e-5 * np.abs(refb).max()


def test_gan_iteration_under_the_opt_in_split_meets_the_fp32_path_tolerances():
    """The opt-in conv math on the whole path (ops.set_conv_math('bf16x3'): the generator's 256 -> 512 and 512 -> 1024 convolutions, forward and
    data gradient, run as six bf16 products on 64 x 64 wave tiles) against the fp64 oracle with the SAME tolerances as
    test_nets_gpu.test_gan_iteration_matches_oracle asserts for the exact-fp32 kernels: losses 2e-5, gradients 3e-4 of each tensor's largest
    entry (+ 1e-6 of the largest gradient), generator.predict 5e-5.  n_pix 256: the two layers' rows (256) fill the 256-row blocks."""
    from gennet_amd import bbh, ops
    from gennet_amd.engine import to_device
    from oracle import nets_ref as N                                                   # noqa: F401
    import test_nets_gpu as T
    n_pix, B = 256, 3
    rng = np.random.RandomState(17)
    ref, nets, event = T._build_gan(n_pix, rng)
    G, D, DG = nets.generator, nets.signal_discriminator, nets.signal_discriminator_on_generator
    ops.prof_enable(True); ops.prof_reset()
    ops.set_conv_math('bf16x3', workspace_gb=0.5)
    try:
        z = T.f32(rng.uniform(-1, 1, (B, 100)))
        fake_ref = ref.generate(z)
        assert T.rel(G.predict(z), fake_ref) < 5e-5
        z2 = T.f32(rng.uniform(-1, 1, (B, 100)))
        g_masks = T.stack_masks(ref.G, z2, rng)
        d_masks2 = T.stack_masks(ref.D, (z2.shape[0], n_pix, 2, 1), rng)
        names = dict(T.masks_by_name(ref.G, g_masks, G.layers)); names.update(T.masks_by_name(ref.D, d_masks2, D.layers))
        cap = {}
        out = DG.train_on_batch(z2, [1] * B, dropout_masks=names, capture=cap)
        out_ref = ref.g_train_on_batch(z2, [1] * B, g_masks, d_masks2, T.decisions_for(ref.D, D.layers, cap))
        del cap
        T.assert_decisions_consistent(ref.D)
        assert abs(out[0] - out_ref[0]) <= 2e-5 * abs(out_ref[0]) and out[1] == pytest.approx(out_ref[1])
        ggr = [p.grad.cpu().numpy() for l in G.layers for p in l.params]
        gmax = max(np.abs(gr).max() for gr in ref.last_g_grads)
        for k, (gq, gr) in enumerate(zip(ggr, ref.last_g_grads)):
            assert np.abs(gq - gr).max() <= 3e-4 * np.abs(gr).max() + 1e-6 * gmax, (k, T.rel(gq, gr))
        used = ops.prof_collect(2)['launches']
    finally:
        ops.set_conv_math()
        ops.prof_enable(False)
    assert used >= 4, used              # predict (2 layers) + train forward (2) + data gradients: the split kernels really ran


@pytest.mark.parametrize('B,L,Cin,Cout,padding', [(2, 700, 64, 128, 'same'),        # 350 output rows: two 256-row blocks, the second ragged; zero guard rows on both sides
                                                  (1, 600, 48, 192, 'valid'),       # 298 rows, 3 chunks (odd), 3 column tiles
                                                  (3, 1030, 32, 64, 'same'),        # 515 rows: three blocks, the last with 3 live rows
                                                  (1, 512, 512, 1024, 'same')])     # the discriminator's / point-estimator's channel counts
def test_bf16x3_stride2_forward_is_fp32_grade(B, L, Cin, Cout, padding):
    """Round 4: the stride-2 forward on the wide split kernel (de-interleaved slab: even input rows, then odd ones; the partial last DMA segment of
    every region is lane-masked).  Same bounds as the unit-stride cases."""
    from gennet_amd import ops
    rng = np.random.RandomState(B * 17 + L + Cin)
    x = (rng.randn(B, L, Cin) * np.exp(rng.randn(B, L, Cin))).astype(np.float32)
    w = (rng.randn(5, Cin, Cout) / np.sqrt(5 * Cin)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, 5, 2, padding)
    ref = np.maximum(K.conv1d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 2, padding), 0.0)
    dev = torch.device('cuda:0')
    xt, wt, bt = (torch.tensor(v).to(dev) for v in (x, w, b))
    y32 = ops.conv1d_fwd(xt, wt, bt, 2, pl, Lout, 'relu').cpu().numpy().astype(np.float64)
    y3 = ops.conv1d_fwd_bf16x3(xt, wt, bt, 2, pl, Lout, 'relu').cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max()
    e32, e3 = np.abs(y32 - ref).max() / scale, np.abs(y3 - ref).max() / scale
    r32, r3 = np.sqrt(np.mean((y32 - ref) ** 2)) / scale, np.sqrt(np.mean((y3 - ref) ** 2)) / scale
    assert e3 <= 2e-5, e3
    assert e3 <= 2.0 * e32 + 1e-7, (e3, e32)
    assert r3 <= 1.5 * r32 + 1e-8, (r3, r32)


def test_discriminator_step_under_the_opt_in_split_meets_the_fp32_path_tolerances():
    """The discriminator's folded width-2 Conv2D(256 -> 512, stride (2, 1)) -- a stride-2 Conv1D with 512 -> 1024 channels -- on the split kernel's
    stride-2 form with the fused dropout epilogue, inside a whole train_on_batch at n_pix 1024 (256 output rows: one full block), against the oracle
    at the tolerances test_nets_gpu asserts for the exact kernels (loss 2e-5, gradients 2e-4)."""
    from gennet_amd import bbh, ops
    import test_nets_gpu as T
    n_pix, B = 1024, 2
    rng = np.random.RandomState(23)
    ref, nets, event = T._build_gan(n_pix, rng)
    D = nets.signal_discriminator
    sX = T.f32(rng.randn(2 * B, n_pix, 2, 1)); sy = [1.0] * B + [0.0] * B
    d_masks = T.stack_masks(ref.D, sX, rng)
    ops.prof_enable(True); ops.prof_reset()
    ops.set_conv_math('bf16x3', workspace_gb=0.5)
    try:
        cap = {}
        out = D.train_on_batch(sX, sy, dropout_masks=T.masks_by_name(ref.D, d_masks, D.layers), capture=cap)
        out_ref = ref.d_train_on_batch(sX, sy, d_masks, T.decisions_for(ref.D, D.layers, cap))
        del cap
        T.assert_decisions_consistent(ref.D)
        assert abs(out[0] - out_ref[0]) <= 2e-5 * abs(out_ref[0]) and out[1] == pytest.approx(out_ref[1])
        dgr = [p.grad.cpu().numpy() for l in D.layers for p in l.params]
        for gq, gr in zip(dgr, ref.last_d_grads):
            assert T.rel(gq, gr) < 2e-4
        used = ops.prof_collect(2)['launches']
    finally:
        ops.set_conv_math()
        ops.prof_enable(False)
    assert used >= 1, used              # the stride-2 forward really ran on the split kernel


def test_random_shapes_under_the_opt_in_split_agree_with_the_exact_kernels():
    """Forty seeded random launches -- forward, data gradient, weight gradient; both strides, both paddings, row counts from 5 to 900, ragged everything --
    under the opt-in math against the exact-fp32 kernels on the same inputs: every result within 4e-6 of the tensor's largest entry (both are fp32-grade
    evaluations of the same sums), whichever of the split kernels (wide, merged, three-stage, narrow, weight gradient) or the exact fallback the shape selects."""
    from gennet_amd import ops
    rng = np.random.RandomState(2024)
    dev = torch.device('cuda:0')
    taken = 0
    for case in range(40):
        B = int(rng.randint(1, 5)); L = int(rng.choice([5, 17, 33, 64, 100, 191, 192, 257, 400, 513, 900]))
        Cin = int(rng.choice([256, 384, 512])); Cout = int(rng.choice([256, 320, 512]))
        stride = int(rng.choice([1, 2])); padding = str(rng.choice(['same', 'valid']))
        if padding == 'valid' and L < 5:
            continue
        Lout, pl = ops.conv_geometry(L, 5, stride, padding)
        x = torch.tensor((rng.randn(B, L, Cin) * np.exp(rng.randn(B, L, 1))).astype(np.float32)).to(dev)
        w = torch.tensor((rng.randn(5, Cin, Cout) / np.sqrt(5 * Cin)).astype(np.float32)).to(dev)
        b = torch.tensor(rng.randn(Cout).astype(np.float32)).to(dev)
        dy = torch.tensor((rng.randn(B, Lout, Cout) * np.exp(rng.randn(B, Lout, 1))).astype(np.float32)).to(dev)
        wt = ops.conv1d_transpose_w(w)

        def run():
            return (ops.conv1d_fwd(x, w, b, stride, pl, Lout, 'relu'), ops.conv1d_dgrad(dy, wt, L, stride, pl), ops.conv1d_wgrad(x, dy, 5, stride, pl)[0])
        exact = [t.cpu().numpy().astype(np.float64) for t in run()]
        ops.prof_enable(True); ops.prof_reset()
        ops.set_conv_math('bf16x3', workspace_gb=0.25)
        try:
            got = [t.cpu().numpy().astype(np.float64) for t in run()]
            taken += ops.prof_collect(2)['launches']
        finally:
            ops.set_conv_math()
            ops.prof_enable(False)
        for name, g_, e_ in zip(('fwd', 'dgrad', 'wgrad'), got, exact):
            scale = max(np.abs(e_).max(), 1e-30)
            assert np.abs(g_ - e_).max() <= 4e-6 * scale, (case, name, B, L, Cin, Cout, stride, padding, np.abs(g_ - e_).max() / scale)
    assert taken >= 60, taken                    # most of the 120 launches did go through the split kernels


@pytest.mark.parametrize('stride', [1, 2])
def test_split_kernels_are_deterministic_at_bench_like_sizes(stride):
    """A race in the staging choreography (counted vmcnt, one barrier per chunk, stages reused every other chunk) would show as run-to-run differences on
    launches that keep every CU busy for several rounds: forward, data gradient and weight gradient of a 512 -> 1024 layer on 48 x 2048 rows, five times
    each, bit for bit."""
    from gennet_amd import ops
    dev = torch.device('cuda:0')
    B, L, Cin, Cout = 48, 2048, 512, 1024
    Lout, pl = ops.conv_geometry(L, 5, stride, 'same')
    x = ops.fill_normal((B, L, Cin), 0.0, 1.0, 11, 0, dev)
    dy = ops.fill_normal((B, Lout, Cout), 0.0, 1.0, 12, 0, dev)
    w = ops.fill_normal((5, Cin, Cout), 0.0, 0.02, 13, 0, dev)
    b = ops.fill_normal((Cout,), 0.0, 1.0, 14, 0, dev)
    wt = ops.conv1d_transpose_w(w)
    ops.prof_enable(True); ops.prof_reset()
    ops.set_conv_math('bf16x3', workspace_gb=2.0)
    try:
        first = None
        for rep in range(5):
            out = (ops.conv1d_fwd(x, w, b, stride, pl, Lout, 'relu'), ops.conv1d_dgrad(dy, wt, L, stride, pl), ops.conv1d_wgrad(x, dy, 5, stride, pl)[0])
            if first is None:
                first = [t.clone() for t in out]
            else:
                for name, t, f in zip(('fwd', 'dgrad', 'wgrad'), out, first):
                    assert torch.equal(t, f), (rep, name, float((t - f).abs().max()))
        assert ops.prof_collect(2)['launches'] == 15
    finally:
        ops.set_conv_math()
        ops.prof_enable(False)
