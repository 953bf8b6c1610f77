"""Edge cases through the C ABI on the GPU: empty batches and zero-length element ranges return empty results instead of
faulting, predict() handles a ragged last batch and inputs shorter than the batch, the smallest shapes every kernel family
accepts, and the largest single allocation the BASELINE configs touch (index arithmetic beyond 2^31 elements)."""
import numpy as np
import pytest
import torch

from oracle import keras_ref as K

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def z(*shape, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=dev())


def test_empty_batch_at_the_c_abi():
    """B = 0: the forward entry points return success without launching; the gradient entry points refuse (an empty sum has no
    caller on the path) -- with an error code, never a fault."""
    from gennet_amd import _lib, ops
    d = z(1)                                              # any valid device pointer: nothing may be read or written through it
    p = d.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    _lib.call('gn_conv1d_fwd', p, p, p, p, 0, 64, 16, 128, 5, 1, 2, 64, 1, 0.0, s)
    _lib.call('gn_conv1d_dgrad', p, p, p, 0, 64, 16, 128, 5, 1, 2, 64, s)
    _lib.call('gn_dense_fwd', p, p, p, p, 0, 100, 256, 0, 0.0, s)
    with pytest.raises(_lib.GennetHipError):
        _lib.call('gn_conv1d_wgrad', p, p, p, p, p, 1 << 20, 0, 64, 16, 128, 5, 1, 2, 64, s)
    with pytest.raises(_lib.GennetHipError):
        _lib.call('gn_dense_bwd', p, p, p, p, p, p, p, 1 << 20, 0, 100, 256, s)
    torch.cuda.synchronize()
    assert float(d[0]) == 0.0


def test_predict_ragged_and_short_batches():
    from gennet_amd import bbh
    rng = np.random.RandomState(2)
    pe = bbh.signal_pe_model(128)
    x = rng.randn(70, 128, 1).astype(np.float32)
    full = pe.predict(x, batch_size=70)
    for bs in (32, 64, 7, 100):                                                    # 70 = 2*32+6 = 64+6 = 10*7; 100 > N
        got = pe.predict(x, batch_size=bs)
        for a, b in zip(full, got):
            assert a.shape == (70, 1) and np.array_equal(a, b)                     # rows are independent: batching cannot change a bit
    one = pe.predict(x[:1])
    assert np.array_equal(one[0], full[0][:1]) and np.array_equal(one[1], full[1][:1])
    none = pe.predict(x[:0])
    assert none[0].shape == (0, 1) and none[1].shape == (0, 1)


def test_smallest_shapes_of_each_conv_family():
    from gennet_amd import ops
    rng = np.random.RandomState(3)
    for (B, L, Cin, Cout, k, s, padding) in ((1, 5, 16, 8, 5, 1, 'valid'),         # one output row, MFMA path, Cout below one tile
                                             (1, 1, 1, 4, 1, 1, 'valid'),          # small-Cin, a single sample
                                             (1, 5, 8, 1, 5, 1, 'same'),           # small-Cout
                                             (1, 2, 16, 16, 5, 2, 'same')):        # stride 2 on two rows: every tap but one is padding
        x = rng.randn(B, L, Cin).astype(np.float32); w = rng.randn(k, Cin, Cout).astype(np.float32); b = rng.randn(Cout).astype(np.float32)
        Lout, pl = ops.conv_geometry(L, k, s, padding)
        ref = K.conv1d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s, padding)
        y = ops.conv1d_fwd(torch.tensor(x).to(dev()), torch.tensor(w).to(dev()), torch.tensor(b).to(dev()), s, pl, Lout).cpu().numpy()
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)


def test_small_cin_conv_strides_beyond_two():
    """The small-Cin kernel stages its input window in a fixed LDS array: strides > 2 get fewer rows per block instead of writing
    past it (Conv1D(16,5,strides=3) on 4 channels and Conv1D(64,5,strides=9) on 1 channel: the shapes ADVICE round 1 named)."""
    from gennet_amd import ops
    rng = np.random.RandomState(4)
    for (B, L, Cin, Cout, k, s, padding) in ((2, 4001, 4, 16, 5, 3, 'same'), (1, 5000, 1, 64, 5, 9, 'valid'), (2, 3000, 3, 8, 5, 5, 'same')):
        x = rng.randn(B, L, Cin).astype(np.float32); w = rng.randn(k, Cin, Cout).astype(np.float32); b = rng.randn(Cout).astype(np.float32)
        Lout, pl = ops.conv_geometry(L, k, s, padding)
        ref = K.conv1d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s, padding)
        y = ops.conv1d_fwd(torch.tensor(x).to(dev()), torch.tensor(w).to(dev()), torch.tensor(b).to(dev()), s, pl, Lout).cpu().numpy()
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)


def test_indexing_beyond_2_to_31_elements():
    """Generator conv5 output at batch 512 holds 512 * 2048 * 1024 = 2^30 elements and its input-gradient slabs more; the first
    dense layer's activations at B = 4096 (config 4 on one rank) 2^30.  Run the streaming kernels on a 2^31 + 2^20 element
    tensor and check both ends (size_t indexing, no 32-bit wrap)."""
    from gennet_amd import ops
    n = (1 << 31) + (1 << 20)
    x = torch.empty(n, dtype=torch.float32, device=dev())
    x[:1024] = -1.0; x[-1024:] = 2.0
    y = ops.act_fwd(x.reshape(1, -1), 'relu').reshape(-1)
    assert float(y[:1024].abs().max()) == 0.0 and float(y[-1024:].min()) == 2.0
    del y
    m = ops.dropout_mask((n,), 0.5, 7, 0, dev())
    frac = float(m[-(1 << 22):].float().mean())
    assert 0.49 < frac < 0.51                                                      # the tail is really drawn, not left untouched
    del m, x
    torch.cuda.empty_cache()
