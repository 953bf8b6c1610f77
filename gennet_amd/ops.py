"""Thin tensor-level wrappers over the C ABI: torch supplies device memory and the current HIP stream, nothing else.

Every function takes / returns contiguous fp32 CUDA tensors in Keras channels-last layout and launches on
torch's current stream.  No function here computes anything itself.
"""
import torch

from . import _lib

ACT = {'linear': 0, None: 0, 'relu': 1, 'relu_max': 2, 'leaky': 3, 'tanh': 4, 'sigmoid': 5}

_ws = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.GennetHipError('gennet_amd ops need CUDA (HIP) tensors; got a %s tensor -- there is no CPU path' % t.device)
        if not t.is_contiguous():
            raise ValueError('non-contiguous tensor passed to a gennet_amd op')


_ws_on_use = None       # engine.StepGraph.capture installs a callback here: a captured graph keeps the RAW ADDRESS of the scratch buffer it
                        # was handed, so it must also keep the buffer alive after a larger request has replaced it in _ws


def workspace(nbytes, device):
    """Stream-ordered scratch, grown on demand and reused (all ops run on one stream per process)."""
    key = (device.type, device.index)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws[key] = buf
    if _ws_on_use is not None:
        _ws_on_use(buf)
    return buf


def same_pad(L, k, s):
    """TF 'SAME': out = ceil(L/s); pad_total = max((out-1)*s + k - L, 0); left = total // 2."""
    out = -(-L // s)
    tot = max((out - 1) * s + k - L, 0)
    return out, tot // 2


def conv_geometry(L, k, stride, padding):
    if padding == 'same':
        return same_pad(L, k, stride)
    if padding == 'valid':
        return (L - k) // stride + 1, 0
    raise ValueError('padding %r' % (padding,))


# ---------------------------------------------------------------------------------------------- conv / dense
def conv1d_fwd(x, w, b, stride, pad_left, Lout, act='linear', act_param=0.0):
    _chk(x, w, b)
    B, L, Cin = x.shape
    k, _, Cout = w.shape
    y = torch.empty((B, Lout, Cout), dtype=torch.float32, device=x.device)
    _lib.call('gn_conv1d_fwd', _p(x), _p(w), _p(b), _p(y), B, L, Cin, Cout, k, stride, pad_left, Lout, ACT[act], float(act_param), _stream())
    return y


def conv1d_fwd_stats(x, w, b, stride, pad_left, Lout):
    """Linear conv forward + the BatchNorm statistics of its output: (y, sums fp64 (2*Cout,) = [sum y | sum y^2]); the sums come out of
    the conv kernel's epilogue where the pipelined kernel runs, from a separate pass otherwise (same values either way)."""
    _chk(x, w, b)
    B, L, Cin = x.shape
    k, _, Cout = w.shape
    y = torch.empty((B, Lout, Cout), dtype=torch.float32, device=x.device)
    sums = torch.empty((2 * Cout,), dtype=torch.float64, device=x.device)
    nb = _lib.size('gn_conv1d_fwd_stats_workspace', B, Lout, Cout)
    ws = workspace(nb, x.device)
    _lib.call('gn_conv1d_fwd_stats', _p(x), _p(w), _p(b), _p(y), _p(sums), _p(ws), ws.numel(), B, L, Cin, Cout, k, stride, pad_left, Lout, _stream())
    return y, sums


_BF16X3_WS = {}


def conv1d_fwd_bf16x3(x, w, b, stride, pad_left, Lout, act='linear', act_param=0.0, resplit=True):
    """EXPERIMENTAL opt-in: conv1d_fwd on the bf16 matrix cores with 3-way split fp32 operands (csrc/conv_bf16x3.hip)."""
    _chk(x, w, b)
    B, L, Cin = x.shape
    k, _, Cout = w.shape
    n = _lib.size('gn_conv1d_bf16x3_workspace', B, L, Cin, Cout, k)
    key = (x.device, n)
    ws = _BF16X3_WS.get(key)
    if ws is None:
        ws = _BF16X3_WS[key] = torch.empty(n, dtype=torch.uint8, device=x.device)
    y = torch.empty((B, Lout, Cout), dtype=torch.float32, device=x.device)
    _lib.call('gn_conv1d_fwd_bf16x3', _p(x), _p(w), _p(b), _p(y), _p(ws), n, B, L, Cin, Cout, k, stride, pad_left, Lout, ACT[act], float(act_param),
              1 if resplit else 0, _stream())
    return y


def conv1d_fwd_wino(x, w, b, pad_left, Lout, act='linear', act_param=0.0):
    """conv1d_fwd of a unit-stride 5-tap layer through the transform-domain kernel directly (csrc/conv_wino.hip; tests and layer benchmarks)."""
    _chk(x, w, b)
    B, L, Cin = x.shape
    k, _, Cout = w.shape
    n = _lib.size('gn_conv1d_wino_workspace', Cin, Cout)
    ws = workspace(n, x.device)
    y = torch.empty((B, Lout, Cout), dtype=torch.float32, device=x.device)
    _lib.call('gn_conv1d_fwd_wino', _p(x), _p(w), _p(b), _p(y), _p(ws), ws.numel(), B, L, Cin, Cout, k, 1, pad_left, Lout, ACT[act], float(act_param), _stream())
    return y


def conv1d_fwd_dropout(x, w, b, mask, stride, pad_left, Lout, act, act_param, rate):
    """conv + activation + inverted dropout in one epilogue; mask: uint8 keep-mask with the shape of the output."""
    _chk(x, w, b, mask)
    B, L, Cin = x.shape
    k, _, Cout = w.shape
    y = torch.empty((B, Lout, Cout), dtype=torch.float32, device=x.device)
    assert mask.numel() == y.numel()
    _lib.call('gn_conv1d_fwd_dropout', _p(x), _p(w), _p(b), _p(mask), _p(y), B, L, Cin, Cout, k, stride, pad_left, Lout, ACT[act], float(act_param),
              float(rate), _stream())
    return y


def act_dropout_bwd(dy, y, mask, act, act_param, rate, inplace=False):
    _chk(dy, y, mask)
    dx = dy if inplace else torch.empty_like(dy)
    _lib.call('gn_act_dropout_bwd', _p(dy), _p(y), _p(mask), _p(dx), dy.numel(), ACT[act], float(act_param), float(rate), _stream())
    return dx


def conv1d_transpose_w(w):
    _chk(w)
    k, Cin, Cout = w.shape
    wt = torch.empty((k, Cout, Cin), dtype=torch.float32, device=w.device)
    _lib.call('gn_conv1d_transpose_w', _p(w), _p(wt), k, Cin, Cout, _stream())
    return wt


def conv1d_dgrad(dy, wt, L, stride, pad_left, prev=None):
    """prev = (y_prev, act, act_param, mask_prev, rate): fuse the producer layer's activation/dropout backward into the epilogue."""
    _chk(dy, wt)
    B, Lout, Cout = dy.shape
    k, _, Cin = wt.shape
    dx = torch.empty((B, L, Cin), dtype=torch.float32, device=dy.device)
    if prev is None:
        _lib.call('gn_conv1d_dgrad', _p(dy), _p(wt), _p(dx), B, L, Cin, Cout, k, stride, pad_left, Lout, _stream())
    else:
        y_prev, act, param, mask, rate = prev
        _chk(y_prev, mask)
        assert y_prev.numel() == dx.numel()
        _lib.call('gn_conv1d_dgrad_fused', _p(dy), _p(wt), _p(dx), B, L, Cin, Cout, k, stride, pad_left, Lout, _p(y_prev), _p(mask), ACT[act], float(param),
                  float(rate), _stream())
    return dx


def can_fuse_dgrad(Cin, Cout):
    return Cin > 4 and Cout > 4


def conv1d_wgrad(x, dy, k, stride, pad_left, dw=None, db=None):
    _chk(x, dy, dw, db)
    B, L, Cin = x.shape
    _, Lout, Cout = dy.shape
    if dw is None:
        dw = torch.empty((k, Cin, Cout), dtype=torch.float32, device=x.device)
    if db is None:
        db = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    nb = _lib.size('gn_conv1d_wgrad_workspace', B, L, Cin, Cout, k, stride, Lout)
    ws = workspace(nb, x.device)
    _lib.call('gn_conv1d_wgrad', _p(x), _p(dy), _p(dw), _p(db), _p(ws), ws.numel(), B, L, Cin, Cout, k, stride, pad_left, Lout, _stream())
    return dw, db


def conv2d_w2_fold(w, b):
    _chk(w, b)
    kh, kw, Cin, Cout = w.shape
    assert kw == 5
    wf = torch.empty((kh, 2 * Cin, 2 * Cout), dtype=torch.float32, device=w.device)
    bf = torch.empty((2 * Cout,), dtype=torch.float32, device=w.device)
    _lib.call('gn_conv2d_w2_fold', _p(w), _p(b), _p(wf), _p(bf), kh, Cin, Cout, _stream())
    return wf, bf


def conv2d_w2_unfold_grad(dwf, dbf, Cin, Cout, dw=None, db=None):
    _chk(dwf, dbf, dw, db)
    kh = dwf.shape[0]
    if dw is None:
        dw = torch.empty((kh, 5, Cin, Cout), dtype=torch.float32, device=dwf.device)
    if db is None:
        db = torch.empty((Cout,), dtype=torch.float32, device=dwf.device)
    _lib.call('gn_conv2d_w2_unfold_grad', _p(dwf), _p(dbf), _p(dw), _p(db), kh, Cin, Cout, _stream())
    return dw, db


def maxpool_h2_fwd(x):
    """MaxPooling2D(pool_size=(2,1)) on (B, H, ...): gn_maxpool_h2_fwd."""
    _chk(x)
    B, H = x.shape[0], x.shape[1]
    R = x.numel() // max(B * H, 1)
    y = torch.empty((B, H // 2) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    _lib.call('gn_maxpool_h2_fwd', _p(x), _p(y), B, H, R, _stream())
    return y


def maxpool_h2_bwd(dy, x):
    _chk(dy, x)
    B, H = x.shape[0], x.shape[1]
    R = x.numel() // max(B * H, 1)
    dx = torch.empty_like(x)
    _lib.call('gn_maxpool_h2_bwd', _p(dy), _p(x), _p(dx), B, H, R, _stream())
    return dx


def tap_groups(k):
    """(G, h): a k-tap Conv1D (k > 5) runs as G = ceil(k/5) tap groups of h = ceil(k/G) taps (gn_conv1d_tap_groups, csrc/tap_fold.hip)."""
    G = (k + 4) // 5
    return G, (k + G - 1) // G


def conv1d_tapfold_x(x, k, pl):
    """More than 5 taps as h taps over G*Cin channels (csrc/tap_fold.hip): the input with its shifted copies beside it, left padding materialised."""
    _chk(x)
    B, L, Cin = x.shape
    x2 = torch.empty((B, L + pl, tap_groups(k)[0] * Cin), dtype=torch.float32, device=x.device)
    _lib.call('gn_conv1d_tapfold_x', _p(x), _p(x2), B, L, Cin, k, pl, _stream())
    return x2


def conv1d_tapunfold_dx(dx2, L, k, pl):
    _chk(dx2)
    B, L2, C2 = dx2.shape
    G = tap_groups(k)[0]
    assert L2 == L + pl and C2 % G == 0
    dx = torch.empty((B, L, C2 // G), dtype=torch.float32, device=dx2.device)
    _lib.call('gn_conv1d_tapunfold_dx', _p(dx2), _p(dx), B, L, C2 // G, k, pl, _stream())
    return dx


def conv1d_tapfold_w(w):
    _chk(w)
    k, Cin, Cout = w.shape
    G, h = tap_groups(k)
    w2 = torch.empty((h, G * Cin, Cout), dtype=torch.float32, device=w.device)
    _lib.call('gn_conv1d_tapfold_w', _p(w), _p(w2), k, Cin, Cout, _stream())
    return w2


def conv1d_tapunfold_dw(dw2, k, dw=None):
    _chk(dw2, dw)
    h, C2, Cout = dw2.shape
    G, hh = tap_groups(k)
    assert h == hh and C2 % G == 0
    if dw is None:
        dw = torch.empty((k, C2 // G, Cout), dtype=torch.float32, device=dw2.device)
    _lib.call('gn_conv1d_tapunfold_dw', _p(dw2), _p(dw), k, C2 // G, Cout, _stream())
    return dw


def conv1d_up2_fold(w, b, stride):
    """UpSampling1D(2) -> Conv1D(5, 'same', stride) as a 3-tap stride-1 conv on the un-upsampled input (gn_conv1d_up2_fold)."""
    _chk(w, b)
    k, Cin, Cout = w.shape
    assert k == 5 and stride in (1, 2)
    Cf = Cout * (2 if stride == 1 else 1)
    wf = torch.empty((3, Cin, Cf), dtype=torch.float32, device=w.device)
    bf = torch.empty((Cf,), dtype=torch.float32, device=w.device)
    _lib.call('gn_conv1d_up2_fold', _p(w), _p(b), _p(wf), _p(bf), Cin, Cout, stride, _stream())
    return wf, bf


def conv1d_up2_unfold_grad(dwf, dbf, Cout, stride, dw=None, db=None):
    _chk(dwf, dbf, dw, db)
    Cin = dwf.shape[1]
    assert dwf.shape[0] == 3 and dwf.shape[2] == Cout * (2 if stride == 1 else 1)
    if dw is None:
        dw = torch.empty((5, Cin, Cout), dtype=torch.float32, device=dwf.device)
    if db is None:
        db = torch.empty((Cout,), dtype=torch.float32, device=dwf.device)
    _lib.call('gn_conv1d_up2_unfold_grad', _p(dwf), _p(dbf), _p(dw), _p(db), Cin, Cout, stride, _stream())
    return dw, db


def dense_fwd(x, w, b, act='linear', act_param=0.0):
    _chk(x, w, b)
    B, n_in = x.shape
    n_out = w.shape[1]
    y = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
    _lib.call('gn_dense_fwd', _p(x), _p(w), _p(b), _p(y), B, n_in, n_out, ACT[act], float(act_param), _stream())
    return y


def dense_bwd(x, w, dy, need_dx=True, dw=None, db=None, prev=None):
    _chk(x, w, dy, dw, db)
    B, n_in = x.shape
    n_out = w.shape[1]
    dx = torch.empty_like(x) if need_dx else None
    if prev is not None:
        _, act, param, mask, rate = prev
        assert need_dx and n_out <= 4
        if dw is None:
            dw = torch.empty_like(w)
        if db is None:
            db = torch.empty((n_out,), dtype=torch.float32, device=x.device)
        _lib.call('gn_dense_bwd_fused', _p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), B, n_in, n_out, _p(mask), ACT[act], float(param), float(rate), _stream())
        return dx, dw, db
    if dw is None:
        dw = torch.empty_like(w)
    if db is None:
        db = torch.empty((n_out,), dtype=torch.float32, device=x.device)
    nb = _lib.size('gn_dense_bwd_workspace', B, n_in, n_out)
    ws = workspace(nb, x.device)
    _lib.call('gn_dense_bwd', _p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), _p(ws), ws.numel(), B, n_in, n_out, _stream())
    return dx, dw, db


# ---------------------------------------------------------------------------------------------- elementwise
def act_fwd(x, act, act_param=0.0):
    _chk(x)
    y = torch.empty_like(x)
    _lib.call('gn_act_fwd', _p(x), _p(y), x.numel(), ACT[act], float(act_param), _stream())
    return y


def act_bwd(dy, y, act, act_param=0.0, inplace=False):
    _chk(dy, y)
    dx = dy if inplace else torch.empty_like(dy)
    _lib.call('gn_act_bwd', _p(dy), _p(y), _p(dx), dy.numel(), ACT[act], float(act_param), _stream())
    return dx


def dropout_mask(shape, rate, seed, offset, device):
    m = torch.empty(shape, dtype=torch.uint8, device=device)
    _lib.call('gn_dropout_mask', _p(m), m.numel(), float(rate), int(seed), int(offset), _stream())
    return m


def conv_fold_bn(w, b, scale, shift):
    """(k, Cin, Cout) kernel and bias with an inference-phase BatchNormalization folded in: conv(x; w', b') = BN_infer(conv(x; w, b))."""
    _chk(w, b, scale, shift)
    w2 = torch.empty_like(w)
    b2 = torch.empty_like(scale)
    _lib.call('gn_conv_fold_bn', _p(w), _p(b), _p(scale), _p(shift), _p(w2), _p(b2), w.numel() // w.shape[-1], w.shape[-1], _stream())
    return w2, b2


_CONV_MATH_WS = [None]


def default_conv_math():
    """What the engine sets at start-up: GENNET_CONV_MATH, 'wino' when unset."""
    import os
    return os.environ.get('GENNET_CONV_MATH', 'wino')


WINO_WS_BYTES = 64 << 20       # transformed kernel of one launch: 6 * Cin * Cout * 4 bytes (25 MB for the generator's 512 -> 1024 layer)


def set_conv_math(mode=None, workspace_gb=7.0, device=None):
    """How the MFMA convolutions compute (process-wide, csrc/capi.hip conv_dispatch):
    'wino' (the engine's default): the unit-stride 5-tap Conv1D forward / data-gradient launches with Cin >= 32, Cin % 8 == 0 and Cout % 64 == 0 run in the
        transform domain (Cook-Toom F(2,5), csrc/conv_wino.hip: 6 fp32 multiplies per two outputs instead of 10, every product an exact fp32 fma on the fp32
        matrix instruction); everything else on the direct kernels.  Chosen by the layer's shape alone (never the batch size).
    'fp32': the direct exact-fp32 MFMA kernels everywhere (results bit-identical to a k-ordered fmaf chain).
    'bf16x3': opt-in experiment -- the launches with at least 256 channels on either side run on the bf16 matrix cores with 3-way split operands
        (csrc/conv_bf16x3.hip, csrc/wgrad_bf16x3.hip); workspace_gb must hold the split operands of the largest such launch."""
    dev = device or torch.device('cuda', torch.cuda.current_device())
    if mode is None:
        mode = default_conv_math()
    if mode == 'fp32':
        _lib.call('gn_set_conv_math', 0, None, 0)
        _CONV_MATH_WS[0] = None
        return
    if mode == 'wino':
        ws = torch.empty(WINO_WS_BYTES, dtype=torch.uint8, device=dev)
        _lib.call('gn_set_conv_math', 2, _p(ws), ws.numel())
        _CONV_MATH_WS[0] = ws
        return
    if mode != 'bf16x3':
        raise ValueError('conv math %r (wino | fp32 | bf16x3)' % (mode,))
    ws = torch.empty(int(workspace_gb * (1 << 30)), dtype=torch.uint8, device=dev)
    _lib.call('gn_set_conv_math', 1, _p(ws), ws.numel())
    _CONV_MATH_WS[0] = ws


import contextlib


@contextlib.contextmanager
def conv_math(mode):
    """`with ops.conv_math('fp32'):` -- the given conv arithmetic for the duration of the block, the process default (default_conv_math) afterwards."""
    set_conv_math(mode)
    try:
        yield
    finally:
        set_conv_math()


def prelu_fwd(x, alpha):
    """x (B, ...), alpha with the shape of one sample."""
    _chk(x, alpha)
    y = torch.empty_like(x)
    _lib.call('gn_prelu_fwd', _p(x), _p(alpha), _p(y), x.shape[0], alpha.numel(), _stream())
    return y


def prelu_bwd(dy, x, alpha, need_dx=True, dalpha=None):
    """-> (dx or None, dalpha); dalpha is written into the given tensor when there is one (the flat gradient buffer)."""
    _chk(dy, x, alpha)
    dx = torch.empty_like(x) if need_dx else None
    if dalpha is None:
        dalpha = torch.empty_like(alpha)
    _lib.call('gn_prelu_bwd', _p(dy), _p(x), _p(alpha), _p(dx), _p(dalpha), x.shape[0], alpha.numel(), _stream())
    return dx, dalpha


def dropout_apply(x, mask, rate):
    _chk(x, mask)
    y = torch.empty_like(x)
    _lib.call('gn_dropout_apply', _p(x), _p(mask), _p(y), x.numel(), float(rate), _stream())
    return y


def upsample2_fwd(x):
    _chk(x)
    B, L, Cc = x.shape
    y = torch.empty((B, 2 * L, Cc), dtype=torch.float32, device=x.device)
    _lib.call('gn_upsample2_fwd', _p(x), _p(y), B, L, Cc, _stream())
    return y


def upsample2_bwd(dy):
    _chk(dy)
    B, L2, Cc = dy.shape
    dx = torch.empty((B, L2 // 2, Cc), dtype=torch.float32, device=dy.device)
    _lib.call('gn_upsample2_bwd', _p(dy), _p(dx), B, L2 // 2, Cc, _stream())
    return dx


def subtract_stack_fwd(x, event):
    _chk(x, event)
    B, n = x.shape[0], x.shape[1]
    img = torch.empty((B, n, 2, 1), dtype=torch.float32, device=x.device)
    _lib.call('gn_subtract_stack_fwd', _p(x), _p(event), _p(img), B, n, _stream())
    return img


def subtract_stack_bwd(dimg):
    _chk(dimg)
    B, n = dimg.shape[0], dimg.shape[1]
    dx = torch.empty((B, n, 1), dtype=torch.float32, device=dimg.device)
    _lib.call('gn_subtract_stack_bwd', _p(dimg), _p(dx), B, n, _stream())
    return dx


def affine_stack_fwd(x, a0, b0, a1, b1):
    """x (B, n, 1) -> img (B, n, 2, 1): [a0*x + b0 | a1*x + b1], b0 / b1 (n,) device vectors or None."""
    _chk(x)
    B, n = x.shape[0], x.shape[1]
    img = torch.empty((B, n, 2, 1), dtype=torch.float32, device=x.device)
    _lib.call('gn_affine_stack_fwd', _p(x), _p(b0), _p(b1), float(a0), float(a1), _p(img), B, n, _stream())
    return img


def affine_stack_bwd(dimg, a0, a1):
    _chk(dimg)
    B, n = dimg.shape[0], dimg.shape[1]
    dx = torch.empty((B, n, 1), dtype=torch.float32, device=dimg.device)
    _lib.call('gn_affine_stack_bwd', _p(dimg), float(a0), float(a1), _p(dx), B, n, _stream())
    return dx


def assemble_d_batch(real, noise, fake, event):
    """[real | fake] discriminator batch (2B, n, 2, 1); fake half reversed like the reference's prepend loop."""
    _chk(real, noise, fake, event)
    B, n = real.shape[0], real.shape[1]
    sX = torch.empty((2 * B, n, 2, 1), dtype=torch.float32, device=real.device)
    _lib.call('gn_assemble_d_batch', _p(real), _p(noise), _p(fake), _p(event), _p(sX), B, n, _stream())
    return sX


def fill_uniform(shape, lo, hi, seed, offset, device):
    t = torch.empty(shape, dtype=torch.float32, device=device)
    _lib.call('gn_fill_uniform', _p(t), t.numel(), float(lo), float(hi), int(seed), int(offset), _stream())
    return t


def fill_normal(shape, mean, std, seed, offset, device):
    t = torch.empty(shape, dtype=torch.float32, device=device)
    if isinstance(std, DevScalar):
        _lib.call('gn_fill_normal_dyn', _p(t), t.numel(), float(mean), std.ptr, int(seed), int(offset), _stream())
    else:
        _lib.call('gn_fill_normal', _p(t), t.numel(), float(mean), float(std), int(seed), int(offset), _stream())
    return t


def gather_rows(src, idx):
    _chk(src, idx)
    rows, width = idx.numel(), src.shape[1]
    out = torch.empty((rows, width), dtype=torch.float32, device=src.device)
    _lib.call('gn_gather_rows', _p(src), _p(idx), _p(out), rows, width, _stream())
    return out


def axpy(y, x, a):
    _chk(y, x)
    _lib.call('gn_axpy', _p(y), _p(x), float(a), y.numel(), _stream())
    return y


# ---------------------------------------------------------------------------------------------- batch norm
def bn_stats(x2d):
    """x2d: (rows, C) view.  Returns fp64 sums tensor (2*C,): [sum x | sum x^2]."""
    _chk(x2d)
    rows, Cc = x2d.shape
    sums = torch.empty((2 * Cc,), dtype=torch.float64, device=x2d.device)
    nb = _lib.size('gn_bn_stats_workspace', rows, Cc)
    ws = workspace(nb, x2d.device)
    _lib.call('gn_bn_stats', _p(x2d), rows, Cc, _p(sums), _p(ws), ws.numel(), _stream())
    return sums


def bn_finalize(sums, count, gamma, beta, eps, momentum, moving_mean, moving_var, zero_debias=None):
    """zero_debias = (biased_mean, biased_var, local_step): TF's assign_moving_average(zero_debias=True) with local_step the already
    incremented update count; None: the plain exponential average."""
    Cc = gamma.numel()
    dev = gamma.device
    scale, shift, smean, sinv = (torch.empty((Cc,), dtype=torch.float32, device=dev) for _ in range(4))
    if zero_debias is not None:
        bm, bv, step = zero_debias
        if isinstance(step, DevScalar):
            _lib.call('gn_bn_finalize_zero_debias_dyn', _p(sums), float(count), _p(gamma), _p(beta), float(eps), float(momentum), _p(moving_mean), _p(moving_var),
                      _p(bm), _p(bv), step.ptr, _p(scale), _p(shift), _p(smean), _p(sinv), Cc, _stream())
        else:
            _lib.call('gn_bn_finalize_zero_debias', _p(sums), float(count), _p(gamma), _p(beta), float(eps), float(momentum), _p(moving_mean), _p(moving_var),
                      _p(bm), _p(bv), int(step), _p(scale), _p(shift), _p(smean), _p(sinv), Cc, _stream())
    else:
        _lib.call('gn_bn_finalize', _p(sums), float(count), _p(gamma), _p(beta), float(eps), float(momentum), _p(moving_mean), _p(moving_var),
                  _p(scale), _p(shift), _p(smean), _p(sinv), Cc, _stream())
    return scale, shift, smean, sinv


def bn_infer_coeffs(gamma, beta, moving_mean, moving_var, eps):
    Cc = gamma.numel()
    scale, shift = (torch.empty((Cc,), dtype=torch.float32, device=gamma.device) for _ in range(2))
    _lib.call('gn_bn_infer_coeffs', _p(gamma), _p(beta), _p(moving_mean), _p(moving_var), float(eps), _p(scale), _p(shift), Cc, _stream())
    return scale, shift


def bn_apply(x2d, scale, shift, mask=None, act='linear', act_param=0.0, rate=0.0):
    _chk(x2d, mask)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    _lib.call('gn_bn_apply', _p(x2d), _p(scale), _p(shift), _p(mask), _p(y), rows, Cc, ACT[act], float(act_param), float(rate), _stream())
    return y


def bn_apply_dropgen(x2d, scale, shift, act, act_param, rate, seed, offset):
    """bn_apply + activation + dropout with the keep-mask drawn in the same pass (dropout_mask's stream); -> (y, mask)."""
    _chk(x2d)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    mask = torch.empty((rows, Cc), dtype=torch.uint8, device=x2d.device)
    _lib.call('gn_bn_apply_dropgen', _p(x2d), _p(scale), _p(shift), _p(mask), _p(y), rows, Cc, ACT[act], float(act_param), float(rate), int(seed), int(offset),
              _stream())
    return y, mask


def bn_bwd_stats(dy2d, y2d, x2d, mask, smean, sinv, act='linear', act_param=0.0, rate=0.0, scale=None, shift=None):
    """scale / shift (the forward's bn_finalize outputs): the activation output is recomputed from x2d and y2d (may be None) is not read."""
    _chk(dy2d, y2d, x2d, mask)
    rows, Cc = x2d.shape
    dsums = torch.empty((2 * Cc,), dtype=torch.float64, device=x2d.device)
    nb = _lib.size('gn_bn_stats_workspace', rows, Cc)
    ws = workspace(nb, x2d.device)
    _lib.call('gn_bn_bwd_stats', _p(dy2d), _p(y2d), _p(x2d), _p(mask), _p(smean), _p(sinv), _p(dsums), _p(ws), ws.numel(), rows, Cc,
              ACT[act], float(act_param), float(rate), _p(scale), _p(shift), _stream())
    return dsums


def bn_bwd_apply(dy2d, y2d, x2d, mask, gamma, smean, sinv, dsums_global, count, dsums_local, dgamma, dbeta, act='linear', act_param=0.0, rate=0.0,
                 scale=None, shift=None):
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    _lib.call('gn_bn_bwd_apply', _p(dy2d), _p(y2d), _p(x2d), _p(mask), _p(gamma), _p(smean), _p(sinv), _p(dsums_global), float(count),
              _p(dsums_local), _p(dx), _p(dgamma), _p(dbeta), rows, Cc, ACT[act], float(act_param), float(rate), _p(scale), _p(shift), _stream())
    return dx


class ConvGrad1(object):
    """The data gradient of a Conv1D(1 filter, k taps, stride 1), NOT materialised: g (B, Lout, 1) the conv's output gradient, w (k, C, 1)
    its kernel, L the conv's input length.  Consumed by bn_bwd_stats_conv1 / bn_bwd_apply_conv1 (gn_bn_bwd_*_conv1)."""

    def __init__(self, g, w, L, pad_left):
        _chk(g, w)
        self.g, self.w, self.L, self.pad_left = g, w, int(L), int(pad_left)
        self.B, self.Lout = int(g.shape[0]), int(g.shape[1])
        self.k, self.C = int(w.shape[0]), int(w.shape[1])
        self.shape = (self.B, self.L, self.C)


def bn_bwd_stats_conv1(cg, x2d, mask, smean, sinv, act, act_param, rate, scale, shift):
    _chk(x2d, mask)
    rows, Cc = x2d.shape
    assert rows == cg.B * cg.L and Cc == cg.C
    dsums = torch.empty((2 * Cc,), dtype=torch.float64, device=x2d.device)
    ws = workspace(_lib.size('gn_bn_stats_workspace', rows, Cc), x2d.device)
    _lib.call('gn_bn_bwd_stats_conv1', _p(cg.g), _p(cg.w), cg.L, cg.Lout, cg.k, cg.pad_left, _p(x2d), _p(mask), _p(smean), _p(sinv), _p(dsums), _p(ws),
              ws.numel(), rows, Cc, ACT[act], float(act_param), float(rate), _p(scale), _p(shift), _stream())
    return dsums


def bn_bwd_apply_conv1(cg, x2d, mask, gamma, smean, sinv, dsums_global, count, dsums_local, dgamma, dbeta, act, act_param, rate, scale, shift):
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    _lib.call('gn_bn_bwd_apply_conv1', _p(cg.g), _p(cg.w), cg.L, cg.Lout, cg.k, cg.pad_left, _p(x2d), _p(mask), _p(gamma), _p(smean), _p(sinv),
              _p(dsums_global), float(count), _p(dsums_local), _p(dx), _p(dgamma), _p(dbeta), rows, Cc, ACT[act], float(act_param), float(rate),
              _p(scale), _p(shift), _stream())
    return dx


# ---------------------------------------------------------------------------------------------- loss / optimizer
def loss(kind, p, y, Bglobal=None):
    """kind 'binary_crossentropy' | 'mean_squared_error'. p, y (B,1). Returns (dp, out[2] = [loss share, hit count])."""
    _chk(p, y)
    B = p.shape[0]
    dp = torch.empty_like(p)
    out = torch.empty((2,), dtype=torch.float32, device=p.device)
    fn = {'binary_crossentropy': 'gn_bce_loss', 'mean_squared_error': 'gn_mse_loss'}[kind]
    _lib.call(fn, _p(p), _p(y), _p(dp), _p(out), B, int(Bglobal or B), _stream())
    return dp, out


def adam_step(p, g, m, v, lr_t, b1, b2, eps):
    """lr_t: a python float, or (inside a captured step graph) the integer device address of a float the host refreshes before every replay."""
    _chk(p, g, m, v)
    if isinstance(lr_t, DevScalar):
        _lib.call('gn_adam_step_dyn', _p(p), _p(g), _p(m), _p(v), p.numel(), lr_t.ptr, float(b1), float(b2), float(eps), _stream())
    else:
        _lib.call('gn_adam_step', _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr_t), float(b1), float(b2), float(eps), _stream())


class DevScalar(object):
    """Address of one step-varying scalar in a step graph's parameter block (engine.StepGraph.slot)."""

    def __init__(self, ptr):
        self.ptr = int(ptr)


def set_rng_base(ptr):
    _lib.call('gn_set_rng_base', None if ptr is None else int(ptr))


# ---------------------------------------------------------------------------------------------- profiling hooks
_PROF_ON = False


def prof_enable(on=True):
    global _PROF_ON
    _PROF_ON = bool(on)
    _lib.call('gn_prof_enable', 1 if on else 0)


def prof_enabled():
    return _PROF_ON


def prof_reset():
    _lib.call('gn_prof_reset')


def prof_collect(kind=-1):
    """kind 0: conv_mfma kernels (forward + data gradient), 1: wgrad_mfma_kernel, 2: bf16x3 conv (opt-in), 3: fused synthesiser, -1: all."""
    import ctypes
    out = (ctypes.c_double * 4)()
    _lib.call('gn_prof_collect', int(kind), ctypes.cast(out, ctypes.c_void_p))
    return {'launches': int(out[0]), 'ms': out[1], 'flop': out[2], 'bytes': out[3]}
