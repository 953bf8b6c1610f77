"""CPU oracle: numpy restatement of the Keras-2.2.4 / TF-1.12 arithmetic used by the BBH hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gennet_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, as the checker.

PARITY UNPINNED against real Keras/TF: keras==2.2.4 / tensorflow-gpu==1.12.0
(/root/reference/requirements.txt:16,47) are third-party, absent from /root/reference and from this
image, and the reference ships no tests, golden vectors or weight files for the BBH nets.  What pins this
file instead: (i) every backward pass is cross-checked against torch-CPU autograd in fp64
(tests/test_oracle_nets.py), (ii) the call sites it follows are cited per function.

Layout follows Keras ``channels_last``: activations (B, L, C); Conv1D kernels (k, Cin, Cout); Conv2D kernels
(kh, kw, Cin, Cout); Dense kernels (in, out).  All functions are dtype-preserving; tests run them in
float64.
"""
import numpy as np

BN_EPS = 1e-3          # keras BatchNormalization default epsilon
K_EPS = 1e-7           # keras.backend.epsilon()
# TF evaluates the BCE clip bounds in the tensor dtype (float32): clip_by_value(p, eps32, 1 - eps32) with
# 1 - eps32 == 0.99999988 (one fp32 ulp below 1), not 0.9999999.  Restated with those fp32 constants.
CLIP_LO = float(np.float32(K_EPS))
CLIP_HI = float(np.float32(1.0) - np.float32(K_EPS))


# ----------------------------------------------------------------------------------------------
# padding  (TF SAME/VALID)
# ----------------------------------------------------------------------------------------------
def same_pad(L, k, s):
    out = -(-L // s)
    tot = max((out - 1) * s + k - L, 0)
    return out, tot // 2, tot - tot // 2


def conv_out_len(L, k, s, padding):
    if padding == 'same':
        return same_pad(L, k, s)[0]
    return (L - k) // s + 1


# ----------------------------------------------------------------------------------------------
# Conv1D  (bbhMahoGANy.py:250-292, :362-394)
# ----------------------------------------------------------------------------------------------
def _pad1d(x, k, s, padding):
    B, L, C = x.shape
    if padding == 'same':
        out, pl, pr = same_pad(L, k, s)
        xp = np.zeros((B, L + pl + pr, C), x.dtype)
        xp[:, pl:pl + L] = x
    else:
        out, pl, pr = (L - k) // s + 1, 0, 0
        xp = x
    return xp, out, pl


def conv1d_fwd(x, W, b, stride=1, padding='valid'):
    """y[b,t,co] = bias[co] + sum_{k,ci} x[b, s*t + k - pl, ci] * W[k,ci,co]  (cross-correlation)."""
    k, Cin, Cout = W.shape
    xp, out, pl = _pad1d(x, k, stride, padding)
    y = np.zeros((x.shape[0], out, Cout), x.dtype)
    for j in range(k):
        xs = xp[:, j:j + stride * (out - 1) + 1:stride]           # (B, out, Cin)
        y += xs @ W[j]
    if b is not None:
        y += b
    return y


def conv1d_bwd(x, W, dy, stride=1, padding='valid'):
    """Returns (dx, dW, db) of conv1d_fwd."""
    k, Cin, Cout = W.shape
    xp, out, pl = _pad1d(x, k, stride, padding)
    dxp = np.zeros_like(xp)
    dW = np.zeros_like(W)
    for j in range(k):
        sl = slice(j, j + stride * (out - 1) + 1, stride)
        xs = xp[:, sl]
        dW[j] = xs.reshape(-1, Cin).T @ dy.reshape(-1, Cout)          # sum over (b, t): one dgemm
        dxp[:, sl] += dy @ W[j].T
    dx = dxp[:, pl:pl + x.shape[1]]
    db = dy.sum(axis=(0, 1))
    return dx, dW, db


# ----------------------------------------------------------------------------------------------
# Conv2D  (bbhMahoGANy.py:439, :447) -- direct definition, used to pin the width-2 fold
# ----------------------------------------------------------------------------------------------
def conv2d_fwd(x, W, b, strides=(1, 1), padding='same'):
    B, H, Wd, C = x.shape
    kh, kw, Cin, Cout = W.shape
    sh, sw = strides
    if padding == 'same':
        oh, pt, pb = same_pad(H, kh, sh)
        ow, pl, pr = same_pad(Wd, kw, sw)
    else:
        oh, pt, pb = (H - kh) // sh + 1, 0, 0
        ow, pl, pr = (Wd - kw) // sw + 1, 0, 0
    xp = np.zeros((B, H + pt + pb, Wd + pl + pr, C), x.dtype)
    xp[:, pt:pt + H, pl:pl + Wd] = x
    y = np.zeros((B, oh, ow, Cout), x.dtype)
    for i in range(kh):
        for j in range(kw):
            xs = xp[:, i:i + sh * (oh - 1) + 1:sh, j:j + sw * (ow - 1) + 1:sw]
            y += xs @ W[i, j]
    if b is not None:
        y += b
    return y


def conv2d_bwd(x, W, dy, strides=(1, 1), padding='same'):
    B, H, Wd, C = x.shape
    kh, kw, Cin, Cout = W.shape
    sh, sw = strides
    if padding == 'same':
        oh, pt, pb = same_pad(H, kh, sh)
        ow, pl, pr = same_pad(Wd, kw, sw)
    else:
        oh, pt, pb = (H - kh) // sh + 1, 0, 0
        ow, pl, pr = (Wd - kw) // sw + 1, 0, 0
    xp = np.zeros((B, H + pt + pb, Wd + pl + pr, C), x.dtype)
    xp[:, pt:pt + H, pl:pl + Wd] = x
    dxp = np.zeros_like(xp)
    dW = np.zeros_like(W)
    for i in range(kh):
        for j in range(kw):
            s0 = slice(i, i + sh * (oh - 1) + 1, sh)
            s1 = slice(j, j + sw * (ow - 1) + 1, sw)
            dW[i, j] = xp[:, s0, s1].reshape(-1, Cin).T @ dy.reshape(-1, Cout)
            dxp[:, s0, s1] += dy @ W[i, j].T
    return dxp[:, pt:pt + H, pl:pl + Wd], dW, dy.sum(axis=(0, 1, 2))


def fold_conv2d_w2(W):
    """Width-2 image, 'same' padding, width-stride 1: Conv2D(kh x 5) == Conv1D over H with
    Cin' = 2*Cin (index w*Cin + c), Cout' = 2*Cout (index w'*Cout + c') and kernel
    W'[kh, w*Cin+c, w'*Cout+c'] = W[kh, kw = w - w' + 2, c, c'].  """
    kh, kw, Cin, Cout = W.shape
    assert kw == 5
    Wf = np.zeros((kh, 2 * Cin, 2 * Cout), W.dtype)
    for w in range(2):
        for wo in range(2):
            Wf[:, w * Cin:(w + 1) * Cin, wo * Cout:(wo + 1) * Cout] = W[:, w - wo + 2]
    return Wf


def unfold_conv2d_w2_grad(dWf, Cin, Cout):
    """Adjoint of fold_conv2d_w2: gradient w.r.t. the (kh,5,Cin,Cout) kernel (dead taps kw=0,4 get 0)."""
    kh = dWf.shape[0]
    dW = np.zeros((kh, 5, Cin, Cout), dWf.dtype)
    for w in range(2):
        for wo in range(2):
            dW[:, w - wo + 2] += dWf[:, w * Cin:(w + 1) * Cin, wo * Cout:(wo + 1) * Cout]
    return dW


# ----------------------------------------------------------------------------------------------
# Dense / UpSampling1D / MyLayer
# ----------------------------------------------------------------------------------------------
def dense_fwd(x, W, b):
    return x @ W + (0 if b is None else b)


def dense_bwd(x, W, dy):
    return dy @ W.T, x.T @ dy, dy.sum(axis=0)


def upsample1d_fwd(x, size=2):
    return np.repeat(x, size, axis=1)


def upsample1d_bwd(dy, size=2):
    B, L, C = dy.shape
    return dy.reshape(B, L // size, size, C).sum(axis=2)


def maxpool_h2_fwd(x):
    """MaxPooling2D(pool_size=(2,1)) (bbhMahoGANy.py:444-490, `maxpool = True`): max over row pairs along axis 1, 'valid' (an odd last row is dropped).
    -> (y, take_second): the routing mask the backward pass uses; a tie goes to the first row (TensorFlow's kernels keep the first maximum)."""
    Ho = x.shape[1] // 2
    a, c = x[:, 0:2 * Ho:2], x[:, 1:2 * Ho:2]
    second = c > a
    return np.where(second, c, a), second


def maxpool_h2_bwd(dy, second, H):
    dx = np.zeros((dy.shape[0], H) + dy.shape[2:], dy.dtype)
    Ho = dy.shape[1]
    dx[:, 0:2 * Ho:2] = np.where(second, 0.0, dy)
    dx[:, 1:2 * Ho:2] = np.where(second, dy, 0.0)
    return dx


def mylayer_fwd(x, const):
    """bbhMahoGANy.py:180-184: stack([x, const - x], axis=2): (B,n,1) -> (B,n,2,1)."""
    return np.stack([x, const - x], axis=2)


def mylayer_bwd(dy):
    return dy[:, :, 0] - dy[:, :, 1]


# ----------------------------------------------------------------------------------------------
# BatchNormalization(momentum) -- SURVEY Appendix B.4 (keras 2.2.4 normalization.py, unpinned)
# ----------------------------------------------------------------------------------------------
def bn_train_fwd(x, gamma, beta, eps=BN_EPS):
    """Normalise over all axes but the last with batch mean / biased variance.
    Returns y, (xhat, inv_std), mean, var (biased)."""
    axes = tuple(range(x.ndim - 1))
    mean = x.mean(axis=axes)
    var = x.var(axis=axes)
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * inv
    return gamma * xhat + beta, (xhat, inv), mean, var


def bn_moving_update(moving_mean, moving_var, mean, var, n, momentum, eps=BN_EPS):
    """Plain exponential moving average, TF's assign_moving_average(zero_debias=False): v -= (v - value)*(1 - m); the variance is first
    scaled by n/(n-(1+eps)) (keras 2.2.4 normalization.py, "sample variance")."""
    var_c = var * (n / (n - (1.0 + eps)))
    d = 1.0 - momentum
    return (moving_mean - (moving_mean - mean) * d, moving_var - (moving_var - var_c) * d)


def bn_moving_update_zero_debias(moving_mean, moving_var, zd, mean, var, n, momentum, eps=BN_EPS):
    """TF 1.12 moving_averages.assign_moving_average(variable, value, decay, zero_debias=True), which is what keras 2.2.4's TF backend
    calls from K.moving_average_update (recollection of tensorflow_backend.py; neither source is in the container):
        biased     -= (biased - value) * (1 - m)           # shadow accumulator, initialised to ZERO whatever the variable holds
        local_step += 1
        variable   -= variable - biased / (1 - m**local_step)
    i.e. the moving statistic is the debiased average of the batch values seen so far and keeps no memory of its 0 / 1 initial value.
    `zd` = [biased_mean, biased_var, local_step] (one per call site of the layer in TF; see layers.BatchNormalization).  Returns the new
    (moving_mean, moving_var, zd)."""
    var_c = var * (n / (n - (1.0 + eps)))
    d = 1.0 - momentum
    bm = zd[0] - (zd[0] - mean) * d
    bv = zd[1] - (zd[1] - var_c) * d
    t = zd[2] + 1
    corr = 1.0 - momentum ** t
    return (moving_mean - (moving_mean - bm / corr), moving_var - (moving_var - bv / corr), [bm, bv, t])


def bn_train_bwd(dy, cache, gamma):
    xhat, inv = cache
    axes = tuple(range(dy.ndim - 1))
    n = np.prod([dy.shape[a] for a in axes])
    dgamma = (dy * xhat).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    dx = (gamma * inv / n) * (n * dy - dbeta - xhat * dgamma)
    return dx, dgamma, dbeta


def bn_infer_fwd(x, gamma, beta, moving_mean, moving_var, eps=BN_EPS):
    return gamma * (x - moving_mean) / np.sqrt(moving_var + eps) + beta


# ----------------------------------------------------------------------------------------------
# activations  (fwd returns y; bwd takes y = the activation OUTPUT)
# ----------------------------------------------------------------------------------------------
def act_fwd(x, kind, param=0.0):
    if kind in ('linear', None):
        return x
    if kind == 'relu':
        return np.maximum(x, 0)
    if kind == 'relu_max':                  # keras ReLU(max_value=param)
        return np.clip(x, 0, param)
    if kind == 'leaky':                     # LeakyReLU(alpha=param)
        return np.where(x > 0, x, param * x)
    if kind == 'tanh':
        return np.tanh(x)
    if kind == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-x))
    raise ValueError(kind)


def act_bwd(dy, y, kind, param=0.0):
    if kind in ('linear', None):
        return dy
    if kind == 'relu':
        return dy * (y > 0)
    if kind == 'relu_max':
        return dy * ((y > 0) & (y < param))
    if kind == 'leaky':
        return dy * np.where(y > 0, 1.0, param)
    if kind == 'tanh':
        return dy * (1 - y * y)
    if kind == 'sigmoid':
        return dy * y * (1 - y)
    raise ValueError(kind)


def dropout_fwd(x, mask, rate):
    """Inverted dropout with an injected keep-mask (1 keep / 0 drop)."""
    return x * mask / (1.0 - rate)


# ----------------------------------------------------------------------------------------------
# losses / metrics  
# ----------------------------------------------------------------------------------------------
SAT_BAND = 1e-4        # min(p, 1 - p) below this: the sample is in the regime where fp32 cannot represent 1 - p to more than a few bits


def bce_loss(p, y, p_impl=None):
    """keras binary_crossentropy (TF backend, from probabilities). p,y: (B,1). Returns (loss, dL/dp).

    p_impl (optional, (B,1) float32): the probabilities the fp32 implementation under test produced.  TF evaluates this loss in the tensor dtype,
    float32: for a sample the discriminator has (nearly) saturated, 1 - p has one or two significant bits there, the clip bound 1 - 1e-7 is ONE fp32
    step below 1, and dL/dp = (sigma(z) - y) / (p (1 - p)) is what fp32 makes of it -- an fp64 evaluation of the same expression differs from it by tens
    of per cent, in a quantity Adam then normalises to a full step.  Like the branch of a ReLU at its kink (nets_ref.Stack.forward `decisions`), that
    is not something a higher-precision oracle can decide: for the samples with min(p, 1 - p) < SAT_BAND the loss term and its gradient are evaluated
    in float32 AT the implementation's p, after checking that this p is the oracle's to two fp32 steps; all other samples stay fp64.
    self-check result in bce_loss.last = (samples in the saturated band, largest |p_impl - p| among them in units of 2^-24)."""
    B = p.shape[0]
    pc = np.clip(p, CLIP_LO, CLIP_HI)
    z = np.log(pc / (1 - pc))
    per = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    inside = (p >= CLIP_LO) & (p <= CLIP_HI)
    dz = (1.0 / (1.0 + np.exp(-z)) - y) / (B * p.shape[-1])
    dp = np.where(inside, dz / (pc * (1 - pc)), 0.0)
    bce_loss.last = (0, 0.0)
    if p_impl is not None:
        sat = np.minimum(p, 1 - p) < SAT_BAND
        if sat.any():
            p32 = np.asarray(p_impl, np.float32).reshape(p.shape)
            y32 = np.asarray(y, np.float32).reshape(p.shape)
            bce_loss.last = (int(sat.sum()), float(np.abs(p32.astype(np.float64) - p)[sat].max() / 2.0 ** -24))
            one, eps = np.float32(1.0), np.float32(K_EPS)
            pc32 = np.minimum(np.maximum(p32, eps), one - eps)
            z32 = np.log(pc32 / (one - pc32))
            per32 = np.maximum(z32, np.float32(0)) - z32 * y32 + np.log1p(np.exp(-np.abs(z32)))
            sg32 = one / (one + np.exp(-z32))
            in32 = (p32 >= eps) & (p32 <= one - eps)
            dp32 = np.where(in32, (sg32 - y32) / (pc32 * (one - pc32)) / np.float32(B * p.shape[-1]), np.float32(0))
            per = np.where(sat, per32.astype(np.float64), per)
            dp = np.where(sat, dp32.astype(np.float64), dp)
    loss = per.mean(axis=-1).mean()
    return loss, dp


def mse_loss(p, y):
    B = p.shape[0]
    d = p - y
    return (d * d).mean(axis=-1).mean(), 2 * d / (B * p.shape[-1])


def binary_accuracy(p, y):
    return float(np.mean(np.round(p) == y))


# ----------------------------------------------------------------------------------------------
# Adam, keras form (bbhMahoGANy.py:1101-1119: lr=9e-5, beta_1=0.5)
# ----------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, t, lr=9e-5, b1=0.5, b2=0.999, eps=K_EPS):
    """t is the 1-based step index AFTER increment. Returns new (p, m, v).
    lr, beta_1 and beta_2 are float32 VARIABLES in Keras (K.variable): the numbers that take part are float32(lr) etc. -- what the
    reference's own Keras files record (training_config of 2_model_version/weight_version/d_model.hdf5: beta_2 0.9990000128746033,
    lr 0.004000000189989805; tests/golden/keras_h5_golden.json), so 1 - beta_2 is 0.00099998713, not 0.001.  epsilon is a python float."""
    lr, b1, b2 = (float(np.float32(x)) for x in (lr, b1, b2))
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


def glorot_uniform(rng, shape, dtype=np.float32):
    """keras VarianceScaling(scale=1, fan_avg, uniform); conv fan_in = receptive*Cin, fan_out = receptive*Cout."""
    if len(shape) == 2:
        fi, fo = shape
    else:
        rec = int(np.prod(shape[:-2]))
        fi, fo = rec * shape[-2], rec * shape[-1]
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)
