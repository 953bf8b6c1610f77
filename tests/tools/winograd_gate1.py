"""Gate 1 of the transform-domain (Winograd / Cook-Toom) fp32 convolution: CPU-only numerics.

Question: does a 1-D F(2,5) convolution (6 multiplies per 2 outputs instead of 10) evaluated in fp32 -- input transform in fp32 registers,
six point-GEMMs as k-ordered fp32 fma chains (what v_mfma_f32_32x32x2_f32 computes), output transform in fp32 -- stay within 4x of the error
the DIRECT k-ordered fp32 fma chain (today's kernels) has against an fp64 evaluation of the same layer, on the BASELINE layer shapes
(bbhMahoGANy.py:259-283 generator, :382-386 PE q branch)?  Same question for the weight gradient in the transposed form
dW = G^T [ sum_tiles (B^T x) . (A dy) ].

TEST INFRASTRUCTURE / measurement tool: nothing under gennet_amd/ imports it.  Run:  python tests/tools/winograd_gate1.py [--quick]
Writes profiles/r05_winograd_gate1.txt when --out is given.
"""
import argparse
import itertools
import sys
import time
from fractions import Fraction

import numpy as np

INF = 'inf'


# ----------------------------------------------------------------------------------------------
# Cook-Toom matrices in exact rationals.  Linear convolution s = C [(Ag g) . (Ad d)] (evaluate, multiply, interpolate); the FIR form
# F(m, r) is its transpose: y = Ad^T [ (Ag g) . (C^T x) ]  ->  AT = Ad^T (m x n), G = Ag (n x r), BT = C^T (n x n), n = m + r - 1.
# ----------------------------------------------------------------------------------------------
def _vander(points, cols):
    rows = []
    for p in points:
        if p == INF:
            rows.append([Fraction(0)] * (cols - 1) + [Fraction(1)])
        else:
            rows.append([Fraction(p) ** j for j in range(cols)])
    return rows


def _inv(M):
    n = len(M)
    A = [list(r) + [Fraction(int(i == j)) for j in range(n)] for i, r in enumerate(M)]
    for c in range(n):
        piv = next(r for r in range(c, n) if A[r][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        pv = A[c][c]
        A[c] = [v / pv for v in A[c]]
        for r in range(n):
            if r != c and A[r][c] != 0:
                f = A[r][c]
                A[r] = [a - f * b for a, b in zip(A[r], A[c])]
    return [r[n:] for r in A]


def cook_toom(m, r, points, scale='bt_int'):
    """Returns (AT, G, BT) as float64 arrays.  scale: 'plain' | 'bt_int' (rows of BT scaled so the row's entries are integers with gcd 1 where
    possible, the inverse factor moved into G -- the familiar integer-looking input transform)."""
    n = m + r - 1
    assert len(points) == n
    Ag = _vander(points, r)
    Ad = _vander(points, m)
    C = _inv(_vander(points, n))                # n x n: coefficients from values
    BT = [[C[j][i] for j in range(n)] for i in range(n)]        # C^T
    G = [list(row) for row in Ag]
    AT = [[Ad[i][j] for i in range(n)] for j in range(m)]
    if scale == 'bt_int':
        from math import gcd
        for i in range(n):
            dens = [v.denominator for v in BT[i]]
            l = 1
            for d in dens:
                l = l * d // gcd(l, d)
            nums = [int(v * l) for v in BT[i]]
            g = 0
            for v in nums:
                g = gcd(g, abs(v))
            f = Fraction(l, g if g else 1)
            BT[i] = [v * f for v in BT[i]]
            G[i] = [v / f for v in G[i]]
    f64 = lambda M: np.array([[float(v) for v in row] for row in M], np.float64)
    return f64(AT), f64(G), f64(BT)


def check_exact(m, r, points):
    AT, G, BT = cook_toom(m, r, points)
    rng = np.random.RandomState(0)
    g = rng.randn(r)
    d = rng.randn(m + r - 1)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[j] * d[i + j] for j in range(r)) for i in range(m)])
    assert np.allclose(y, ref, rtol=1e-10, atol=1e-10), (points, y, ref)


# ----------------------------------------------------------------------------------------------
# fp32 emulation.  fma(a, b, c) with a, b, c fp32: the product is exact in fp64, the sum rounds once to 53 bits and once to 24 (double
# rounding differs from a true fma in ~2^-29 of the cases: irrelevant for error statistics).
# ----------------------------------------------------------------------------------------------
def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def lin32(M, xs):
    """rows of M applied to the list of fp32 arrays xs with one fp32 fma per non-zero entry, left to right (what a register transform does)."""
    out = []
    for row in M:
        acc = None
        for c, x in zip(row, xs):
            if c == 0.0:
                continue
            cf = np.float32(c)
            assert float(cf) == c or abs(float(cf) - c) < 1e-7 * abs(c)
            if acc is None:
                acc = (np.float32(c) * x).astype(np.float32) if c != 1.0 else x.copy()
            else:
                acc = fma32(np.full((), cf, np.float32), x, acc)
        out.append(acc if acc is not None else np.zeros_like(xs[0]))
    return out


def chain_gemm32(A, Bm):
    """C[i, j] = k-ordered fp32 fma chain of sum_k A[i, k] * B[k, j] (the exact-fp32 MFMA kernels' arithmetic)."""
    acc = np.zeros((A.shape[0], Bm.shape[1]), np.float32)
    A64 = A.astype(np.float64)
    B64 = Bm.astype(np.float64)
    for k in range(A.shape[1]):
        acc = (A64[:, k, None] * B64[None, k, :] + acc.astype(np.float64)).astype(np.float32)
    return acc


def direct_fwd32(x, W):
    """x (L + r - 1, Cin) fp32 (already padded), W (r, Cin, Cout) fp32 -> (L, Cout): one chain over (tap, channel) in the kernels' order
    (channel chunks of 8 outermost, taps, then channels: the order does not matter for the statistics)."""
    r, Cin, Cout = W.shape
    L = x.shape[0] - r + 1
    A = np.concatenate([x[j:j + L] for j in range(r)], axis=1)              # (L, r*Cin)
    Bm = W.reshape(r * Cin, Cout)
    return chain_gemm32(A, Bm)


def wino_fwd32(x, W, m, mats):
    AT, G, BT = mats
    r, Cin, Cout = W.shape
    n = m + r - 1
    L = x.shape[0] - r + 1
    assert L % m == 0
    T = L // m
    U = np.einsum('pj,jck->pck', G, W.astype(np.float64)).astype(np.float32)           # weight transform: fp64 pass, rounded once
    d = [x[j:j + m * T:m] if False else x[j + m * np.arange(T)] for j in range(n)]     # d[j][t, c] = x[m t + j, c]
    V = lin32(BT, d)
    Mp = [chain_gemm32(V[p], U[p]) for p in range(n)]
    Y = lin32(AT, Mp)                                                                  # m arrays (T, Cout)
    y = np.empty((L, Cout), np.float32)
    for i in range(m):
        y[i::m] = Y[i]
    return y


def direct_wgrad32(x, dy, r):
    """dW[j] = sum_t x[t + j]^T dy[t]: chain over t."""
    L = dy.shape[0]
    return np.stack([chain_gemm32(np.ascontiguousarray(x[j:j + L].T), dy) for j in range(r)])


def wino_wgrad32(x, dy, r, m, mats):
    """dW = G^T [ sum_tiles (B^T x_tile) . (A dy_tile) ]: both operand transforms in fp32 registers, six chains over the tiles, the final
    G^T (5 x 6 per (ci, co)) in fp64 on the fp32 sums (it rides on the split-K reduce pass)."""
    AT, G, BT = mats
    n = m + r - 1
    L = dy.shape[0]
    T = L // m
    d = [x[j + m * np.arange(T)] for j in range(n)]
    V = lin32(BT, d)                                                # n x (T, Cin)
    e = [dy[i + m * np.arange(T)] for i in range(m)]
    Dp = lin32(AT.T, e)                                             # n x (T, Cout)
    Q = np.stack([chain_gemm32(np.ascontiguousarray(V[p].T), Dp[p]) for p in range(n)])      # (n, Cin, Cout)
    return np.einsum('pj,pck->jck', G, Q.astype(np.float64)).astype(np.float32)


def err(a, ref):
    d = a.astype(np.float64) - ref
    s = np.sqrt(np.mean(ref ** 2))
    return np.abs(d).max() / s, np.sqrt(np.mean(d ** 2)) / s


POINT_SETS_6 = {
    '0,1,-1,2,-2,inf': [0, 1, -1, 2, -2, INF],
    '0,1,-1,1/2,-1/2,inf': [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), INF],
    '0,1,-1,1/2,-2,inf': [0, 1, -1, Fraction(1, 2), -2, INF],
    '0,1,-1,2,-1/2,inf': [0, 1, -1, 2, Fraction(-1, 2), INF],
}
POINT_SETS_4 = {'0,1,-1,inf': [0, 1, -1, INF]}


def run(out):
    def say(*a):
        line = ' '.join(str(v) for v in a)
        print(line, flush=True)
        out.append(line)

    for ps in POINT_SETS_6.values():
        check_exact(2, 5, ps)
        check_exact(4, 3, ps)
    check_exact(2, 3, POINT_SETS_4['0,1,-1,inf'])
    say('# Cook-Toom matrices exact (checked in fp64 against the direct sum).')
    for name, ps in POINT_SETS_6.items():
        AT, G, BT = cook_toom(2, 5, ps)
        say('F(2,5) points', name)
        say('  BT =', np.array2string(BT, max_line_width=200, precision=5).replace('\n', '\n       '))
        say('  G  =', np.array2string(G, max_line_width=200, precision=5).replace('\n', '\n       '))
        say('  AT =', np.array2string(AT, max_line_width=200, precision=5).replace('\n', '\n       '))

    rng = np.random.RandomState(5)
    shapes = [('G 128->256', 128, 256), ('G 256->512', 256, 512), ('G 512->1024', 512, 1024), ('PE q 64->128', 64, 128), ('PE q 128->256', 128, 256)]
    L = 64
    say('\n# forward / data gradient (same linear map).  Activations: tanh outputs with Dropout(0.2) as the generator has them; weights glorot-uniform.')
    say('# error = |fp32 result - fp64 result| / rms(fp64 result); columns: max, rms.  L = %d positions, one batch element.' % L)
    results = {}
    for lname, Cin, Cout in shapes:
        lim = np.sqrt(6.0 / (5 * (Cin + Cout)))
        W = rng.uniform(-lim, lim, (5, Cin, Cout)).astype(np.float32)
        x = np.tanh(rng.randn(L + 4, Cin)) * (rng.rand(L + 4, Cin) > 0.2) / 0.8
        x = x.astype(np.float32)
        A = np.concatenate([x[j:j + L] for j in range(5)], axis=1).astype(np.float64)
        ref = A @ W.reshape(5 * Cin, Cout).astype(np.float64)
        t0 = time.time()
        e_dir = err(direct_fwd32(x, W), ref)
        say('%-14s direct fp32 chain     max %.3e rms %.3e   (%.1fs)' % (lname, e_dir[0], e_dir[1], time.time() - t0))
        for name, ps in POINT_SETS_6.items():
            mats = cook_toom(2, 5, ps)
            e = err(wino_fwd32(x, W, 2, mats), ref)
            say('%-14s F(2,5) %-20s max %.3e rms %.3e   ratio to direct: max %.2f rms %.2f' % (lname, name, e[0], e[1], e[0] / e_dir[0], e[1] / e_dir[1]))
            results[('fwd', lname, name)] = (e[0] / e_dir[0], e[1] / e_dir[1])

    say('\n# weight gradient, transposed form.  K = B * L rows; here 2048 rows (the error of a chain grows with its length; the kernels split K into slabs')
    say('# of ~2048-8192 rows per partial sum and reduce the partials in a second pass).  x as above, dy ~ N(0, 1).')
    Lw = 2048
    for lname, Cin, Cout in [('G 128->256', 128, 256), ('G 512->1024 (tile 128x128)', 128, 128)]:
        x = (np.tanh(rng.randn(Lw + 4, Cin)) * (rng.rand(Lw + 4, Cin) > 0.2) / 0.8).astype(np.float32)
        dy = rng.randn(Lw, Cout).astype(np.float32)
        ref = np.stack([x[j:j + Lw].astype(np.float64).T @ dy.astype(np.float64) for j in range(5)])
        e_dir = err(direct_wgrad32(x, dy, 5), ref)
        say('%-28s direct fp32 chain     max %.3e rms %.3e' % (lname, e_dir[0], e_dir[1]))
        for name, ps in POINT_SETS_6.items():
            mats = cook_toom(2, 5, ps)
            e = err(wino_wgrad32(x, dy, 5, 2, mats), ref)
            say('%-28s F(2,5)^T %-20s max %.3e rms %.3e   ratio to direct: max %.2f rms %.2f' % (lname, name, e[0], e[1], e[0] / e_dir[0], e[1] / e_dir[1]))
            results[('wgrad', lname, name)] = (e[0] / e_dir[0], e[1] / e_dir[1])

    say('\n# folded 3-tap layers (UpSampling1D folded into the conv): F(2,3), 4 multiplies per 2 outputs instead of 6')
    for lname, Cin, Cout in [('G up-fold 256->64', 256, 64), ('G up-fold 64->2x128', 64, 256)]:
        lim = np.sqrt(6.0 / (5 * (Cin + Cout)))
        W = rng.uniform(-lim, lim, (3, Cin, Cout)).astype(np.float32)
        x = (np.tanh(rng.randn(L + 2, Cin)) * (rng.rand(L + 2, Cin) > 0.2) / 0.8).astype(np.float32)
        A = np.concatenate([x[j:j + L] for j in range(3)], axis=1).astype(np.float64)
        ref = A @ W.reshape(3 * Cin, Cout).astype(np.float64)
        e_dir = err(direct_fwd32(x, W), ref)
        mats = cook_toom(2, 3, POINT_SETS_4['0,1,-1,inf'])
        e = err(wino_fwd32(x, W, 2, mats), ref)
        say('%-20s direct max %.3e rms %.3e | F(2,3) max %.3e rms %.3e  ratio max %.2f rms %.2f' % (lname, e_dir[0], e_dir[1], e[0], e[1], e[0] / e_dir[0], e[1] / e_dir[1]))
    return results

# ----------------------------------------------------------------------------------------------
# Net level: one generator update through the frozen discriminator (bbhMahoGANy.py:1296) and one CNN train_on_batch (:1165) of the fp64
# oracle, with the stride-1 5-tap convolutions (Cin >= 128 in G, the q branch's 64 -> 128 -> 256) replaced by the fp32 emulations
# above (forward, data gradient, weight gradient).  Bounds: the GPU parity tests' own (tests/test_nets_gpu.py): loss 1e-5 relative,
# gradients 1e-4 of the largest entry of each tensor.
# ----------------------------------------------------------------------------------------------
def net_level(say, points_name='0,1,-1,1/2,-2,inf'):
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from oracle import keras_ref as K
    from oracle import nets_ref as N
    mats = cook_toom(2, 5, POINT_SETS_6[points_name])
    orig_fwd, orig_bwd = K.conv1d_fwd, K.conv1d_bwd

    def eligible(W, stride):
        return W.shape[0] == 5 and stride == 1 and W.shape[1] >= 64 and W.shape[2] >= 64

    def make(mode):
        def fwd(x, W, b, stride=1, padding='valid'):
            if not eligible(W, stride):
                return orig_fwd(x, W, b, stride, padding)
            xp, out, pl = K._pad1d(x, 5, 1, padding)
            oe = out + (out & 1)
            xpe = np.zeros((x.shape[0], oe + 4, x.shape[2]), np.float32)
            xpe[:, :xp.shape[1]] = xp
            W32 = W.astype(np.float32)
            y = np.stack([(wino_fwd32(xpe[i], W32, 2, mats) if mode == 'wino' else direct_fwd32(xpe[i], W32))[:out] for i in range(x.shape[0])]).astype(np.float64)
            return y + b if b is not None else y

        def bwd(x, W, dy, stride=1, padding='valid'):
            if not eligible(W, stride):
                return orig_bwd(x, W, dy, stride, padding)
            B, L, Cin = x.shape
            xp, out, pl = K._pad1d(x, 5, 1, padding)
            oe = out + (out & 1)
            xpe = np.zeros((B, oe + 4, Cin), np.float32)
            xpe[:, :xp.shape[1]] = xp
            dye = np.zeros((B, oe, W.shape[2]), np.float32)
            dye[:, :out] = dy
            # weight gradient: one chain per batch element (a K-split), partials summed in fp64 (the reduce pass)
            dW = np.zeros(W.shape, np.float64)
            for i in range(B):
                dW += (wino_wgrad32(xpe[i], dye[i], 5, 2, mats) if mode == 'wino' else direct_wgrad32(xpe[i], dye[i], 5)).astype(np.float64)
            # data gradient: full correlation of dy with the flipped, transposed kernel
            Wt = np.ascontiguousarray(W[::-1].transpose(0, 2, 1)).astype(np.float32)
            Lp = xp.shape[1]
            Le = Lp + (Lp & 1)
            dyp = np.zeros((B, Le + 4, W.shape[2]), np.float32)
            dyp[:, 4:4 + out] = dy
            dxp = np.stack([(wino_fwd32(dyp[i], Wt, 2, mats) if mode == 'wino' else direct_fwd32(dyp[i], Wt))[:Lp] for i in range(B)]).astype(np.float64)
            return dxp[:, pl:pl + L], dW, dy.sum(axis=(0, 1))
        return fwd, bwd

    def relg(a, b):
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)

    def run_gan(mode):
        if mode:
            K.conv1d_fwd, K.conv1d_bwd = make(mode)
        try:
            n_pix, B = 64, 4
            rng = np.random.RandomState(11)
            gan = N.GAN(n_pix, rng.randn(n_pix), rng=np.random.RandomState(2))
            for st in (gan.G, gan.D):
                for p in st.params:
                    p[...] = p.astype(np.float32).astype(np.float64)
            z = rng.uniform(-1, 1, (B, 100))
            mr = np.random.RandomState(3)
            gm, h = {}, None
            # dropout masks by walking shapes
            shapes = {}
            x = z
            for li, s in enumerate(gan.G.spec):
                if s[0] == 'dense': x = np.zeros((B, s[2]))
                elif s[0] == 'reshape': x = np.zeros((B,) + tuple(s[1]))
                elif s[0] == 'up': x = np.zeros((B, x.shape[1] * 2, x.shape[2]))
                elif s[0] == 'conv1d': x = np.zeros((B, K.conv_out_len(x.shape[1], s[3], s[4], s[5]), s[2]))
                elif s[0] == 'drop': gm[li] = (mr.rand(*x.shape) >= s[1]).astype(np.float64)
            dm = {2: (mr.rand(B, n_pix // 2, 2, 256) >= 0.4).astype(np.float64), 5: (mr.rand(B, n_pix // 4, 2, 512) >= 0.4).astype(np.float64)}
            out = gan.g_train_on_batch(z, [1.0] * B, gm, dm)
            return out[0], gan.last_g_grads
        finally:
            K.conv1d_fwd, K.conv1d_bwd = orig_fwd, orig_bwd

    def run_pe(mode):
        if mode:
            K.conv1d_fwd, K.conv1d_bwd = make(mode)
        try:
            n_pix, B = 128, 4
            rng = np.random.RandomState(12)
            pe = N.PENet(n_pix, rng=np.random.RandomState(1))
            for st in (pe.mc, pe.q):
                for p in st.params:
                    p[...] = p.astype(np.float32).astype(np.float64)
            x = rng.randn(B, n_pix, 1)
            out = pe.train_on_batch(x, rng.uniform(20, 35, B), rng.uniform(0.5, 1, B))
            return out[0], pe.last_grads
        finally:
            K.conv1d_fwd, K.conv1d_bwd = orig_fwd, orig_bwd

    say('\n# net level (points %s): fp64 oracle vs the same oracle with the eligible convolutions computed in emulated fp32' % points_name)
    ok = True
    for nm, fn in (('generator update through frozen D, n_pix 64, B 4', run_gan), ('CNN train_on_batch, n_pix 128, B 4', run_pe)):
        l0, g0 = fn(None)
        for mode in ('direct', 'wino'):
            l1, g1 = fn(mode)
            worst = max(relg(a, b) for a, b in zip(g1, g0) if b is not None and np.abs(b).max() > 1e-12)
            lr = abs(l1 - l0) / abs(l0)
            say('%-50s %-6s loss rel %.2e (bound 1e-5)   worst gradient tensor %.2e of its largest entry (bound 1e-4)' % (nm, mode, lr, worst))
            if mode == 'wino' and not (lr <= 1e-5 and worst <= 1e-4):
                ok = False
    say('net-level gate: %s' % ('PASS' if ok else 'FAIL'))
    return ok



if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out')
    a = ap.parse_args()
    lines = []
    res = run(lines)
    worst = {}
    for (kind, lname, name), (rmax, rrms) in res.items():
        w = worst.setdefault(name, [0.0, 0.0])
        w[0] = max(w[0], rmax)
        w[1] = max(w[1], rrms)
    print('\n# worst ratio to the direct chain over all layers, per point set (gate: <= 4):')
    lines.append('\n# worst ratio to the direct chain over all layers, per point set (gate: <= 4):')
    for name, w in worst.items():
        s = '%-22s max-error ratio %.2f, rms-error ratio %.2f -> %s' % (name, w[0], w[1], 'PASS' if max(w) <= 4 else 'FAIL')
        print(s)
        lines.append(s)
    net_level(lambda *v: (print(*v, flush=True), lines.append(' '.join(str(x) for x in v))))
    if a.out:
        open(a.out, 'w').write('\n'.join(lines) + '\n')
