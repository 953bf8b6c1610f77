#!/usr/bin/env python
"""Experimental bf16x3 convolution (csrc/conv_bf16x3.hip): error against an fp64 reference next to the exact-fp32 MFMA path,
and throughput on the dominant layer shapes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gennet_amd import ops  # noqa: E402


def ref64(x, w, b, s, pl, Lout):
    B, L, Cin = x.shape
    k, _, Cout = w.shape
    xp = np.zeros((B, L + 2 * k + s * Lout, Cin)); xp[:, k:k + L] = x
    y = np.zeros((B, Lout, Cout))
    for j in range(k):
        idx = k + s * np.arange(Lout) + j - pl
        y += xp[:, idx] @ w[j]
    return y + b


def main():
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(0)
    for (B, L, Cin, Cout, k, s, padding) in ((2, 300, 64, 128, 5, 1, 'same'), (3, 277, 48, 256, 5, 1, 'valid'), (2, 200, 32, 128, 3, 1, 'valid'), (1, 256, 512, 1024, 5, 1, 'same')):
        x = (rng.randn(B, L, Cin) * np.exp(rng.randn(B, L, Cin))).astype(np.float32)
        w = (rng.randn(k, Cin, Cout) / np.sqrt(k * Cin)).astype(np.float32); b = rng.randn(Cout).astype(np.float32)
        Lout, pl = ops.conv_geometry(L, k, s, padding)
        r = ref64(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s, pl, Lout)
        xt, wt, bt = (torch.tensor(v).to(dev) for v in (x, w, b))
        y32 = ops.conv1d_fwd(xt, wt, bt, s, pl, Lout).cpu().numpy()
        y3 = ops.conv1d_fwd_bf16x3(xt, wt, bt, s, pl, Lout).cpu().numpy()
        sc = np.abs(r).max()
        print('B%d L%d %d->%d k%d s%d %s: max|err|/max|y|  fp32-mfma %.3e   bf16x3 %.3e   rms  %.3e  %.3e' % (
            B, L, Cin, Cout, k, s, padding, np.abs(y32 - r).max() / sc, np.abs(y3 - r).max() / sc,
            np.sqrt(np.mean((y32 - r) ** 2)) / sc, np.sqrt(np.mean((y3 - r) ** 2)) / sc), flush=True)
    for (B, L, Cin, Cout, s, padding) in ((64, 2048, 512, 1024, 1, 'same'), (64, 2048, 256, 512, 1, 'same')):
        x = torch.randn(B, L, Cin, device=dev); w = torch.randn(5, Cin, Cout, device=dev) * 0.02; b = torch.zeros(Cout, device=dev)
        Lout, pl = ops.conv_geometry(L, 5, s, padding)
        flop = 2.0 * B * Lout * 5 * Cin * Cout
        for name, fn in (('fp32-mfma', lambda: ops.conv1d_fwd(x, w, b, s, pl, Lout, 'relu')),
                         ('bf16x3 (conv only)', lambda: ops.conv1d_fwd_bf16x3(x, w, b, s, pl, Lout, 'relu', resplit=False)),
                         ('bf16x3 (+split)', lambda: ops.conv1d_fwd_bf16x3(x, w, b, s, pl, Lout, 'relu', resplit=True))):
            ops.conv1d_fwd_bf16x3(x, w, b, s, pl, Lout, 'relu')
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print('B%d L%d %d->%d s%d  %-20s %.3f ms  %.1f TFLOP/s (fp32-equivalent)' % (B, L, Cin, Cout, s, name, ms, flop / ms / 1e9), flush=True)


if __name__ == '__main__':
    main()
