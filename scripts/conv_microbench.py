#!/usr/bin/env python
"""Micro-benchmark of the MFMA convolution kernels on one layer shape (default: the dominant generator layer
Conv1D(512 -> 1024, k5, same) on (B, 2048, 512)); prints achieved TFLOP/s from HIP events.  Used under rocprofv3 --pmc.

  python scripts/conv_microbench.py [--B 64] [--L 2048] [--cin 512] [--cout 1024] [--stride 1] [--padding same] [--iters 10] [--what fwd|dgrad|wgrad|all]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--L', type=int, default=2048)
    ap.add_argument('--cin', type=int, default=512)
    ap.add_argument('--cout', type=int, default=1024)
    ap.add_argument('--stride', type=int, default=1)
    ap.add_argument('--padding', default='same')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--what', default='all')
    a = ap.parse_args()
    from gennet_amd import ops
    dev = torch.device('cuda:0')
    x = torch.randn(a.B, a.L, a.cin, device=dev)
    w = torch.randn(5, a.cin, a.cout, device=dev) * 0.02
    b = torch.zeros(a.cout, device=dev)
    Lout, pl = ops.conv_geometry(a.L, 5, a.stride, a.padding)
    dy = torch.randn(a.B, Lout, a.cout, device=dev)
    wt = ops.conv1d_transpose_w(w)
    flop = 2.0 * a.B * Lout * 5 * a.cin * a.cout
    runs = {'fwd': lambda: ops.conv1d_fwd(x, w, b, a.stride, pl, Lout, 'relu'),
            'dgrad': lambda: ops.conv1d_dgrad(dy, wt, a.L, a.stride, pl),
            'wgrad': lambda: ops.conv1d_wgrad(x, dy, 5, a.stride, pl)}
    for name, fn in runs.items():
        if a.what not in ('all', name):
            continue
        fn(); fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print('%-6s B=%d L=%d %d->%d s=%d: %.3f ms  %.1f TFLOP/s' % (name, a.B, a.L, a.cin, a.cout, a.stride, ms, flop / ms / 1e9), flush=True)


if __name__ == '__main__':
    main()
