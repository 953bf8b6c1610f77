#!/usr/bin/env python
"""Per-layer timing of the MFMA convolution kernels at the shapes and batch sizes bench.py runs (n_pix = 2048):
forward, data gradient and weight gradient of every Conv1D / folded Conv2D with Cin >= 5 of the three networks.
Prints ms, TFLOP/s and each layer's share of the summed MFMA time of one bench step -- shows which shapes sit below the
kernel's large-layer rate.

  python scripts/layer_sweep.py [--iters 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, B, L, Cin, Cout, k, stride, padding, launches of fwd / dgrad / wgrad per bench step)
# G runs forward 3x per GAN iteration (predict, and inside the G step; the D step uses the predicted batch), backward once.
LAYERS = [
    # the two UpSampling1D -> Conv1D pairs run folded: 3 taps, stride 1, on the un-upsampled tensor (gn_conv1d_up2_fold)
    ('G  Up2+conv 256->64 s2 (3-tap fold)', 512, 1024, 256, 64, 3, 1, 'same', 2, 1, 1),
    ('G  Up2+conv 64->128 (3-tap, 2 phases)', 512, 1024, 64, 256, 3, 1, 'same', 2, 1, 1),
    ('G  conv 128->256', 512, 2048, 128, 256, 5, 1, 'same', 2, 1, 1),
    ('G  conv 256->512', 512, 2048, 256, 512, 5, 1, 'same', 2, 1, 1),
    ('G  conv 512->1024', 512, 2048, 512, 1024, 5, 1, 'same', 2, 1, 1),
    ('D  folded conv2 512->1024 s2 (2B)', 1024, 1024, 512, 1024, 5, 2, 'same', 1, 1, 1),
    ('D  folded conv2 512->1024 s2 (B)', 512, 1024, 512, 1024, 5, 2, 'same', 1, 1, 0),
    ('PE mc 64->128 s2', 256, 1024, 64, 128, 5, 2, 'valid', 2, 2, 2),
    ('PE mc 128->256 s2', 256, 510, 128, 256, 5, 2, 'valid', 2, 2, 2),
    ('PE mc 256->512 s2', 256, 253, 256, 512, 5, 2, 'valid', 2, 2, 2),
    ('PE q 64->128', 256, 2048, 64, 128, 5, 1, 'valid', 2, 2, 2),
    ('PE q 128->256', 256, 2044, 128, 256, 5, 1, 'valid', 2, 2, 2),
    ('PE q 256->512 s2', 256, 2040, 256, 512, 5, 2, 'valid', 2, 2, 2),
    ('PE q 512->1024 s2', 256, 1018, 512, 1024, 5, 2, 'valid', 2, 2, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--math', default=None, help="conv arithmetic: wino (the engine's default) | fp32 (direct kernels only) | bf16x3")
    a = ap.parse_args()
    from gennet_amd import engine, ops
    dev = engine.device()                 # sets the process's conv arithmetic (GENNET_CONV_MATH, 'wino' when unset), as every model does
    if a.math:
        ops.set_conv_math(a.math)
    print('conv math: %s' % (a.math or ops.default_conv_math()))
    rows = []
    for name, B, L, cin, cout, k, s, padding, nf, nd, nw in LAYERS:
        x = torch.randn(B, L, cin, device=dev)
        w = torch.randn(k, cin, cout, device=dev) * 0.02
        b = torch.zeros(cout, device=dev)
        Lout, pl = ops.conv_geometry(L, k, s, padding)
        dy = torch.randn(B, Lout, cout, device=dev)
        wt = ops.conv1d_transpose_w(w)
        flop = 2.0 * B * Lout * k * cin * cout
        runs = (('fwd', nf, lambda: ops.conv1d_fwd(x, w, b, s, pl, Lout, 'relu')),
                ('dgrad', nd, lambda: ops.conv1d_dgrad(dy, wt, L, s, pl)),
                ('wgrad', nw, lambda: ops.conv1d_wgrad(x, dy, k, s, pl)))
        for what, count, fn in runs:
            fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            rows.append((name, what, ms, flop / ms / 1e9, count))
        del x, w, dy, wt
    total = sum(ms * c for _, _, ms, _, c in rows)
    print('%-40s %-6s %9s %9s %7s %7s' % ('layer', 'pass', 'ms', 'TFLOP/s', 'x/step', 'share'))
    for name, what, ms, tf, c in rows:
        print('%-40s %-6s %9.3f %9.1f %7d %6.1f%%' % (name, what, ms, tf, c, 100.0 * ms * c / total), flush=True)
    print('sum of MFMA conv time per bench step: %.1f ms' % total)


if __name__ == '__main__':
    main()
