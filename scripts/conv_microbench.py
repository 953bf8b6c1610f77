#!/usr/bin/env python
"""Time single conv launches (forward, data gradient, weight gradient) at given shapes: us per launch and TFLOP/s.  The kernel-selection
switches of csrc/conv_pipe.hip are read once per process from the environment (GN_CONV_NONARROW, GN_CONV_NARROW_BELOW, GN_CONV_NARROW_BLOCKS),
so run the script once per setting.
  python scripts/conv_microbench.py                       # the MFMA layers of the three nets at batch 8, n_pix 1024
  python scripts/conv_microbench.py B L Cin Cout k stride  # one shape"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gennet_amd import ops  # noqa: E402
from gennet_amd.engine import device  # noqa: E402

# (name, B, L, Cin, Cout, k, stride, pad_left, Lout) at batch 8, n_pix 1024
SHAPES = [
    ('G conv 128->256', 8, 1024, 128, 256, 5, 1, 2, 1024), ('G conv 256->512', 8, 1024, 256, 512, 5, 1, 2, 1024), ('G conv 512->1024', 8, 1024, 512, 1024, 5, 1, 2, 1024),
    ('D conv2 folded (2B)', 16, 512, 512, 1024, 5, 2, 1, 256), ('PE q 128->256', 8, 1020, 128, 256, 5, 1, 0, 1016), ('PE q 256->512 s2', 8, 1016, 256, 512, 5, 2, 0, 506),
    ('PE q 512->1024 s2', 8, 506, 512, 1024, 5, 2, 0, 251), ('PE mc 256->512 s2', 8, 125, 256, 512, 5, 2, 0, 61),
]


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    shapes = SHAPES
    if len(sys.argv) == 7:
        B, L, Cin, Cout, k, s = [int(v) for v in sys.argv[1:]]
        shapes = [('arg', B, L, Cin, Cout, k, s, 0, (L - k) // s + 1)]
    dev = device()
    print('%-24s %10s %10s %10s   (us per launch | TFLOP/s)' % ('layer', 'fwd', 'dgrad', 'wgrad'))
    for name, B, L, Cin, Cout, k, stride, pl, Lout in shapes:
        x = ops.fill_normal((B, L, Cin), 0.0, 1.0, 1, 0, dev)
        w = ops.fill_uniform((k, Cin, Cout), -0.05, 0.05, 2, 0, dev)
        b = torch.zeros(Cout, device=dev)
        y = ops.conv1d_fwd(x, w, b, stride, pl, Lout)
        wt = ops.conv1d_transpose_w(w)
        flop = 2.0 * B * Lout * k * Cin * Cout
        tf = timeit(lambda: ops.conv1d_fwd(x, w, b, stride, pl, Lout, 'relu'))
        td = timeit(lambda: ops.conv1d_dgrad(y, wt, L, stride, pl))
        tw = timeit(lambda: ops.conv1d_wgrad(x, y, k, stride, pl))
        print('%-24s %5.0f|%5.1f %5.0f|%5.1f %5.0f|%5.1f' % (name, tf * 1e6, flop / tf / 1e12, td * 1e6, flop / td / 1e12, tw * 1e6, flop / tw / 1e12), flush=True)


if __name__ == '__main__':
    main()
