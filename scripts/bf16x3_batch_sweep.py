#!/usr/bin/env python
"""Opt-in bf16-split convolution (csrc/conv_bf16x3.hip) at the bench's batch sizes: exact-fp32 kernel, split kernel alone, split kernel + its split passes.
    python scripts/bf16x3_batch_sweep.py [--short-taps]       (GN_BF16X3_NARROW=1: round 1's 64 x 32 wave tiles)
--short-taps: the 3- and 2-tap launches a stride-2 data gradient consists of (profiles/r04_bf16x3_short_taps.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gennet_amd import ops
dev = torch.device('cuda:0')
SHORT = '--short-taps' in sys.argv
CASES = ((512, 1024, 1024, 512, 3), (512, 1024, 1024, 512, 2), (512, 1024, 512, 256, 3), (512, 1024, 512, 256, 2)) if SHORT else \
        ((64, 2048, 512, 1024, 5), (256, 2048, 512, 1024, 5), (512, 2048, 512, 1024, 5), (512, 2048, 256, 512, 5), (512, 2048, 1024, 512, 5))
for (B, L, Cin, Cout, k) in CASES:
    x = torch.randn(B, L, Cin, device=dev); w = torch.randn(k, Cin, Cout, device=dev) * 0.02; b = torch.zeros(Cout, device=dev)
    Lout, pl = ops.conv_geometry(L, k, 1, 'same')
    flop = 2.0 * B * Lout * k * Cin * Cout
    for name, fn in (('fp32-mfma', lambda: ops.conv1d_fwd(x, w, b, 1, pl, Lout, 'relu')),
                     ('bf16x3 (conv only)', lambda: ops.conv1d_fwd_bf16x3(x, w, b, 1, pl, Lout, 'relu', resplit=False)),
                     ('bf16x3 (+split)', lambda: ops.conv1d_fwd_bf16x3(x, w, b, 1, pl, Lout, 'relu', resplit=True))):
        ops.conv1d_fwd_bf16x3(x, w, b, 1, pl, Lout, 'relu')
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print('k%d B%d L%d %d->%d  %-20s %.3f ms  %.1f TFLOP/s' % (k, B, L, Cin, Cout, name, ms, flop / ms / 1e9), flush=True)
    del x, w
