#!/usr/bin/env python
"""Opt-in bf16-split weight gradient (csrc/wgrad_bf16x3.hip) against the exact kernel: whole call (split passes + kernel + reduce + bias gradient) and the
MFMA kernel alone.    python scripts/bf16x3_wgrad_sweep.py [--s2]        (GN_WGBF_ABL=1|2|4: timing ablations; profiles/r04_bf16x3_wgrad.txt)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gennet_amd import ops
dev = torch.device('cuda:0')
import sys as _s
ST = 2 if '--s2' in _s.argv else 1
for (B, L, Cin, Cout) in (((256, 2048, 512, 1024), (256, 2048, 256, 512)) if ST == 2 else ((128, 2048, 512, 1024), (128, 2048, 256, 512))):
    x = torch.randn(B, L, Cin, device=dev); dy = torch.randn(B, L // ST, Cout, device=dev)
    flop = 2.0 * B * (L // ST) * 5 * Cin * Cout
    for name in ('fp32', 'bf16x3'):
        ops.set_conv_math(name, workspace_gb=8) if name != 'fp32' else ops.set_conv_math('fp32')
        ops.conv1d_wgrad(x, dy, 5, ST, 2 if ST == 1 else 1); torch.cuda.synchronize()
        ops.prof_enable(True); ops.prof_reset()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.conv1d_wgrad(x, dy, 5, ST, 2 if ST == 1 else 1)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        k = ops.prof_collect(2 if name != 'fp32' else 1)
        ops.prof_enable(False)
        if not k['launches']:
            print('s%d wgrad B%d L%d %d->%d  %-8s not taken (below the size threshold): %.3f ms on the exact kernel' % (ST, B, L, Cin, Cout, name, ms), flush=True)
            continue
        km = k['ms'] / k['launches']
        print('s%d ' % ST + 'wgrad B%d L%d %d->%d  %-8s %.3f ms whole (%.1f TFLOP/s), kernel alone %.3f ms (%.1f TFLOP/s)' % (B, L, Cin, Cout, name, ms, flop / ms / 1e9, km, flop / km / 1e9), flush=True)
    ops.set_conv_math('fp32')
    del x, dy
