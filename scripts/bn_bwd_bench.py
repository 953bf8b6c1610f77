#!/usr/bin/env python
"""Times the generator's last BatchNormalization backward at the BASELINE size (B 512, L 2048, C 1024) both ways: the output conv's data
gradient materialised (gn_conv1d_dgrad + gn_bn_bwd_stats + gn_bn_bwd_apply) and formed on the fly (gn_bn_bwd_*_conv1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from gennet_amd import ops
    B, L, C = int(os.environ.get('B', 512)), 2048, 1024
    act = os.environ.get('ACT', 'tanh')
    dev = torch.device('cuda:0')
    x = torch.randn(B * L, C, device=dev)
    mask = (torch.rand(B * L, C, device=dev) >= 0.2).to(torch.uint8)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
    sums = ops.bn_stats(x)
    scale, shift, smean, sinv = ops.bn_finalize(sums, B * L, gamma, beta, 1e-3, 0.99, mm, mv)
    g = torch.randn(B, L, 1, device=dev); w = torch.randn(5, C, 1, device=dev) * 0.1
    wt = ops.conv1d_transpose_w(w)
    dgamma = torch.empty(C, device=dev); dbeta = torch.empty(C, device=dev)

    def materialised():
        dz = ops.conv1d_dgrad(g, wt, L, 1, 2).reshape(B * L, C)
        ds = ops.bn_bwd_stats(dz, None, x, mask, smean, sinv, act, 0.0, 0.2, scale, shift)
        return ops.bn_bwd_apply(dz, None, x, mask, gamma, smean, sinv, ds, B * L, ds, dgamma, dbeta, act, 0.0, 0.2, scale, shift)

    def lazy():
        cg = ops.ConvGrad1(g, w, L, 2)
        ds = ops.bn_bwd_stats_conv1(cg, x, mask, smean, sinv, act, 0.0, 0.2, scale, shift)
        return ops.bn_bwd_apply_conv1(cg, x, mask, gamma, smean, sinv, ds, B * L, ds, dgamma, dbeta, act, 0.0, 0.2, scale, shift)

    for name, fn in (('materialised', materialised), ('on the fly', lazy)):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        print('%-13s %.3f ms' % (name, e0.elapsed_time(e1) / 5), flush=True)


if __name__ == '__main__':
    main()
