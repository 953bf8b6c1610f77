#!/usr/bin/env python
"""Workload for rocprofv3 at the reference's own operating point (batch 8, n_pix 1024): N eager CNN steps and N eager GAN iterations.
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b8 -o b8 -- python3 scripts/small_batch_profile.py 1024 8 20"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gennet_amd import bbh, engine, ops  # noqa: E402


def main():
    n_pix = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    which = sys.argv[4] if len(sys.argv) > 4 else 'both'
    dev = engine.device()
    engine.set_init_seed(1); engine.set_device_seed(1); random.seed(1); np.random.seed(1)
    images = ops.fill_normal((5000, n_pix), 0.0, 1.0, 77, 0, dev)
    pars = torch.stack([ops.fill_uniform((5000,), 20.0, 35.0, 78, 0, dev), ops.fill_uniform((5000,), 0.5, 1.0, 79, 0, dev)], dim=1).contiguous()
    bank = bbh.DeviceBank(images, pars)
    ev = np.random.RandomState(5).randn(n_pix, 1).astype(np.float32)
    nets = bbh.build_and_compile(ev, n_pix)
    event = engine.to_device(ev.reshape(-1))
    for _ in range(n):
        if which in ('both', 'cnn'):
            bbh.pe_train_step(nets.signal_pe, bank, B)
        if which in ('both', 'gan'):
            bbh.gan_train_step(nets, bank, event, B)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
