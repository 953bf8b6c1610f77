#!/usr/bin/env python
"""Step latency at the reference's own defaults (batch_size = pe_batch_size = 8, n_pix = 1024: bbhMahoGANy.py:100-107), where the
loops are launch-bound rather than MFMA-bound: wall time per CNN train step and per GAN iteration, eager (one ctypes call per kernel) and as
replayed hipGraphs (bbh.GraphedPEStep / GraphedGANStep), with and without the per-step device -> host read of the losses."""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gennet_amd import bbh, engine, ops  # noqa: E402


def main():
    n_pix = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = engine.device()
    engine.set_init_seed(1); engine.set_device_seed(1); random.seed(1); np.random.seed(1)
    images = ops.fill_normal((5000, n_pix), 0.0, 1.0, 77, 0, dev)
    pars = torch.stack([ops.fill_uniform((5000,), 20.0, 35.0, 78, 0, dev), ops.fill_uniform((5000,), 0.5, 1.0, 79, 0, dev)], dim=1).contiguous()
    bank = bbh.DeviceBank(images, pars)
    ev = np.random.RandomState(5).randn(n_pix, 1).astype(np.float32)
    nets = bbh.build_and_compile(ev, n_pix)
    event = engine.to_device(ev.reshape(-1))
    gpe, ggan = bbh.GraphedPEStep(nets.signal_pe, bank, B), bbh.GraphedGANStep(nets, bank, event, B)
    rows = (('CNN train_on_batch, eager', lambda: bbh.pe_train_step(nets.signal_pe, bank, B)),
            ('GAN iteration, eager', lambda: bbh.gan_train_step(nets, bank, event, B)),
            ('CNN train_on_batch, hipGraph replay', gpe),
            ('GAN iteration, hipGraph replay', ggan),
            ('CNN train_on_batch, hipGraph replay, losses not read', lambda: gpe(want_losses=False)),
            ('GAN iteration, hipGraph replay, losses not read', lambda: ggan(want_losses=False)))
    for name, fn in rows:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print('%-55s n_pix=%d batch=%d: %.3f ms per step, %.0f waveforms/s' % (name, n_pix, B, 1e3 * dt, B / dt), flush=True)


if __name__ == '__main__':
    main()
