#!/usr/bin/env python
"""The CNN half of DESIGN 6a's question, asked the same way as gan_dynamics_cpu.py asks the GAN half: does an INDEPENDENT implementation
(oracle/torch_ref.PENet: torch-CPU autograd) trained with the reference's loop body (bbhMahoGANy.py:1153-1168: batch 8, lr 9e-5, noise N(0, sigma),
sigma ~ U(0, 5), on the first int(B / 8) rows) on the SAME bank reach the same held-out error as the HIP path (cnn_dynamics_gpu.py)?
Logs the mean |error| in (mc, q) on 1000 held-out templates every --log steps.  TEST INFRASTRUCTURE (imports oracle/); CPU only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gan_dynamics_common as C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n-pix', type=int, default=256)
    ap.add_argument('--bank', type=int, default=11000)
    ap.add_argument('--heldout', type=int, default=1000)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=20000)
    ap.add_argument('--log', type=int, default=2000)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--out', default='gpurun_out/dyn/cnn_cpu.json')
    a = ap.parse_args()
    import random
    import torch
    from oracle import torch_ref as T
    if a.threads:
        torch.set_num_threads(a.threads)
    prob = C.make_problem(a.n_pix, a.bank, a.seed)
    n_train = a.bank - a.heldout
    bank = torch.as_tensor(prob['bank'][:n_train]); pars = torch.as_tensor(prob['pars'][:n_train])
    hx = torch.as_tensor(prob['bank'][n_train:]).reshape(-1, a.n_pix, 1); hy = prob['pars'][n_train:].astype(np.float64)
    torch.manual_seed(a.seed); random.seed(a.seed); np.random.seed(a.seed)
    pe = T.PENet(a.n_pix, seed=1 + a.seed)
    out = {'implementation': 'oracle/torch_ref.PENet (torch %s CPU autograd, %d threads)' % (torch.__version__, torch.get_num_threads()),
           'config': {k: getattr(a, k) for k in ('n_pix', 'bank', 'heldout', 'batch', 'steps', 'seed')}, 'prior_std [mc, q]': [float(hy[:, 0].std()), float(hy[:, 1].std())],
           'trajectory': []}
    t0 = time.time()
    n_noisy = int(a.batch / 8)
    for step in range(a.steps + 1):
        idx = random.sample(range(n_train), a.batch)
        x = bank[idx].clone()
        sigma = float(np.random.uniform(0, 5))
        if n_noisy > 0:
            x[:n_noisy] += sigma * torch.randn(n_noisy, a.n_pix)
        r = pe.train_on_batch(x.reshape(a.batch, a.n_pix, 1), pars[idx, 0], pars[idx, 1])
        if step % a.log == 0:
            with torch.no_grad():
                pm, pq = pe.forward(hx)
            e = [float(np.abs(hy[:, 0] - pm.numpy().reshape(-1)).mean()), float(np.abs(hy[:, 1] - pq.numpy().reshape(-1)).mean())]
            rec = {'step': step, 'loss [total, mc, q]': r, 'mean_abs_error_heldout [mc, q]': e, 'seconds': time.time() - t0}
            out['trajectory'].append(rec)
            print(json.dumps(rec), flush=True)
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            with open(a.out + '.tmp', 'w') as fh:
                json.dump(out, fh, indent=1)
            os.replace(a.out + '.tmp', a.out)


if __name__ == '__main__':
    main()
