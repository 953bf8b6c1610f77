#!/usr/bin/env python
"""The HIP path's CNN trajectory on the problem of cnn_dynamics_cpu.py (same bank, same held-out set, same loop body = the product's own
bbh.pe_train_step): mean |error| in (mc, q) on the held-out templates every --log steps.  TEST INFRASTRUCTURE.
    python tests/tools/cnn_dynamics_gpu.py --steps 20000 --out gpurun_out/dyn/cnn_gpu.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
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
    ap.add_argument('--out', default='gpurun_out/dyn/cnn_gpu.json')
    a = ap.parse_args()
    import random
    import torch
    from gennet_amd import bbh, engine
    prob = C.make_problem(a.n_pix, a.bank, a.seed)
    n_train = a.bank - a.heldout
    engine.set_init_seed(1 + a.seed); engine.set_device_seed(1000 + a.seed)
    random.seed(a.seed); np.random.seed(a.seed)
    bank = bbh.DeviceBank(prob['bank'][:n_train], prob['pars'][:n_train])
    hx = engine.to_device(prob['bank'][n_train:]).reshape(-1, a.n_pix, 1); hy = prob['pars'][n_train:].astype(np.float64)
    pe = bbh.signal_pe_model(a.n_pix)
    pe.compile(loss='mean_squared_error', optimizer=engine.Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])
    out = {'implementation': 'gennet_amd HIP path (bbh.pe_train_step, eager)', 'config': {k: getattr(a, k) for k in ('n_pix', 'bank', 'heldout', 'batch', 'steps', 'seed')},
           'prior_std [mc, q]': [float(hy[:, 0].std()), float(hy[:, 1].std())], 'trajectory': []}
    t0 = time.time()
    for step in range(a.steps + 1):
        r = bbh.pe_train_step(pe, bank, a.batch)
        if step % a.log == 0:
            p = pe.predict_device(hx, batch_size=250)
            e = [float(np.abs(hy[:, k] - p[k].cpu().numpy().reshape(-1)).mean()) for k in range(2)]
            rec = {'step': step, 'loss [total, mc, q]': [float(v) for v in r[:3]], 'mean_abs_error_heldout [mc, q]': e, 'seconds': time.time() - t0}
            out['trajectory'].append(rec)
            print(json.dumps(rec), flush=True)
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as fh:
        json.dump(out, fh, indent=1)


if __name__ == '__main__':
    main()
