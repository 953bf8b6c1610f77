#!/usr/bin/env python
"""The HIP path's GAN trajectory on the problem of gan_dynamics_common.make_problem (same bank, same event as gan_dynamics_cpu.py), logged the
same way: VERDICT r3 next-round item 1a/1c.  The loop body is the product's own bbh.gan_train_step (bbhMahoGANy.py:1241-1299); the two
diagnostic switches are scripts/validate_posterior.build_nets'.  TEST INFRASTRUCTURE (imports oracle/ through the common module).

    python tests/tools/gan_dynamics_gpu.py --iters 15000 --out gpurun_out/dyn/gpu_default.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'scripts'))
import gan_dynamics_common as C  # noqa: E402


def run(a):
    import random
    import torch
    import validate_posterior as V
    from gennet_amd import bbh, engine, ops
    prob = C.make_problem(a.n_pix, a.bank, a.seed)
    engine.set_init_seed(2 + a.seed); engine.set_device_seed(1000 + a.seed)
    random.seed(a.seed); np.random.seed(a.seed)
    n = a.n_pix
    nets = V.build_nets(prob['event'].reshape(n, 1), n, lr=a.lr, moving_average=a.moving_average, bce_on_logit=a.bce_on_logit, do_pe=False)
    bank = bbh.DeviceBank(prob['bank'], prob['pars'])
    ev = engine.to_device(prob['event'])
    out = {'implementation': 'gennet_amd HIP path (bbh.gan_train_step, eager)',
           'config': {'n_pix': n, 'bank': a.bank, 'batch': a.batch, 'iters': a.iters, 'lr': a.lr, 'seed': a.seed, 'moving_average': a.moving_average,
                      'bce_grad': 'on_logit (diagnostic)' if a.bce_on_logit else 'clip'},
           'event': {'optimal_snr': prob['snr'], 'template_scale_g': prob['g']}, 'trajectory': []}
    win = []
    t0 = time.time()
    real64 = bank.images[:64].contiguous()
    for it in range(a.iters + 1):
        r = bbh.gan_train_step(nets, bank, ev, a.batch)
        win.append(r)
        if it % a.log == 0:
            seed, off = engine.device_rng().take(256 * 100)
            z = ops.fill_uniform((256, 100), -1.0, 1.0, seed, off, engine.device())
            w = nets.generator.predict_device(z, batch_size=64)
            seed, off = engine.device_rng().take(64 * n)
            noise = ops.fill_normal((64, n, 1), 0.0, 1.0, seed, off, engine.device())
            sX, _ = bbh.assemble_discriminator_batch(real64, noise, w[:64].contiguous(), ev)
            pd = nets.signal_discriminator.predict_device(sX, batch_size=64).reshape(-1).double().cpu().numpy()
            if not a.bce_on_logit:
                pc = np.clip(pd, 1e-30, 1.0)
                pd = np.log(pc) - np.log1p(-np.minimum(pc, 1 - 1e-16))           # logit of the fp32 probability (+-inf-safe: saturates near +-36)
            lr_, lf = pd[:64], pd[64:]
            ov, rms = C.waveform_stats(w.cpu().numpy(), prob['clean'])
            m = np.mean(win, axis=0)
            rec = {'iteration': it, 'sg_loss': r[0], 'sd_loss': r[2], 'sg_acc': r[1], 'sd_acc': r[3],
                   'window_mean [sg_loss, sd_loss, sg_acc, sd_acc]': [float(m[0]), float(m[2]), float(m[1]), float(m[3])],
                   'waveform_rms': rms, 'waveform_overlap_with_clean_event': ov,
                   'D_logit_on_fake [mean, min, max] (inference phase)': [float(lf.mean()), float(lf.min()), float(lf.max())],
                   'D_logit_on_real [mean, min, max] (inference phase)': [float(lr_.mean()), float(lr_.min()), float(lr_.max())],
                   'seconds': time.time() - t0}
            out['trajectory'].append(rec)
            print(json.dumps(rec), flush=True)
            win = []
    torch.cuda.synchronize()
    out['seconds'] = time.time() - t0
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, 'w') as fh:
            json.dump(out, fh, indent=1)
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--n-pix', type=int, default=256)
    ap.add_argument('--bank', type=int, default=2000)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=15000)
    ap.add_argument('--log', type=int, default=500)
    ap.add_argument('--lr', type=float, default=9e-5)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--moving-average', default='tf_zero_debias', choices=('tf_zero_debias', 'ema'))
    ap.add_argument('--bce-on-logit', action='store_true')
    ap.add_argument('--out', default='gpurun_out/dyn/gpu_default.json')
    return ap.parse_args(argv)


if __name__ == '__main__':
    run(parse())
