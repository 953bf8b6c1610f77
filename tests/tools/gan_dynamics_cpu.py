#!/usr/bin/env python
"""Does an INDEPENDENT implementation reproduce the GAN collapse of DESIGN 6a?  

Runs the reference's GAN loop body (bbhMahoGANy.py:1241-1299) with the torch-autograd CPU port (oracle/torch_ref.GAN: torch's own autograd and
oneDNN kernels, no line shared with gennet_amd's engine or HIP kernels) at n_pix 256, batch 8, lr 9e-5 on the problem of
gan_dynamics_common.make_problem, and logs every --log iterations: sd_loss / sg_loss / sd_acc / sg_acc of the last iteration and their means over
the window, the rms of generator.predict on 256 latent draws and its overlap with the clean event, the discriminator's logits on real / fake rows.
The HIP path's trajectory on the same problem comes from gan_dynamics_gpu.py.

Options are the two recollected (dagger) semantics the outcome could hinge on:
  --moving-average tf_zero_debias | ema      BatchNormalization's moving statistics
  --bce-grad clip | noclip                   Keras 2.2.4's binary cross-entropy from clipped probabilities | DIAGNOSTIC: cross-entropy on the logit

TEST INFRASTRUCTURE (imports oracle/).  CPU only; ~0.3-0.6 s per iteration on 8 threads at n_pix 256.
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
    ap.add_argument('--bank', type=int, default=2000)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=15000)
    ap.add_argument('--log', type=int, default=500)
    ap.add_argument('--lr', type=float, default=9e-5)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--moving-average', default='tf_zero_debias', choices=('tf_zero_debias', 'ema'))
    ap.add_argument('--bce-grad', default='clip', choices=('clip', 'noclip'))
    ap.add_argument('--out', default='gpurun_out/gan_dynamics_cpu.json')
    a = ap.parse_args()
    import random
    import torch
    from oracle import torch_ref as T
    if a.threads:
        torch.set_num_threads(a.threads)
    prob = C.make_problem(a.n_pix, a.bank, a.seed)
    torch.manual_seed(a.seed)
    random.seed(a.seed)
    gan = T.GAN(a.n_pix, prob['event'], seed=2 + a.seed, lr=a.lr, moving_average=a.moving_average, bce_grad=a.bce_grad)
    bank = torch.as_tensor(prob['bank'])
    ev = torch.as_tensor(prob['event']).reshape(1, a.n_pix, 1)
    out = {'implementation': 'oracle/torch_ref.GAN (torch %s CPU autograd, %d threads)' % (torch.__version__, torch.get_num_threads()),
           'config': {k: getattr(a, k.replace('-', '_')) for k in ('n_pix', 'bank', 'batch', 'iters', 'lr', 'seed', 'moving_average', 'bce_grad')},
           'event': {'optimal_snr': prob['snr'], 'template_scale_g': prob['g']}, 'trajectory': []}
    win = []
    t0 = time.time()
    for it in range(a.iters + 1):
        idx = random.sample(range(a.bank), a.batch)                      # bbhMahoGANy.py:1244
        sg, sd = gan.iteration(bank[idx], a.batch)
        win.append([sg, sd] + gan.last_acc)
        if it % a.log == 0:
            with torch.no_grad():
                w = gan.G(torch.rand(256, 100) * 2 - 1, False)
                img_f = torch.stack([w[:64], ev - w[:64]], dim=2)
                img_r = torch.stack([bank[:64].reshape(64, a.n_pix, 1), torch.randn(64, a.n_pix, 1)], dim=2)
                lf = gan.D_logit(img_f, False).reshape(-1)
                lr_ = gan.D_logit(img_r, False).reshape(-1)
            ov, rms = C.waveform_stats(w.numpy(), prob['clean'])
            m = np.mean(win, axis=0)
            rec = {'iteration': it, 'sg_loss': sg, 'sd_loss': sd, 'sg_acc': gan.last_acc[0], 'sd_acc': gan.last_acc[1],
                   'window_mean [sg_loss, sd_loss, sg_acc, sd_acc]': [float(v) for v in m], 'waveform_rms': rms, 'waveform_overlap_with_clean_event': ov,
                   'D_logit_on_fake [mean, min, max] (inference phase)': [float(lf.mean()), float(lf.min()), float(lf.max())],
                   'D_logit_on_real [mean, min, max] (inference phase)': [float(lr_.mean()), float(lr_.min()), float(lr_.max())],
                   'seconds': time.time() - t0}
            out['trajectory'].append(rec)
            print(json.dumps(rec), flush=True)
            win = []
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            with open(a.out + '.tmp', 'w') as fh:
                json.dump(out, fh, indent=1)
            os.replace(a.out + '.tmp', a.out)


if __name__ == '__main__':
    main()
