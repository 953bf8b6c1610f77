#!/usr/bin/env python
"""Known-answer validation of the product's RESULT: does the CNN + GAN pipeline recover the posterior on (mc, q)?

The reference scores its own run every `cadence` iterations (bbhMahoGANy.py:1330-1356: 4000 generator draws -> signal_pe -> overlap with a
lalinference posterior through overlap_tests, :811-873), states its CNN accuracy yard-stick (`pe_std = [0.0219, 0.0057]`, :1345) and, in its
prototypes, validates against an EXACT grid posterior (tests/burstMahoGANy.py:716-725).  lalinference cannot run here, but for this
project's own signal model the exact answer is computable: with data d = g h(mc, q, idx) + n, n ~ N(0, 1) white (what the loops add,
:1161, :1277), the hunt_constrain prior p(mc, q) ~ 1 / (mc q) on the box (log-uniform component masses, gw_template_maker.py:327-339) and idx
uniform on the convert_beta window,

    p(mc, q | d)  ~  p(mc, q)  sum_idx  exp( d . g h(mc, q, idx)  -  |g h(mc, q, idx)|^2 / 2 )

evaluated on a (mc, q) grid with the SAME synthesiser that makes the training templates (templates.Synth).  The script

  1. builds the bank (n_pix 1024, the script's default; 50 000 templates as gw_template_maker.py:60) and the event = held-out (36, 29)
     template + N(0, 1);
  2. trains the CNN point-estimator with the reference's loop body and reports mean |error| on 4000 held-out templates next to pe_std;
  3. trains the GAN with the reference's loop body; at every cadence draws 4000 posterior samples (bbh.posterior_samples) and scores them
     against samples of the exact posterior with overlap_tests' beta and the KS tests;
  4. writes everything to --out (profiles/r03_posterior_validation.json).

The likelihood sums and the posterior sampling below are the CHECKER (torch reductions on the device, numpy on the host); the pipeline
under test is the package's own path.  One GPU call: the budget is split by --cnn-seconds / --gan-seconds.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PE_STD_REFERENCE = [0.02185649964844209, 0.005701401364171313]        # bbhMahoGANy.py:1345


def exact_posterior(syn, d, g, lo, hi, n_mc=300, n_q=100, chunk=8192):
    """Grid posterior on (mc, q) in [20, 35] x [0.5, 1], idx marginalised over [lo, hi).  Returns (mc axis, q axis, normalised pmf (n_mc, n_q))."""
    import torch
    from gennet_amd import templates as T
    mc = 20.0 + (np.arange(n_mc) + 0.5) * (15.0 / n_mc)
    q = 0.5 + (np.arange(n_q) + 0.5) * (0.5 / n_q)
    MC, Q = np.meshgrid(mc, q, indexing='ij')
    m1, m2 = T.m1m2_from_mc_q(MC.reshape(-1), Q.reshape(-1))
    ok = (m1 + m2 < 100.0) & (m1 > 5.0) & (m2 > 5.0) & (m1 <= 95.0)
    logprior = np.where(ok, -np.log(MC.reshape(-1)) - np.log(Q.reshape(-1)), -np.inf)
    dd = torch.as_tensor(np.asarray(d, np.float64)).to(syn.scale.device)
    npts = m1.size
    logl = torch.full((npts,), -float('inf'), dtype=torch.float64, device=dd.device)
    for idx in range(lo, hi):
        for s in range(0, npts, chunk):
            h, _ = syn.templates(m1[s:s + chunk], m2[s:s + chunk], np.full(min(chunk, npts - s), idx), g=g, dtype=torch.float64)
            ll = (h * dd).sum(1) - 0.5 * (h * h).sum(1)
            logl[s:s + chunk] = torch.logaddexp(logl[s:s + chunk], ll)
    lp = logl.cpu().numpy() + logprior
    lp -= lp.max()
    p = np.exp(lp)
    p /= p.sum()
    return mc, q, p.reshape(n_mc, n_q)


def sample_grid(mc, q, pmf, n, rng):
    """n samples of the grid posterior, uniform inside the chosen cell: (2, n)."""
    flat = rng.choice(pmf.size, size=n, p=pmf.reshape(-1))
    i, j = np.unravel_index(flat, pmf.shape)
    dmc, dq = mc[1] - mc[0], q[1] - q[0]
    return np.array([mc[i] + (rng.rand(n) - 0.5) * dmc, q[j] + (rng.rand(n) - 0.5) * dq])


def moments(s):
    return {'mc_mean': float(np.mean(s[0])), 'mc_std': float(np.std(s[0])), 'q_mean': float(np.mean(s[1])), 'q_std': float(np.std(s[1])),
            'corr': float(np.corrcoef(s[0], s[1])[0, 1]) if np.std(s[0]) > 0 and np.std(s[1]) > 0 else None}


def build_nets(event, fs, lr=9e-5, chi_loss=False, moving_average='tf_zero_debias', bce_on_logit=False, do_pe=True):
    """bbh.build_and_compile, plus the two DIAGNOSTIC switches of VERDICT r3 item 1c.  Neither is reachable from the product surface:
      moving_average 'ema'   every BatchNormalization built here uses the plain exponential moving average (tf.keras; layers.BatchNormalization
                             already offers it per layer) instead of the Keras-2.2.4 default 'tf_zero_debias';
      bce_on_logit           the discriminator's final sigmoid is made linear and binary cross-entropy is evaluated ON THE LOGIT
                             (loss = softplus(z) - z y, dL/dz = (sigmoid(z) - y) / B): no clip, and no fp32 sigmoid that rounds to exactly 0 / 1, so
                             the gradient is never zeroed.  Keras 2.2.4 cannot do this (binary_crossentropy from probabilities, Appendix B.8).  The
                             arithmetic is a few torch ops on the device, monkey-patched over ops.loss for THIS PROCESS only -- a checker's tool."""
    import torch
    from gennet_amd import bbh, layers, ops
    saved_ma, saved_d = layers.BN_MOVING_AVERAGE, bbh.signal_discriminator_model
    layers.BN_MOVING_AVERAGE = moving_average
    if bce_on_logit:
        def d_linear(n_pix=1024):
            m = saved_d(n_pix)
            assert m.layers[-1].act_spec[0] == 'sigmoid'
            m.layers[-1].act_spec = ('linear', 0.0)
            return m
        bbh.signal_discriminator_model = d_linear
        if not getattr(ops.loss, '_on_logit', False):
            plain = ops.loss

            def loss_on_logit(kind, p, y, Bglobal=None):
                if kind != 'binary_crossentropy':
                    return plain(kind, p, y, Bglobal)
                B = float(Bglobal or p.shape[0])
                per = torch.clamp(p, min=0) - p * y + torch.log1p(torch.exp(-p.abs()))
                return (torch.sigmoid(p) - y) / B, torch.stack([per.sum() / B, ((p > 0).float() == y).float().sum()])
            loss_on_logit._on_logit = True
            ops.loss = loss_on_logit
    try:
        return bbh.build_and_compile(event, fs, lr=lr, chi_loss=chi_loss, do_pe=do_pe)
    finally:
        layers.BN_MOVING_AVERAGE, bbh.signal_discriminator_model = saved_ma, saved_d


def rail_report(y_q, p_q, lo=0.0, hi=1.0):
    """ReLU(max_value=1) on the q head (bbhMahoGANy.py:400): fraction of predictions sitting exactly on a rail, and the mean |error| without them."""
    y_q, p_q = np.asarray(y_q, np.float64).reshape(-1), np.asarray(p_q, np.float64).reshape(-1)
    on0, on1 = p_q <= lo, p_q >= hi
    free = ~(on0 | on1)
    return {'fraction_at_0': float(on0.mean()), 'fraction_at_1': float(on1.mean()),
            'mean_abs_error_q_off_rail': float(np.abs(y_q - p_q)[free].mean()) if free.any() else None,
            'mean_abs_error_q_on_rail': float(np.abs(y_q - p_q)[~free].mean()) if (~free).any() else None,
            'true_q_mean_of_rows_at_1': float(y_q[on1].mean()) if on1.any() else None}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--n-pix', type=int, default=1024)
    ap.add_argument('--bank', type=int, default=50000)
    ap.add_argument('--snr', type=float, default=0.0, help='optimal SNR of the event-like template after scaling; 0: scale templates to unit variance '
                                                           '(the role of gw_norm_constant) and report the SNR that gives')
    ap.add_argument('--pe-batch', type=int, default=256)
    ap.add_argument('--pe-iter', type=int, default=16000, help='500 000 x 8 waveforms of the reference (bbhMahoGANy.py:86-89) = 15 625 steps of 256')
    ap.add_argument('--cnn-seconds', type=float, default=300.0)
    ap.add_argument('--pe-settle-steps', type=int, default=0, help='DIAGNOSTIC, not in the reference: after the CNN loop, this many more steps at lr / 10, '
                                                                     'and the held-out error again -- separates the optimiser\'s last-iterate noise from what the net has learnt')
    ap.add_argument('--gan-batch', type=int, default=32)
    ap.add_argument('--gan-iter', type=int, default=125000, help='500 000 x 8 waveforms of the reference = 125 000 iterations of 32')
    ap.add_argument('--gan-seconds', type=float, default=600.0)
    ap.add_argument('--cadence', type=int, default=5000)
    ap.add_argument('--fine-until', type=int, default=0, help='score every --fine-cadence iterations up to this iteration (the early trajectory)')
    ap.add_argument('--fine-cadence', type=int, default=500)
    ap.add_argument('--predict-batch', type=int, default=32, help='chunk of generator.predict inside the GAN iteration')
    ap.add_argument('--lr', type=float, default=9e-5)
    ap.add_argument('--moving-average', default='tf_zero_debias', choices=('tf_zero_debias', 'ema'), help='DIAGNOSTIC (build_nets): BatchNormalization moving statistics')
    ap.add_argument('--bce-on-logit', action='store_true', help='DIAGNOSTIC (build_nets): cross-entropy on the discriminator logit, gradient never zeroed; not a product mode')
    ap.add_argument('--chi-loss', action='store_true', help='chi_loss (bbhMahoGANy.py:97, :1106-1109): the generator trains on chisquare_Loss instead of binary cross-entropy')
    ap.add_argument('--save-pe', default='', help='write the trained CNN to this .h5 file (Keras layout)')
    ap.add_argument('--load-pe', default='', help='skip the CNN loop and load the CNN from this .h5 file')
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--graph', action='store_true', help='run the train steps as captured hipGraphs (engine.GraphedStep)')
    ap.add_argument('--out', default='gpurun_out/posterior_validation.json')
    return ap.parse_args(argv)


def run(args):
    """The whole validation for one configuration; returns the result dictionary (and writes it to args.out when that is set)."""
    import random
    import torch
    from gennet_amd import bbh, engine, ops, posterior, templates as T
    t_start = time.time()
    fs = args.n_pix
    N = 4 * fs
    engine.set_init_seed(args.seed); engine.set_device_seed(1000 + args.seed)
    random.seed(args.seed); np.random.seed(args.seed)
    f = np.arange(N // 2 + 1) * 0.25
    psd = 1e-46 * ((np.maximum(f, 10.0) / 150.0) ** -4.0 + 2.0 + 2.0 * (f / 150.0) ** 2.0)
    psd[f < 10.0] = 0.0
    syn = T.Synth(fs, 4, psd)
    lo, hi = T.convert_beta([0.45, 0.55], fs, 4)
    ev_raw, _ = syn.templates([36.0], [29.0], [N // 2], dtype=torch.float64)
    ob = T.OnlineBank(fs, 4, psd, seed=7 + args.seed, noise=None)
    probe, _ = ob.draw(4096)
    if args.snr > 0:
        g = args.snr / float(torch.sqrt((ev_raw ** 2).sum()))
    else:
        g = 1.0 / float(probe.std())
    ob.g = g
    snr = g * float(torch.sqrt((ev_raw ** 2).sum()))
    images, pars = ob.draw(args.bank)
    bank = bbh.DeviceBank(images, pars)
    held_x, held_y = ob.draw(4000)                                                       # held-out templates for the CNN read-out
    noise = np.random.RandomState(100 + args.seed).randn(fs)
    event = (ev_raw.cpu().numpy()[0] * g + noise).astype(np.float32)                    # d = g h + n
    truth = [float((36.0 * 29.0) ** 0.6 / 65.0 ** 0.2), 29.0 / 36.0]
    out = {'config': {k: getattr(args, k) for k in ('n_pix', 'bank', 'pe_batch', 'pe_iter', 'gan_batch', 'gan_iter', 'cadence', 'lr', 'seed', 'graph', 'chi_loss', 'load_pe', 'moving_average', 'bce_on_logit')},
           'event': {'m1': 36.0, 'm2': 29.0, 'idx': N // 2, 'mc': truth[0], 'q': truth[1], 'optimal_snr': snr, 'template_scale_g': g,
                     'noise': 'N(0,1), RandomState(%d)' % (100 + args.seed)},
           'reference_yardstick_pe_std': PE_STD_REFERENCE}
    print('event: mc %.3f q %.3f optimal SNR %.2f (scale %.4g); bank %d x %d' % (truth[0], truth[1], snr, g, bank.n, fs), flush=True)

    # ---- exact posterior (the known answer)
    t0 = time.time()
    mc_ax, q_ax, pmf = exact_posterior(syn, event.astype(np.float64), g, lo, hi)
    rng = np.random.RandomState(5)
    exact = sample_grid(mc_ax, q_ax, pmf, 3907, rng)                                     # as many samples as the lalinference posterior had (:61)
    out['exact_posterior'] = dict(moments(exact), grid=[len(mc_ax), len(q_ax)], idx_window=[lo, hi], seconds=time.time() - t0,
                                  map_mc=float(mc_ax[np.unravel_index(pmf.argmax(), pmf.shape)[0]]), map_q=float(q_ax[np.unravel_index(pmf.argmax(), pmf.shape)[1]]),
                                  self_overlap_beta=float(posterior.overlap_tests([exact[0].reshape(-1, 1), exact[1].reshape(-1, 1)],
                                                                                  sample_grid(mc_ax, q_ax, pmf, 4000, rng))[2]))
    print('exact posterior: %s (%.1f s)' % (json.dumps(out['exact_posterior']), time.time() - t0), flush=True)

    # ---- networks
    nets = build_nets(event.reshape(fs, 1), fs, lr=args.lr, chi_loss=args.chi_loss, moving_average=args.moving_average, bce_on_logit=args.bce_on_logit)
    ev_dev = engine.to_device(event)
    pe_step = gan_step = None
    if args.graph:
        pe_step = bbh.GraphedPEStep(nets.signal_pe, bank, args.pe_batch)
        gan_step = bbh.GraphedGANStep(nets, bank, ev_dev, args.gan_batch, predict_batch=args.predict_batch)

    # ---- CNN point-estimator: the reference's loop body (bbhMahoGANy.py:1153-1168)
    t0 = time.time()
    hist = []
    i = 0
    if args.load_pe:
        nets.signal_pe.load_weights(args.load_pe)
        args.pe_iter = 0
    while i < args.pe_iter and time.time() - t0 < args.cnn_seconds:
        r = pe_step(want_losses=(i % 1000 == 0)) if pe_step else bbh.pe_train_step(nets.signal_pe, bank, args.pe_batch)
        if i % 1000 == 0:
            hist.append([i] + [float(v) for v in r[:3]])
            print('cnn %6d: total %.4f mc %.4f q %.5f  (%.0f s)' % (i, r[0], r[1], r[2], time.time() - t0), flush=True)
        i += 1
    torch.cuda.synchronize()
    t_cnn = time.time() - t0
    p = nets.signal_pe.predict_device(held_x.reshape(-1, fs, 1), batch_size=256)
    hy = held_y.cpu().numpy().astype(np.float64)
    err = [np.abs(hy[:, k] - p[k].cpu().numpy().reshape(-1)) for k in range(2)]
    rms_tr, std_tr = bbh.pe_accuracy(nets.signal_pe, bank)
    out['cnn'] = {'steps': i, 'batch': args.pe_batch, 'waveforms': i * args.pe_batch, 'seconds': t_cnn, 'waveforms_per_s': i * args.pe_batch / t_cnn,
                  'mean_abs_error_heldout [mc, q]': [float(e.mean()) for e in err], 'median_abs_error_heldout [mc, q]': [float(np.median(e)) for e in err],
                  'mean_abs_error_training_4000 [mc, q] (the reference read-out, :1184-1196)': std_tr, 'mse_training_4000 [mc, q]': rms_tr,
                  'prior_std [mc, q]': [float(hy[:, 0].std()), float(hy[:, 1].std())],
                  'q_head_rails (ReLU(max_value=1), :400)': rail_report(hy[:, 1], p[1].cpu().numpy()), 'loss_history [step, total, mc, q]': hist}
    if args.save_pe:
        nets.signal_pe.save_weights(args.save_pe, True)
    if args.pe_settle_steps > 0:
        nets.signal_pe.optimizer.lr = float(np.float32(args.lr / 10.0))
        for _ in range(args.pe_settle_steps):
            bbh.pe_train_step(nets.signal_pe, bank, args.pe_batch)
        p2 = nets.signal_pe.predict_device(held_x.reshape(-1, fs, 1), batch_size=256)
        err2 = [np.abs(hy[:, k] - p2[k].cpu().numpy().reshape(-1)) for k in range(2)]
        out['cnn']['diagnostic_after_settling'] = {'extra_steps': args.pe_settle_steps, 'lr': args.lr / 10.0,
                                                   'mean_abs_error_heldout [mc, q]': [float(e.mean()) for e in err2],
                                                   'over_reference_pe_std [mc, q]': [float(err2[k].mean()) / PE_STD_REFERENCE[k] for k in range(2)]}
    print('cnn done: %s' % json.dumps({k: v for k, v in out['cnn'].items() if 'history' not in k}), flush=True)

    # ---- GAN: the reference's loop body (:1241-1299), scored at every cadence (:1330-1356)
    def score(tag):
        pe_s, waves = bbh.posterior_samples(nets, 4000, predict_batch=500)
        s = np.array([pe_s[0].reshape(-1), pe_s[1].reshape(-1)], np.float64)
        rec = dict(moments(s), iteration=tag)
        ov = bbh.posterior_overlap(pe_s, exact)
        if ov is not None:
            ks, ad, beta = ov
            rec.update(beta=float(beta), ks_stat=[float(ks[0][0]), float(ks[1][0])], ks_p=[float(ks[0][1]), float(ks[1][1])])
        w = waves.reshape(4000, fs)
        rec['waveform_overlap_with_clean_event'] = float(np.mean((w @ (ev_raw.cpu().numpy()[0] * g)) / (np.linalg.norm(w, axis=1) * snr + 1e-30)))
        rec['waveform_rms'] = float(np.sqrt(np.mean(w ** 2)))
        return rec
    t0 = time.time()
    traj = []
    it = 0
    while it < args.gan_iter and time.time() - t0 < args.gan_seconds:
        at_cadence = it % args.cadence == 0 or (it < args.fine_until and it % args.fine_cadence == 0)
        r = gan_step(want_losses=at_cadence) if gan_step else bbh.gan_train_step(nets, bank, ev_dev, args.gan_batch, predict_batch=args.predict_batch)
        if at_cadence:
            rec = score(it)
            rec.update(sg_loss=float(r[0]), sg_acc=float(r[1]), sd_loss=float(r[2]), sd_acc=float(r[3]), seconds=time.time() - t0)
            traj.append(rec)
            print('gan %s' % json.dumps(rec), flush=True)
        it += 1
    torch.cuda.synchronize()
    t_gan = time.time() - t0
    final = score(it)
    out['gan'] = {'iterations': it, 'batch': args.gan_batch, 'waveforms': it * args.gan_batch, 'seconds': t_gan, 'iterations_per_s': it / t_gan,
                  'final': final, 'best_beta': max([r.get('beta', 0.0) for r in traj + [final]]),
                  'best_beta_iteration': max(traj + [final], key=lambda r: r.get('beta', 0.0))['iteration'],
                  'best_waveform_overlap': max([r['waveform_overlap_with_clean_event'] for r in traj + [final]]), 'trajectory': traj}
    out['total_seconds'] = time.time() - t_start
    # tolerance statement: the CNN against the reference's yard-stick; the GAN posterior against the exact one
    ex = out['exact_posterior']
    out['verdict'] = {
        'cnn_mean_abs_error_vs_pe_std': [out['cnn']['mean_abs_error_heldout [mc, q]'][k] / PE_STD_REFERENCE[k] for k in range(2)],
        'gan_mean_offset_in_exact_sigmas [mc, q]': [(final['mc_mean'] - ex['mc_mean']) / ex['mc_std'], (final['q_mean'] - ex['q_mean']) / ex['q_std']],
        'gan_width_ratio [mc, q]': [final['mc_std'] / ex['mc_std'], final['q_std'] / ex['q_std']],
        'beta_final': final.get('beta'), 'beta_of_two_exact_sample_sets': ex['self_overlap_beta']}
    print('verdict: %s' % json.dumps(out['verdict']), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as fh:
            json.dump(out, fh, indent=1)
        print('wrote', args.out)
    return out


if __name__ == '__main__':
    run(parse())
