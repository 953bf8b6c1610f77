#!/usr/bin/env python
"""Counterpart of bbhMahoGANy.main() (bbhMahoGANy.py:959-1382) on the gennet_amd engine: load the ts/pars template files,
train the CNN point-estimator, then the generator/discriminator pair, periodically pushing generator draws through the CNN
and pickling the posterior samples.  File names, data preparation and loop bodies follow the reference; the plotting /
KDE-overlap reporting block (:541-957, :1176-1231, :1302-1359) is host-side reporting and is out of scope.

Single GPU:  python scripts/bbh_train.py --templates templates/ --tag _srate-1024hz_oversamp --n-pix 1024
8 GPUs:      python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bbh_train.py ...
             (batch sizes are per GPU; the global batch is sharded as in SURVEY 8e)
"""
import argparse
import os
import pickle
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--templates', default='templates/')
    ap.add_argument('--event-name', default='gw150914')
    ap.add_argument('--training-num', type=int, default=50000)
    ap.add_argument('--tag', default='_srate-1024hz_oversamp')
    ap.add_argument('--n-pix', type=int, default=1024)
    ap.add_argument('--batch-size', type=int, default=8)
    ap.add_argument('--pe-batch-size', type=int, default=8)
    ap.add_argument('--pe-iter', type=int, default=500000)
    ap.add_argument('--max-iter', type=int, default=500000)
    ap.add_argument('--cadence', type=int, default=100)
    ap.add_argument('--lr', type=float, default=9e-5)
    ap.add_argument('--event-scale', type=float, default=817.98, help='bbhMahoGANy.py:1028-1029 scales the event by this literal')
    ap.add_argument('--out', default='.')
    ap.add_argument('--chi-loss', action='store_true', help='chi_loss (:97): the generator trains on chisquare_Loss (:146-162) instead of binary cross-entropy')
    ap.add_argument('--n-sig', type=float, default=1.0, help='n_sig (:85), the noise standard deviation in chisquare_Loss')
    ap.add_argument('--cnn-noise-frac', type=float, default=1.0 / 8.0, help='cnn_noise_frac (:113): fraction of each CNN batch that gets noise added')
    ap.add_argument('--retrain-pe-mod', action='store_true', help='retrain_pe_mod (:104, :1145): load best_models/signal_pe.h5 and keep training it')
    ap.add_argument('--n-noise-real', type=int, default=1, help='noise realisations per sampled template in the GAN loop (:107)')
    ap.add_argument('--graph', action='store_true', help='replay the two loop bodies as captured hipGraphs (single GPU, n_noise_real 1): the same numbers bit for '
                                                       'bit, no per-iteration host synchronisation between the read-outs')
    ap.add_argument('--pe-cadence', type=int, default=1000, help='CNN progress read-out every so many iterations (:1176, :1200)')
    ap.add_argument('--old-model', action='store_true', help='do_old_model (:1133-1138): start all four networks from the files of an earlier run in --out')
    ap.add_argument('--only-old-pe-model', action='store_true', help='do_only_old_pe_model (:1141-1142): load best_models/signal_pe.h5 and skip the CNN loop')
    ap.add_argument('--sanity-check', default=None,
                    help='the (n, n_pix) pickle of scripts/make_posterior_templates.py: with --lalinf-posterior the CNN read-out of it is scored (:1224-1229)')
    ap.add_argument('--lalinf-posterior', default=None,
                    help='mc_q pickle of scripts/get_lalinf_pars.py: the overlap of the GAN posterior with it is scored at every cadence (:1345-1356)')
    args = ap.parse_args()

    from gennet_amd import bbh, dist, engine, hostio, templates as T
    dp = dist.init()
    rank, world = (dp.rank, dp.world_size) if dp else (0, 1)
    engine.set_init_seed(1)
    engine.set_device_seed(1000)              # one device stream: every rank takes the counters of its rows of the global draw
    random.seed(1); np.random.seed(1)

    base = '%s%s' % (args.templates, args.event_name)
    ts, par = T.load_ts_pars('%s_ts_0_%sSamp%s.sav' % (base, args.training_num, args.tag), '%s_params_0_%sSamp%s.sav' % (base, args.training_num, args.tag))
    images, labels, _, signal_pars = T.training_arrays(ts, par)                       # :1007-1014, :1036, :1053-1055
    with open('data/%s0%s.sav' % (args.event_name, args.tag), 'rb') as f:            # :1027-1028
        noise_signal = np.reshape(pickle.load(f, encoding='latin1') * args.event_scale, (args.n_pix, 1))

    nets = bbh.build_and_compile(noise_signal, args.n_pix, lr=args.lr, data_parallel=dp, chi_loss=args.chi_loss, n_sig=args.n_sig)
    if dp:
        for m in (nets.generator, nets.signal_discriminator, nets.signal_pe):
            dp.sync_model(m)
    bank = bbh.DeviceBank(images, labels)
    event = engine.to_device(noise_signal.reshape(-1))
    os.makedirs(os.path.join(args.out, 'best_models'), exist_ok=True)
    os.makedirs(os.path.join(args.out, 'GAN_posterior_samples'), exist_ok=True)

    lalinf_pars, beta_score_hist = None, []
    if args.lalinf_posterior:                                                         # :1016-1020 (lalinf_pars)
        with open(args.lalinf_posterior, 'rb') as f:
            lalinf_pars = np.asarray(pickle.load(f, encoding='latin1'), np.float64)
    sanity = None
    if args.sanity_check:                                                             # :1225-1226
        with open(args.sanity_check, 'rb') as f:
            sanity = engine.to_device(np.asarray(pickle.load(f, encoding='latin1'), np.float32).reshape(-1, args.n_pix, 1))
    if args.old_model:                                                                # :1133-1138
        nets.signal_pe.load_weights(os.path.join(args.out, 'best_models/signal_pe.h5'))
        nets.signal_discriminator.load_weights(os.path.join(args.out, 'discriminator.h5'))
        nets.signal_discriminator_on_generator.load_weights(os.path.join(args.out, 'signal_dis_on_gen.h5'))
        nets.generator.load_weights(os.path.join(args.out, 'generator.h5'))
    if args.only_old_pe_model or args.retrain_pe_mod:                                 # :1141-1142
        nets.signal_pe.load_weights(os.path.join(args.out, 'best_models/signal_pe.h5'))
    # serialisation + file writes of the cadence blocks leave the loop's thread; the `with` block closes the writer -- every queued
    # file on disk, or its error raised -- on normal exit AND when a loop dies (KeyboardInterrupt included)
    with hostio.BackgroundWriter() as bg:
        skip_pe = args.only_old_pe_model and not args.retrain_pe_mod                      # :1145
        graphed = args.graph and dp is None and args.n_noise_real == 1
        pe_step = bbh.GraphedPEStep(nets.signal_pe, bank, args.pe_batch_size, cnn_noise_frac=args.cnn_noise_frac) if graphed else None
        gan_step = bbh.GraphedGANStep(nets, bank, event, args.batch_size) if graphed else None
        for i in range(0 if skip_pe else args.pe_iter):                                   # :1153-1173
            if graphed:
                pe_loss = pe_step(want_losses=(i % args.pe_cadence == 0 or i % 1000 == 0))
            else:
                pe_loss = bbh.pe_train_step(nets.signal_pe, bank, args.pe_batch_size, cnn_noise_frac=args.cnn_noise_frac, rank=rank, world=world)
            if i % 5000 == 0 and i > 0 and rank == 0:
                nets.signal_pe.save(os.path.join(args.out, 'best_models/signal_pe.h5'), True, writer=bg)
            if i % args.pe_cadence == 0 and i > 0:                                        # :1176-1196
                rms, pe_std = bbh.pe_accuracy(nets.signal_pe, bank)                       # every rank: it draws from the shared host index stream
                if rank == 0:
                    print('%d: [PE loss: %f, acc: %f, RMS: %f,%f] mean |error| (mc, q): %f, %f' % (i, pe_loss[0], pe_loss[1], rms[0], rms[1], pe_std[0], pe_std[1]),
                          flush=True)
            elif i % args.pe_cadence == 0 and rank == 0:
                print('%d: [PE loss: %f, acc: %f]' % (i, pe_loss[0], pe_loss[1]), flush=True)
            if i % 1000 == 0 and i > 0:                                                   # :1200-1229: its OWN random.sample(.., 4000) -- at the default
                rms, pe_std = bbh.pe_accuracy(nets.signal_pe, bank)                       # pe_cadence = 1000 both blocks run and the host stream advances twice
                if rank == 0:
                    print('%d: [PE loss: %f, acc: %f, RMS: %f,%f] mean |error| (mc, q): %f, %f' % (i, pe_loss[0], pe_loss[1], rms[0], rms[1], pe_std[0], pe_std[1]),
                          flush=True)
                    if sanity is not None and lalinf_pars is not None:
                        score = bbh.posterior_overlap([p.cpu().numpy() for p in nets.signal_pe.predict_device(sanity)], lalinf_pars)
                        if score is not None:
                            print('%d: [CNN sanity check vs lalinference: overlap beta %f]' % (i, score[2]), flush=True)
        if not skip_pe and args.pe_iter > 0 and rank == 0:
            nets.signal_pe.save(os.path.join(args.out, 'best_models/signal_pe.h5'), True, writer=bg)     # so that --old-model finds the trained CNN
        print('Completed CNN PE')

        for i in range(args.max_iter):                                                    # :1241-1382
            if graphed:
                l = gan_step(want_losses=(i % args.cadence == 0))
            else:
                l = bbh.gan_train_step(nets, bank, event, args.batch_size, rank=rank, world=world, n_noise_real=args.n_noise_real)
            if i % args.cadence == 0 and i > 0 and rank == 0:
                print('%d: [sD loss: %f, acc: %f]  [sG loss: %f, acc: %f]' % (i, l[2], l[3], l[0], l[1]), flush=True)
                pe_samples, waves = bbh.posterior_samples(nets, 4000)                     # :1330-1343
                score = bbh.posterior_overlap(pe_samples, lalinf_pars) if lalinf_pars is not None else None          # :1345-1356 (the plot is not produced)
                if score is not None:
                    ks, ad, beta = score
                    beta_score_hist.append(float(beta))
                    print('%d: [posterior overlap beta: %f, KS p (mc, q): %g, %g]' % (i, beta, ks[0][1], ks[1][1]), flush=True)
                    bg.pickle(list(beta_score_hist), os.path.join(args.out, 'beta_score_hist.sav'), protocol=2)
                bg.pickle(pe_samples, os.path.join(args.out, 'gan_pe_samples.sav'), protocol=2)
                bg.pickle(waves, os.path.join(args.out, 'gan_pe_waveforms.sav'), protocol=2)
                nets.generator.save_weights(os.path.join(args.out, 'generator.h5'), True, writer=bg)
                nets.signal_discriminator.save_weights(os.path.join(args.out, 'discriminator.h5'), True, writer=bg)
                nets.signal_discriminator_on_generator.save_weights(os.path.join(args.out, 'signal_dis_on_gen.h5'), True, writer=bg)
                bg.pickle(pe_samples, os.path.join(args.out, 'GAN_posterior_samples/posterior_samples_%05d.sav' % i), protocol=pickle.DEFAULT_PROTOCOL)


if __name__ == '__main__':
    main()
