#!/usr/bin/env python
"""Counterpart of gw_template_maker.main() (gw_template_maker.py:743-865): same argparse flags, same output files.

The reference reads three lalinference .dat files (freqData, freqDataWithInjection, PSD; :753-767) that are not shipped with
it.  This script takes them when given (--freq-data / --freq-data-inj / --psd-file) and otherwise falls back to a synthetic
event: an analytic aLIGO-like PSD, coloured noise from gen_noise and the GW150914-like (36, 29) template as the injection.
Everything else follows main(): whiten the event in the frequency domain, gw_norm_constant = 1/std of the whitened noisy event
(:782), central 1-s crops (:790-791), sim_data blocks with mdist 'hunt_constrain' and beta [0.45, 0.55] (:806), x gw_norm_constant
(:813-814), ts / params / event pickles (:842-863).
"""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parser(prog='make_templates.py', extra=None):
    p = argparse.ArgumentParser(prog=prog, description='generates GW template banks on MI355X (gennet_amd)')
    p.add_argument('-N', '--Nsamp', type=int, default=50000, help='the number of samples')
    p.add_argument('-Nn', '--Nnoise', type=int, default=0, help='the number of noise realisations per signal, if 0 then signal only')
    p.add_argument('-Nb', '--Nblock', type=int, default=50000, help='the number of training samples per output file')
    p.add_argument('-f', '--fsample', type=int, default=1024, help='the sampling frequency (Hz)')
    p.add_argument('-T', '--Tobs', type=int, default=2, help='the observation duration (sec)')
    p.add_argument('-I', '--detectors', type=str, nargs='+', default=['H1'], help='the detectors to use')
    p.add_argument('-b', '--basename', type=str, default='templates/', help='output file path and basename')
    p.add_argument('-m', '--mdist', type=str, default='astro', help='mass distribution for training (astro, hunt_constrain)')
    p.add_argument('-z', '--seed', type=int, default=1, help='the random seed')
    p.add_argument('--freq-data', default=None)
    p.add_argument('--freq-data-inj', default=None)
    p.add_argument('--psd-file', default=None)
    p.add_argument('--peak-offset', type=int, default=11, help='the alignment constant of gw_template_maker.py:554')
    p.add_argument('--tag', default=None)
    if extra:
        extra(p)
    return p.parse_args()


def prepare_event(args, T):
    """gw_template_maker.py:753-791 (and, line for line the same, lalinf_post_waveform_maker.py:759-791): the event in noise and the noise-free
    event whitened in the frequency domain, gw_norm_constant = 1/std of the whitened noisy event, their central 1-s crops, the PSD.
    Returns (psd, wht_wvf, h_t, gw_norm_constant, fs, safeTobs, tag)."""
    if args.seed > 0:
        np.random.seed(args.seed)
    fs = args.fsample
    safeTobs = T.safe * args.Tobs
    N = fs * safeTobs
    Nf = N // 2 + 1
    tag = args.tag if args.tag is not None else '_srate-%dhz_oversamp' % fs

    if args.freq_data and args.freq_data_inj and args.psd_file:
        noise_f = np.loadtxt(args.freq_data)[:, 1:]
        sig_f = np.loadtxt(args.freq_data_inj)[:, 1:]
        noise_f = noise_f[:, 0] + 1j * noise_f[:, 1]
        sig_f = sig_f[:, 0] + 1j * sig_f[:, 1]
        sig_f[np.isnan(sig_f)] = 0
        noise_f[np.isnan(noise_f)] = 0
        psd = np.loadtxt(args.psd_file)[:, 1]
        h_f = sig_f - noise_f
    else:
        f = np.arange(Nf) / float(safeTobs)
        x = np.where(f > 0, f, 1.0) / 215.0
        psd = np.where(f >= 10.0, 1e-49 * (x ** -4.14 - 5.0 / (x * x) + 111.0 * (1 - x * x + 0.5 * x ** 4) / (1 + 0.5 * x * x)), 0.0)
        syn = T.Synth(fs, safeTobs, np.ones(Nf))           # unit "PSD": un-whitened event-like template in the time domain
        hp_t, hc_t, _ = syn.series([36.0], [29.0])
        ht = (hp_t * syn.Fp + hc_t * syn.Fc).cpu().numpy().reshape(N) / np.sqrt(2.0 / fs)
        h_f = np.fft.rfft(ht)
        noise_f = np.fft.rfft(T.gen_noise(fs, safeTobs, psd))
        sig_f = noise_f + h_f

    wht_wvf = np.fft.irfft(T.whiten_data(sig_f, safeTobs, fs, psd, 'fd'), N)
    h_t = np.fft.irfft(T.whiten_data(h_f, safeTobs, fs, psd, 'fd'), N)
    gw_norm_constant = 1.0 / np.std(wht_wvf)
    c0, c1 = int((safeTobs / 2) * fs - fs / 2.0), int((safeTobs / 2) * fs + fs / 2.0)
    return psd, wht_wvf[c0:c1], h_t[c0:c1], gw_norm_constant, fs, safeTobs, tag


def main():
    from gennet_amd import templates as T
    args = parser()
    psd, wht_wvf, h_t, gw_norm_constant, fs, safeTobs, tag = prepare_event(args, T)
    event_name = 'gw150914'

    os.makedirs(os.path.dirname(args.basename) or '.', exist_ok=True)
    os.makedirs('data', exist_ok=True)
    nblock = int(np.ceil(float(args.Nsamp) / float(args.Nblock)))
    for i in range(nblock):
        ts, par = T.sim_data(fs, safeTobs, psd, args.detectors, args.Nnoise, size=args.Nblock, mdist='hunt_constrain', beta=[0.45, 0.55], peak_off=args.peak_offset)
        ts[0] = ts[0] * gw_norm_constant
        if i != nblock - 1:
            ts[0] = ts[0][:-1]
            ts[1] = ts[1][:-1]
            par = par[:-1]
        tp, pp = T.save_ts_pars(args.basename, event_name, i, args.Nsamp, tag, ts, par)
        with open('data/%s%d%s.sav' % (event_name, i, tag), 'wb') as fh:
            pickle.dump(wht_wvf, fh, protocol=2)
        with open('data/%s_data%s.pkl' % (event_name, tag), 'wb') as fh:
            pickle.dump(h_t, fh, protocol=2)
        print('block %d/%d: %s %s (%d templates, gw_norm_constant %.6g)' % (i + 1, nblock, tp, pp, ts[0].shape[0], gw_norm_constant))
    print('success')


if __name__ == '__main__':
    main()
