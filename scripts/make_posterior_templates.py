#!/usr/bin/env python
"""Counterpart of lalinf_post_waveform_maker.main() (lalinf_post_waveform_maker.py:748-843): noise-free whitened waveforms whose component
masses come row by row from lalinference posterior samples -- the CNN "sanity check" set of the trainer (bbhMahoGANy.py:1228-1231).

Same flags and event handling as scripts/make_templates.py (the reference's two main() functions share those lines), plus the posterior
file: the (2, n) m1_m2 pickle written by scripts/get_lalinf_pars.py (data/get_lalinf_pars.py:65-67).  As in the reference (:385) the pair
handed to the waveform generator is [row 1, row 0] of that file; with --mc-q-file the chirp mass label comes from the posterior column
(:404) instead of being recomputed from the masses.  Output: data/<event>_cnn_sanity_check_ts_mass-time-vary<tag>.sav, float64 (n, fs), the
last row the event-like (36, 29) template (:831-836).  The percentile plot (:807-829) is reporting and not produced.

  python scripts/make_posterior_templates.py --posterior data/gw150914_m1_m2_lainf_post_srate-2048.sav -f 2048 -T 1 -N 3907 -Nb 3907
"""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def extra(p):
    p.add_argument('--posterior', required=True, help='the m1_m2 pickle of scripts/get_lalinf_pars.py')
    p.add_argument('--mc-q-file', default=None, help='optional mc_q pickle: chirp-mass labels from the posterior column')
    p.add_argument('--batch-size', type=int, default=3907, help='lalinf_post_waveform_maker.py:694 generates batch_size - 1 posterior rows')


def main():
    from gennet_amd import templates as T
    from make_templates import parser, prepare_event
    args = parser('make_posterior_templates.py', extra)
    psd, wht_wvf, h_t, gw_norm_constant, fs, safeTobs, tag = prepare_event(args, T)
    event_name = 'gw150914'
    with open(args.posterior, 'rb') as f:
        m12 = np.asarray(pickle.load(f, encoding='latin1'), np.float64)
    if m12.ndim != 2 or m12.shape[0] != 2:
        raise SystemExit('%s: expected a pickled (2, n) array [post_m1, post_m2]' % args.posterior)
    post_mc = None
    if args.mc_q_file:
        with open(args.mc_q_file, 'rb') as f:
            post_mc = np.asarray(pickle.load(f, encoding='latin1'), np.float64)[0]
    os.makedirs('data', exist_ok=True)
    nblock = int(np.ceil(float(args.Nsamp) / float(args.Nblock)))
    for i in range(nblock):
        ts, par = T.sim_data_posterior(fs, safeTobs, psd, m12[1], m12[0], post_mc, dets=args.detectors, size=args.Nblock, beta=[0.45, 0.55],
                                       batch_size=args.batch_size, peak_off=args.peak_offset)
        path = 'data/%s_cnn_sanity_check_ts_mass-time-vary%s.sav' % (event_name, tag)
        arr = T.save_sanity_check(path, ts, gw_norm_constant)
        print('block %d/%d: %s (%d waveforms x %d samples, gw_norm_constant %.6g)' % (i + 1, nblock, path, arr.shape[0], arr.shape[1], gw_norm_constant))
    print('success')


if __name__ == '__main__':
    main()
