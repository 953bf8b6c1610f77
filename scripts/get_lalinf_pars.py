#!/usr/bin/env python
"""Counterpart of BBH_version/data/get_lalinf_pars.py: lalinference posterior columns (chirp mass mc, mass ratio q) -> the pickles the
trainer and the posterior-driven template maker read:

    <event>_m1_m2_lainf_post_<tag>.sav     np.array([post_m1, post_m2])      (get_lalinf_pars.py:52-67; file name as the reference spells it)
    <event>_mc_M_lainf_post_<tag>.sav      np.array([post_mc, post_M])       (:69-86)
    <event>_mc_q_lalinf_post_<tag>.sav     np.array([post_mc, post_q])       (:88-91)

The reference solves two equations per posterior row with sympy (minutes for a few thousand rows); gennet_amd.templates.lalinf_pars is their
closed form (pinned by tests/golden/lalinf_pars_golden.npz).  Defaults follow the reference's switches (:47-49): m1_m2 and mc_q on, mc_M off.

Input: the reference reads a lalinference `posterior_samples.hdf5` through pandas / PyTables, neither of which is a dependency here; give the
two columns instead as an .npz with arrays `mc` and `q`, a pickle of a (2, n) array [mc, q], or a text file with two columns.

  python scripts/get_lalinf_pars.py --posterior post.npz --event-name gw150914 --tag srate-2048 --out data/
"""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_columns(path):
    if path.endswith('.npz'):
        z = np.load(path)
        return np.asarray(z['mc'], np.float64), np.asarray(z['q'], np.float64)
    if path.endswith(('.sav', '.pkl', '.pickle')):
        with open(path, 'rb') as f:
            a = np.asarray(pickle.load(f, encoding='latin1'), np.float64)
        if a.ndim != 2 or a.shape[0] != 2:
            raise SystemExit('%s: expected a pickled (2, n) array [mc, q], got shape %r' % (path, a.shape))
        return a[0], a[1]
    a = np.loadtxt(path, ndmin=2)
    if a.shape[1] != 2:
        raise SystemExit('%s: expected two columns (mc, q)' % path)
    return a[:, 0], a[:, 1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--posterior', required=True)
    ap.add_argument('--event-name', default='gw150914')
    ap.add_argument('--tag', default='srate-2048')
    ap.add_argument('--out', default='.')
    ap.add_argument('--m1m2', type=int, default=1, help='write the m1_m2 file (do_m1m2, :47)')
    ap.add_argument('--mc-M', type=int, default=0, help='write the mc_M file (do_mc_M, :48)')
    ap.add_argument('--mc-q', type=int, default=1, help='write the mc_q file (do_mc_q, :49)')
    a = ap.parse_args()
    from gennet_amd import templates as T
    mc, q = read_columns(a.posterior)
    pars = T.lalinf_pars(mc, q)
    os.makedirs(a.out, exist_ok=True)
    for on, key, stem in ((a.m1m2, 'm1_m2', '%s_m1_m2_lainf_post_%s.sav'), (a.mc_M, 'mc_M', '%s_mc_M_lainf_post_%s.sav'),
                          (a.mc_q, 'mc_q', '%s_mc_q_lalinf_post_%s.sav')):
        if on:
            path = os.path.join(a.out, stem % (a.event_name, a.tag))
            with open(path, 'wb') as f:
                pickle.dump(pars[key], f, protocol=2)
            print('%s: %s array of %d posterior rows' % (path, key, pars[key].shape[1]))


if __name__ == '__main__':
    main()
