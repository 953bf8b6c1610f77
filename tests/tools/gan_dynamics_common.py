"""The GAN-dynamics study's problem, built on the CPU from the oracle synthesiser so that the CPU port (gan_dynamics_cpu.py) and the HIP path
(gan_dynamics_gpu.py) train on the SAME bank and the SAME event.

TEST INFRASTRUCTURE: imports oracle/ (allowed under tests/ only).  Event construction is scripts/validate_posterior.py's: templates scaled to unit
variance (the role of gw_norm_constant, gw_template_maker.py:782), event = held-out (36, 29) template at idx N/2 + N(0,1) from RandomState(100 + seed)
(bbhMahoGANy.py:1027-1029: the event file is a whitened noisy template), PSD = the validation script's analytic curve.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_problem(n_pix=256, bank=2000, seed=1):
    """Returns dict(bank (bank, n_pix) f32, pars (bank, 2) [mc, q], event (n_pix,) f32, clean (n_pix,) f64 scaled, g, snr)."""
    from oracle import synth_ref as S
    fs, T_obs = int(n_pix), 4
    N = T_obs * fs
    f = np.arange(N // 2 + 1) * 0.25
    psd = 1e-46 * ((np.maximum(f, 10.0) / 150.0) ** -4.0 + 2.0 + 2.0 * (f / 150.0) ** 2.0)
    psd[f < 10.0] = 0.0
    Fp, Fc = S.antenna_response(S.EVENT_TIME, S.RA, S.DEC, S.PSI)
    st = np.random.get_state()
    np.random.seed(seed)
    rows, pars = [], []
    for _ in range(bank):
        p = S.gen_par(fs, T_obs, 'hunt_constrain', (0.45, 0.55), False)
        rows.append(S.gen_bbh(fs, T_obs, psd, p, Fp, Fc)[0])
        pars.append([p.mc, p.m2 / p.m1])
    np.random.set_state(st)
    rows = np.asarray(rows)
    g = 1.0 / rows.std()
    m1, m2 = 36.0, 29.0
    M = m1 + m2
    eta = m1 * m2 / M ** 2
    ev = S.bbhparams(M * eta ** 0.6, M, eta, m1, m2, S.RA, S.DEC, S.IOTA, S.PHI, S.PSI, N // 2, None, None)
    clean = S.gen_bbh(fs, T_obs, psd, ev, Fp, Fc)[0] * g
    noise = np.random.RandomState(100 + seed).randn(fs)
    return {'bank': (rows * g).astype(np.float32), 'pars': np.asarray(pars, np.float32), 'event': (clean + noise).astype(np.float32), 'clean': clean,
            'g': float(g), 'snr': float(np.sqrt((clean ** 2).sum())), 'n_pix': fs}


def waveform_stats(w, clean):
    """w (n, n_pix) generator outputs: (mean normalised overlap with the clean event, rms)."""
    w = np.asarray(w, np.float64).reshape(len(w), -1)
    ov = (w @ clean) / (np.linalg.norm(w, axis=1) * np.linalg.norm(clean) + 1e-30)
    return float(ov.mean()), float(np.sqrt(np.mean(w ** 2)))
