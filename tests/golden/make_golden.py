#!/usr/bin/env python
"""Generates tests/golden/synth_golden.npz by RUNNING the reference's own pure-numpy helpers.

The reference files are Python 2 and import LALSuite, so they cannot be imported; but these functions are plain numpy:
their source lines are read from /root/reference at run time (never copied into this repo), exec-ed under Python 3 and
fed seeded inputs.  Only inputs and outputs (data) are stored.  Run in the build container (the reference does not
exist on the GPU box):      python tests/golden/make_golden.py

  tukey            gw_template_maker.py:87-113
  convert_beta     gw_template_maker.py:133-159   (module global safe = 2, :54)
  gen_noise        gw_template_maker.py:161-193
  whiten_data      gw_template_maker.py:243-286
  hunt_constrain   gw_template_maker.py:329-338   (body of gen_masses, dedented; the py2 print on :328 is skipped)
"""
import os
import textwrap

import numpy as np

REF = '/root/reference/BBH_version/gw_template_maker.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'synth_golden.npz')


def grab(lines, a, b):
    return ''.join(lines[a - 1:b])


def main():
    lines = open(REF).read().splitlines(True)
    ns = {'np': np, 'safe': 2}
    for a, b in ((87, 113), (133, 159), (161, 193), (243, 286)):
        exec(compile(grab(lines, a, b), '%s:%d-%d' % (REF, a, b), 'exec'), ns)
    body = textwrap.dedent(grab(lines, 329, 338))
    src = 'def hunt_constrain(m_min, M_max):\n    flag = False\n' + textwrap.indent(body, '    ') + '    return m12, mc, eta\n'
    exec(compile(src, '%s:329-338' % REF, 'exec'), ns)

    out = {}
    # tukey: the window sizes gen_bbh builds (int(16/15 * N/2), alpha 1/8) and whiten_data('td') builds (N, alpha 1/8)
    for fs in (1024, 2048, 4096):
        N = 4 * fs
        out['tukey_%d' % int((16.0 / 15.0) * N / 2)] = ns['tukey'](int((16.0 / 15.0) * N / 2), alpha=1.0 / 8.0)
    out['tukey_64_half'] = ns['tukey'](64, alpha=0.5)
    out['tukey_4096_eighth'] = ns['tukey'](4096, alpha=1.0 / 8.0)
    # convert_beta index constants
    cb = []
    for fs in (1024, 2048, 4096):
        for beta in ([0.45, 0.55], [0.5, 0.5], [0.75, 0.95]):
            cb.append([fs, beta[0], beta[1]] + list(ns['convert_beta'](beta, fs, 4)))
    out['convert_beta'] = np.array(cb)
    # whiten_data, both flags, on seeded data with a PSD that has zero bins
    rng = np.random.RandomState(11)
    fs, T = 256, 4
    N = fs * T; Nf = N // 2 + 1
    psd = np.abs(rng.randn(Nf)) * 1e-3 + 1e-4
    psd[:5] = 0.0; psd[40] = 0.0
    xf = rng.randn(Nf) + 1j * rng.randn(Nf)
    out['wh_psd'] = psd
    out['wh_fd_in'] = xf.copy()
    out['wh_fd_out'] = ns['whiten_data'](xf.copy(), T, fs, psd, 'fd')
    xt = rng.randn(N)
    out['wh_td_in'] = xt.copy()
    out['wh_td_out'] = ns['whiten_data'](xt.copy(), T, fs, psd, 'td')
    # gen_noise with the legacy global stream
    np.random.seed(7)
    out['noise_psd'] = psd
    out['noise_out'] = ns['gen_noise'](fs, T, psd)
    np.random.seed(7)
    out['noise_normals'] = np.random.normal(0, 1, 2 * Nf)          # the draws it consumed: re block then im block
    # hunt_constrain rejection sampler: 200 accepted draws and the stream position afterwards
    np.random.seed(1)
    acc = []
    for _ in range(200):
        m12, mc, eta = ns['hunt_constrain'](5.0, 100.0)
        acc.append([m12[0], m12[1], mc, eta])
    out['hunt_seed1'] = np.array(acc)
    out['hunt_seed1_next_uniform'] = np.random.uniform(0, 1, 3)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, {k: np.asarray(v).shape for k, v in out.items()})
    posterior_golden()


def posterior_golden():
    """overlap_tests (bbhMahoGANy.py:811-873) executed as written, on seeded samples, with scipy KDEs built the way
    make_contour_plot does (:790: gaussian_kde(dataset), dataset = np.array([x, y]))."""
    import warnings
    from scipy.stats import anderson_ksamp, gaussian_kde, ks_2samp
    lines = open('/root/reference/BBH_version/bbhMahoGANy.py').read().splitlines(True)
    ns = {'np': np, 'ks_2samp': ks_2samp, 'anderson_ksamp': anderson_ksamp, 'comb_pe_model': False}
    exec(compile(grab(lines, 811, 873), 'bbhMahoGANy.py:811-873', 'exec'), ns)
    rng = np.random.RandomState(21)
    n_pe, n_lal = 4000, 3907
    cov = np.array([[0.6, 0.012], [0.012, 0.002]])
    pe = rng.multivariate_normal([30.2, 0.80], cov, n_pe)
    lal = rng.multivariate_normal([30.0, 0.79], cov * 1.3, n_lal)
    pred_samp = [pe[:, 0:1].astype(np.float32), pe[:, 1:2].astype(np.float32)]          # what signal_pe.predict returns
    lalinf_samp = np.array([lal[:, 0], lal[:, 1]])
    k_cnn = gaussian_kde(np.array([pred_samp[0].reshape(-1), pred_samp[1].reshape(-1)]))
    k_lal = gaussian_kde(lalinf_samp)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ks, ad, beta = ns['overlap_tests'](pred_samp, lalinf_samp, [30.0, 0.79], k_cnn, k_lal)
    probe = np.vstack([rng.uniform(27, 33, 50), rng.uniform(0.6, 1.0, 50)])
    out = {'pred_mc': pred_samp[0], 'pred_q': pred_samp[1], 'lalinf': lalinf_samp, 'beta': beta,
           'ks': np.array([[ks[0][0], ks[0][1]], [ks[1][0], ks[1][1]]], dtype=np.float64),
           'ad_stat': np.array([ad[0][0], ad[1][0]]), 'probe': probe, 'probe_pdf_cnn': k_cnn.pdf(probe), 'probe_pdf_lal': k_lal.pdf(probe)}
    path = os.path.join(os.path.dirname(OUT), 'posterior_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, 'beta =', beta)


if __name__ == '__main__':
    main()
