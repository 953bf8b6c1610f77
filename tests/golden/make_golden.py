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

indexing_golden.npz (sample indexing: north_star's "bit-exact for sample indexing" rests on these):
  bbhparams        gw_template_maker.py:69-85
  gen_masses       gw_template_maker.py:289-370
  gen_par          gw_template_maker.py:372-460   (draw order of the legacy MT19937 stream, idx, the event-like branch)
  gen_bbh          gw_template_maker.py:462, :493-498, :518-547, :553-575 executed as written; the LALSuite lines :499-516 is replaced by
                   SUPPLIED frequency-domain spectra hp.data.data / hc.data.data, and the make_bbh call :551 (pylal antenna response +
                   LAL time delay; its return value :630 is the un-shifted ht = hp*Fp + hc*Fc, hp, hc) by that expression with SUPPLIED
                   constants Fp, Fc.  Everything else -- whiten_data('fd'), irfft, roll, ref_idx = argmax, sidx, the window placement,
                   the python slice [ref_idx - par.idx - 11:], zero fill, window -- is the reference's own text.
  sim_data         gw_template_maker.py:632-740   executed as written (Nnoise = 0) on top of the functions above: the crop :695, the trim
                   quirk :718-722, the permutation :725-727, the event-like template appended last :730-739
masses_golden.npz: gen_masses :289-370 for 'astro', 'hunt_constrain', 'gh', 'metric' on seeded streams (values and stream position)
lalinf_pars_golden.npz (data/get_lalinf_pars.py, the posterior-column conversion in front of the posterior-driven synthesiser, row n3):
  the do_m1m2 and do_mc_M loops (:52-65, :69-84, located by their first / last statement) executed as written with sympy on supplied
  post_mc / post_q columns
posterior_mode_golden.npz (the posterior-driven synthesiser, lalinf_post_waveform_maker.py, row n3):
  bbhparams :71-87, tukey :89-116, convert_beta :136-163, whiten_data :247-290, gen_par :356-475 (masses from SUPPLIED gan_post /
  all_lalinf_posteriors['mc'] rows; the randint it draws BEFORE the gw_tmp branch overrides idx), gen_bbh :477, :508-513, :534-561,
  :566-585, :587 as written with the LALSuite lines :514-532 replaced by SUPPLIED spectra and the make_bbh call :564 by its return value
  hp*Fp + hc*Fc (:628, :647), and the whole of sim_data :649-746 -- run over TWO consecutive blocks of one seeded stream, as main()'s
  nblock loop does (:799-805)
SAFETY: this script exec()s text read from /root/reference (public, untrusted content).  Run it ONLY in the sandboxed build container.  Every
file is pinned by SHA-256 before anything of it is executed (REF_SHA256), and every executed range asserts its first and last statement
(ANCHORS), so a shifted or altered upstream file fails here instead of silently executing other statements.
Run-time text handling (nothing of it is stored): tabs expanded to 8 columns (the files mix tabs and spaces: Python 2 semantics), and
lines that are Python 2 `print '...'` statements (with or without the `if verb:` prefix) replaced by `pass`.  Integer `/` in those
lines only ever divides even ints by 2 or feeds int(): Python 3's true division gives the same values for T_obs = 4 and even fs.
"""
import hashlib
import os
import re
import textwrap
import time

import numpy as np

REF = '/root/reference/BBH_version/gw_template_maker.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'synth_golden.npz')


REF_DIR = '/root/reference/BBH_version/'
REF_SHA256 = {
    'gw_template_maker.py': 'c0463d49c136e3a0510c74dda90b8ed3d4b4eede5b7d9728d3a7fee46fdf8fda',
    'lalinf_post_waveform_maker.py': '25da70c5237d590d342d5edde8d633e55695e907ae44f8767e76dbfb67bb5716',
    'bbhMahoGANy.py': 'ca60921f79dd095532b46314dd515d7890eb9c6cf42452d6f4dd06a464dd40c2',
    'data/get_lalinf_pars.py': '2c1e2f9bf8a1736e574ac865b4a478e2498137e7287ebd757961c4d68bcf92b6',
}
# first / last statement (stripped prefix) of every line range executed below
ANCHORS = {
    'gw_template_maker.py': {(69, 85): ('class bbhparams:', 'self.SNR = SNR'), (87, 113): ('def tukey(M,alpha=0.5):', 'return np.array(w[:M])'),
                             (133, 159): ('def convert_beta(beta,fs,T_obs):', 'return low_idx,high_idx'),
                             (161, 193): ('def gen_noise(fs,T_obs,psd):', 'return x'),
                             (243, 286): ('def whiten_data(data,duration,sample_rate,psd,flag=', 'return xf'),
                             (289, 370): ('def gen_masses(m_min=5.0,M_max=100.0,mdist=', 'exit(1)'),
                             (329, 338): ('new_m_min = m_min', 'mc = np.sum(m12)*eta**(3.0/5.0)'),
                             (372, 460): ('def gen_par(fs,T_obs,mdist=', 'return par'),
                             (462, 462): ('def gen_bbh(fs,T_obs,psds,dets=', 'def gen_bbh('), (493, 498): ('N = T_obs * fs', 'f_max = fs/2'),
                             (518, 546): ('whiten_hp = whiten_data(hp.data.data', 'j = 0'),
                             (553, 573): ('# place signal into timeseries', 'hc[j,:] *= win'), (575, 575): ('return ts, hp, hc, ts', 'return ts, hp, hc, ts'),
                             (632, 740): ('def sim_data(fs,T_obs,psds,dets=', 'return [ts, yval], temp')},
    'lalinf_post_waveform_maker.py': {(71, 87): ('class bbhparams:', 'self.SNR = SNR'), (89, 116): ('def tukey(M,alpha=0.5):', 'return np.array(w[:M])'),
                                      (136, 163): ('def convert_beta(beta,fs,T_obs):', 'return low_idx,high_idx'),
                                      (247, 290): ('def whiten_data(data,duration,sample_rate,psd,flag=', 'return xf'),
                                      (356, 475): ('def gen_par(fs,T_obs,index,mdist=', 'return par'),
                                      (477, 477): ('def gen_bbh(fs,T_obs,idx,psds,dets=', 'def gen_bbh('), (508, 513): ('N = T_obs * fs', 'f_max = fs/2'),
                                      (534, 561): ('# whiten waveform in frequency domain', 'for det in dets:'),
                                      (566, 585): ('# place signal into timeseries', 'hc[j,:] *= win'), (587, 587): ('return ts, hp, hc', 'return ts, hp, hc'),
                                      (649, 746): ('def sim_data(fs,T_obs,psds,dets=', 'return [ts, yval], temp')},
    'bbhMahoGANy.py': {(811, 873): ('def overlap_tests(pred_samp,lalinf_samp,true_vals,kernel_cnn,kernel_lalinf):', 'return ks_score, ad_score, beta_score')},
}


def ref_lines(name):
    """The lines of one reference file, after checking the file is byte for byte the one the line numbers in this script were read from."""
    raw = open(REF_DIR + name, 'rb').read()
    got = hashlib.sha256(raw).hexdigest()
    if got != REF_SHA256[name]:
        raise SystemExit('%s%s: sha256 %s differs from the pinned %s -- the line ranges in this script no longer apply; nothing executed'
                         % (REF_DIR, name, got, REF_SHA256[name]))
    lines = raw.decode('utf-8').splitlines(True)
    for (a, b), (first, last) in ANCHORS.get(name, {}).items():
        if not lines[a - 1].strip().startswith(first) or not lines[b - 1].strip().startswith(last):
            raise SystemExit('%s:%d-%d does not start / end with the expected statements (%r ... %r)' % (name, a, b, first, last))
    return lines


def grab(lines, a, b):
    return ''.join(lines[a - 1:b])


def main():
    lines = ref_lines('gw_template_maker.py')
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
    # gen_noise -> whiten_data('td') composed (BASELINE configs[4]: coloured noise whitened with the same PSD), the reference's two functions
    # as written, at N = 1024 and N = 8192; the normals consumed are stored so that the fused kernel can be fed the same draws
    for fs_, seed_ in ((256, 17), (2048, 18)):
        N_ = fs_ * T; Nf_ = N_ // 2 + 1
        prng = np.random.RandomState(seed_)
        psd_ = np.abs(prng.randn(Nf_)) * 1e-3 + 1e-4
        psd_[:5] = 0.0; psd_[Nf_ // 3] = 0.0
        np.random.seed(seed_)
        x_ = ns['gen_noise'](fs_, T, psd_)
        np.random.seed(seed_)
        out['nchain_%d_normals' % fs_] = np.random.normal(0, 1, 2 * Nf_)
        out['nchain_%d_psd' % fs_] = psd_
        out['nchain_%d_out' % fs_] = ns['whiten_data'](x_.copy(), T, fs_, psd_, 'td')
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
    indexing_golden()
    lalinf_pars_golden()
    masses_golden()
    posterior_mode_golden()


_PY2_PRINT = re.compile(r"^(\s*)(if verb:\s*)?print\s+'")


def ref_text(lines, a, b):
    """Reference lines a..b (1-based, inclusive) as Python-3-parsable text: tabs expanded, py2 print statements -> pass."""
    out = []
    for ln in lines[a - 1:b]:
        ln = ln.expandtabs(8)
        m = _PY2_PRINT.match(ln)
        out.append(m.group(1) + 'pass\n' if m else ln)
    return ''.join(out)


class _FD(object):
    """Shape of a LAL COMPLEX16FrequencySeries as gen_bbh reads it: .data.data is the complex spectrum."""

    def __init__(self, a):
        self.data = type('d', (), {})()
        self.data.data = a


def reference_namespace():
    """The reference's synthesiser functions, executed from its own text, with the two LALSuite touch points supplied by the caller
    through ns['_supplied_fd_waveform'](par, fs, T_obs) -> (hp, hc) and ns['_Fp'], ns['_Fc']."""
    lines = ref_lines('gw_template_maker.py')
    ns = {'np': np, 'time': time, 'safe': 2, 'verb': False, 'gw_tmp': True, 'do_time_grid': False, 'N_time_grid': 25, 'sample_num': 50000,
          '_captured': {}}
    for a, b in ((69, 85), (87, 113), (133, 159), (243, 286), (289, 370), (372, 460)):
        exec(compile(ref_text(lines, a, b), '%s:%d-%d' % (REF, a, b), 'exec'), ns)
    src = (ref_text(lines, 462, 462) + ref_text(lines, 493, 498)
           + '    hp, hc = _supplied_fd_waveform(par, fs, T_obs)        # stands in for :499-516 (lalsimulation)\n'
           + ref_text(lines, 518, 546)
           + '    for det in dets:\n'
           + '        ht_shift, hp_shift, hc_shift = orig_hp*_Fp + orig_hc*_Fc, orig_hp, orig_hc     # stands in for :551 (make_bbh -> :613, :630)\n'
           + ref_text(lines, 553, 573)
           + '    _captured.update(ref_idx=int(ref_idx), sidx=sidx, win=win.copy(), start=int(ref_idx-par.idx-11))\n'
           + ref_text(lines, 575, 575))
    assert lines[548 - 1].strip() == 'for det in dets:' and 'make_bbh' in lines[551 - 1] and lines[575 - 1].strip() == 'return ts, hp, hc, ts'
    exec(compile(src, '%s:gen_bbh' % REF, 'exec'), ns)
    exec(compile(ref_text(lines, 632, 740), '%s:632-740' % REF, 'exec'), ns)
    return ns


def indexing_golden():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from oracle import synth_ref as S          # only for the SUPPLIED inputs (this project's chirp model, PSD curve and Fp/Fc constants)
    ns = reference_namespace()
    Fp, Fc = S.antenna_response(S.EVENT_TIME, S.RA, S.DEC, S.PSI)
    ns['_Fp'], ns['_Fc'] = Fp, Fc
    out = {'Fp': Fp, 'Fc': Fc}
    T_obs = 4

    # --- gen_par: draw order and values, both branches, three sample rates; stream position afterwards
    rows = []
    for fs, seed in ((1024, 1), (2048, 2), (4096, 3), (256, 4)):
        np.random.seed(seed)
        for k in range(40):
            p = ns['gen_par'](fs, T_obs, mdist='hunt_constrain', beta=[0.45, 0.55], gw_tmp=(k % 13 == 12))
            rows.append([fs, seed, p.mc, p.M, p.eta, p.m1, p.m2, p.ra, p.dec, p.iota, p.phi, p.psi, p.idx])
        rows.append([fs, seed, -1] + list(np.random.uniform(0, 1, 3)) + [0] * 7)          # marker row: the next three uniforms
    out['gen_par_rows'] = np.array(rows, dtype=np.float64)
    np.random.seed(5)
    out['gen_par_beta_75_95'] = np.array([ns['gen_par'](1024, T_obs, mdist='hunt_constrain', beta=[0.75, 0.95]).idx for _ in range(20)])

    # --- gen_bbh downstream of the spectra: chirps from this project's model AND adversarial random spectra (ref_idx anywhere, so the
    # python slice start goes negative / beyond N - len(window) and the zero fill is exercised), per sample rate
    cases = []
    for fs in (256, 1024, 2048, 4096):
        N = fs * T_obs; Nf = N // 2 + 1
        psd = S.analytic_psd(Nf, 1.0 / T_obs)
        rng = np.random.RandomState(100 + fs)
        np.random.seed(fs)
        n_chirp = 4 if fs <= 2048 else 2
        for k in range(n_chirp + 3):
            par = ns['gen_par'](fs, T_obs, mdist='hunt_constrain', beta=[0.45, 0.55], gw_tmp=(k == 1))
            if k < n_chirp:
                hp, hc = S.chirp_fd(par.m1, par.m2, Nf, 1.0 / T_obs, iota=par.iota, phi=par.phi)
            else:                                                  # random spectra scaled like a strain spectrum
                hp = (rng.randn(Nf) + 1j * rng.randn(Nf)) * 1e-24; hc = (rng.randn(Nf) + 1j * rng.randn(Nf)) * 1e-24
                par.idx = [3, N - 5, N // 2][k - n_chirp]          # slide start far negative / far positive / central
            ns['_supplied_fd_waveform'] = lambda par_, fs_, T_, hp=hp, hc=hc: (_FD(hp.copy()), _FD(hc.copy()))
            ts, hp_t, hc_t, _ = ns['gen_bbh'](fs, T_obs, psd, dets=['H1'], beta=[0.45, 0.55], par=par, gw_tmp=(k == 1))
            cap = dict(ns['_captured'])
            cases.append((fs, par.idx, cap['ref_idx'], cap['sidx'], cap['start'], hp, hc, ts[0], cap['win']))
    out['bbh_meta'] = np.array([[c[0], c[1], c[2], c[3], c[4]] for c in cases], dtype=np.int64)       # fs, idx, ref_idx, sidx, slice start
    for n, c in enumerate(cases):
        out['bbh_hp_%02d' % n] = c[5]; out['bbh_hc_%02d' % n] = c[6]
        fs = c[0]
        out['bbh_crop_%02d' % n] = c[7][int(1.5 * fs):int(2.5 * fs)]
        nz = np.flatnonzero(c[7])
        out['bbh_support_%02d' % n] = np.array([nz.min() if nz.size else -1, nz.max() if nz.size else -1, float(np.abs(c[7]).sum()), float((c[7] ** 2).sum())])
    wins = {}
    for c in cases:
        wins.setdefault(c[0], c[8])
    for fs, w in wins.items():                                     # window placement: first/last non-zero sample, flat region, checksum
        nz = np.flatnonzero(w); flat = np.flatnonzero(w == 1.0)
        out['win_%d' % fs] = np.array([nz.min(), nz.max(), flat.min(), flat.max(), w.sum()])

    # --- sim_data, whole, seeded, with chirps from this project's model as the supplied spectra
    for fs, size, seed in ((256, 14, 1), (1024, 6, 1)):
        N = fs * T_obs; Nf = N // 2 + 1
        psd = S.analytic_psd(Nf, 1.0 / T_obs)
        ns['_supplied_fd_waveform'] = lambda par_, fs_, T_: tuple(_FD(a) for a in S.chirp_fd(par_.m1, par_.m2, fs_ * T_ // 2 + 1, 1.0 / T_, iota=par_.iota, phi=par_.phi))
        np.random.seed(seed)
        (ts, yval), pars = ns['sim_data'](fs, T_obs, psd, dets=['H1'], Nnoise=0, size=size, mdist='hunt_constrain', beta=[0.45, 0.55])
        out['sim_%d_next_uniform' % fs] = np.random.uniform(0, 1, 3)
        out['sim_%d_ts' % fs] = ts
        out['sim_%d_yval' % fs] = yval
        out['sim_%d_pars' % fs] = np.array([[p.mc, p.M, p.eta, p.m1, p.m2, p.ra, p.dec, p.iota, p.phi, p.psi, p.idx] for p in pars])
        out['sim_%d_meta' % fs] = np.array([fs, size, seed])
    path = os.path.join(os.path.dirname(OUT), 'indexing_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(cases), 'gen_bbh cases')


def posterior_golden():
    """overlap_tests (bbhMahoGANy.py:811-873) executed as written, on seeded samples, with scipy KDEs built the way
    make_contour_plot does (:790: gaussian_kde(dataset), dataset = np.array([x, y]))."""
    import warnings
    from scipy.stats import anderson_ksamp, gaussian_kde, ks_2samp
    lines = ref_lines('bbhMahoGANy.py')
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


def masses_golden():
    """gen_masses (gw_template_maker.py:289-370) executed as written for all four distributions on seeded legacy streams: 25 draws each,
    and the next uniform afterwards (stream position)."""
    ns = reference_namespace()
    out = {}
    for k, mdist in enumerate(('astro', 'hunt_constrain', 'gh', 'metric')):
        np.random.seed(40 + k)
        rows = []
        for _ in range(25):
            m12, mc, eta = ns['gen_masses'](5.0, 100.0, mdist)
            rows.append([float(np.asarray(m12)[0]), float(np.asarray(m12)[1]), float(np.asarray(mc).reshape(-1)[0]), float(np.asarray(eta).reshape(-1)[0])])
        out[mdist] = np.array(rows)
        out[mdist + '_next'] = np.random.uniform(0, 1, 2)
    dst = os.path.join(os.path.dirname(OUT), 'masses_golden.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, {k: v.shape for k, v in out.items()})


def lalinf_pars_golden():
    """data/get_lalinf_pars.py: the two conversion loops (`if do_m1m2:` ... `lalinf_pars = np.array([post_m1,post_m2])` and
    `if do_mc_M:` ... `lalinf_pars = np.array([post_mc,post_M])`) executed as written -- sympy's solve on the reference's own equations --
    on SUPPLIED posterior columns post_mc / post_q (the script reads them from a lalinference HDF5 file that is not in the repository);
    the pickle.dump lines that follow each block are not executed.  Line ranges are located by their first / last statement at run
    time.  Stores inputs and the two (2, n) arrays."""
    from sympy import Eq, Symbol, solve
    lines = ref_lines('data/get_lalinf_pars.py')

    def block(first, last):
        a = next(i for i, ln in enumerate(lines) if ln.startswith(first))
        b = next(i for i, ln in enumerate(lines) if i > a and ln.strip().startswith(last))
        return ''.join(lines[a:b + 1]), a + 1, b + 1

    rng = np.random.RandomState(12)
    post_mc = np.concatenate([rng.uniform(24.0, 34.0, 5), [30.0]])
    post_q = np.concatenate([rng.uniform(0.5, 1.0, 5), [1.0]])
    out = {'post_mc': post_mc, 'post_q': post_q}
    for flag, first, last, key in (('do_m1m2', 'if do_m1m2:', 'lalinf_pars = np.array([post_m1,post_m2])', 'm1_m2'),
                                   ('do_mc_M', 'if do_mc_M:', 'lalinf_pars = np.array([post_mc,post_M])', 'mc_M')):
        text, a, b = block(first, last)
        ns = {'np': np, 'Eq': Eq, 'Symbol': Symbol, 'solve': solve, 'post_mc': post_mc, 'post_q': post_q, flag: True, 'print': lambda *x: None}
        exec(compile(text, 'get_lalinf_pars.py:%d-%d' % (a, b), 'exec'), ns)
        out[key] = np.asarray(ns['lalinf_pars'], np.float64)
        out[key + '_lines'] = np.array([a, b])
    dst = os.path.join(os.path.dirname(OUT), 'lalinf_pars_golden.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, out['m1_m2'][:, :2], out['m1_m2_lines'], out['mc_M_lines'])


def posterior_mode_namespace(gan_post, post_mc, batch_size):
    """lalinf_post_waveform_maker.py's synthesiser functions executed from its own text.  Module globals the functions read (:58-68) are
    supplied: gan_post (n, 2) = the TRANSPOSED m1_m2 file (:65, :68), all_lalinf_posteriors['mc'] (:67), batch_size (:61), safe, verb, gw_tmp."""
    name = 'lalinf_post_waveform_maker.py'
    lines = ref_lines(name)
    ns = {'np': np, 'time': time, 'safe': 2, 'verb': False, 'gw_tmp': True, 'batch_size': batch_size, 'gan_post': gan_post,
          'all_lalinf_posteriors': {'mc': post_mc}, '_captured': []}
    for a, b in ((71, 87), (89, 116), (136, 163), (247, 290), (356, 475)):
        exec(compile(ref_text(lines, a, b), '%s:%d-%d' % (name, a, b), 'exec'), ns)
    assert 'lalsimulation.IMRPhenomPv2' in lines[514 - 1] and 'approximant)' in lines[532 - 1] and 'make_bbh(' in lines[564 - 1]
    src = (ref_text(lines, 477, 477) + ref_text(lines, 508, 513)
           + '    hp, hc = _supplied_fd_waveform(par, fs, T_obs)        # stands in for :514-532 (lalsimulation; :516 reads the undefined lal.PC_SIi)\n'
           + ref_text(lines, 534, 561)
           + '        ht_shift, hp_shift, hc_shift = orig_hp*_Fp + orig_hc*_Fc, orig_hp, orig_hc     # stands in for :564 (make_bbh -> :628, :647)\n'
           + ref_text(lines, 566, 585)
           + '    _captured.append((int(ref_idx), int(par.idx)))\n'
           + ref_text(lines, 587, 587))
    exec(compile(src, '%s:gen_bbh' % name, 'exec'), ns)
    exec(compile(ref_text(lines, 649, 746), '%s:649-746' % name, 'exec'), ns)
    return ns


def posterior_mode_golden():
    """gen_par / gen_bbh / sim_data of lalinf_post_waveform_maker.py executed as written over TWO consecutive blocks of one seeded legacy
    stream (main()'s nblock loop, :799-805, calls sim_data again without reseeding): per block the parameters in returned order, the
    cropped series, and the stream position afterwards.  The event-like row's gen_par call draws a randint that the gw_tmp branch then
    discards (:440-444 before :460-461), so block 2's idx values depend on it."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from oracle import synth_ref as S          # only for the SUPPLIED inputs (this project's chirp model, PSD curve and Fp/Fc constants)
    Fp, Fc = S.antenna_response(S.EVENT_TIME, S.RA, S.DEC, S.PSI)
    out = {'Fp': Fp, 'Fc': Fc}
    T_obs = 4
    for fs, n_post, size, batch_size, seed in ((256, 12, 9, 3907, 3), (512, 10, 40, 7, 4)):
        rng = np.random.RandomState(50 + fs)
        post_mc = rng.uniform(26.0, 32.0, n_post); post_q = rng.uniform(0.6, 1.0, n_post)
        heavy = post_mc * (1.0 + post_q) ** 0.2 / post_q ** 0.6
        m1_m2_file = np.array([post_q * heavy, heavy])              # the m1_m2 pickle: row 0 the lighter mass (get_lalinf_pars.py:52-67)
        gan_post = np.transpose(m1_m2_file)                         # :68
        ns = posterior_mode_namespace(gan_post, post_mc, batch_size)
        ns['_Fp'], ns['_Fc'] = Fp, Fc
        ns['_supplied_fd_waveform'] = lambda par_, fs_, T_: tuple(_FD(a) for a in S.chirp_fd(par_.m1, par_.m2, fs_ * T_ // 2 + 1, 1.0 / T_, iota=par_.iota, phi=par_.phi))
        N = fs * T_obs
        psd = S.analytic_psd(N // 2 + 1, 1.0 / T_obs)
        np.random.seed(seed)
        key = 'pm_%d_' % fs
        out[key + 'm1_m2_file'] = m1_m2_file; out[key + 'post_mc'] = post_mc; out[key + 'meta'] = np.array([fs, size, batch_size, seed])
        for blk in range(2):
            (ts, yval), pars = ns['sim_data'](fs, T_obs, psd, dets=['H1'], Nnoise=0, size=size, mdist='hunt_constrain', beta=[0.45, 0.55])
            out[key + 'ts_%d' % blk] = ts
            out[key + 'yval_%d' % blk] = yval
            out[key + 'pars_%d' % blk] = np.array([[p.mc, p.M, p.eta, p.m1, p.m2, p.ra, p.dec, p.iota, p.phi, p.psi, p.idx] for p in pars])
            st = np.random.get_state()
            out[key + 'next_uniform_%d' % blk] = np.random.uniform(0, 1, 3)
            np.random.set_state(st)                                 # peeking must not move the stream between the blocks
        out[key + 'ref_idx'] = np.array(ns['_captured'])
    path = os.path.join(os.path.dirname(OUT), 'posterior_mode_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
