"""GPU parity of the template synthesiser (gennet_amd/templates.py + csrc/synth.hip).

Against (i) the golden vectors produced by the reference's own numpy functions (tests/golden/synth_golden.npz) and
(ii) the fp64 oracle (oracle/synth_ref.py).  Bars: sample indices (ref_idx, crop placement, parameter draws) bit-exact;
everything downstream of h~(f) <= 1e-12 relative (fp64 FFT butterfly order differs from pocketfft's); the closed-form
chirp itself <= 1e-9 relative (phase ~1e3 rad amplifies last-bit differences of cbrt/pow/sincos between libm and the device).
"""
import os

import numpy as np
import pytest
import torch

from oracle import synth_ref as S

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'synth_golden.npz'))


def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def dev64(a):
    from gennet_amd.engine import device
    return torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(device())


def c2dev(z):
    return dev64(np.stack([z.real, z.imag], axis=-1))


@pytest.mark.parametrize("N", [16, 64, 1024, 4096, 8192, 16384])
def test_irfft_rfft_match_numpy(N):
    from gennet_amd import templates as T
    rng = np.random.RandomState(N)
    nb = 3
    X = rng.randn(nb, N // 2 + 1) + 1j * rng.randn(nb, N // 2 + 1)        # DC / Nyquist carry imaginary parts that must be ignored
    x = T.irfft(c2dev(X), N).cpu().numpy()
    assert rel(x, np.fft.irfft(X, N)) < 1e-13
    xr = rng.randn(nb, N)
    Xf = T.rfft(dev64(xr)).cpu().numpy()
    assert rel(Xf[..., 0] + 1j * Xf[..., 1], np.fft.rfft(xr)) < 1e-13
    # round trip at full size: irfft(rfft(x)) == x
    assert rel(T.irfft(T.rfft(dev64(xr)), N).cpu().numpy(), xr) < 1e-13


def test_whiten_data_matches_reference_golden():
    from gennet_amd import templates as T
    psd = G['wh_psd']
    out = T.whiten_data(G['wh_fd_in'], 4, 256, psd, 'fd')
    assert np.array_equal(out, G['wh_fd_out'])                              # one multiply per component: bit-exact
    assert rel(T.whiten_data(G['wh_td_in'], 4, 256, psd, 'td'), G['wh_td_out']) < 1e-12


def test_gen_noise_matches_reference_golden():
    from gennet_amd import templates as T
    np.random.seed(7)
    x = T.gen_noise(256, 4, G['noise_psd'].copy())
    assert rel(x, G['noise_out']) < 1e-12
    nxt = np.random.normal(0, 1, 3)                                          # stream position: exactly 2*Nf normals consumed
    np.random.seed(7)
    np.random.normal(0, 1, 513); np.random.normal(0, 1, 513)
    assert np.array_equal(nxt, np.random.normal(0, 1, 3))


def test_gen_noise_device_statistics():
    """Philox path: coloured noise with std ~ sqrt(psd*fs/2) (SURVEY Appendix D identity), DC-free, reproducible."""
    from gennet_amd import templates as T
    fs, Tobs = 1024, 4
    Nf = fs * Tobs // 2 + 1
    psd = np.full(Nf, 2.0e-3)
    x = T.gen_noise_device(fs, Tobs, psd, 64, seed=5).cpu().numpy()
    assert abs(x.std() - np.sqrt(2.0e-3 * fs / 2)) < 0.01 * np.sqrt(2.0e-3 * fs / 2)
    assert abs(x.mean()) < 5e-3
    x2 = T.gen_noise_device(fs, Tobs, psd, 64, seed=5).cpu().numpy()
    assert np.array_equal(x, x2)


@pytest.mark.parametrize("fs", [1024, 2048])
def test_chirp_kernel_matches_oracle_model(fs):
    from gennet_amd import templates as T
    Tobs = 4
    Nf = fs * Tobs // 2 + 1
    psd = S.analytic_psd(Nf, 1.0 / Tobs)
    syn = T.Synth(fs, Tobs, psd)
    m1 = np.array([36.0, 30.1, 45.3, 22.0]); m2 = np.array([29.0, 20.2, 24.9, 21.5])
    _, _, (hp, hc) = syn.series(m1, m2)
    hp = hp.cpu().numpy(); hc = hc.cpu().numpy()
    sc = S.whiten_scale(psd, fs); sc[0] = 0
    for b in range(4):
        rp, rc = S.chirp_fd(m1[b], m2[b], Nf, 1.0 / Tobs)
        assert rel(hp[b, :, 0] + 1j * hp[b, :, 1], rp * sc) < 1e-9
        assert rel(hc[b, :, 0] + 1j * hc[b, :, 1], rc * sc) < 1e-9
        assert np.all(hp[b, rp == 0] == 0)                                  # band edges identical


@pytest.mark.parametrize("fs", [256, 1024, 2048])
def test_align_crop_bit_exact_indexing_given_same_spectra(fs):
    """Downstream of h~(f): feed the ORACLE's whitened spectra to the device irFFT + align/crop; ref_idx must be exact and the
    crop must equal the oracle's to 1e-12."""
    from gennet_amd import templates as T
    Tobs = 4
    N = fs * Tobs; Nf = N // 2 + 1
    psd = S.analytic_psd(Nf, 1.0 / Tobs)
    syn = T.Synth(fs, Tobs, psd)
    np.random.seed(fs)
    pars = [S.gen_par(fs, Tobs) for _ in range(6)]
    sp = [S.chirp_fd(p.m1, p.m2, Nf, 1.0 / Tobs) for p in pars]
    whp = np.array([S.whiten_data(a, Tobs, fs, psd, 'fd') for a, _ in sp]); whc = np.array([S.whiten_data(b, Tobs, fs, psd, 'fd') for _, b in sp])
    hp_t = T.irfft(c2dev(whp), N); hc_t = T.irfft(c2dev(whc), N)
    Fp, Fc = S.antenna_response(S.EVENT_TIME, S.RA, S.DEC, S.PSI)
    assert (syn.Fp, syn.Fc) == (Fp, Fc)
    out, ref = syn.align(hp_t, hc_t, [p.idx for p in pars], int(1.5 * fs), fs, Fp, Fc)
    out = out.cpu().numpy(); ref = ref.cpu().numpy()
    for b, p in enumerate(pars):
        crop, ref_idx = S.align_crop(np.fft.irfft(whp[b], N), np.fft.irfft(whc[b], N), p.idx, fs, Fp, Fc)
        assert ref[b] == ref_idx
        assert rel(out[b], crop) < 1e-12


IG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'indexing_golden.npz'))


def test_align_crop_matches_reference_executed_fixtures():
    """The sample-indexing contract against what the REFERENCE'S OWN gen_bbh text produced (tests/golden/indexing_golden.npz:
    gw_template_maker.py:518-547, :553-575 executed with supplied spectra): ref_idx and the slice start exact, the cropped series
    <= 1e-12 -- chirps and the adversarial random spectra with negative / far-positive python slice starts, fs 256 ... 4096."""
    from gennet_amd import templates as T
    Fp, Fc = float(IG['Fp']), float(IG['Fc'])
    for n, (fs, idx, ref_idx, sidx, start) in enumerate(IG['bbh_meta']):
        fs = int(fs); N = 4 * fs
        psd = S.analytic_psd(N // 2 + 1, 0.25)
        syn = T.Synth(fs, 4, psd)
        assert (syn.Fp, syn.Fc) == (Fp, Fc)
        whp = T.whiten_data(IG['bbh_hp_%02d' % n], 4, fs, psd, 'fd'); whc = T.whiten_data(IG['bbh_hc_%02d' % n], 4, fs, psd, 'fd')
        hp_t = T.irfft(c2dev(whp[None]), N); hc_t = T.irfft(c2dev(whc[None]), N)
        out, ref = syn.align(hp_t, hc_t, [int(idx)], int(1.5 * fs), fs, Fp, Fc)
        assert int(ref.cpu()[0]) == ref_idx and int(ref.cpu()[0]) - int(idx) - syn.peak_off == start
        assert rel(out.cpu().numpy()[0], IG['bbh_crop_%02d' % n]) < 1e-12
        # the full windowed series (gen_bbh's return value): support and checksums of the reference's ts
        full, _ = syn.align(hp_t, hc_t, [int(idx)], 0, N, Fp, Fc)
        tw = T.tukey(int((16.0 / 15.0) * N / 2), alpha=1.0 / 8.0)
        win = np.zeros(N); a = int((N - tw.size) / 2); win[a:a + tw.size] = tw
        ts = full.cpu().numpy()[0] * win
        lo, hi, l1, l2 = IG['bbh_support_%02d' % n]
        nz = np.flatnonzero(ts)
        assert (nz.min() if nz.size else -1, nz.max() if nz.size else -1) == (lo, hi)
        assert abs(np.abs(ts).sum() - l1) <= 1e-11 * l1 and abs((ts ** 2).sum() - l2) <= 1e-11 * l2


@pytest.mark.parametrize("fs", [256, 1024])
def test_sim_data_matches_reference_executed_fixture(fs):
    """templates.sim_data against the reference's sim_data (gw_template_maker.py:632-740 executed as written on the same supplied
    waveform model): identical draws, shuffle and event-like row (exact), same host-RNG position afterwards, series to 1e-9 (the
    chirp evaluation: device pow/sincos vs libm, see the module docstring; downstream of the spectra the bar is 1e-12, test above)."""
    from gennet_amd import templates as T
    _, size, seed = [int(v) for v in IG['sim_%d_meta' % fs]]
    psd = S.analytic_psd(fs * 2 + 1, 0.25)
    np.random.seed(seed)
    (ts, yval), pars = T.sim_data(fs, 4, psd, ['H1'], 0, size, 'hunt_constrain', [0.45, 0.55])
    assert np.array_equal(np.random.uniform(0, 1, 3), IG['sim_%d_next_uniform' % fs])
    got = np.array([[p.mc, p.M, p.eta, p.m1, p.m2, p.ra, p.dec, p.iota, p.phi, p.psi, p.idx] for p in pars])
    assert np.array_equal(got, IG['sim_%d_pars' % fs]) and np.array_equal(yval, IG['sim_%d_yval' % fs])
    assert ts.shape == IG['sim_%d_ts' % fs].shape and rel(ts, IG['sim_%d_ts' % fs]) < 1e-9


@pytest.mark.parametrize("fs", [256, 512, 1024, 2048, 4096])
def test_fused_synthesiser_equals_separate_kernels_and_oracle(fs):
    """gn_synth_templates (spectrum, both inverse FFTs, arg-max, slide and crop in one workgroup, nothing through HBM) against the
    separate kernels (chirp -> irFFT x2 -> align/crop): ref_idx exact, crops <= 1e-12; against the fp64 oracle <= 1e-9 (chirp
    evaluation, see the module docstring); fp32 output = the fp64 output rounded once.  Masses include light systems whose spectrum
    reaches the Nyquist bin at low sample rates (the (-1)^n term) and heavy ones that end far below it."""
    from gennet_amd import templates as T
    Tobs = 4
    N = fs * Tobs
    psd = S.analytic_psd(N // 2 + 1, 1.0 / Tobs)
    syn = T.Synth(fs, Tobs, psd)
    np.random.seed(fs + 1)
    pars = [T.gen_par(fs, Tobs, mdist='hunt_constrain', beta=[0.45, 0.55]) for _ in range(37)]
    m1 = np.array([p.m1 for p in pars] + [6.0, 7.5, 90.0, 36.0]); m2 = np.array([p.m2 for p in pars] + [5.5, 6.0, 9.0, 29.0])
    lo, hi = T.convert_beta([0.45, 0.55], fs, Tobs)
    idx = np.array([p.idx for p in pars] + [lo, hi - 1, N // 2, 3])                   # the last slide start is far from the usual window
    a, ra = syn.templates(m1, m2, idx, g=1.7, fused=True)
    b, rb = syn.templates(m1, m2, idx, g=1.7, fused=False)
    assert torch.equal(ra, rb) and a.shape == b.shape == (41, fs) and a.dtype == torch.float64
    assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-12
    for k in range(41):
        assert rel(a[k].cpu().numpy(), b[k].cpu().numpy()) < 1e-11, k                # per row too (rows differ in scale)
    a32, r32 = syn.templates(m1, m2, idx, g=1.7, dtype=torch.float32)
    assert torch.equal(r32, ra) and torch.equal(a32, a.float())
    Fp, Fc = syn.Fp, syn.Fc
    for k in (0, 5, 36, 37, 38, 39, 40):
        p = S.bbhparams(0, m1[k] + m2[k], 0, m1[k], m2[k], S.RA, S.DEC, S.IOTA, S.PHI, S.PSI, int(idx[k]), None, None)
        crop, ref_idx = S.gen_bbh(fs, Tobs, psd, p, Fp, Fc)
        assert int(ra[k]) == ref_idx
        assert rel(a[k].cpu().numpy() / 1.7, crop) < 1e-9
    if fs == 256:
        hp, _ = S.chirp_fd(6.0, 5.5, N // 2 + 1, 1.0 / Tobs)
        assert hp[-1] != 0                                                             # the Nyquist-bin term was really exercised


def test_prior_drawn_inside_the_kernel():
    """gn_synth_templates_prior (BASELINE config 5): masses and idx drawn by each workgroup from a Philox stream.  (i) the returned
    parameters fed to the explicit-parameter kernel give bit-identical rows, (ii) every draw obeys the reference's hunt_constrain
    rule and the idx window, (iii) a batch is a pure function of (seed, counter) and disjoint counter ranges give different draws,
    (iv) the accepted (mc, q) follow the same distribution as the host rejection sampler (two-sample KS, alpha 1e-3)."""
    from scipy.stats import ks_2samp
    from gennet_amd import templates as T
    fs, Tobs = 1024, 4
    psd = S.analytic_psd(fs * 2 + 1, 0.25)
    syn = T.Synth(fs, Tobs, psd)
    lo, hi = T.convert_beta([0.45, 0.55], fs, Tobs)
    nb = 4096
    out, labels, ref, mm, idx = syn.templates_prior(nb, seed=7, counter=0, idx_lo=lo, idx_hi=hi, g=1.3, dtype=torch.float64, want_params=True)
    m = mm.cpu().numpy(); ix = idx.cpu().numpy(); lab = labels.cpu().numpy()
    again, ref2 = syn.templates(m[:, 0], m[:, 1], ix, g=1.3)
    assert torch.equal(out, again) and torch.equal(ref, ref2)
    m1, m2 = m[:, 0], m[:, 1]
    eta = m1 * m2 / (m1 + m2) ** 2; mc = (m1 + m2) * eta ** 0.6
    assert np.all((m1 + m2 < 100) & (m1 > 5) & (m2 > 5) & (m1 >= m2) & (m2 / m1 >= 0.5) & (mc >= 20) & (mc <= 35))
    assert ix.min() >= lo and ix.max() < hi and len(np.unique(ix)) > 0.8 * (hi - lo)
    assert np.allclose(lab[:, 0], mc, rtol=1e-6) and np.allclose(lab[:, 1], m2 / m1, rtol=1e-6)
    o2, l2, _ = syn.templates_prior(64, 7, 0, lo, hi, g=1.3, dtype=torch.float64)
    assert torch.equal(o2, out[:64]) and torch.equal(l2, labels[:64])                       # same (seed, counter): same templates
    o3, _, _ = syn.templates_prior(64, 7, 64 * T.Synth.PRIOR_TRIALS, lo, hi, g=1.3, dtype=torch.float64)
    assert torch.equal(o3, out[64:128])                                                      # counter ranges tile: b-th template of a later call
    o4, _, _ = syn.templates_prior(64, 8, 0, lo, hi, g=1.3, dtype=torch.float64)
    assert not torch.equal(o4, o2)
    host = T.OnlineBank(fs, Tobs, psd, seed=3, prior='host')
    h1, h2 = host.draw_masses(nb)
    heta = h1 * h2 / (h1 + h2) ** 2
    assert ks_2samp(mc, (h1 + h2) * heta ** 0.6).pvalue > 1e-3 and ks_2samp(m2 / m1, h2 / h1).pvalue > 1e-3
    assert ks_2samp(ix, host.rng.randint(lo, hi, nb)).pvalue > 1e-3


def test_fused_synthesiser_empty_and_unsupported_lengths():
    from gennet_amd import _lib, templates as T
    syn = T.Synth(256, 4, S.analytic_psd(513, 0.25))
    out, ref = syn.templates(np.zeros(0), np.zeros(0), np.zeros(0, np.int32))
    assert out.shape == (0, 256) and ref.shape == (0,)
    syn64 = T.Synth(64, 4, np.ones(129))                                               # N = 256: below the fused kernel's range -> separate kernels
    o, r = syn64.templates([30.0], [25.0], [130])
    assert o.shape == (1, 64) and torch.isfinite(o).all()
    with pytest.raises(_lib.GennetHipError):
        syn64.templates([30.0], [25.0], [130], fused=True)


def test_align_python_slice_semantics_edge_cases():
    """start = ref_idx - idx - 11 < 0 (python slices from the end) and start near N (zero fill), on crafted series."""
    from gennet_amd import templates as T
    fs, Tobs = 64, 4
    N = fs * Tobs
    psd = np.ones(N // 2 + 1)
    syn = T.Synth(fs, Tobs, psd)
    rng = np.random.RandomState(2)
    hp = rng.randn(3, N); hc = rng.randn(3, N)
    peaks = [5, N - 3, 100]                    # rolled-frame peak positions
    for b, pk in enumerate(peaks):
        hp[b, (pk + fs) % N] = 50.0            # roll by -fs moves sample s to s - fs
    idxs = [40, 10, 100]
    out, ref = syn.align(dev64(hp), dev64(hc), idxs, 0, N, 0.3, -0.7)
    out = out.cpu().numpy(); ref = ref.cpu().numpy()
    for b in range(3):
        hp_r = np.roll(hp[b], -fs); hc_r = np.roll(hc[b], -fs)
        ri = int(np.argmax(hp_r ** 2 + hc_r ** 2))
        assert ref[b] == ri == peaks[b]
        tmp = (hp_r * 0.3 + hc_r * -0.7)[int(ri - idxs[b] - 11):]
        ts = np.zeros(N); ts[:min(len(tmp), N)] = tmp[:N]
        assert np.array_equal(out[b], ts)


def test_argmax_first_maximum_tie_break():
    from gennet_amd import templates as T
    fs, Tobs = 64, 4
    N = fs * Tobs
    syn = T.Synth(fs, Tobs, np.ones(N // 2 + 1))
    hp = np.zeros((1, N)); hc = np.zeros((1, N))
    hp[0, 200] = 2.0; hp[0, 90] = -2.0; hc[0, 150] = 2.0                    # three equal maxima of hp^2 + hc^2
    _, ref = syn.align(dev64(hp), dev64(hc), [0], 0, N, 1.0, 1.0)
    assert ref.cpu().numpy()[0] == int(np.argmax(np.roll(hp[0], -fs) ** 2 + np.roll(hc[0], -fs) ** 2))


def test_sim_data_seeded_matches_oracle():
    """Whole synthesiser, seed 1 (gw_template_maker.py:128): identical parameter draws (exact), identical shuffle, event-like
    template last; time series to 1e-9 (limited by the chirp evaluation, see module docstring)."""
    from gennet_amd import templates as T
    fs, Tobs, size = 1024, 4, 12
    psd = S.analytic_psd(fs * Tobs // 2 + 1, 1.0 / Tobs)
    np.random.seed(1)
    (ts_ref, y_ref), par_ref = S.sim_data(fs, Tobs, psd, size, 'hunt_constrain', (0.45, 0.55))
    next_ref = np.random.uniform(0, 1, 2)
    np.random.seed(1)
    (ts, y), par = T.sim_data(fs, Tobs, psd, ['H1'], 0, size, 'hunt_constrain', [0.45, 0.55])
    assert np.array_equal(np.random.uniform(0, 1, 2), next_ref)             # same number of host RNG draws consumed
    assert ts.shape == (size, 1, fs) and ts.dtype == np.float64 and np.array_equal(y, y_ref)
    for a, b in zip(par, par_ref):
        assert (a.mc, a.M, a.eta, a.m1, a.m2, a.idx, a.ra, a.dec, a.iota, a.phi, a.psi) == (b.mc, b.M, b.eta, b.m1, b.m2, b.idx, b.ra, b.dec, b.iota, b.phi, b.psi)
    assert (par[-1].m1, par[-1].m2, par[-1].idx) == (36.0, 29.0, 2048)
    assert rel(ts, ts_ref) < 1e-9


def test_gen_bbh_single_template_surface():
    from gennet_amd import templates as T
    fs, Tobs = 512, 4
    N = fs * Tobs
    psd = S.analytic_psd(N // 2 + 1, 1.0 / Tobs)
    np.random.seed(4)
    p = T.gen_par(fs, Tobs, mdist='hunt_constrain', beta=[0.45, 0.55])
    ts, hp, hc, ts2 = T.gen_bbh(fs, Tobs, psd, ['H1'], [0.45, 0.55], p)
    assert ts.shape == hp.shape == hc.shape == (1, N) and ts is ts2
    crop, _ = S.gen_bbh(fs, Tobs, psd, p)
    assert rel(ts[0, int(1.5 * fs):int(2.5 * fs)], crop) < 1e-9
    Fp, Fc = T.antenna_response(float(T.event_time), p.ra, p.dec, p.psi)
    mid = slice(int(1.5 * fs), int(2.5 * fs))
    assert rel(ts[0, mid], hp[0, mid] * Fp + hc[0, mid] * Fc) < 1e-13
    ht, _, _ = T.make_bbh(hp, hc, fs, p.ra, p.dec, p.psi, 'H1')
    assert np.array_equal(ht, hp * Fp + hc * Fc)


def test_full_size_property_envelope_peak_lands_on_idx():
    """BASELINE size (fs = 2048, N = 8192), size-independent property: after the slide the envelope hp^2 + hc^2 peaks exactly at
    crop-relative sample idx + 11 - 1.5 fs ([830, 1239) at 2 kHz)."""
    from gennet_amd import templates as T
    fs, Tobs = 2048, 4
    psd = S.analytic_psd(fs * Tobs // 2 + 1, 1.0 / Tobs)
    syn = T.Synth(fs, Tobs, psd)
    np.random.seed(9)
    pars = [T.gen_par(fs, Tobs, mdist='hunt_constrain', beta=[0.45, 0.55]) for _ in range(256)]
    idx = np.array([p.idx for p in pars])
    hp_t, hc_t, _ = syn.series([p.m1 for p in pars], [p.m2 for p in pars])
    c0 = int(1.5 * fs)
    a, ref = syn.align(hp_t, hc_t, idx, c0, fs, 1.0, 0.0)
    b, _ = syn.align(hp_t, hc_t, idx, c0, fs, 0.0, 1.0)
    env = (a * a + b * b).cpu().numpy()
    assert np.array_equal(env.argmax(axis=1), idx + 11 - c0)
    assert idx.min() >= 3891 and idx.max() < 4300
    ts, ref2 = syn.templates([p.m1 for p in pars], [p.m2 for p in pars], idx)
    assert torch.equal(ref, ref2) and ts.shape == (256, fs) and torch.isfinite(ts).all()


@pytest.mark.parametrize("fs", [256, 512])
def test_posterior_driven_mode_matches_reference_execution(fs):
    """lalinf_post_waveform_maker mode against the reference's own gen_par / gen_bbh / sim_data executed over two
    consecutive blocks of one seeded stream (tests/golden/posterior_mode_golden.npz): parameters, idx draws, shuffle and the event-like row
    exact, stream position after each block exact, series <= 1e-9 (the closed-form chirp on the device against libm, as everywhere)."""
    from gennet_amd import templates as T
    PM = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'posterior_mode_golden.npz'))
    key = 'pm_%d_' % fs
    _, size, batch_size, seed = [int(v) for v in PM[key + 'meta']]
    f = PM[key + 'm1_m2_file']
    N = fs * 4
    psd = S.analytic_psd(N // 2 + 1, 0.25)
    m1, m2 = T.m1m2_from_mc_q(PM[key + 'post_mc'], f[0] / f[1])
    assert np.allclose(m1, f[1], rtol=1e-13) and np.allclose(m2, f[0], rtol=1e-13)      # the file's row 1 is the heavier mass
    np.random.seed(seed)
    for blk in range(2):
        (ts, y), pars = T.sim_data_posterior(fs, 4, psd, f[1], f[0], PM[key + 'post_mc'], size=size, batch_size=batch_size)
        got = np.array([[p.mc, p.M, p.eta, p.m1, p.m2, p.ra, p.dec, p.iota, p.phi, p.psi, p.idx] for p in pars])
        assert np.array_equal(got, PM[key + 'pars_%d' % blk])
        assert (pars[-1].m1, pars[-1].m2, pars[-1].idx) == (36.0, 29.0, N // 2 - 4)
        assert ts.shape == PM[key + 'ts_%d' % blk].shape and np.array_equal(y, PM[key + 'yval_%d' % blk])
        assert rel(ts, PM[key + 'ts_%d' % blk]) < 1e-9
        st = np.random.get_state()
        assert np.array_equal(np.random.uniform(0, 1, 3), PM[key + 'next_uniform_%d' % blk])
        np.random.set_state(st)


# ---------------------------------------------------------------------------------------------------------------------
# fused coloured noise (BASELINE configs[4]): gen_noise -> whiten_data('td') -> crop in one kernel (csrc/noise_chain.h)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fs", [256, 2048])
def test_fused_noise_matches_the_reference_composition(fs):
    """gn_noise_whitened fed the SAME normals against whiten_data(gen_noise(...), 'td') as the reference's own two functions produced it
    (tests/golden/synth_golden.npz: nchain_*), whole series (crop = [0, N)) and the central second, <= 1e-12; a PSD with zero bins."""
    from gennet_amd import templates as T
    psd = G['nchain_%d_psd' % fs]
    N = 4 * fs
    ns = T.NoiseSynth(fs, 4, psd)
    ref = G['nchain_%d_out' % fs]
    nrm = G['nchain_%d_normals' % fs].reshape(1, -1)
    full = ns.draw(1, normals=nrm, dtype=torch.float64, crop=(0, N)).cpu().numpy()[0]
    assert rel(full, ref) < 1e-12
    mid = ns.draw(1, normals=nrm, dtype=torch.float64).cpu().numpy()[0]
    assert np.array_equal(mid, full[int(1.5 * fs):int(2.5 * fs)])
    # three rows at once, fp32 output added to a template row: out = float(add + noise)
    nrm3 = np.stack([nrm[0], -nrm[0], nrm[0][::-1]])
    add = dev64(np.random.RandomState(1).randn(3, fs))
    o32 = ns.draw(3, normals=nrm3, add=add, dtype=torch.float32)
    o64 = ns.draw(3, normals=nrm3, dtype=torch.float64)
    assert torch.equal(o32, (add + o64).float())
    assert rel(o64[0].cpu().numpy(), mid) == 0 and rel(o64[1].cpu().numpy(), -mid) < 1e-13          # the chain is linear in the normals
    assert ns.draw(0).shape == (0, fs)


@pytest.mark.parametrize("fs", [1024, 4096])
def test_fused_noise_philox_stream_against_the_oracle_chain(fs):
    """Philox mode: the normals the kernel drew (normals_out) pushed through the numpy oracle (gen_noise's spectrum construction, irfft,
    whiten_data('td')) reproduce its output to 1e-12; the stream is a pure function of (seed, counter, row); statistics of the whitened
    series (std ~ sqrt(mean tukey^2) = 0.96 over the whole series, ~1 inside the flat part of the window)."""
    from gennet_amd import templates as T
    N = 4 * fs; Nf = N // 2 + 1
    psd = S.analytic_psd(Nf, 0.25)
    ns = T.NoiseSynth(fs, 4, psd)
    out, nrm = ns.draw(4, seed=5, counter=1000, dtype=torch.float64, crop=(0, N), want_normals=True)
    out = out.cpu().numpy(); nrm = nrm.cpu().numpy()
    amp = np.sqrt(0.25 * 4 * psd); amp[psd == 0.0] = 0.0
    for b in range(4):
        re = amp * nrm[b, :Nf]; im = amp * nrm[b, Nf:]
        re[0] = 0.0; im[0] = 0.0
        x = N * np.fft.irfft(re + 1j * im) * 0.25
        assert rel(out[b], S.whiten_data(x, 4, fs, psd, 'td')) < 1e-12
    assert abs(nrm.mean()) < 0.02 and abs(nrm.std() - 1.0) < 0.02
    again = ns.draw(4, seed=5, counter=1000, dtype=torch.float64, crop=(0, N)).cpu().numpy()
    assert np.array_equal(out, again)
    shifted = ns.draw(3, seed=5, counter=1000 + Nf, dtype=torch.float64, crop=(0, N)).cpu().numpy()
    assert np.array_equal(shifted, out[1:])                            # row b of a launch = row 0 of a launch whose counter is b * Nf further
    assert not np.array_equal(ns.draw(1, seed=6, counter=1000, dtype=torch.float64, crop=(0, N)).cpu().numpy()[0], out[0])
    mid = out[:, int(1.5 * fs):int(2.5 * fs)]
    live = psd > 0                                                      # bins below the PSD floor carry no noise: expected variance = live fraction
    assert abs(mid.std() - np.sqrt(live[1:].mean())) < 0.05


@pytest.mark.parametrize("fs", [512, 2048, 4096])
def test_template_plus_noise_in_one_launch_equals_the_two_kernels(fs):
    """gn_synth_templates_noise (template and coloured noise by the same workgroup, crop waiting in registers) against the template kernel
    writing fp64 rows + gn_noise_whitened adding to them, same Philox stream: bit-identical, fp64 and fp32, given parameters and prior mode."""
    from gennet_amd import templates as T
    N = 4 * fs; Nf = N // 2 + 1
    psd = S.analytic_psd(Nf, 0.25)
    syn = T.Synth(fs, 4, psd); ns = T.NoiseSynth(fs, 4, psd)
    np.random.seed(fs)
    pars = [T.gen_par(fs, 4, mdist='hunt_constrain', beta=[0.45, 0.55]) for _ in range(5)]
    m1 = [p.m1 for p in pars]; m2 = [p.m2 for p in pars]; idx = [p.idx for p in pars]
    g = 817.98
    ts, ref = syn.templates(m1, m2, idx, g=g, dtype=torch.float64)
    two64 = ns.draw(5, seed=9, counter=77, add=ts, dtype=torch.float64)
    two32 = ns.draw(5, seed=9, counter=77, add=ts, dtype=torch.float32)
    one64, ref1 = syn.templates_noise(m1, m2, idx, ns, 9, 77, g=g, dtype=torch.float64)
    one32, _ = syn.templates_noise(m1, m2, idx, ns, 9, 77, g=g, dtype=torch.float32)
    assert torch.equal(ref, ref1) and torch.equal(one64, two64) and torch.equal(one32, two32)
    assert float((two64 - ts).std()) > 0.5                              # the noise really is there
    lo, hi = T.convert_beta([0.45, 0.55], fs, 4)
    p_ts, p_lab, p_ref = syn.templates_prior(6, 3, 0, lo, hi, g=g, dtype=torch.float64)
    p_two = ns.draw(6, seed=4, counter=11, add=p_ts, dtype=torch.float32)
    p_one, p_lab1, p_ref1 = syn.templates_prior(6, 3, 0, lo, hi, g=g, dtype=torch.float32, noise=ns, noise_seed=4, noise_counter=11)
    assert torch.equal(p_one, p_two) and torch.equal(p_lab, p_lab1) and torch.equal(p_ref, p_ref1)


def test_online_bank_config5_fs4096():
    """BASELINE config 5 (srate 4096): on-GPU synthesis inside the loop.  Size-independent properties at full size: labels obey the
    prior box, the envelope peak lands inside the crop-relative window [1649, 2468) + peak offset, draws are
    reproducible per seed and differ across seeds (rank streams), coloured + whitened noise has ~unit variance."""
    from gennet_amd import templates as T
    fs, Tobs = 4096, 4
    psd = S.analytic_psd(fs * Tobs // 2 + 1, 1.0 / Tobs)
    ob = T.OnlineBank(fs, Tobs, psd, gw_norm_constant=1.0, seed=3)
    x, y = ob.draw(64)
    assert x.shape == (64, fs) and x.dtype == torch.float32 and y.shape == (64, 2) and torch.isfinite(x).all()
    yl = y.cpu().numpy()
    assert np.all((yl[:, 0] >= 20) & (yl[:, 0] <= 35) & (yl[:, 1] >= 0.5) & (yl[:, 1] <= 1.0))
    peak = x.abs().argmax(dim=1).cpu().numpy()
    assert np.all((peak >= 7782 + 11 - 6144 - 40) & (peak < 8601 + 11 - 6144 + 40))
    x2, y2 = T.OnlineBank(fs, Tobs, psd, seed=3).draw(64)
    assert torch.equal(x, x2) and torch.equal(y, y2)
    x3, _ = T.OnlineBank(fs, Tobs, psd, seed=4).draw(64)
    assert not torch.equal(x, x3)
    obn = T.OnlineBank(fs, Tobs, psd, seed=3, noise='coloured')
    xn, yn = obn.draw(64)
    assert torch.equal(yn, y)                                # same prior stream: the same templates underneath
    resid = (xn - x).cpu().numpy()
    assert abs(resid.std() - 0.99) < 0.06                    # whitened coloured noise has unit variance inside the flat part of the Tukey window
    xn2, _ = T.OnlineBank(fs, Tobs, psd, seed=3, noise='coloured').draw(64)
    assert torch.equal(xn, xn2)
    xh, _ = T.OnlineBank(fs, Tobs, psd, seed=3, noise='coloured', prior='host').draw(8)      # host-drawn parameters: template kernel + noise kernel
    assert xh.shape == (8, fs) and torch.isfinite(xh).all()


def test_pe_train_step_online_runs_at_fs2048():
    from gennet_amd import bbh, engine, templates as T
    fs = 2048
    psd = S.analytic_psd(fs * 4 // 2 + 1, 0.25)
    ob = T.OnlineBank(fs, 4, psd, seed=1, noise='white')
    pe = bbh.signal_pe_model(fs)
    pe.compile(loss='mean_squared_error', optimizer=engine.Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])
    l0 = bbh.pe_train_step_online(pe, ob, 16)
    l1 = bbh.pe_train_step_online(pe, ob, 16)
    assert len(l0) == 5 and np.isfinite(l0).all() and np.isfinite(l1).all()
