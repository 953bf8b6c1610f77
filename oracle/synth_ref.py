"""CPU oracle: numpy (fp64) restatement of the template synthesiser, BBH_version/gw_template_maker.py.

TEST INFRASTRUCTURE ONLY (see keras_ref.py header).

Pinned by tests/test_oracle_synth.py against tests/golden/synth_golden.npz, which tests/golden/make_golden.py
produced by executing the reference's own pure-numpy functions: tukey (:87-113), convert_beta (:133-159), gen_noise
(:161-193), whiten_data (:243-286), hunt_constrain sampler (:329-338).

PARITY UNPINNED for the two LALSuite calls (lal / lalsimulation / pylal: third-party, version not pinned by the reference,
absent here, no tests in the reference):
  * SimInspiralChooseFDWaveform(IMRPhenomPv2) (:507-516) -> `chirp_fd` below is THIS PROJECT'S OWN closed-form non-spinning
    inspiral-merger-ringdown model in the phenomenological (Ajith et al. 2008 "PhenomA") functional form; it is not
    LAL-identical and is not claimed to be.  Everything downstream of h~(f) is pinned.
  * pylal.antenna.response (:612) -> `antenna_response` restates the textbook detector-tensor formula for LHO; for the
    fixed arguments of the reference it yields two constants (Fp, Fc).
"""
import numpy as np

SAFE = 2                      # gw_template_maker.py:54
MTSUN_SI = 4.925491025543576e-06
MPC_SEC = 3.085677581491367e22 / 299792458.0      # one megaparsec in seconds
EVENT_TIME = 1126259462.0     # gw_template_maker.py:62
RA, DEC, IOTA, PHI, PSI = 2.21535724066, -1.23649695537, 2.5, 1.5, 1.75   # :433-437
F_LOW, DIST_MPC = 40.0, 410.0  # :495, :500


class bbhparams(object):
    """gw_template_maker.py:69-85."""

    def __init__(self, mc, M, eta, m1, m2, ra, dec, iota, phi, psi, idx, snr, SNR):
        self.mc, self.M, self.eta, self.m1, self.m2 = mc, M, eta, m1, m2
        self.ra, self.dec, self.iota, self.phi, self.psi = ra, dec, iota, phi, psi
        self.idx, self.snr, self.SNR = idx, snr, SNR


# ------------------------------------------------------------------------------------------------ windows / indices
def tukey(M, alpha=0.5):
    """:87-113 (scipy's Tukey): cosine tapers of floor(alpha*(M-1)/2)+1 samples each side, ones in between."""
    n = np.arange(M)
    width = int(np.floor(alpha * (M - 1) / 2.0))
    left = n[:width + 1]
    right = n[M - width - 1:]
    w = np.ones(M)
    w[:width + 1] = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * left / alpha / (M - 1))))
    w[M - width - 1:] = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * right / alpha / (M - 1))))
    return w


def convert_beta(beta, fs, T_obs):
    """:133-159: beta fractions of the central window -> sample indices of the safe (x2) window."""
    nb = np.array([beta[0] + 0.5 * SAFE - 0.5, beta[1] + 0.5 * SAFE - 0.5]) / SAFE
    return int(T_obs * fs * nb[0]), int(T_obs * fs * nb[1])


# ------------------------------------------------------------------------------------------------ whitening / noise
def whiten_scale(psd, sample_rate):
    """sqrt(2*invpsd/fs) with invpsd = 0 where psd <= 0 (:273-276)."""
    inv = np.zeros(psd.size)
    pos = psd > 0.0
    inv[pos] = 1.0 / psd[pos]
    return np.sqrt(2.0 * inv / sample_rate)


def whiten_data(data, duration, sample_rate, psd, flag='td'):
    """:243-286."""
    if flag == 'td':
        xf = np.fft.rfft(tukey(duration * sample_rate, alpha=1.0 / 8.0) * data)
    else:
        xf = np.array(data, dtype=complex)
    xf = xf * whiten_scale(psd, sample_rate)
    xf[0] = 0.0
    return np.fft.irfft(xf) if flag == 'td' else xf


def gen_noise(fs, T_obs, psd, normals=None):
    """:161-193.  `normals` (2*Nf standard normals: re block then im block) replaces the two np.random.normal calls."""
    N = T_obs * fs
    Nf = N // 2 + 1
    df = 1.0 / T_obs
    amp = np.sqrt(0.25 * T_obs * psd)
    amp[psd == 0.0] = 0.0
    if normals is None:
        re = amp * np.random.normal(0, 1, Nf)
        im = amp * np.random.normal(0, 1, Nf)
    else:
        re, im = amp * normals[:Nf], amp * normals[Nf:2 * Nf]
    re[0] = 0.0
    im[0] = 0.0
    return N * np.fft.irfft(re + 1j * im) * df


# ------------------------------------------------------------------------------------------------ parameter draws
def gen_masses(m_min=5.0, M_max=100.0, mdist='hunt_constrain'):
    """:289-339: log-uniform component masses by rejection; 'hunt_constrain' adds q >= 0.5 and 20 <= mc <= 35.
    Two uniforms are consumed per trial (np.random.uniform(0,1,2))."""
    log_m_max = np.log(M_max - m_min)
    while True:
        m12 = np.exp(np.log(m_min) + np.random.uniform(0, 1, 2) * (log_m_max - np.log(m_min)))
        eta = m12[0] * m12[1] / (m12[0] + m12[1]) ** 2
        mc = np.sum(m12) * eta ** (3.0 / 5.0)
        ok = (np.sum(m12) < M_max) and np.all(m12 > m_min) and (m12[0] >= m12[1])
        if mdist == 'hunt_constrain':
            ok = ok and (m12[1] / m12[0] >= 0.5) and (20.0 <= mc <= 35.0)
        elif mdist != 'astro':
            raise ValueError('mass distribution %r is not on the hot path' % mdist)
        if ok:
            return m12, mc, eta


def gen_par(fs, T_obs, mdist='hunt_constrain', beta=(0.45, 0.55), gw_tmp=False):
    """:372-460.  RNG order: masses (2 uniforms per trial), then five rand() draws (iota, psi, phi, ra, dec) whose values are
    overwritten by constants, then one randint(low, high) unless low == high."""
    m12, mc, eta = gen_masses(5.0, 100.0, mdist)
    M = np.sum(m12)
    for _ in range(5):
        np.random.rand()
    if gw_tmp:
        beta = [0.5, 0.5]
    lo, hi = convert_beta(beta, fs, T_obs)
    idx = lo if lo == hi else int(np.random.randint(lo, hi, 1)[0])
    if gw_tmp:
        m1, m2 = 36.0, 29.0
        eta = m1 * m2 / (m1 + m2) ** 2
        M = m1 + m2
        mc = M * eta ** (3.0 / 5.0)
        return bbhparams(mc, M, eta, m1, m2, RA, DEC, IOTA, PHI, PSI, idx, None, None)
    return bbhparams(mc, M, eta, m12[0], m12[1], RA, DEC, IOTA, PHI, PSI, idx, None, None)


# ------------------------------------------------------------------------------------------------ detector response
def gmst_rad(gps):
    """Greenwich mean sidereal time (IAU 1982 polynomial) for a GPS time; GPS-UTC = 17 s at the event epoch."""
    utc = gps - 17.0
    jd = utc / 86400.0 + 2444244.5            # GPS epoch 1980-01-06 00:00 UTC = JD 2444244.5
    t = (jd - 2451545.0) / 36525.0
    sec = 67310.54841 + (876600.0 * 3600.0 + 8640184.812866) * t + 0.093104 * t * t - 6.2e-6 * t ** 3
    return (sec % 86400.0) * (2.0 * np.pi / 86400.0)


LHO_X = np.array([-0.22389266154, 0.79983062746, 0.55690487831])
LHO_Y = np.array([-0.91397818574, 0.02609403989, -0.40492342125])


def antenna_response(gps, ra, dec, psi):
    """F+ and Fx of LIGO Hanford for a source at (ra, dec), polarisation psi (radians): D_ij = (x_i x_j - y_i y_j)/2
    contracted with the polarisation tensors built from the wave-frame basis (Anderson et al. 2001 convention)."""
    D = 0.5 * (np.outer(LHO_X, LHO_X) - np.outer(LHO_Y, LHO_Y))
    gha = gmst_rad(gps) - ra
    cg, sg, cd, sd, cp, sp = np.cos(gha), np.sin(gha), np.cos(dec), np.sin(dec), np.cos(psi), np.sin(psi)
    X = np.array([-cp * sg - sp * cg * sd, -cp * cg + sp * sg * sd, sp * cd])
    Y = np.array([sp * sg - cp * cg * sd, sp * cg + cp * sg * sd, cp * cd])
    Fp = X @ D @ X - Y @ D @ Y
    Fc = X @ D @ Y + Y @ D @ X
    return Fp, Fc


# ------------------------------------------------------------------------------------------------ FD chirp (own model)
_PHENOM_F = {  # pi*M*f_k = a*eta^2 + b*eta + c
    'merg': (2.9740e-1, 4.4810e-2, 9.5560e-2), 'ring': (5.9411e-1, 8.9794e-2, 1.9111e-1),
    'sigma': (5.0801e-1, 7.7515e-2, 2.2369e-2), 'cut': (8.4845e-1, 1.2848e-1, 2.7299e-1)}
_PHENOM_PSI = {  # psi_k = (x*eta^2 + y*eta + z)/eta, multiplying (pi*M*f)^((k-5)/3)
    0: (1.7516e-1, 7.9483e-2, -7.2390e-2), 2: (-5.1571e1, -1.7595e1, 1.3253e1), 3: (6.5866e2, 1.7803e2, -1.5972e2),
    4: (-3.9031e3, -7.7493e2, 8.8195e2), 6: (-2.4874e4, -1.4892e3, 4.4588e3), 7: (2.5196e4, 3.3970e2, -3.9573e3)}
PSI_ORDERS = (0, 2, 3, 4, 6, 7)


def chirp_coeffs(m1, m2, dist_mpc):
    """Per-template constants of the chirp model (the same numbers the HIP kernel receives)."""
    M = m1 + m2
    eta = m1 * m2 / (M * M)
    piM = np.pi * M * MTSUN_SI
    fk = {k: (a * eta * eta + b * eta + c) / piM for k, (a, b, c) in _PHENOM_F.items()}
    psi = np.array([(x * eta * eta + y * eta + z) / eta for (x, y, z) in (_PHENOM_PSI[k] for k in PSI_ORDERS)])
    amp0 = (M * MTSUN_SI) ** (5.0 / 6.0) / (dist_mpc * MPC_SEC * np.pi ** (2.0 / 3.0)) * np.sqrt(5.0 * eta / 24.0) * fk['merg'] ** (-7.0 / 6.0)
    # time shift that puts the stationary-phase time of f_ring at t = 0:  t0 = -(1/2pi) d/df [sum_k psi_k (piM f)^((k-5)/3)] at f_ring
    v = (piM * fk['ring']) ** (1.0 / 3.0)
    dsum = sum(p * ((k - 5) / 3.0) * v ** (k - 5) / fk['ring'] for p, k in zip(psi, PSI_ORDERS))
    t0 = -dsum / (2.0 * np.pi)
    return {'piM': piM, 'f_merg': fk['merg'], 'f_ring': fk['ring'], 'sigma': fk['sigma'], 'f_cut': fk['cut'], 'psi': psi, 'amp0': amp0, 't0': t0}


def chirp_fd(m1, m2, Nf, df, f_low=F_LOW, dist_mpc=DIST_MPC, iota=IOTA, phi=PHI):
    """h~+(f), h~x(f) on the grid f = k*df, k = 0..Nf-1: zero below f_low, at DC and from f_cut upwards."""
    c = chirp_coeffs(m1, m2, dist_mpc)
    f = np.arange(Nf) * df
    live = (f >= f_low) & (f > 0) & (f < c['f_cut'])
    fs_ = np.where(live, f, 1.0)
    v = np.cbrt(c['piM'] * fs_)
    v2 = v * v
    iv = 1.0 / v
    pw = {0: iv ** 5, 2: iv ** 3, 3: iv * iv, 4: iv, 6: v, 7: v2}
    phase = 2.0 * np.pi * fs_ * c['t0'] + 2.0 * phi
    for p, k in zip(c['psi'], PSI_ORDERS):
        phase = phase + p * pw[k]
    r = fs_ / c['f_merg']
    lor = (1.0 / (2.0 * np.pi)) * c['sigma'] / ((fs_ - c['f_ring']) ** 2 + 0.25 * c['sigma'] ** 2)
    wnorm = (np.pi * c['sigma'] / 2.0) * (c['f_ring'] / c['f_merg']) ** (-2.0 / 3.0)
    shape = np.where(fs_ < c['f_merg'], r ** (-7.0 / 6.0), np.where(fs_ < c['f_ring'], r ** (-2.0 / 3.0), wnorm * lor))
    amp = np.where(live, c['amp0'] * shape, 0.0)
    h = amp * (np.cos(phase) - 1j * np.sin(phase))
    ci = np.cos(iota)
    return 0.5 * (1.0 + ci * ci) * h, (-1j * ci) * h


# ------------------------------------------------------------------------------------------------ gen_bbh / sim_data
PEAK_OFFSET = 11              # gw_template_maker.py:554 ("use 21 if sampling at 2kHz" -- a config field here)


def align_crop(hp_t, hc_t, idx, fs, Fp, Fc, peak_off=PEAK_OFFSET):
    """:521-575 + :695 given the two whitened time series (length N = 4 fs): roll by -fs, ref_idx = argmax(hp^2 + hc^2),
    ht = hp*Fp + hc*Fc, slide by ref_idx - idx - peak_off (python slice semantics), zero-fill, Tukey window, crop
    [1.5 fs, 2.5 fs).  Returns (crop, ref_idx)."""
    N = hp_t.size
    hp_r = np.roll(hp_t, -int(fs))
    hc_r = np.roll(hc_t, -int(fs))
    ref_idx = int(np.argmax(hp_r ** 2 + hc_r ** 2))
    ht = hp_r * Fp + hc_r * Fc
    tmp = ht[int(ref_idx - idx - peak_off):]
    ts = np.zeros(N)
    if len(tmp) < N:
        ts[:len(tmp)] = tmp
    else:
        ts[:] = tmp[:N]
    win = np.zeros(N)
    tw = tukey(int((16.0 / 15.0) * N / SAFE), alpha=1.0 / 8.0)
    a = int((N - tw.size) / 2)
    win[a:a + tw.size] = tw
    ts = ts * win
    T_obs = N // fs
    return ts[int((T_obs / 2) * fs - fs / 2):int((T_obs / 2) * fs + fs / 2)], ref_idx


def gen_bbh(fs, T_obs, psd, par, Fp=None, Fc=None, peak_off=PEAK_OFFSET):
    """:462-575 for one detector: FD waveform -> whiten ('fd') -> irfft -> align -> crop.  Returns (crop (fs,), ref_idx)."""
    N = T_obs * fs
    hp, hc = chirp_fd(par.m1, par.m2, N // 2 + 1, 1.0 / T_obs, iota=par.iota, phi=par.phi)
    whp = whiten_data(hp, T_obs, fs, psd, 'fd')
    whc = whiten_data(hc, T_obs, fs, psd, 'fd')
    if Fp is None:
        Fp, Fc = antenna_response(EVENT_TIME, par.ra, par.dec, par.psi)
    return align_crop(np.fft.irfft(whp, N), np.fft.irfft(whc, N), par.idx, fs, Fp, Fc, peak_off)


def sim_data(fs, T_obs, psd, size, mdist='hunt_constrain', beta=(0.45, 0.55), gw_tmp=True, peak_off=PEAK_OFFSET):
    """:632-740 with Nnoise = 0, do_time_grid = False: (size-1) random templates, np.random.permutation shuffle, then the
    event-like (36, 29) template appended last.  Returns ([ts (size,1,fs), yval], pars)."""
    n_rand = size - 1 if gw_tmp else size
    ts, par = [], []
    for _ in range(n_rand):
        p = gen_par(fs, T_obs, mdist, beta, False)
        ts.append(gen_bbh(fs, T_obs, psd, p, peak_off=peak_off)[0].reshape(1, -1))
        par.append(p)
    ts = np.array(ts).reshape(n_rand, 1, fs)
    perm = np.random.permutation(n_rand)
    par = [par[i] for i in perm]
    ts = ts[perm]
    yval = np.ones(n_rand, dtype=int)
    if gw_tmp:
        p = gen_par(fs, T_obs, mdist, beta, True)
        ts = np.concatenate((ts, gen_bbh(fs, T_obs, psd, p, peak_off=peak_off)[0].reshape(1, 1, fs)))
        par.append(p)
        yval = np.append(yval, 1)
    return [ts, yval], par


def gen_par_posterior(fs, T_obs, index, gan_post, post_mc, beta=(0.75, 0.95), gw_tmp=False):
    """lalinf_post_waveform_maker.py:356-475: m12 = [gan_post[index,1], gan_post[index,0]] (:385), mc from the posterior column (:404),
    fixed angles (:433-437), ONE randint (:440-444) drawn BEFORE the gw_tmp branch (:460-473) replaces idx / masses."""
    m12 = [gan_post[index, 1], gan_post[index, 0]]
    eta = m12[0] * m12[1] / (m12[0] + m12[1]) ** 2
    mc = post_mc[index]
    M = np.sum(m12)
    low_idx, high_idx = convert_beta(list(beta), fs, T_obs)
    idx = low_idx if low_idx == high_idx else int(np.random.randint(low_idx, high_idx, 1)[0])
    par = bbhparams(mc, M, eta, m12[0], m12[1], RA, DEC, IOTA, PHI, PSI, idx, None, None)
    if gw_tmp:
        idx = int((T_obs * fs) / 2) - 4
        m1, m2 = 36.0, 29.0
        eta = m1 * m2 / (m1 + m2) ** 2
        M = m1 + m2
        par = bbhparams(M * eta ** (3.0 / 5.0), M, eta, m1, m2, RA, DEC, IOTA, PHI, PSI, idx, None, None)
    return par


def sim_data_posterior(fs, T_obs, psd, gan_post, post_mc, size, batch_size=3907, beta=(0.45, 0.55), gw_tmp=True, peak_off=PEAK_OFFSET):
    """lalinf_post_waveform_maker.py:649-746 with Nnoise = 0: posterior rows 0..size-2 (stopping at batch_size-1 waveforms, :718-721),
    permutation (:730), event-like template from gen_par(.., cnt, gw_tmp=True) last (:735-744)."""
    if gw_tmp:
        size = size - 1
    ts, par = [], []
    cnt = 0
    while cnt < size:
        p = gen_par_posterior(fs, T_obs, cnt, gan_post, post_mc, beta, False)
        ts.append(gen_bbh(fs, T_obs, psd, p, peak_off=peak_off)[0].reshape(1, -1))
        par.append(p)
        cnt += 1
        if len(ts) == batch_size - 1:
            size = batch_size - 1
            break
    ts = np.array(ts)[:size]
    par = par[:size]
    perm = np.random.permutation(size)
    par = [par[i] for i in perm]
    ts = ts[perm]
    yval = np.ones(size, dtype=int)
    if gw_tmp:
        p = gen_par_posterior(fs, T_obs, cnt, gan_post, post_mc, beta, True)
        ts = np.concatenate((ts, gen_bbh(fs, T_obs, psd, p, peak_off=peak_off)[0].reshape(1, 1, fs)))
        par.append(p)
        yval = np.append(yval, 1)
    return [ts, yval], par


def analytic_psd(Nf, df, f_floor=10.0):
    """A fixed aLIGO-like analytic noise curve on the Nf grid (synthetic stand-in for the lalinference PSD file,
    which the reference does not ship): zero below f_floor so that whiten_data's psd<=0 handling is exercised."""
    f = np.arange(Nf) * df
    x = np.where(f > 0, f, 1.0) / 215.0
    s = 1e-49 * (x ** -4.14 - 5.0 / (x * x) + 111.0 * (1 - x * x + 0.5 * x ** 4) / (1 + 0.5 * x * x))
    return np.where(f >= f_floor, s, 0.0)
