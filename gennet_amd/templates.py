"""Host-side mirror of BBH_version/gw_template_maker.py: the waveform + noise batch synthesiser and its ts/pars files.

Same function names, argument meaning and return shapes as the reference (SURVEY section 8b "Synth surface"); the reference's
per-template python loop (gw_template_maker.py:676, one LAL call + two irFFTs + three discarded spline fits per template,
serial) becomes batched HIP kernels: FD chirp + PSD whitening -> batched irFFT (whole transform in LDS) -> argmax-align,
antenna combination and 1-s crop, all fp64 like numpy, nothing touching host memory until the bank is pickled.

The frequency-domain waveform is NOT LAL's IMRPhenomPv2 (LALSuite cannot be reproduced offline): `gn_chirp_fd_whitened` is
this project's own closed-form non-spinning inspiral-merger-ringdown chirp (PhenomA functional form).  Everything
downstream of h~(f) follows the reference sample for sample, including the integer alignment arithmetic of :554.

Random draws that define the bank (masses, the five discarded angles, idx, the final permutation) stay on the host numpy
legacy stream in the reference's order, so a seeded run selects exactly the reference's parameters.
"""
import pickle

import numpy as np
import torch

from . import _lib
from .engine import device

safe = 2                       # gw_template_maker.py:54
gw_tmp = True                  # :56
event_time = '1126259462'      # :62
PEAK_OFFSET = 11               # :554 (comment there: "use 21 if sampling at 2kHz") -- a parameter of gen_bbh / sim_data here
F_LOW, DIST_MPC = 40.0, 410.0  # :495, :500
RA, DEC, IOTA, PHI, PSI = 2.21535724066, -1.23649695537, 2.5, 1.5, 1.75    # :433-437


class bbhparams(object):
    """gw_template_maker.py:69-85 (the trainer's copy, bbhMahoGANy.py:129-144, adds fmin; readers use .mc, .m1, .m2 only)."""

    def __init__(self, mc, M, eta, m1, m2, ra, dec, iota, phi, psi, idx, snr, SNR):
        self.mc, self.M, self.eta, self.m1, self.m2 = mc, M, eta, m1, m2
        self.ra, self.dec, self.iota, self.phi, self.psi = ra, dec, iota, phi, psi
        self.idx, self.snr, self.SNR = idx, snr, SNR


# --------------------------------------------------------------------------------------------------- small host helpers
def tukey(M, alpha=0.5):
    """gw_template_maker.py:87-113."""
    n = np.arange(M)
    width = int(np.floor(alpha * (M - 1) / 2.0))
    w = np.ones(M)
    w[:width + 1] = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n[:width + 1] / alpha / (M - 1))))
    w[M - width - 1:] = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n[M - width - 1:] / alpha / (M - 1))))
    return w


def convert_beta(beta, fs, T_obs):
    """gw_template_maker.py:133-159."""
    newbeta = np.array([(beta[0] + 0.5 * safe - 0.5), (beta[1] + 0.5 * safe - 0.5)]) / safe
    return int(T_obs * fs * newbeta[0]), int(T_obs * fs * newbeta[1])


def _whiten_scale(psd, sample_rate):
    """sqrt(2*invpsd/fs), invpsd = 0 where psd <= 0, and 0 at DC (whiten_data sets xf[0] = 0): :273-279."""
    psd = np.asarray(psd, np.float64)
    inv = np.zeros(psd.size)
    pos = psd > 0.0
    inv[pos] = 1.0 / psd[pos]
    s = np.sqrt(2.0 * inv / sample_rate)
    s[0] = 0.0
    return s


def _gmst_rad(gps):
    utc = gps - 17.0
    t = ((utc / 86400.0 + 2444244.5) - 2451545.0) / 36525.0
    sec = 67310.54841 + (876600.0 * 3600.0 + 8640184.812866) * t + 0.093104 * t * t - 6.2e-6 * t ** 3
    return (sec % 86400.0) * (2.0 * np.pi / 86400.0)


_LHO_X = np.array([-0.22389266154, 0.79983062746, 0.55690487831])
_LHO_Y = np.array([-0.91397818574, 0.02609403989, -0.40492342125])


def antenna_response(gps, ra, dec, psi, det='H1'):
    """Stand-in for pylal.antenna.response (:612): F+, Fx of LIGO Hanford from the detector tensor (not LAL-verified)."""
    if det != 'H1':
        raise NotImplementedError('detector %r (the reference fixes H1)' % det)
    D = 0.5 * (np.outer(_LHO_X, _LHO_X) - np.outer(_LHO_Y, _LHO_Y))
    gha = _gmst_rad(gps) - ra
    cg, sg, cd, sd, cp, sp = np.cos(gha), np.sin(gha), np.cos(dec), np.sin(dec), np.cos(psi), np.sin(psi)
    X = np.array([-cp * sg - sp * cg * sd, -cp * cg + sp * sg * sd, sp * cd])
    Y = np.array([sp * sg - cp * cg * sd, sp * cg + cp * sg * sd, cp * cd])
    return X @ D @ X - Y @ D @ Y, X @ D @ Y + Y @ D @ X


# --------------------------------------------------------------------------------------------------- device plumbing
def _s():
    return torch.cuda.current_stream().cuda_stream


def _d64(a):
    return torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(device())


_TW = {}


def twiddles(N):
    """exp(+2 pi i k / N), k < N/2, fp64, built once on the host and cached in HBM."""
    key = (N, device().index)
    if key not in _TW:
        k = np.arange(N // 2)
        w = np.exp(2j * np.pi * k / N)
        _TW[key] = _d64(np.stack([w.real, w.imag], axis=1))
    return _TW[key]


def irfft(X, N):
    """X: (nb, N/2+1, 2) fp64 device tensor (re, im) -> (nb, N) fp64."""
    nb = X.shape[0]
    out = torch.empty((nb, N), dtype=torch.float64, device=X.device)
    _lib.call('gn_irfft_f64', X.data_ptr(), out.data_ptr(), twiddles(N).data_ptr(), nb, N, _s())
    return out


def rfft(x):
    nb, N = x.shape
    X = torch.empty((nb, N // 2 + 1, 2), dtype=torch.float64, device=x.device)
    _lib.call('gn_rfft_f64', x.data_ptr(), X.data_ptr(), twiddles(N).data_ptr(), nb, N, _s())
    return X


def _mul(x, w, complex_x):
    _lib.call('gn_mul_f64', x.data_ptr(), w.data_ptr(), x.numel(), w.numel(), 1 if complex_x else 0, _s())
    return x


# --------------------------------------------------------------------------------------------------- reference surface
def whiten_data(data, duration, sample_rate, psd, flag='td'):
    """gw_template_maker.py:243-286.  numpy in / numpy out; arithmetic on the device."""
    scale = _d64(_whiten_scale(psd, sample_rate))
    if flag == 'td':
        N = int(duration * sample_rate)
        x = _d64(np.asarray(data, np.float64).reshape(1, N))
        _mul(x, _d64(tukey(N, alpha=1.0 / 8.0)), False)
        X = _mul(rfft(x), scale, True)
        return irfft(X, N).cpu().numpy().reshape(N)
    xf = np.asarray(data, np.complex128)
    X = _d64(np.stack([xf.real, xf.imag], axis=1).reshape(1, xf.size, 2))
    _mul(X, scale, True)
    o = X.cpu().numpy().reshape(xf.size, 2)
    return o[:, 0] + 1j * o[:, 1]


def gen_noise(fs, T_obs, psd):
    """gw_template_maker.py:161-193: the 2*Nf standard normals come from the legacy numpy stream (re block, then im block),
    exactly as in the reference; spectrum scaling and the irFFT run on the device."""
    N = T_obs * fs
    Nf = N // 2 + 1
    df = 1.0 / T_obs
    psd = np.asarray(psd, np.float64)
    amp = np.sqrt(0.25 * T_obs * psd)
    amp[psd == 0.0] = 0.0
    amp[0] = 0.0
    re = np.random.normal(0, 1, Nf)
    im = np.random.normal(0, 1, Nf)
    X = _d64(np.stack([re, im], axis=1).reshape(1, Nf, 2))
    _mul(X, _d64(amp), True)
    x = irfft(X, N)
    _lib.call('gn_scale_f64', x.data_ptr(), float(N), x.numel(), _s())      # x = N*irfft(.)*df, two multiplies like numpy
    _lib.call('gn_scale_f64', x.data_ptr(), float(df), x.numel(), _s())
    return x.cpu().numpy().reshape(N)


def gen_noise_device(fs, T_obs, psd, nb, seed, offset=0):
    """nb coloured-noise realisations drawn on the device (Philox normals): (nb, N) fp64 tensor in HBM.  Same spectrum
    construction as gen_noise; the random stream differs from numpy's by construction (statistical parity only)."""
    N = T_obs * fs
    Nf = N // 2 + 1
    psd = np.asarray(psd, np.float64)
    amp = np.sqrt(0.25 * T_obs * psd)
    amp[psd == 0.0] = 0.0
    X = torch.empty((nb, Nf, 2), dtype=torch.float64, device=device())
    _lib.call('gn_noise_fd', _d64(amp).data_ptr(), X.data_ptr(), nb, Nf, int(seed), int(offset), _s())
    x = irfft(X, N)
    _lib.call('gn_scale_f64', x.data_ptr(), float(N) * (1.0 / T_obs), x.numel(), _s())
    return x


class NoiseSynth(object):
    """gen_noise (gw_template_maker.py:161-193) -> whiten_data(flag='td') (:243-286) -> central crop, fused: ONE kernel, one workgroup per row,
    every N-point real transform an N/2-point complex transform in LDS, fp64 like numpy (gn_noise_whitened; BASELINE configs[4]).  Keeps
    the three tables of one (fs, T_obs, psd) in HBM: amp = sqrt(0.25 T psd) (0 where psd == 0), the whitening scale, the Tukey(1/8) window."""

    def __init__(self, fs, T_obs, psd):
        self.fs, self.T_obs = int(fs), int(T_obs)
        self.N = self.fs * self.T_obs
        self.Nf = self.N // 2 + 1
        if self.N not in Synth.FUSED_N:
            raise NotImplementedError('NoiseSynth: series length %d is outside the fused kernel\'s range %r' % (self.N, Synth.FUSED_N))
        psd = np.asarray(psd, np.float64)
        assert psd.size == self.Nf, 'psd must have N/2+1 = %d bins' % self.Nf
        amp = np.sqrt(0.25 * T_obs * psd)
        amp[psd == 0.0] = 0.0
        self.amp = _d64(amp)
        self.wscale = _d64(_whiten_scale(psd, fs))
        self.win = _d64(tukey(self.N, alpha=1.0 / 8.0))
        twiddles(self.N)

    def draw(self, nb, seed=0, counter=0, normals=None, add=None, dtype=torch.float32, crop=None, want_normals=False):
        """nb whitened-noise rows -> (nb, crop_len) device tensor of `dtype` (fp32 or fp64).  normals: (nb, 2 Nf) numpy array of standard
        normals per row in numpy's draw order [re block | im block] (what gen_noise consumes from the legacy stream), or None: Philox
        stream (seed, counter), which the caller advances by nb * Nf.  add: (nb, crop_len) fp64 device rows the noise is added to.
        crop = (start, length), default the central second [1.5 fs, 2.5 fs).  want_normals: also return the (nb, 2 Nf) normals used."""
        c0, cl = crop if crop is not None else (int((self.T_obs / 2) * self.fs - self.fs / 2), self.fs)
        out = torch.empty((nb, cl), dtype=dtype, device=device())
        nin = None
        if normals is not None:
            nin = _d64(np.asarray(normals, np.float64).reshape(nb, 2 * self.Nf))
        nout = torch.empty((nb, 2 * self.Nf), dtype=torch.float64, device=device()) if want_normals else None
        if add is not None:
            assert add.dtype == torch.float64 and tuple(add.shape) == (nb, cl) and add.is_contiguous()
        if nb:
            _lib.call('gn_noise_whitened', self.amp.data_ptr(), self.wscale.data_ptr(), self.win.data_ptr(), twiddles(self.N).data_ptr(),
                      None if nin is None else nin.data_ptr(), None if nout is None else nout.data_ptr(), None if add is None else add.data_ptr(),
                      out.data_ptr() if dtype == torch.float64 else None, out.data_ptr() if dtype == torch.float32 else None, nb, self.N, c0, cl,
                      1.0 / self.T_obs, int(seed), int(counter), _s())
        return (out, nout) if want_normals else out


def gen_masses(m_min=5.0, M_max=100.0, mdist='astro'):
    """gw_template_maker.py:289-370, all four distributions, same draws from the numpy legacy stream per rejection trial:
    'astro' / 'hunt_constrain' (:312-339: two log-uniform component masses per trial; the latter adds q >= 0.5, 20 <= mc <= 35),
    'gh' (:341-351: q ~ U(1, 10), m2 ~ U(5, 75), m1 = q m2, both below 75), 'metric' (:353-367: total mass and eta from the
    template-bank-metric densities).  Returns (m12, mc, eta) with mc, eta as floats (the reference's 'metric' branch returns them as
    1-element arrays).  Pinned by tests/golden/masses_golden.npz (the reference's function executed on seeded streams)."""
    if mdist in ('astro', 'hunt_constrain'):
        log_m_max = np.log(M_max - m_min)
        while True:
            m12 = np.exp(np.log(m_min) + np.random.uniform(0, 1, 2) * (log_m_max - np.log(m_min)))
            eta = m12[0] * m12[1] / (m12[0] + m12[1]) ** 2
            mc = np.sum(m12) * eta ** (3.0 / 5.0)
            flag = (np.sum(m12) < M_max) and np.all(m12 > m_min) and (m12[0] >= m12[1])
            if mdist == 'hunt_constrain':
                flag = flag and (m12[1] / m12[0] >= 0.5) and (mc >= 20.0) and (mc <= 35.0)
            if flag:
                return m12, mc, eta
    if mdist == 'gh':
        m12 = np.zeros(2)
        while True:
            q = np.random.uniform(1.0, 10.0, 1)
            m12[1] = np.random.uniform(5.0, 75.0, 1)[0]
            m12[0] = m12[1] * q[0]
            if np.all(m12 < 75.0) and np.all(m12 > 5.0) and (m12[0] >= m12[1]):
                eta = m12[0] * m12[1] / (m12[0] + m12[1]) ** 2
                return m12, np.sum(m12) * eta ** (3.0 / 5.0), eta
    if mdist == 'metric':
        M_min = 2.0 * m_min
        eta_min = m_min * (M_max - m_min) / M_max ** 2
        while True:
            # 1-element arrays as in the reference: numpy's array power and its scalar power differ in the last bit
            M = (M_min ** (-7.0 / 3.0) - np.random.uniform(0, 1, 1) * (M_min ** (-7.0 / 3.0) - M_max ** (-7.0 / 3.0))) ** (-3.0 / 7.0)
            eta = (eta_min ** (-2.0) - np.random.uniform(0, 1, 1) * (eta_min ** (-2.0) - 16.0)) ** (-1.0 / 2.0)
            m12 = np.zeros(2)
            m12[0] = (0.5 * M + M * np.sqrt(0.25 - eta))[0]
            m12[1] = (M - m12[0])[0]
            if (np.sum(m12) < M_max) and np.all(m12 > m_min) and (m12[0] >= m12[1]):
                return m12, float((np.sum(m12) * eta ** (3.0 / 5.0))[0]), float(eta[0])
    raise ValueError('unknown mass distribution %r (astro, hunt_constrain, gh, metric; gw_template_maker.py:368-370 prints and exits)' % (mdist,))


def gen_par(fs, T_obs, mdist='astro', beta=[0.75, 0.95], gw_tmp=False):
    """gw_template_maker.py:372-460 (same RNG consumption: masses, five discarded rand(), one randint unless low == high)."""
    m12, mc, eta = gen_masses(5.0, 100.0, mdist=mdist)
    M = np.sum(m12)
    for _ in range(5):              # iota, psi, phi, ra, dec are drawn and then overwritten by constants (:403-416 vs :433-437)
        np.random.rand()
    if gw_tmp:
        beta = [0.5, 0.5]
    low_idx, high_idx = convert_beta(beta, fs, T_obs)
    idx = low_idx if low_idx == high_idx else int(np.random.randint(low_idx, high_idx, 1)[0])
    if gw_tmp:
        m1, m2 = 36.0, 29.0
        eta = m1 * m2 / (m1 + m2) ** 2
        M = m1 + m2
        return bbhparams(M * eta ** (3.0 / 5.0), M, eta, m1, m2, RA, DEC, IOTA, PHI, PSI, idx, None, None)
    return bbhparams(mc, M, eta, m12[0], m12[1], RA, DEC, IOTA, PHI, PSI, idx, None, None)


class Synth(object):
    """Batched gen_bbh for one (fs, T_obs, psd): keeps the whitening scale and FFT twiddles in HBM."""

    def __init__(self, fs, T_obs, psd, det='H1', peak_off=PEAK_OFFSET, f_low=F_LOW, dist_mpc=DIST_MPC):
        self.fs, self.T_obs = int(fs), int(T_obs)
        self.N = self.fs * self.T_obs
        self.Nf = self.N // 2 + 1
        self.scale = _d64(_whiten_scale(psd, fs))
        assert self.scale.numel() == self.Nf, 'psd must have N/2+1 = %d bins' % self.Nf
        self.Fp, self.Fc = antenna_response(float(event_time), RA, DEC, PSI, det)
        self.peak_off, self.f_low, self.dist_mpc = int(peak_off), float(f_low), float(dist_mpc)
        twiddles(self.N)

    def series(self, m1, m2, iota=IOTA, phi=PHI):
        """Whitened time-domain polarisations irfft(whiten(h~)) (un-rolled): two (nb, N) fp64 device tensors."""
        m1 = _d64(np.atleast_1d(m1)); m2 = _d64(np.atleast_1d(m2))
        nb = m1.numel()
        hp = torch.empty((nb, self.Nf, 2), dtype=torch.float64, device=device())
        hc = torch.empty_like(hp)
        _lib.call('gn_chirp_fd_whitened', m1.data_ptr(), m2.data_ptr(), self.scale.data_ptr(), hp.data_ptr(), hc.data_ptr(), nb, self.Nf,
                  1.0 / self.T_obs, self.f_low, self.dist_mpc, float(iota), float(phi), _s())
        return irfft(hp, self.N), irfft(hc, self.N), (hp, hc)

    def align(self, hp_t, hc_t, idx, crop0, crop_len, Fp, Fc, g=1.0):
        nb = hp_t.shape[0]
        idx_t = torch.as_tensor(np.ascontiguousarray(np.atleast_1d(idx), np.int32)).to(device())
        out = torch.empty((nb, crop_len), dtype=torch.float64, device=device())
        ref = torch.empty((nb,), dtype=torch.int32, device=device())
        _lib.call('gn_align_crop', hp_t.data_ptr(), hc_t.data_ptr(), idx_t.data_ptr(), out.data_ptr(), ref.data_ptr(), nb, self.N, self.fs,
                  int(crop0), int(crop_len), self.peak_off, float(Fp), float(Fc), float(g), _s())
        return out, ref

    FUSED_N = (1024, 2048, 4096, 8192, 16384)

    def templates(self, m1, m2, idx, g=1.0, chunk=4096, dtype=torch.float64, fused=None):
        """Central 1-s crops [1.5 fs, 2.5 fs) of the detector strain for a batch of (m1, m2, idx): (nb, fs) device tensor (fp64, or
        fp32 with dtype=torch.float32) and the reference indices (nb,) int32.  (gen_bbh + the crop of sim_data :695; the Tukey window
        equals 1.0 there.)  One fused kernel per batch (gn_synth_templates: spectrum, both inverse FFTs, arg-max, slide and crop in
        LDS); fused=False, or a series length the fused kernel does not cover, runs the separate kernels (chirp -> irFFT x2 -> align)."""
        m1 = np.atleast_1d(np.asarray(m1, np.float64)); m2 = np.atleast_1d(np.asarray(m2, np.float64)); idx = np.atleast_1d(idx)
        c0 = int((self.T_obs / 2) * self.fs - self.fs / 2)
        if fused is None:
            fused = self.N in self.FUSED_N
        if fused:
            nb = m1.size
            out = torch.empty((nb, self.fs), dtype=dtype, device=device())
            ref = torch.empty((nb,), dtype=torch.int32, device=device())
            if nb == 0:
                return out, ref
            m1d, m2d = _d64(m1), _d64(m2)
            idx_t = torch.as_tensor(np.ascontiguousarray(idx, np.int32)).to(device())
            o64 = out.data_ptr() if dtype == torch.float64 else None
            o32 = out.data_ptr() if dtype == torch.float32 else None
            _lib.call('gn_synth_templates', m1d.data_ptr(), m2d.data_ptr(), idx_t.data_ptr(), self.scale.data_ptr(), twiddles(self.N).data_ptr(), o64, o32,
                      ref.data_ptr(), nb, self.N, self.fs, c0, self.fs, self.peak_off, 1.0 / self.T_obs, self.f_low, self.dist_mpc, float(IOTA), float(PHI),
                      float(self.Fp), float(self.Fc), float(g), _s())
            return out, ref
        if m1.size == 0:
            return torch.empty((0, self.fs), dtype=dtype, device=device()), torch.empty((0,), dtype=torch.int32, device=device())
        outs, refs = [], []
        for s in range(0, m1.size, chunk):
            hp_t, hc_t, _ = self.series(m1[s:s + chunk], m2[s:s + chunk])
            o, r = self.align(hp_t, hc_t, idx[s:s + chunk], c0, self.fs, self.Fp, self.Fc, g)
            outs.append(o); refs.append(r)
        out = torch.cat(outs)
        if dtype == torch.float32:
            o32 = torch.empty(out.shape, dtype=torch.float32, device=out.device)
            _lib.call('gn_f64_to_f32', out.data_ptr(), o32.data_ptr(), 1.0, out.numel(), _s())
            out = o32
        return out, torch.cat(refs)

    PRIOR_TRIALS = 1024           # Philox counters one template of templates_prior consumes (csrc/synth_fused.hip: kPriorTrials)

    def templates_prior(self, nb, seed, counter, idx_lo, idx_hi, g=1.0, dtype=torch.float32, want_params=False, noise=None, noise_seed=0, noise_counter=0):
        """nb templates whose (m1, m2, idx) are drawn from the hunt_constrain prior INSIDE the synthesis kernel (gn_synth_templates_prior):
        returns (crops (nb, fs), labels (nb, 2) = [mc, m2/m1] fp32, ref_idx) -- plus (m1m2 (nb, 2) f64, idx (nb,) i32) with want_params.
        A pure function of (seed, counter); the caller advances counter by nb * PRIOR_TRIALS.
        noise: a NoiseSynth of the same (fs, T_obs, psd) -- the same launch then adds PSD-coloured, whitened noise from the Philox stream
        (noise_seed, noise_counter) to every row (gn_synth_templates_noise; the caller advances noise_counter by nb * Nf)."""
        if self.N not in self.FUSED_N:
            raise NotImplementedError('templates_prior: series length %d is outside the fused kernel\'s range %r' % (self.N, self.FUSED_N))
        out = torch.empty((nb, self.fs), dtype=dtype, device=device())
        labels = torch.empty((nb, 2), dtype=torch.float32, device=device())
        ref = torch.empty((nb,), dtype=torch.int32, device=device())
        mo = torch.empty((nb, 2), dtype=torch.float64, device=device()) if want_params else None
        io = torch.empty((nb,), dtype=torch.int32, device=device()) if want_params else None
        if nb:
            c0 = int((self.T_obs / 2) * self.fs - self.fs / 2)
            o64 = out.data_ptr() if dtype == torch.float64 else None
            o32 = out.data_ptr() if dtype == torch.float32 else None
            mo_p, io_p = None if mo is None else mo.data_ptr(), None if io is None else io.data_ptr()
            if noise is None:
                _lib.call('gn_synth_templates_prior', self.scale.data_ptr(), twiddles(self.N).data_ptr(), o64, o32, labels.data_ptr(), mo_p, io_p,
                          ref.data_ptr(), nb, self.N, self.fs, c0, self.fs, self.peak_off, 1.0 / self.T_obs, self.f_low,
                          self.dist_mpc, float(IOTA), float(PHI), float(self.Fp), float(self.Fc), float(g), int(seed), int(counter), int(idx_lo), int(idx_hi),
                          5.0, 100.0, _s())
            else:
                assert (noise.fs, noise.T_obs) == (self.fs, self.T_obs)
                _lib.call('gn_synth_templates_noise', None, None, None, self.scale.data_ptr(), twiddles(self.N).data_ptr(), noise.amp.data_ptr(),
                          noise.win.data_ptr(), o64, o32, labels.data_ptr(), mo_p, io_p, ref.data_ptr(), nb, self.N, self.fs, c0, self.fs, self.peak_off,
                          1.0 / self.T_obs, self.f_low, self.dist_mpc, float(IOTA), float(PHI), float(self.Fp), float(self.Fc), float(g), int(seed),
                          int(counter), int(idx_lo), int(idx_hi), 5.0, 100.0, int(noise_seed), int(noise_counter), None, _s())
        return (out, labels, ref, mo, io) if want_params else (out, labels, ref)

    def templates_noise(self, m1, m2, idx, noise, noise_seed, noise_counter, g=1.0, dtype=torch.float32, want_normals=False):
        """templates() with the coloured-noise chain run by the same workgroups (gn_synth_templates_noise, given parameters): (nb, fs) rows
        = template * g + whitened noise of stream (noise_seed, noise_counter), and ref_idx; want_normals adds the (nb, 2 Nf) normals used."""
        m1 = np.atleast_1d(np.asarray(m1, np.float64)); m2 = np.atleast_1d(np.asarray(m2, np.float64)); idx = np.atleast_1d(idx)
        nb = m1.size
        out = torch.empty((nb, self.fs), dtype=dtype, device=device())
        ref = torch.empty((nb,), dtype=torch.int32, device=device())
        nout = torch.empty((nb, 2 * noise.Nf), dtype=torch.float64, device=device()) if want_normals else None
        if nb:
            c0 = int((self.T_obs / 2) * self.fs - self.fs / 2)
            m1d, m2d = _d64(m1), _d64(m2)
            idx_t = torch.as_tensor(np.ascontiguousarray(idx, np.int32)).to(device())
            _lib.call('gn_synth_templates_noise', m1d.data_ptr(), m2d.data_ptr(), idx_t.data_ptr(), self.scale.data_ptr(), twiddles(self.N).data_ptr(),
                      noise.amp.data_ptr(), noise.win.data_ptr(), out.data_ptr() if dtype == torch.float64 else None,
                      out.data_ptr() if dtype == torch.float32 else None, None, None, None, ref.data_ptr(), nb, self.N, self.fs, c0, self.fs, self.peak_off,
                      1.0 / self.T_obs, self.f_low, self.dist_mpc, float(IOTA), float(PHI), float(self.Fp), float(self.Fc), float(g), 0, 0, 0, 0, 5.0, 100.0,
                      int(noise_seed), int(noise_counter), None if nout is None else nout.data_ptr(), _s())
        return (out, ref, nout) if want_normals else (out, ref)


def make_bbh(hp, hc, fs, ra, dec, psi, det):
    """gw_template_maker.py:577-630: antenna combination.  The reference also fits and evaluates three time-shift splines and
    then returns the UN-shifted series (:621-630); only what it returns is computed here."""
    Fp, Fc = antenna_response(float(event_time), ra, dec, psi, det)
    return hp * Fp + hc * Fc, hp, hc


def gen_bbh(fs, T_obs, psds, dets=['H1'], beta=[0.75, 0.95], par=None, gw_tmp=False, peak_off=PEAK_OFFSET):
    """gw_template_maker.py:462-575 for one template: returns (ts, hp, hc, ts), each (1, N): the windowed, slid series."""
    syn = Synth(fs, T_obs, psds, dets[0], peak_off)
    hp_t, hc_t, _ = syn.series([par.m1], [par.m2], par.iota, par.phi)
    N = syn.N
    tw = tukey(int((16.0 / 15.0) * N / safe), alpha=1.0 / 8.0)
    win = np.zeros(N)
    a = int((N - tw.size) / 2)
    win[a:a + tw.size] = tw
    win_d = _d64(win)
    Fp, Fc = antenna_response(float(event_time), par.ra, par.dec, par.psi, dets[0])
    outs = []
    for fp, fc in ((Fp, Fc), (1.0, 0.0), (0.0, 1.0)):
        o, _ = syn.align(hp_t, hc_t, [par.idx], 0, N, fp, fc)
        outs.append(_mul(o, win_d, False).cpu().numpy().reshape(1, N))
    return outs[0], outs[1], outs[2], outs[0]


def sim_data(fs, T_obs, psds, dets=['H1'], Nnoise=25, size=1000, mdist='astro', beta=[0.75, 0.95], peak_off=PEAK_OFFSET, to_host=True):
    """gw_template_maker.py:632-740 with Nnoise = 0 (what main() passes, :806; the Nnoise > 0 branch of the reference would
    raise on an array psd, :687) and do_time_grid = False.  Returns ([ts (size, 1, fs), yval], pars): size-1 random
    templates shuffled by np.random.permutation, then the GW150914-like (36, 29) template appended last."""
    if Nnoise > 0:
        raise NotImplementedError('sim_data(Nnoise > 0): the reference ships noise-free templates (Nnoise = 0)')
    n_rand = size - 1 if gw_tmp else size
    pars = [gen_par(fs, T_obs, mdist=mdist, beta=beta, gw_tmp=False) for _ in range(n_rand)]
    syn = Synth(fs, T_obs, psds, dets[0], peak_off)
    ts, _ = syn.templates([p.m1 for p in pars], [p.m2 for p in pars], [p.idx for p in pars])
    perm = np.random.permutation(n_rand)
    pars = [pars[i] for i in perm]
    ts = ts[torch.as_tensor(perm).to(ts.device)]
    if gw_tmp:
        p = gen_par(fs, T_obs, mdist=mdist, beta=beta, gw_tmp=True)
        ev, _ = syn.templates([p.m1], [p.m2], [p.idx])
        ts = torch.cat([ts, ev])
        pars.append(p)
    yval = np.ones(len(pars), dtype=int)
    ts = ts.reshape(len(pars), 1, int(fs))
    return [ts.cpu().numpy() if to_host else ts, yval], pars


# --------------------------------------------------------------------------------------------------- file layout
class _ParsPickler(pickle.Pickler):
    pass


def save_ts_pars(basename, event_name, i, sample_num, tag, ts, pars):
    """Writes <basename><event>_ts_<i>_<sample_num>Samp<tag>.sav and ..._params_... (gw_template_maker.py:842-850), pickle
    protocol 2 (what py2 cPickle.HIGHEST_PROTOCOL is).  bbhparams instances are pickled as `__main__.bbhparams`, the name the
    reference writer (run as a script) stores and the reference reader (bbhMahoGANy.py:129, :972-973) resolves."""
    ts_path = '%s%s_ts_%s_%sSamp%s.sav' % (basename, event_name, i, sample_num, tag)
    par_path = '%s%s_params_%s_%sSamp%s.sav' % (basename, event_name, i, sample_num, tag)
    with open(ts_path, 'wb') as f:
        pickle.dump([np.asarray(ts[0], np.float64), np.asarray(ts[1])], f, protocol=2)
    old = bbhparams.__module__
    import sys
    main_mod = sys.modules['__main__']
    had = getattr(main_mod, 'bbhparams', None)
    try:
        bbhparams.__module__ = '__main__'
        setattr(main_mod, 'bbhparams', bbhparams)
        with open(par_path, 'wb') as f:
            pickle.dump(list(pars), f, protocol=2)
    finally:
        bbhparams.__module__ = old
        if had is None:
            delattr(main_mod, 'bbhparams')
        else:
            setattr(main_mod, 'bbhparams', had)
    return ts_path, par_path


class _RefUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == 'bbhparams':
            return bbhparams
        return pickle.Unpickler.find_class(self, module, name)


def load_ts_pars(ts_path, par_path):
    """Reads the two files (py2 cPickle or ours): returns ([ts, yval], [bbhparams])."""
    with open(ts_path, 'rb') as f:
        ts = pickle.load(f, encoding='latin1')
    with open(par_path, 'rb') as f:
        pars = _RefUnpickler(f, encoding='latin1').load()
    return ts, list(pars)


def training_arrays(ts, pars):
    """bbhMahoGANy.py:1007-1014, :1036, :1053-1055: images (Ns-1, n_pix), labels [[mc, m2/m1]], with the last (event-like)
    template split off as the event."""
    images = np.reshape(ts[0], (ts[0].shape[0], ts[0].shape[2]))
    labels = np.array([[k.mc, (k.m2 / k.m1)] for k in pars])
    return images[:-1], labels[:-1], images[-1], labels[-1]


# --------------------------------------------------------------------------------------------------- posterior-driven mode
def m1m2_from_mc_q(mc, q):
    """Component masses of a posterior row, heavier first: q = m2/m1 <= 1, mc = (m1 m2)^(3/5)/(m1+m2)^(1/5)
    -> m1 = mc (1+q)^(1/5) / q^(3/5), m2 = q m1.  This is the pair lalinf_post_waveform_maker.py:385 forms (m12 = [column 1, column 0] of
    the m1_m2 file); the file itself holds them the other way round, see lalinf_pars."""
    mc = np.asarray(mc, np.float64); q = np.asarray(q, np.float64)
    m1 = mc * (1.0 + q) ** 0.2 / q ** 0.6
    return m1, q * m1


def lalinf_pars(post_mc, post_q):
    """data/get_lalinf_pars.py:52-91 in closed form (the reference solves the two equations per row with sympy): the three arrays it
    pickles from the lalinference posterior columns mc and q.
      'm1_m2': [post_m1, post_m2] with post_m1 = mc (1 + 1/q)^(1/5) q^(3/5)  (root of :58, which sets m2 = m1/q)
                                       post_m2 = mc (1 + q)^(1/5) / q^(3/5)   (root of :62, which sets m1 = q m2)
               -- for q <= 1 the FIRST row is the lighter mass, as in the reference's file
      'mc_M' : [mc, post_m1 + post_m2]     (:69-84)
      'mc_q' : [mc, q]                     (:88-91)
    Pinned by tests/golden/lalinf_pars_golden.npz (the reference's loops executed on supplied columns)."""
    mc = np.asarray(post_mc, np.float64).reshape(-1); q = np.asarray(post_q, np.float64).reshape(-1)
    if mc.shape != q.shape or (q <= 0).any() or (mc <= 0).any():
        raise ValueError('lalinf_pars: mc and q must be positive columns of equal length')
    m1 = mc * (1.0 + 1.0 / q) ** 0.2 * q ** 0.6
    m2 = mc * (1.0 + q) ** 0.2 / q ** 0.6
    return {'m1_m2': np.array([m1, m2]), 'mc_M': np.array([mc, m1 + m2]), 'mc_q': np.array([mc, q])}


def gen_par_posterior(fs, T_obs, index, post_m1, post_m2, post_mc=None, beta=[0.75, 0.95], gw_tmp=False):
    """gen_par of lalinf_post_waveform_maker.py:356-475: masses of posterior row `index` (m12 = [gan_post[index,1], gan_post[index,0]], :385 --
    here post_m1 the heavier, post_m2 the lighter column), chirp mass from the posterior column (:404), the fixed angles (:433-437), and ONE
    randint from the numpy legacy stream (:440-444) -- drawn for the event-like template too, whose gw_tmp branch then replaces idx by
    N/2 - 4 and the masses by (36, 29) (:460-473).  Pinned by tests/golden/posterior_mode_golden.npz."""
    m1, m2 = float(post_m1[index]), float(post_m2[index])
    eta = m1 * m2 / (m1 + m2) ** 2
    mc = post_mc[index] if post_mc is not None else (m1 + m2) * eta ** (3.0 / 5.0)
    low_idx, high_idx = convert_beta(beta, fs, T_obs)
    idx = low_idx if low_idx == high_idx else int(np.random.randint(low_idx, high_idx, 1)[0])
    if gw_tmp:
        m1, m2 = 36.0, 29.0
        eta = m1 * m2 / (m1 + m2) ** 2
        return bbhparams((m1 + m2) * eta ** (3.0 / 5.0), m1 + m2, eta, m1, m2, RA, DEC, IOTA, PHI, PSI, int((T_obs * fs) / 2) - 4, None, None)
    return bbhparams(mc, np.sum([m1, m2]), eta, m1, m2, RA, DEC, IOTA, PHI, PSI, idx, None, None)


def posterior_block_pars(fs, T_obs, post_m1, post_m2, post_mc=None, size=None, beta=[0.45, 0.55], batch_size=3907):
    """The host half of one sim_data block of lalinf_post_waveform_maker.py:690-746 -- every draw it takes from the numpy legacy stream, in
    its order: one gen_par per posterior row (at most size-1 with gw_tmp, and at most batch_size-1, :718-721), the permutation (:730),
    then the event-like row's gen_par (:738), which reads posterior row `cnt` and draws a randint of its own.
    Returns (pars in generation order, perm, event-like par or None)."""
    n = len(post_m1) if size is None else (size - 1 if gw_tmp else size)
    n = min(n, batch_size - 1)
    if n + (1 if gw_tmp else 0) > len(post_m1):
        raise IndexError('posterior has %d rows, the block reads row %d (lalinf_post_waveform_maker.py:385)' % (len(post_m1), n))
    pars = [gen_par_posterior(fs, T_obs, k, post_m1, post_m2, post_mc, beta=beta, gw_tmp=False) for k in range(n)]
    perm = np.random.permutation(n)
    ev = gen_par_posterior(fs, T_obs, n, post_m1, post_m2, post_mc, beta=beta, gw_tmp=True) if gw_tmp else None
    return pars, perm, ev


def sim_data_posterior(fs, T_obs, psds, post_m1, post_m2, post_mc=None, dets=['H1'], size=None, beta=[0.45, 0.55], batch_size=3907,
                       peak_off=PEAK_OFFSET, to_host=True):
    """lalinf_post_waveform_maker.py sim_data / gen_par (:356-475, :649-746): the same synthesiser fed with component masses
    taken row by row from posterior samples instead of the prior (the CNN "sanity check" set, bbhMahoGANy.py:1228-1231).
    Parameter handling and the consumption of the numpy legacy stream follow the reference draw for draw (posterior_block_pars), so
    consecutive blocks of one seeded run (main()'s nblock loop, :799-805) select the reference's idx values.
    Returns ([ts (n,1,fs), yval], pars)."""
    post_m1 = np.asarray(post_m1, np.float64); post_m2 = np.asarray(post_m2, np.float64)
    pars, perm, ev = posterior_block_pars(fs, T_obs, post_m1, post_m2, post_mc, size, beta, batch_size)
    syn = Synth(fs, T_obs, psds, dets[0], peak_off)
    ts, _ = syn.templates([p.m1 for p in pars], [p.m2 for p in pars], [p.idx for p in pars])
    pars = [pars[i] for i in perm]
    ts = ts[torch.as_tensor(perm).to(ts.device)]
    if ev is not None:
        evt, _ = syn.templates([ev.m1], [ev.m2], [ev.idx])
        ts = torch.cat([ts, evt])
        pars.append(ev)
    ts = ts.reshape(len(pars), 1, int(fs))
    return [ts.cpu().numpy() if to_host else ts, np.ones(len(pars), dtype=int)], pars


def save_sanity_check(path, ts, gw_norm_constant=1.0):
    """data/<event>_cnn_sanity_check_ts_mass-time-vary<tag>.sav (lalinf_post_waveform_maker.py:812-837): float64 (n, fs)."""
    arr = np.reshape(np.asarray(ts[0], np.float64) * gw_norm_constant, (ts[0].shape[0], ts[0].shape[2]))
    with open(path, 'wb') as f:
        pickle.dump(arr, f, protocol=2)
    return arr


# --------------------------------------------------------------------------------------------------- on-GPU synthesis inside the train loop
class OnlineBank(object):
    """BASELINE config 5: templates (and, optionally, PSD-coloured noise whitened with the same PSD) synthesised on the GPU inside
    the training loop instead of being read from a stored bank -- no template, and no template parameter, ever touches host memory.

    Every draw() is a fresh batch from the prior, made by ONE kernel (gn_synth_templates_prior): each workgroup draws its masses by
    the hunt_constrain rejection rule and its idx from randint(convert_beta(beta)) out of a counter-based Philox stream, evaluates
    the chirp, whitens, transforms, aligns and crops, scales by gw_norm_constant and writes the fp32 row and its labels [mc, m2/m1].
    The stream is (seed, counter): data-parallel ranks pass different seeds and never exchange anything; two banks with
    the same seed produce the same batches.  noise='coloured' adds gen_noise (Philox) -> whiten_data('td') -> central crop, generated and
    whitened inside the same kernel launch (gn_synth_templates_noise; NoiseSynth);
    noise='white' adds N(0,1) as the reference's train loops do (bbhMahoGANy.py:1161, :1277).  Series lengths the fused kernel does
    not cover (N < 1024) draw the parameters on the host (prior='host' forces that everywhere)."""

    def __init__(self, fs, T_obs=4, psd=None, gw_norm_constant=1.0, beta=(0.45, 0.55), seed=1, noise=None, peak_off=PEAK_OFFSET, prior=None):
        self.fs, self.T_obs = int(fs), int(T_obs)
        self.N = self.fs * self.T_obs
        self.psd = np.asarray(psd, np.float64)
        self.syn = Synth(fs, T_obs, self.psd, 'H1', peak_off)
        self.g = float(gw_norm_constant)
        self.lo, self.hi = convert_beta(list(beta), fs, T_obs)
        self.rng = np.random.RandomState(seed)
        self.noise = noise
        self.seed = int(seed)
        self.counter = 0                   # noise stream position
        self.prior_counter = 0             # parameter stream position (seed + 2^32: disjoint from the noise stream)
        self.n_pix = self.fs
        self.prior = prior or ('device' if self.N in Synth.FUSED_N else 'host')
        self.nsyn = NoiseSynth(fs, T_obs, self.psd) if (noise == 'coloured' and self.N in Synth.FUSED_N) else None
        self._win = _d64(tukey(self.N, alpha=1.0 / 8.0)) if (noise == 'coloured' and self.nsyn is None) else None

    def draw_masses(self, n):
        """hunt_constrain prior (gw_template_maker.py:327-339), vectorised rejection sampling on the host (prior='host')."""
        out1, out2 = [], []
        lo, span = np.log(5.0), np.log(95.0) - np.log(5.0)
        need = n
        while need > 0:
            m = np.exp(lo + self.rng.uniform(0, 1, (4 * need + 64, 2)) * span)
            a, b = m[:, 0], m[:, 1]
            eta = a * b / (a + b) ** 2
            mc = (a + b) * eta ** 0.6
            ok = (a + b < 100.0) & (a > 5.0) & (b > 5.0) & (a >= b) & (b / a >= 0.5) & (mc >= 20.0) & (mc <= 35.0)
            out1.append(a[ok][:need]); out2.append(b[ok][:need])
            need -= len(out1[-1])
        return np.concatenate(out1), np.concatenate(out2)

    def _templates(self, batch, dtype):
        if self.prior == 'device':
            out, labels, _ = self.syn.templates_prior(batch, self.seed + (1 << 32), self.prior_counter, self.lo, self.hi, g=self.g, dtype=dtype)
            self.prior_counter += batch * Synth.PRIOR_TRIALS
            return out, labels
        m1, m2 = self.draw_masses(batch)
        idx = self.rng.randint(self.lo, self.hi, batch) if self.hi > self.lo else np.full(batch, self.lo)
        out, _ = self.syn.templates(m1, m2, idx, g=self.g, dtype=dtype)
        eta = m1 * m2 / (m1 + m2) ** 2
        labels = torch.as_tensor(np.stack([(m1 + m2) * eta ** 0.6, m2 / m1], axis=1).astype(np.float32)).to(device())
        return out, labels

    def draw_clean(self, batch):
        """-> (noise-free templates (batch, fs) fp32, labels): what the GAN loop's real images carry in column 0 (bbhMahoGANy.py:1277-1284)."""
        return self._templates(batch, torch.float32)

    def draw_noise(self, batch):
        """-> (batch, fs) fp32 noise rows of this bank's kind: 'coloured' = gen_noise -> whiten_data('td') -> crop in one launch
        (gn_noise_whitened), otherwise N(0,1) (what the reference's loops draw, :1161, :1277).  The GAN loop's noise column."""
        if self.noise == 'coloured' and self.nsyn is not None:
            out = self.nsyn.draw(batch, self.seed, self.counter, dtype=torch.float32)
            self.counter += batch * (self.N // 2 + 1)
            return out
        out = torch.empty((batch, self.fs), dtype=torch.float32, device=device())
        _lib.call('gn_fill_normal', out.data_ptr(), out.numel(), 0.0, 1.0, self.seed, self.counter, _s())
        self.counter += out.numel()
        return out

    def draw(self, batch):
        """-> (images (batch, fs) fp32 = template + this bank's noise, labels (batch, 2) fp32), both device tensors.
        noise='coloured': ONE launch per batch when the prior is drawn on the device (gn_synth_templates_noise: prior -> chirp -> inverse
        transforms -> align -> crop, then gen_noise -> whiten_data('td') -> crop in the same workgroup, sum written as fp32); with host-drawn
        parameters the template kernel writes fp64 rows and the fused noise kernel adds to them.  Series lengths below 1024 run the
        separate fp64 kernels."""
        c0 = int((self.T_obs / 2) * self.fs - self.fs / 2)
        if self.noise == 'coloured' and self.nsyn is not None:
            Nf = self.N // 2 + 1
            if self.prior == 'device':
                out, labels, _ = self.syn.templates_prior(batch, self.seed + (1 << 32), self.prior_counter, self.lo, self.hi, g=self.g, dtype=torch.float32,
                                                          noise=self.nsyn, noise_seed=self.seed, noise_counter=self.counter)
                self.prior_counter += batch * Synth.PRIOR_TRIALS
            else:
                ts, labels = self._templates(batch, torch.float64)
                out = self.nsyn.draw(batch, self.seed, self.counter, add=ts.contiguous(), dtype=torch.float32)
            self.counter += batch * Nf
            return out, labels
        if self.noise == 'coloured':
            ts, labels = self._templates(batch, torch.float64)
            nz = gen_noise_device(self.fs, self.T_obs, self.psd, batch, self.seed, self.counter)
            self.counter += batch * (self.N // 2 + 1)
            _mul(nz, self._win, False)
            X = _mul(rfft(nz), self.syn.scale, True)
            ts = (ts + irfft(X, self.N)[:, c0:c0 + self.fs]).contiguous()
            out = torch.empty((batch, self.fs), dtype=torch.float32, device=device())
            _lib.call('gn_f64_to_f32', ts.data_ptr(), out.data_ptr(), 1.0, ts.numel(), _s())
        else:
            out, labels = self._templates(batch, torch.float32)       # fp32 rows straight from the fused kernel
        if self.noise == 'white':
            _lib.call('gn_axpy', out.data_ptr(), self.draw_noise(batch).data_ptr(), 1.0, out.numel(), _s())
        return out, labels
