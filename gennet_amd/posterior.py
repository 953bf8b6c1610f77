"""Posterior read-out scoring, the numbers bbhMahoGANy.py prints every `cadence` iterations:
`overlap_tests` (bbhMahoGANy.py:811-873) = two-sample KS and Anderson-Darling tests per parameter plus the KDE overlap
"beta score" of the CNN/GAN samples against the lalinference posterior samples.

The two KDE evaluations on the 100 x 100 grid (2 x 10^4 points x ~4000 samples each) run on the device (gn_kde2d_pdf);
bandwidth, covariance and normalisation follow scipy.stats.gaussian_kde (Scott's rule), which is what the reference builds
in make_contour_plot (:790).  The KS / AD tests are the same scipy.stats calls the reference makes (host-side statistics on
~8000 scalars; reporting, not arithmetic of the hot path).
"""
import numpy as np
import torch

from . import _lib
from .engine import device


class GaussianKDE(object):
    """scipy.stats.gaussian_kde(dataset) for a (2, n) dataset: factor = n^(-1/6) (Scott), covariance = factor^2 * cov(dataset),
    pdf(points) evaluated by the HIP kernel."""

    def __init__(self, dataset):
        self.dataset = np.atleast_2d(np.asarray(dataset, np.float64))
        assert self.dataset.shape[0] == 2, 'the BBH read-out is two-dimensional (mc, q)'
        self.d, self.n = self.dataset.shape
        self.factor = self.n ** (-1.0 / (self.d + 4))
        self.covariance = np.atleast_2d(np.cov(self.dataset, rowvar=1, bias=False)) * self.factor ** 2
        self.inv_cov = np.linalg.inv(self.covariance)
        self.norm = 1.0 / (np.sqrt(np.linalg.det(2 * np.pi * self.covariance)) * self.n)
        self._dev = None

    def pdf(self, points):
        pts = np.ascontiguousarray(np.atleast_2d(np.asarray(points, np.float64)))
        assert pts.shape[0] == 2
        if self._dev is None:
            self._dev = torch.as_tensor(np.ascontiguousarray(self.dataset)).to(device())
        p = torch.as_tensor(pts).to(device())
        out = torch.empty(pts.shape[1], dtype=torch.float64, device=device())
        _lib.call('gn_kde2d_pdf', self._dev.data_ptr(), self.n, p.data_ptr(), pts.shape[1], float(self.inv_cov[0, 0]), float(self.inv_cov[0, 1]),
                  float(self.inv_cov[1, 1]), float(self.norm), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out.cpu().numpy()

    __call__ = pdf


def make_kernels(pred_samp, lalinf_samp):
    """The two KDEs plot_pe_samples builds through make_contour_plot (bbhMahoGANy.py:660-678, :790)."""
    k_cnn = GaussianKDE(np.array([np.reshape(pred_samp[0], (-1,)), np.reshape(pred_samp[1], (-1,))]))
    k_lal = GaussianKDE(np.asarray(lalinf_samp)[:2])
    return k_cnn, k_lal


def overlap_tests(pred_samp, lalinf_samp, true_vals=None, kernel_cnn=None, kernel_lalinf=None):
    """bbhMahoGANy.py:811-873 for the two-network read-out (comb_pe_model = False): pred_samp = [mc (n,1), q (n,1)],
    lalinf_samp (2, m).  Returns (ks_score, ad_score, beta_score)."""
    from scipy.stats import anderson_ksamp, ks_2samp
    if kernel_cnn is None or kernel_lalinf is None:
        kernel_cnn, kernel_lalinf = make_kernels(pred_samp, lalinf_samp)
    p0 = np.reshape(pred_samp[0], (-1,)); p1 = np.reshape(pred_samp[1], (-1,))
    l0 = np.asarray(lalinf_samp[0][:]); l1 = np.asarray(lalinf_samp[1][:])
    ks_score = np.array([ks_2samp(p0, l0), ks_2samp(p1, l1)])
    ad_score = [anderson_ksamp([p0, l0]), anderson_ksamp([p1, l1])]
    comb_mc = np.concatenate((np.reshape(pred_samp[0], (-1, 1)), l0.reshape(-1, 1)))
    comb_q = np.concatenate((np.reshape(pred_samp[1], (-1, 1)), l1.reshape(-1, 1)))
    X, Y = np.mgrid[np.min(comb_mc):np.max(comb_mc):100j, np.min(comb_q):np.max(comb_q):100j]
    positions = np.vstack([X.ravel(), Y.ravel()])
    cnn_pdf = kernel_cnn.pdf(positions)
    lalinf_pdf = kernel_lalinf.pdf(positions)
    beta_score = np.divide(np.sum(cnn_pdf * lalinf_pdf), np.sqrt(np.sum(cnn_pdf ** 2) * np.sum(lalinf_pdf ** 2)))
    return ks_score, ad_score, beta_score
