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


def score_grid(lo, hi, n=100):
    """(2, n*n) evaluation points: n x n nodes spanning [lo[0], hi[0]] x [lo[1], hi[1]] inclusive, first coordinate slowest -- the point set
    and order of the reference's 100j x 100j mesh (bbhMahoGANy.py:858-859); the order only matters for bit-for-bit sums."""
    axes = [lo[k] + np.arange(n) * ((hi[k] - lo[k]) / (n - 1)) for k in range(2)]      # node k = lo + k * step: the same doubles the mesh holds
    return np.stack([np.repeat(axes[0], n), np.tile(axes[1], n)])


def pdf_overlap(pa, pb):
    """Normalised inner product of two densities sampled on the same points: sum(a b) / sqrt(sum(a^2) sum(b^2)) (:870)."""
    return np.divide(np.sum(pa * pb), np.sqrt(np.sum(pa ** 2) * np.sum(pb ** 2)))


def overlap_tests(pred_samp, lalinf_samp, true_vals=None, kernel_cnn=None, kernel_lalinf=None):
    """bbhMahoGANy.py:811-873 for the two-network read-out (comb_pe_model = False): pred_samp = [mc (n,1), q (n,1)],
    lalinf_samp (2, m).  Returns (ks_score, ad_score, beta_score): per-parameter two-sample KS and Anderson-Darling tests, and the overlap of
    the two KDEs on a 100 x 100 grid spanning the joint range of both sample sets."""
    from scipy.stats import anderson_ksamp, ks_2samp
    if kernel_cnn is None or kernel_lalinf is None:
        kernel_cnn, kernel_lalinf = make_kernels(pred_samp, lalinf_samp)
    ours = [np.reshape(pred_samp[k], (-1,)) for k in range(2)]
    theirs = [np.asarray(lalinf_samp[k]).reshape(-1) for k in range(2)]
    ks_score = np.array([ks_2samp(ours[k], theirs[k]) for k in range(2)])
    ad_score = [anderson_ksamp([ours[k], theirs[k]]) for k in range(2)]
    both = [np.concatenate((ours[k], theirs[k])) for k in range(2)]
    pts = score_grid([b.min() for b in both], [b.max() for b in both])
    return ks_score, ad_score, pdf_overlap(kernel_cnn.pdf(pts), kernel_lalinf.pdf(pts))
