"""CPU oracle for the posterior read-out score: numpy restatement of overlap_tests (bbhMahoGANy.py:811-873, comb_pe_model = False)
and of the scipy.stats.gaussian_kde formulas it relies on.  TEST INFRASTRUCTURE ONLY (see keras_ref.py header).

Pinned by tests/golden/posterior_golden.npz, which tests/golden/make_golden.py produced by executing the reference's own
overlap_tests (with scipy KDEs built as make_contour_plot does, :790) on seeded samples.
"""
import numpy as np


def kde_pdf(dataset, points):
    """scipy.stats.gaussian_kde(dataset).pdf(points): Scott factor n^(-1/(d+4)), full covariance, Gaussian kernels."""
    dataset = np.atleast_2d(np.asarray(dataset, np.float64)); points = np.atleast_2d(np.asarray(points, np.float64))
    d, n = dataset.shape
    cov = np.atleast_2d(np.cov(dataset, rowvar=1, bias=False)) * (n ** (-1.0 / (d + 4))) ** 2
    inv = np.linalg.inv(cov)
    norm = 1.0 / (np.sqrt(np.linalg.det(2 * np.pi * cov)) * n)
    out = np.zeros(points.shape[1])
    for s in range(0, points.shape[1], 512):
        diff = points[:, None, s:s + 512] - dataset[:, :, None]                # (d, n, chunk)
        e = np.einsum('inp,ij,jnp->np', diff, inv, diff)
        out[s:s + 512] = np.exp(-0.5 * e).sum(axis=0)
    return out * norm


def beta_score(pred_samp, lalinf_samp):
    """The KDE overlap of :853-870: both pdfs on a 100 x 100 grid spanning the combined samples; normalised inner product."""
    p0 = np.reshape(pred_samp[0], (-1,)); p1 = np.reshape(pred_samp[1], (-1,))
    l0 = np.asarray(lalinf_samp[0]); l1 = np.asarray(lalinf_samp[1])
    comb_mc = np.concatenate((p0, l0)); comb_q = np.concatenate((p1, l1))
    X, Y = np.mgrid[np.min(comb_mc):np.max(comb_mc):100j, np.min(comb_q):np.max(comb_q):100j]
    pos = np.vstack([X.ravel(), Y.ravel()])
    a = kde_pdf(np.array([p0, p1]), pos)
    b = kde_pdf(np.array([l0, l1]), pos)
    return np.sum(a * b) / np.sqrt(np.sum(a ** 2) * np.sum(b ** 2))
