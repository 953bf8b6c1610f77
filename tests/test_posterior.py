"""Posterior read-out score: oracle pinned to the reference's own overlap_tests (golden), GPU KDE vs both."""
import os

import numpy as np
import pytest

from oracle import posterior_ref as P

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'posterior_golden.npz'))


def test_oracle_matches_reference_overlap_tests():
    pred = [G['pred_mc'], G['pred_q']]
    assert abs(P.beta_score(pred, G['lalinf']) - float(G['beta'])) <= 1e-12
    d_cnn = np.array([pred[0].reshape(-1), pred[1].reshape(-1)])
    assert np.allclose(P.kde_pdf(d_cnn, G['probe']), G['probe_pdf_cnn'], rtol=1e-12, atol=0)
    assert np.allclose(P.kde_pdf(G['lalinf'], G['probe']), G['probe_pdf_lal'], rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_gpu_overlap_tests_match_reference_golden():
    from gennet_amd import posterior
    pred = [G['pred_mc'], G['pred_q']]
    k_cnn, k_lal = posterior.make_kernels(pred, G['lalinf'])
    assert np.allclose(k_cnn.pdf(G['probe']), G['probe_pdf_cnn'], rtol=1e-12, atol=0)        # fp64 KDE kernel vs scipy
    assert np.allclose(k_lal.pdf(G['probe']), G['probe_pdf_lal'], rtol=1e-12, atol=0)
    ks, ad, beta = posterior.overlap_tests(pred, G['lalinf'], [30.0, 0.79], k_cnn, k_lal)
    assert abs(beta - float(G['beta'])) <= 1e-12                                                # reference: 0.97203794690...
    assert np.allclose(np.array([[ks[0][0], ks[0][1]], [ks[1][0], ks[1][1]]], dtype=np.float64), G['ks'], rtol=1e-12)
    assert np.allclose([ad[0][0], ad[1][0]], G['ad_stat'], rtol=1e-12)
    # identical sample sets overlap completely
    _, _, one = posterior.overlap_tests([G['lalinf'][0].reshape(-1, 1), G['lalinf'][1].reshape(-1, 1)], G['lalinf'])
    assert abs(one - 1.0) < 1e-12


@pytest.mark.gpu
def test_posterior_samples_pipeline_shapes():
    """bbhMahoGANy.py:1330-1343: generator.predict(4000 latent draws) -> signal_pe.predict -> [mc (n,1), q (n,1)]."""
    from gennet_amd import bbh
    nets = bbh.build_and_compile(np.zeros((64, 1), np.float32), 64)
    pe, waves = bbh.posterior_samples(nets, 96)
    assert waves.shape == (96, 64, 1) and pe[0].shape == (96, 1) and pe[1].shape == (96, 1)
    assert pe[0].dtype == np.float32 and np.isfinite(pe[0]).all() and np.all((pe[1] >= 0) & (pe[1] <= 1))


def test_score_grid_is_the_100j_mesh():
    """posterior.score_grid produces the point set and order of np.mgrid[a:b:100j, c:d:100j] -> vstack(ravel) (the mesh of bbhMahoGANy.py:858-859)
    bit for bit (host helper, no device call)."""
    from gennet_amd import posterior
    rng = np.random.RandomState(3)
    for _ in range(20):
        lo = rng.randn(2) * 10; hi = lo + np.abs(rng.randn(2)) * 5 + 1e-3
        X, Y = np.mgrid[lo[0]:hi[0]:100j, lo[1]:hi[1]:100j]
        assert np.array_equal(posterior.score_grid(lo, hi), np.vstack([X.ravel(), Y.ravel()]))
    a = rng.rand(50); b = rng.rand(50)
    assert posterior.pdf_overlap(a, a) == pytest.approx(1.0) and 0 < posterior.pdf_overlap(a, b) < 1
