"""Reduced known-answer validation of the product's result (row x2): scripts/validate_posterior.py at a budget of about a minute.

The full-budget run (the reference's 4 M waveforms per loop, bbhMahoGANy.py:86-89) is recorded in profiles/r03_posterior_validation.json;
this test keeps the machinery honest and puts thresholds on what a short run must already show:
  * the exact grid posterior (the known answer, computed with the synthesiser itself) contains the true parameters and is reproducible;
  * the CNN point-estimator beats the prior-mean predictor by a wide margin on held-out templates;
  * the generator, which only ever sees the NOISY event through the discriminator, produces waveforms that overlap the CLEAN event,
    and the (mc) read-out of its samples lands near the exact posterior's mean.
Thresholds are stated per assert; they are far looser than the reference's own yard-stick (pe_std, :1345) because the budget is 1 / 30 of it.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_run_recovers_the_event_and_localises_chirp_mass():
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import validate_posterior as V
    args = V.parse(['--bank', '20000', '--pe-batch', '128', '--pe-iter', '2500', '--cnn-seconds', '60', '--gan-batch', '8', '--gan-iter', '2000',
                    '--gan-seconds', '60', '--cadence', '500', '--graph', '--seed', '1', '--out', ''])
    out = V.run(args)
    ex, ev = out['exact_posterior'], out['event']
    # the known answer: true (mc, q) inside 2.5 sigma of the exact posterior, whose two independent sample sets overlap completely
    assert abs(ex['mc_mean'] - ev['mc']) < 2.5 * ex['mc_std'] and abs(ex['q_mean'] - ev['q']) < 2.5 * ex['q_std']
    assert ex['self_overlap_beta'] > 0.99 and 0.1 < ex['mc_std'] < 2.0 and ev['optimal_snr'] > 8
    # CNN: mean |error| on 4000 held-out templates below 20 % (mc) / 65 % (q) of the prior's standard deviation after 2500 steps of 128.  (Round 5: 15 % / 50 %
    # turned out to be one seed's luck -- profiles/r05_winograd_gate.txt: the same short run over seeds 1-4 gives 0.31-0.80 in mc, the direct kernels' seed 3
    # above the old bound, and 0.049-0.063 / 0.050-0.085 in q under the direct / transform-domain kernels; the prior's deviations are 4.30 / 0.144.)
    cnn = out['cnn']
    assert cnn['steps'] == 2500
    e_mc, e_q = cnn['mean_abs_error_heldout [mc, q]']
    s_mc, s_q = cnn['prior_std [mc, q]']
    assert e_mc < 0.2 * s_mc and e_q < 0.65 * s_q, (e_mc, e_q, s_mc, s_q)
    # the q head's ReLU(max_value=1) rails (bbhMahoGANy.py:400) are reported, and a trained head is not pinned to them: the prior has q in [0.5, 1], so
    # nothing may sit at 0 and at most a third of the held-out rows (those with q near 1) at the upper rail
    rails = cnn['q_head_rails (ReLU(max_value=1), :400)']
    assert rails['fraction_at_0'] == 0.0 and rails['fraction_at_1'] < 0.34, rails
    assert rails['mean_abs_error_q_off_rail'] is not None and rails['mean_abs_error_q_off_rail'] < 0.65 * s_q, rails
    # GAN after 2000 iterations of batch 8.  Round 4 settled what happens later (the discriminator wins outright and the pair falls into a
    # saturated state of Keras 2.2.4's binary cross-entropy between iteration ~3 000 and ~12 000, in the HIP path AND in the independent torch-CPU port,
    # profiles/r04_gan_dynamics_*.json), so the test scores the state the loop is in BEFORE that and asserts that it is the non-saturated one:
    #  * the generator's waveforms overlap the clean event it only ever saw through the discriminator (normalised inner product > 0.35 -- the recorded
    #    runs show 0.52-0.93 at this point, an untrained generator ~0) and have the event's scale (rms < 3; the saturated states sit at 7-17);
    #  * neither loss is pinned at a clip value of the cross-entropy: sg_loss < 15 (16.1 = -log 1e-7 is the generator-starved state), sd_loss < 1
    #    (7.97 = -0.5 log 1e-7 is the discriminator-starved state), and the generator still receives a gradient (sg_loss finite and > 0);
    #  * the chirp-mass read-out of 4000 draws sits within 5 solar masses of the exact posterior's mean (recorded: 1.1-3.3 with this test's short CNN
    #    training; the prior spans 15, the untrained generator reads 15-21 away).
    gan = out['gan']
    assert gan['iterations'] == 2000
    assert gan['trajectory'][0]['waveform_overlap_with_clean_event'] < 0.2 and abs(gan['trajectory'][0]['mc_mean'] - ex['mc_mean']) > 8
    fin = gan['final']
    assert fin['waveform_overlap_with_clean_event'] > 0.35 and fin['waveform_rms'] < 3.0, fin
    late = [r for r in gan['trajectory'] if r['iteration'] >= 1000]
    assert late and all(0.0 < r['sg_loss'] < 15.0 and r['sd_loss'] < 1.0 and r['waveform_rms'] < 3.0 for r in late), late
    assert abs(fin['mc_mean'] - ex['mc_mean']) < 5.0, (fin, ex)
    assert np.isfinite([fin['q_mean'], fin['mc_std']]).all()
