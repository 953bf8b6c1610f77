"""End to end, as a BBH_version/ user would run it: scripts/make_templates.py (gw_template_maker.main, gw_template_maker.py:743-865)
writes the ts / params / event pickles, scripts/bbh_train.py (bbhMahoGANy.main, bbhMahoGANy.py:959-1382) reads them, trains the CNN
and the GAN on the GPU and leaves the reference's output files behind."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_template_maker_then_trainer_leave_the_reference_files(tmp_path):
    fs, n = 256, 64
    out = run([os.path.join(ROOT, 'scripts/make_templates.py'), '-N', str(n), '-Nb', str(n), '-f', str(fs), '-T', '1', '-m', 'hunt_constrain',
               '-z', '1', '-b', 'templates/'], str(tmp_path))
    assert 'success' in out
    tag = '_srate-%dhz_oversamp' % fs
    tsf = tmp_path / 'templates' / ('gw150914_ts_0_%dSamp%s.sav' % (n, tag))
    prf = tmp_path / 'templates' / ('gw150914_params_0_%dSamp%s.sav' % (n, tag))
    evf = tmp_path / 'data' / ('gw1509140%s.sav' % tag)
    assert tsf.exists() and prf.exists() and evf.exists()
    with open(str(tsf), 'rb') as f:
        ts = pickle.load(f, encoding='latin1')
    assert np.asarray(ts[0]).shape == (n, 1, fs) and np.isfinite(np.asarray(ts[0])).all()          # Appendix D layout: [ts (Ns,1,fs), yval]
    with open(str(evf), 'rb') as f:
        ev = pickle.load(f, encoding='latin1')
    assert np.asarray(ev).shape == (fs,)

    rng = np.random.RandomState(2)
    with open(str(tmp_path / 'lalinf_mc_q.sav'), 'wb') as f:                                       # stand-in for the lalinference posterior (mc, q)
        pickle.dump(np.array([rng.normal(30.0, 1.0, 500), rng.normal(0.8, 0.05, 500)]), f, protocol=2)
    out = run([os.path.join(ROOT, 'scripts/bbh_train.py'), '--templates', 'templates/', '--training-num', str(n), '--tag', tag, '--n-pix', str(fs),
               '--batch-size', '4', '--pe-batch-size', '8', '--pe-iter', '400', '--pe-cadence', '200', '--lr', '1e-3', '--max-iter', '7', '--cadence', '3', '--event-scale', '1.0', '--out', 'run',
               '--lalinf-posterior', 'lalinf_mc_q.sav'], str(tmp_path))
    assert 'Completed CNN PE' in out and '[sD loss:' in out and 'RMS:' in out and 'mean |error| (mc, q):' in out
    # do_old_model / do_only_old_pe_model (bbhMahoGANy.py:1133-1142): a second run starts from the files of the first and skips the CNN loop
    out2 = run([os.path.join(ROOT, 'scripts/bbh_train.py'), '--templates', 'templates/', '--training-num', str(n), '--tag', tag, '--n-pix', str(fs),
                '--batch-size', '4', '--max-iter', '4', '--cadence', '1', '--event-scale', '1.0', '--out', 'run', '--old-model', '--only-old-pe-model', '--graph'],
               str(tmp_path))
    assert 'Completed CNN PE' in out2 and 'PE loss' not in out2 and '[sD loss:' in out2
    if 'posterior overlap beta' in out:             # scored only once both read-outs vary (bbhMahoGANy.py:1352); a few hundred steps may not get there
        with open(str(tmp_path / 'run' / 'beta_score_hist.sav'), 'rb') as f:
            hist = pickle.load(f)
        assert 1 <= len(hist) <= 2 and all(0.0 <= b <= 1.0 for b in hist)
    for name in ('generator.h5', 'discriminator.h5', 'signal_dis_on_gen.h5', 'gan_pe_samples.sav', 'gan_pe_waveforms.sav',
                 'GAN_posterior_samples/posterior_samples_00006.sav'):
        assert (tmp_path / 'run' / name).exists(), name
    assert not list((tmp_path / 'run').glob('**/*.tmp'))                                           # every write was moved into place
    with open(str(tmp_path / 'run' / 'gan_pe_samples.sav'), 'rb') as f:
        pe = np.asarray(pickle.load(f, encoding='latin1'))
    assert pe.shape[0] == 2 and pe.shape[1] == 4000 and np.isfinite(pe).all()                       # (mc, q) x 4000 generator draws (:1330-1343)
    # the .h5 files are Keras-layout HDF5 the package reads back
    sys.path.insert(0, ROOT)
    from gennet_amd import bbh
    g = bbh.generator_model(fs)
    g.load_weights(str(tmp_path / 'run' / 'generator.h5'))
    assert all(np.isfinite(w).all() for w in g.get_weights())


@pytest.mark.gpu
def test_posterior_columns_to_sanity_check_set(tmp_path):
    """data/get_lalinf_pars.py -> lalinf_post_waveform_maker.main(): posterior columns -> m1_m2 / mc_q pickles -> the CNN sanity-check set."""
    fs, n = 256, 12
    rng = np.random.RandomState(3)
    np.savez(str(tmp_path / 'post.npz'), mc=rng.uniform(26, 32, n), q=rng.uniform(0.6, 1.0, n))
    run([os.path.join(ROOT, 'scripts/get_lalinf_pars.py'), '--posterior', 'post.npz', '--tag', 'srate-%d' % fs, '--out', 'data'], str(tmp_path))
    out = run([os.path.join(ROOT, 'scripts/make_posterior_templates.py'), '--posterior', 'data/gw150914_m1_m2_lainf_post_srate-%d.sav' % fs,
               '--mc-q-file', 'data/gw150914_mc_q_lalinf_post_srate-%d.sav' % fs, '-f', str(fs), '-T', '1', '-N', '9', '-Nb', '9', '-z', '1'], str(tmp_path))
    assert 'success' in out
    with open(str(tmp_path / 'data' / ('gw150914_cnn_sanity_check_ts_mass-time-vary_srate-%dhz_oversamp.sav' % fs)), 'rb') as f:
        ts = pickle.load(f)
    assert ts.shape == (9, fs) and ts.dtype == np.float64 and np.isfinite(ts).all() and np.abs(ts).max() > 0         # 8 posterior rows + the event-like template


@pytest.mark.gpu
def test_cadence_overlap_score_of_the_trainer():
    """bbh.posterior_overlap (bbhMahoGANy.py:1345-1356): None while a read-out is constant, else overlap_tests' triple with beta in [0, 1]
    and beta = 1 for identical sample sets."""
    sys.path.insert(0, ROOT)
    from gennet_amd import bbh
    rng = np.random.RandomState(4)
    lal = np.array([rng.normal(30.0, 1.0, 600), rng.normal(0.8, 0.05, 600)])
    pe = [rng.normal(30.2, 1.1, (500, 1)).astype(np.float32), rng.normal(0.79, 0.05, (500, 1)).astype(np.float32)]
    assert bbh.posterior_overlap([np.zeros((500, 1), np.float32), pe[1]], lal) is None
    ks, ad, beta = bbh.posterior_overlap(pe, lal)
    assert 0.5 < beta <= 1.0 and 0.0 <= ks[0][1] <= 1.0
    same = bbh.posterior_overlap([lal[0].reshape(-1, 1), lal[1].reshape(-1, 1)], lal)
    assert abs(same[2] - 1.0) < 1e-9
