"""End-to-end behaviour on the GPU: the CNN point-estimator actually LEARNS (mc, q) from templates synthesised on the device
(BASELINE configs 1 and 5 at reduced size): what the reference's CNN loop (bbhMahoGANy.py:1153-1168) is for."""
import numpy as np
import pytest

from oracle import synth_ref as S

pytestmark = pytest.mark.gpu


def test_point_estimator_learns_chirp_mass_from_online_templates():
    from gennet_amd import bbh, engine, templates as T
    fs, B, steps = 1024, 64, 800
    engine.set_init_seed(3); engine.set_device_seed(11)
    psd = S.analytic_psd(fs * 4 // 2 + 1, 0.25)
    ob = T.OnlineBank(fs, 4, psd, seed=2, noise=None)
    x0, y0 = ob.draw(256)
    ob.g = 1.0 / float(x0.std())                              # unit-variance inputs, the role of gw_norm_constant (gw_template_maker.py:782)
    pe = bbh.signal_pe_model(fs)
    pe.compile(loss='mean_squared_error', optimizer=engine.Adam(lr=3e-4, beta_1=0.5), metrics=['accuracy'])
    rng = np.random.RandomState(0)
    hist = [bbh.pe_train_step_online(pe, ob, B, cnn_noise_frac=0.0, nprng=rng) for _ in range(steps)]
    first = np.mean([h[1] for h in hist[:5]]); last = np.mean([h[1] for h in hist[-100:]])
    prior_var = float(np.var(y0[:, 0].cpu().numpy()))          # what a constant predictor at the prior mean would score
    print('mc loss first %.1f last %.3f prior variance %.2f; q loss first %.3f last %.4f' % (first, last, prior_var, np.mean([h[2] for h in hist[:5]]), np.mean([h[2] for h in hist[-100:]])))
    assert first > 300.0                                       # starts near E[mc^2]
    assert last < 0.25 * prior_var                             # and ends far below the trivial predictor (measured: 0.3-1.7 against 17.8)
    xt, yt = ob.draw(256)
    mc_hat, q_hat = pe.predict(xt.reshape(256, fs, 1).cpu().numpy())
    err = mc_hat[:, 0] - yt[:, 0].cpu().numpy()
    assert np.sqrt(np.mean(err ** 2)) < 0.5 * np.sqrt(prior_var)
    assert np.corrcoef(mc_hat[:, 0], yt[:, 0].cpu().numpy())[0, 1] > 0.9
    # the q head ends in ReLU(max_value=1) (bbhMahoGANy.py:400): outputs stay in [0, 1].  (Under Adam it usually overshoots 1 in the
    # first steps and then sits in the clipped region, where keras' clip gradient is 0 -- a property of the reference's
    # architecture that this engine reproduces; no claim is made on q here.)
    assert (q_hat >= 0).all() and (q_hat <= 1).all()


def test_gan_loop_runs_stably_and_discriminator_learns():
    """120 GAN iterations at a reduced size (n_pix 256, batch 16) through bbh.gan_train_step, the loop body of bbhMahoGANy.py:1243-1299:
    every loss stays finite, BatchNorm moving statistics stay finite and positive, and the discriminator -- which sees clean
    templates + N(0,1) noise against generator output + (event - generator output) -- separates them better than chance."""
    import random
    from gennet_amd import bbh, engine, templates as T
    n_pix, B = 256, 16
    engine.set_init_seed(5); engine.set_device_seed(21); random.seed(3); np.random.seed(3)
    psd = S.analytic_psd(n_pix * 4 // 2 + 1, 0.25)
    ob = T.OnlineBank(n_pix, 4, psd, seed=4, noise=None)
    x0, y0 = ob.draw(512)
    x0 = x0 / x0.std()
    bank = bbh.DeviceBank(x0.contiguous(), y0.contiguous())
    ev = (x0[0].cpu().numpy() + np.random.RandomState(1).randn(n_pix)).astype(np.float32).reshape(n_pix, 1)
    nets = bbh.build_and_compile(ev, n_pix, lr=2e-4)
    event = engine.to_device(ev.reshape(-1))
    hist = np.array([bbh.gan_train_step(nets, bank, event, B) for _ in range(120)])
    assert np.isfinite(hist).all()
    sg_loss, sg_acc, sd_loss, sd_acc = hist.T
    assert sd_acc[-40:].mean() > 0.6, sd_acc[-40:].mean()                        # D does better than a coin
    assert sd_loss[-40:].mean() < sd_loss[:10].mean()                            # and its loss came down
    for l in nets.generator.layers:
        if l.__class__.__name__ == 'BatchNormalization':
            mv = l.moving_variance.numpy(); mm = l.moving_mean.numpy()
            assert np.isfinite(mv).all() and np.isfinite(mm).all() and (mv > 0).all()
    fake = nets.generator.predict(np.random.RandomState(2).uniform(-1, 1, (8, 100)).astype(np.float32))
    assert fake.shape == (8, n_pix, 1) and np.isfinite(fake).all()
