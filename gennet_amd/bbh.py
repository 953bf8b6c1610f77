"""Host-side mirror of BBH_version/bbhMahoGANy.py: model builders, compile wiring and the two training loops.

Function names, arguments and semantics follow the reference script; its module-level globals (bbhMahoGANy.py:84-113)
become the fields of `Config`.  Everything numerical runs in the HIP kernels behind gennet_amd.ops.

  generator_model                       bbhMahoGANy.py:212-295
  signal_pe_model (two-branch)          :356-404   (the comb_pe_model=True branch is dead code in the reference: it reads an
                                                    undefined `batchnorm`, :318 -- not provided)
  signal_discriminator_model            :408-498   (active configuration num_lays=2, batchnorm=False, maxpool=False)
  data_subtraction_model / MyLayer      :164-210
  generator_after_subtracting_noise     :500-519
  generator_containing_signal_discriminator :521-539
  set_trainable                         :797-809
  build_and_compile                     :1089-1119
  pe_train_step / gan_train_step        :1153-1168 / :1241-1299
"""
import random as _pyrandom

import numpy as np
import torch

from . import ops
from .engine import Adam, Input, Model, Sequential, StepGraph, capturing, device, device_rng, to_device
from .layers import (Activation, BatchNormalization, Conv1D, Conv2D, Dense, Dropout, Flatten, LeakyReLU, MaxPooling2D, MyLayer, ReLU,
                     Reshape, UpSampling1D)


class Config(object):
    """bbhMahoGANy.py:84-113 (only the fields the hot path reads)."""

    def __init__(self, **kw):
        self.n_pix = 1024
        self.n_sig = 1.0
        self.batch_size = 8
        self.pe_batch_size = 8
        self.lr = 9e-5
        self.n_noise_real = 1
        self.cnn_noise_frac = 1.0 / 8.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError('unknown config field %r' % k)
            setattr(self, k, v)


def generator_model(n_pix=1024, filtsize=5):
    """bbhMahoGANy.py:212-295.  filtsize: the reference's edit-the-file knob (:228, `filtsize = 5 # 10 is best`); 1..10."""
    model = Sequential(name='generator')
    act, momentum, drate, padding, weights = 'tanh', 0.99, 0.2, 'same', 'glorot_uniform'
    model.add(Dense(256 * 1 * int(n_pix / 2), kernel_initializer=weights, input_shape=(100,)))
    model.add(BatchNormalization(momentum=momentum))
    model.add(Activation(act))
    model.add(Dropout(drate))
    model.add(Reshape((int(n_pix / 2), 256)))
    for i, (filters, strides, up) in enumerate(((64, 2, True), (128, 1, True), (256, 1, False), (512, 1, False), (1024, 1, False))):
        if up:
            model.add(UpSampling1D(size=2))
        model.add(Conv1D(filters, filtsize, kernel_initializer=weights, strides=strides, padding=padding))
        model.add(BatchNormalization(momentum=momentum))
        model.add(Activation(act))
        model.add(Dropout(drate))
    model.add(Conv1D(1, filtsize, padding=padding))
    model.add(Activation('linear'))
    model._config = ('generator_model', n_pix, filtsize)
    return model


def signal_pe_model(n_pix=1024):
    """bbhMahoGANy.py:356-404: chirp-mass branch and inverse-mass-ratio branch sharing one input."""
    inputs = Input(shape=(n_pix, 1))
    act = 'relu'
    mc_branch = Conv1D(64, 5, strides=2, padding='same')(inputs)
    mc_branch = Activation(act)(mc_branch)
    for filters in (128, 256, 512):
        mc_branch = Conv1D(filters, 5, strides=2)(mc_branch)
        mc_branch = Activation(act)(mc_branch)
    mc_branch = Flatten()(mc_branch)
    mc_branch = Dense(1)(mc_branch)
    mc_branch = Activation('relu')(mc_branch)

    q_branch = Conv1D(64, 5, strides=1, padding='same')(inputs)
    q_branch = Activation(act)(q_branch)
    for filters, strides in ((128, 1), (256, 1), (512, 2), (1024, 2)):
        q_branch = Conv1D(filters, 5, strides=strides)(q_branch)
        q_branch = Activation(act)(q_branch)
    q_branch = Flatten()(q_branch)
    q_branch = Dense(1)(q_branch)
    q_branch = ReLU(max_value=1.0)(q_branch)
    model = Model(inputs=inputs, outputs=[mc_branch, q_branch], name='pe net')
    model._config = ('signal_pe_model', n_pix)
    return model


def signal_discriminator_model(n_pix=1024, num_lays=2, batchnorm=False, maxpool=False):
    """bbhMahoGANy.py:408-498.  Defaults = the active configuration (:424-426); num_lays 1..6, batchnorm and maxpool are the reference's edit-the-file
    knobs, layer order as written there (layer 2 normalises AFTER its activation, :449-450; layers 3-6 before, :457-458)."""
    drate, alpha, padding, weights, filtsize, n_neuron_scale, momentum = 0.4, 0.2, 'same', 'glorot_uniform', (5, 5), 4, 0.99
    if not 1 <= num_lays <= 6:
        raise ValueError('num_lays %r (the reference writes out layers 1..6, bbhMahoGANy.py:436-490)' % (num_lays,))
    model = Sequential(name='signal_discriminator')
    for i, (filters, strides) in enumerate(((64 * n_neuron_scale, (2, 1)), (128 * n_neuron_scale, (2, 1)), (256, (1, 1)), (512, (1, 1)), (1024, (1, 1)),
                                            (1024, (1, 1)))[:num_lays]):
        kw = {'input_shape': (n_pix, 2, 1)} if i == 0 else {}
        model.add(Conv2D(filters, filtsize, kernel_initializer=weights, strides=strides, padding=padding, **kw))
        if batchnorm and i >= 2:
            model.add(BatchNormalization(momentum=momentum))
        model.add(LeakyReLU(alpha=alpha))
        if batchnorm and i == 1:
            model.add(BatchNormalization(momentum=momentum))
        model.add(Dropout(drate))
        if maxpool:
            model.add(MaxPooling2D(pool_size=(2, 1)))
    model.add(Flatten())
    model.add(Dense(1))
    model.add(Activation('sigmoid'))
    model._config = ('signal_discriminator_model', n_pix, num_lays, batchnorm, maxpool)
    return model


def data_subtraction_model(noise_signal, npix):
    """bbhMahoGANy.py:190-210."""
    model = Sequential(name='data_subtraction')
    model.add(MyLayer(noise_signal, input_shape=(npix, 1)))
    return model


def generator_after_subtracting_noise(generator, data_subtraction):
    """bbhMahoGANy.py:500-519."""
    model = Sequential()
    model.add(generator)
    model.add(data_subtraction)
    return model


def generator_containing_signal_discriminator(generator, signal_discriminator):
    """bbhMahoGANy.py:521-539."""
    model = Sequential()
    model.add(generator)
    model.add(signal_discriminator)
    return model


def set_trainable(model, trainable):
    """bbhMahoGANy.py:797-809."""
    model.trainable = trainable
    for layer in model.layers:
        layer.trainable = trainable


def model_from_config(cfg):
    return {'generator_model': generator_model, 'signal_pe_model': signal_pe_model,
            'signal_discriminator_model': signal_discriminator_model}[cfg[0]](*cfg[1:])


class Nets(object):
    pass


def chisquare_loss(n_sig=1.0):
    """chisquare_Loss of bbhMahoGANy.py:146-162 for a noise standard deviation n_sig (:85), written with the backend facade exactly as the
    script writes it; Model.compile traces it once and lowers it to the squared-error kernel with the scale 1 / n_sig^2."""
    from .keras import backend as K

    def chisquare_Loss(yTrue, yPred):
        return K.sum(K.square(yTrue - yPred) / (n_sig ** 2), axis=-1)
    return chisquare_Loss


def build_and_compile(noise_signal, n_pix, lr=9e-5, do_pe=True, data_parallel=None, chi_loss=False, n_sig=1.0, filtsize=5, d_config=None):
    """bbhMahoGANy.py:1089-1119, in the reference's order (the order fixes which weights each compiled model trains):
    the combined model is compiled while the discriminator is frozen, the discriminator after it is unfrozen.
    chi_loss (:97, :1106-1109): the combined model trains on chisquare_Loss instead of binary cross-entropy; filtsize (:228): the generator's filter size; d_config (:424-426): the discriminator's
    num_lays / batchnorm / maxpool."""
    nets = Nets()
    nets.generator = generator_model(n_pix, filtsize)
    nets.signal_discriminator = signal_discriminator_model(n_pix, **(d_config or {}))
    nets.data_subtraction = data_subtraction_model(noise_signal, n_pix)
    nets.signal_pe = signal_pe_model(n_pix) if do_pe else None
    dp = data_parallel
    nets.data_subtraction_on_generator = generator_after_subtracting_noise(nets.generator, nets.data_subtraction)
    nets.data_subtraction_on_generator.compile(loss='binary_crossentropy', optimizer=Adam(lr=lr, beta_1=0.5), metrics=['accuracy'], data_parallel=dp)
    nets.signal_discriminator_on_generator = generator_containing_signal_discriminator(nets.data_subtraction_on_generator, nets.signal_discriminator)
    set_trainable(nets.signal_discriminator, False)
    nets.signal_discriminator_on_generator.compile(loss=chisquare_loss(n_sig) if chi_loss else 'binary_crossentropy', optimizer=Adam(lr=lr, beta_1=0.5),
                                                   metrics=['accuracy'], data_parallel=dp)
    set_trainable(nets.signal_discriminator, True)
    nets.signal_discriminator.compile(loss='binary_crossentropy', optimizer=Adam(lr=lr, beta_1=0.5), metrics=['accuracy'], data_parallel=dp)
    if do_pe:
        nets.signal_pe.compile(loss='mean_squared_error', optimizer=Adam(lr=lr, beta_1=0.5), metrics=['accuracy'], data_parallel=dp)
    return nets


# --------------------------------------------------------------------------------------------------------------
# training steps.  The template bank, labels and event stay resident in HBM; batch selection uses the host python
# `random` stream exactly like the reference, everything else (noise, latent vectors, assembly) happens on the device.
# --------------------------------------------------------------------------------------------------------------
class DeviceBank(object):
    """signal_train_images (Ns, n_pix) and signal_train_pars (Ns, 2) = [mc, m2/m1] in HBM (bbhMahoGANy.py:1007-1014)."""

    def __init__(self, images, pars):
        self.images = to_device(np.asarray(images, np.float32)) if not isinstance(images, torch.Tensor) else images.to(device()).float().contiguous()
        self.pars = to_device(np.asarray(pars, np.float32)) if not isinstance(pars, torch.Tensor) else pars.to(device()).float().contiguous()
        self.n = self.images.shape[0]
        self.n_pix = self.images.shape[1]

    def sample(self, batch, rng=_pyrandom, rank=0, world=1):
        return torch.tensor(sample_indices(self.n, batch, rng, rank, world), dtype=torch.int64, device=device())


def sample_indices(n, batch, rng=_pyrandom, rank=0, world=1):
    """random.sample of `batch*world` distinct template indices (bbhMahoGANy.py:1156, :1244) from the host stream that every
    rank advances identically; rank r keeps rows [r*batch, (r+1)*batch), so the union over ranks is exactly the batch a
    single process would have drawn for the global batch size."""
    idx = rng.sample(range(n), batch * world)
    return idx[rank * batch:(rank + 1) * batch]


def _draw_rows(fill, rows, row_len, rank, world, *args):
    """fill((rows, row_len), *args, seed, offset, device) for this rank's rows [rank * rows, (rank + 1) * rows) of a global draw of world * rows
    rows: the counters of ONE stream that every rank advances alike (engine.PhiloxStream.take_rows) -- N ranks draw what one process would."""
    seed, offs = device_rng().take_rows(row_len, [(rank * rows, rows)], rows * world)
    return fill((rows, row_len), *args, seed, offs[0], device())


def pe_train_step(signal_pe, bank, batch, cnn_noise_frac=1.0 / 8.0, rng=_pyrandom, nprng=np.random, rank=0, world=1):
    """One iteration of the CNN loop, bbhMahoGANy.py:1155-1165.  Noise N(0, sigma), sigma ~ U(0,5) drawn once per batch
    on the host stream, is added to the first int(B*cnn_noise_frac) rows (the reference hard-codes the length 1024 at
    :1161; here it is n_pix)."""
    it = bank.sample(batch, rng, rank, world)
    x = ops.gather_rows(bank.images, it)
    y = ops.gather_rows(bank.pars, it)
    # the first int(B * frac) rows of the GLOBAL batch are the noisy ones (:1161); rank r holds global rows [r * batch, (r + 1) * batch)
    n_noisy_global = int(batch * world * cnn_noise_frac)
    n_noisy = max(0, min(batch, n_noisy_global - rank * batch))
    sigma = float(nprng.uniform(0, 5))
    if n_noisy_global > 0:
        seed, offs = device_rng().take_rows(bank.n_pix, [(rank * batch, n_noisy)] if n_noisy > 0 else [], n_noisy_global)
        if n_noisy > 0:
            noise = ops.fill_normal((n_noisy, bank.n_pix), 0.0, sigma, seed, offs[0], device())
            ops.axpy(x[:n_noisy], noise, 1.0)
    return signal_pe.train_on_batch(x.reshape(batch, bank.n_pix, 1), [y[:, 0].contiguous(), y[:, 1].contiguous()])


def pe_train_step_online(signal_pe, online_bank, batch, cnn_noise_frac=1.0 / 8.0, nprng=np.random):
    """The CNN loop body with templates synthesised on the GPU for every batch (BASELINE config 5; templates.OnlineBank) instead of
    gathered from a stored bank; otherwise identical to pe_train_step."""
    x, y = online_bank.draw(batch)
    n_noisy = int(batch * cnn_noise_frac)
    sigma = float(nprng.uniform(0, 5))
    if n_noisy > 0:
        seed, off = device_rng().take(n_noisy * online_bank.n_pix)
        ops.axpy(x[:n_noisy], ops.fill_normal((n_noisy, online_bank.n_pix), 0.0, sigma, seed, off, device()), 1.0)
    return signal_pe.train_on_batch(x.reshape(batch, online_bank.n_pix, 1), [y[:, 0].contiguous(), y[:, 1].contiguous()])


def assemble_discriminator_batch(real, noise, fake, event):
    """bbhMahoGANy.py:1268-1289 on the device, vectorised (the reference's np.append loop is O(B^2) host copies):
    real images [template | N(0,1) noise], fake images [G(z) | event - G(z)] in REVERSED sample order (:1271 prepends),
    labels 1...1 0...0."""
    B, n = real.shape[0], real.shape[1]
    sX = ops.assemble_d_batch(real.reshape(B, n).contiguous(), noise.reshape(B, n).contiguous(), fake.reshape(B, n).contiguous(), event)
    sy = torch.cat([torch.ones(B, device=real.device), torch.zeros(B, device=real.device)])
    return sX, sy


def gan_train_step_online(nets, online_bank, event, batch, predict_batch=32):
    """The GAN loop body with the real half of the discriminator batch synthesised on the GPU for this iteration (BASELINE config 5;
    templates.OnlineBank) instead of gathered from a stored bank: column 0 = noise-free templates from the prior, column 1 = the bank's
    noise (PSD-coloured and whitened by the fused kernel for noise='coloured', N(0,1) otherwise); otherwise identical to gan_train_step."""
    real, _ = online_bank.draw_clean(batch)
    noise = online_bank.draw_noise(batch) if online_bank.noise == 'coloured' else None
    return gan_train_step(nets, None, event, batch, predict_batch=predict_batch, real=real, noise=noise)


def gan_train_step(nets, bank, event, batch, rng=_pyrandom, rank=0, world=1, predict_batch=32, real=None, n_noise_real=1, noise=None):
    """One iteration of the GAN loop, bbhMahoGANy.py:1243-1299.  Returns [sg_loss, sg_acc, sd_loss, sd_acc] (:1299).
    n_noise_real (:107, default 1): noise realisations per sampled template -- the `batch` sampled templates are stacked n_noise_real times
    (:1280-1283) and every other batch of the iteration (latents, fakes, noise, labels) has batch * n_noise_real rows.
    noise: (rows, n_pix) device tensor for column 1 of the real images (:1277; default: fresh N(0,1) drawn here)."""
    if real is None:
        it = bank.sample(batch, rng, rank, world)
        real = ops.gather_rows(bank.images, it)
    if n_noise_real > 1:
        real = real.repeat(int(n_noise_real), 1)                      # whole copies one after another, as np.concatenate builds them
        batch = batch * int(n_noise_real)
    n = real.shape[1]
    z = _draw_rows(ops.fill_uniform, batch, 100, rank, world, -1.0, 1.0)
    fake = nets.generator.predict_device(z, batch_size=predict_batch)                    # inference phase (:1248)
    if noise is None:
        noise = _draw_rows(ops.fill_normal, batch, n, rank, world, 0.0, 1.0).reshape(batch, n, 1)
    elif tuple(noise.shape[:2]) != (batch, n):
        raise ValueError('gan_train_step: noise has shape %r, the real half has (%d, %d)' % (tuple(noise.shape), batch, n))
    sX, sy = assemble_discriminator_batch(real, noise, fake, event)
    # rows of the global discriminator batch [real (world * batch) | fake REVERSED (world * batch)] this rank holds: its real rows, and -- its own fakes
    # reversed are a contiguous run of the globally reversed fake half -- the mirrored rank's block of it
    row_map = None if world == 1 else ([(rank * batch, batch), (world * batch + (world - 1 - rank) * batch, batch)], 2 * world * batch)
    sd_loss = nets.signal_discriminator.train_on_batch(sX, sy, row_map=row_map)
    z = _draw_rows(ops.fill_uniform, batch, 100, rank, world, -1.0, 1.0)
    sg_loss = nets.signal_discriminator_on_generator.train_on_batch(z, torch.ones(batch, device=device()))
    return [sg_loss[0], sg_loss[1], sd_loss[0], sd_loss[1]]


# --------------------------------------------------------------------------------------------------------------
# the two loop bodies as captured hipGraphs (the reference's own operating point, batch_size = pe_batch_size = 8, n_pix = 1024,
# bbhMahoGANy.py:84-89, is launch-bound: a few hundred launches of microseconds each per iteration)
# --------------------------------------------------------------------------------------------------------------
class _GraphedStep(object):
    """Call 1 runs the step eagerly (binds optimizer state, triggers every lazy initialisation); call 2 captures the step into a
    hipGraph (engine.StepGraph) and replays it; later calls replay (under data parallelism every call runs the eager body: the collectives of
    dist.DataParallel are not captured).  The sequence of results is bit-identical to calling the eager step
    every time: same host index stream, same Philox stream positions, same Adam / BatchNormalization step counts.
    want_losses=False skips the device -> host read of the loss statistics (and with it the per-step synchronisation)."""

    def __init__(self, bank, batch, rng, rank, world):
        self.bank, self.batch, self.rng, self.rank, self.world = bank, int(batch), rng, rank, world
        self.calls = 0
        self.sg = None
        self.eager_only = False            # under data parallelism: the collectives are not captured, every call runs the eager body
        self.it = torch.zeros(self.batch, dtype=torch.int64, device=device())
        self.it_host = torch.zeros(self.batch, dtype=torch.int64).pin_memory()

    def _stage_indices(self):
        idx = sample_indices(self.bank.n, self.batch, self.rng, self.rank, self.world)
        self.it_host.numpy()[:] = idx
        self.it.copy_(self.it_host, non_blocking=True)

    def __call__(self, want_losses=True):
        self.calls += 1
        if self.calls == 1 or self.eager_only:
            return self._eager()
        if self.sg is None:
            sg = StepGraph()
            self._stage_indices()
            torch.cuda.synchronize()
            sg.capture(self._body)
            self.sg = sg                   # only a graph whose capture succeeded is ever replayed
        else:
            self.sg.wait_inputs_consumed()
            self._stage_indices()
        outs = self.sg.replay()
        return self._result(outs) if want_losses else None


class GraphedPEStep(_GraphedStep):
    """pe_train_step (bbhMahoGANy.py:1155-1165) as a replayed hipGraph."""

    def __init__(self, signal_pe, bank, batch, cnn_noise_frac=1.0 / 8.0, rng=_pyrandom, nprng=np.random, rank=0, world=1):
        _GraphedStep.__init__(self, bank, batch, rng, rank, world)
        self.eager_only = signal_pe.data_parallel is not None
        self.model, self.frac, self.nprng = signal_pe, cnn_noise_frac, nprng

    def _eager(self):
        return pe_train_step(self.model, self.bank, self.batch, self.frac, self.rng, self.nprng, self.rank, self.world)

    def _body(self):
        b, n = self.batch, self.bank.n_pix
        x = ops.gather_rows(self.bank.images, self.it)
        y = ops.gather_rows(self.bank.pars, self.it)
        n_noisy = int(b * self.frac)
        sigma = capturing().slot('f', lambda: float(self.nprng.uniform(0, 5)))          # drawn once per replay, like :1161's np.random.uniform(0, 5)
        if n_noisy > 0:
            seed, off = device_rng().take(n_noisy * n)
            ops.axpy(x[:n_noisy], ops.fill_normal((n_noisy, n), 0.0, sigma, seed, off, device()), 1.0)
        return self.model.train_on_batch_device([x.reshape(b, n, 1)], [y[:, 0].contiguous().reshape(b, 1), y[:, 1].contiguous().reshape(b, 1)])

    def _result(self, stats):
        return self.model.train_result(stats, self.batch)


class GraphedGANStep(_GraphedStep):
    """gan_train_step (bbhMahoGANy.py:1243-1299) -- latent draw, generator.predict, noise, batch assembly, discriminator step, latent draw,
    generator step through the frozen discriminator -- as ONE replayed hipGraph."""

    def __init__(self, nets, bank, event, batch, rng=_pyrandom, rank=0, world=1, predict_batch=32):
        _GraphedStep.__init__(self, bank, batch, rng, rank, world)
        self.eager_only = nets.signal_discriminator.data_parallel is not None
        self.nets, self.event, self.predict_batch = nets, event, predict_batch

    def _eager(self):
        return gan_train_step(self.nets, self.bank, self.event, self.batch, self.rng, self.rank, self.world, self.predict_batch)

    def _body(self):
        nets, b, n = self.nets, self.batch, self.bank.n_pix
        real = ops.gather_rows(self.bank.images, self.it)
        seed, off = device_rng().take(b * 100)
        z = ops.fill_uniform((b, 100), -1.0, 1.0, seed, off, device())
        fake = nets.generator.predict_device(z, batch_size=self.predict_batch)
        seed, off = device_rng().take(b * n)
        noise = ops.fill_normal((b, n, 1), 0.0, 1.0, seed, off, device())
        sX, sy = assemble_discriminator_batch(real, noise, fake, self.event)
        sd = nets.signal_discriminator.train_on_batch_device([sX], [sy.reshape(2 * b, 1)])
        seed, off = device_rng().take(b * 100)
        z = ops.fill_uniform((b, 100), -1.0, 1.0, seed, off, device())
        sg = nets.signal_discriminator_on_generator.train_on_batch_device([z], [torch.ones(b, 1, device=device())])
        return sg, sd

    def _result(self, outs):
        sg = self.nets.signal_discriminator_on_generator.train_result(outs[0], self.batch)
        sd = self.nets.signal_discriminator.train_result(outs[1], 2 * self.batch)
        return [sg[0], sg[1], sd[0], sd[1]]


def posterior_samples(nets, n_samples=4000, predict_batch=32):
    """bbhMahoGANy.py:1330-1343: generator.predict(U(-1,1)[n,100]) -> signal_pe.predict -> [mc (n,1), q (n,1)]."""
    seed, off = device_rng().take(n_samples * 100)
    z = ops.fill_uniform((n_samples, 100), -1.0, 1.0, seed, off, device())
    wave = nets.generator.predict_device(z, batch_size=predict_batch)
    pe = nets.signal_pe.predict_device(wave, batch_size=predict_batch)
    return [p.cpu().numpy() for p in pe], wave.cpu().numpy()


def posterior_overlap(pe_samples, lalinf_pars):
    """bbhMahoGANy.py:1345-1356 without the plot: the overlap of the GAN posterior [mc (n,1), q (n,1)] with the lalinference posterior
    (2, m) -- (ks_score, ad_score, beta_score) of overlap_tests (:811-873) -- or None while either read-out is still constant ("if results
    aren't terrible", :1352: both variances non-zero)."""
    from . import posterior
    if np.var(pe_samples[0]) == 0 or np.var(pe_samples[1]) == 0:
        return None
    return posterior.overlap_tests(pe_samples, np.asarray(lalinf_pars, np.float64))


def pe_accuracy(signal_pe, bank, n_eval=4000, rng=_pyrandom, predict_batch=32):
    """The CNN loop's progress read-out, bbhMahoGANy.py:1184-1196 / :1208-1222: predict on `n_eval` randomly chosen training templates
    (random.sample from the host stream) and return (rms, pe_std) -- per parameter the mean squared and the mean absolute difference
    between the training labels [mc, q] and the two read-outs.  (The reference's `rms` line indexes the label ROWS k = 0, 1 instead of the
    parameter columns, :1189; this is the per-parameter quantity its message announces.)"""
    n = min(int(n_eval), bank.n)
    it = torch.tensor(rng.sample(range(bank.n), n), dtype=torch.int64, device=device())
    x = ops.gather_rows(bank.images, it).reshape(n, bank.n_pix, 1)
    y = ops.gather_rows(bank.pars, it).cpu().numpy().astype(np.float64)
    pe = [p.cpu().numpy().astype(np.float64).reshape(-1) for p in signal_pe.predict_device(x, batch_size=predict_batch)]
    rms = [float(np.mean((y[:, k] - pe[k]) ** 2)) for k in range(2)]
    pe_std = [float(np.mean(np.abs(y[:, k] - pe[k]))) for k in range(2)]
    return rms, pe_std

