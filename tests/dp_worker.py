"""Worker for the data-parallel equivalence tests (launched by torch.distributed.run with WORLD_SIZE ranks, or directly with
WORLD_SIZE=1 as the single-process reference).

mode 'cpu'  : gloo on CPU tensors; the arithmetic is the fp64 oracle, the communication is gennet_amd.dist -- checks the
              data-parallel MATH of SURVEY 8e (loss normalised by the global batch + SUM all-reduce of gradients; SyncBN
              statistics and their backward sums; rank-sliced host sampling) without a GPU.
mode 'gpu'  : gloo on CUDA tensors, every rank on cuda:0 (one-GPU box; RCCL refuses two ranks on one device): the real HIP
              path, N ranks x B/N rows versus 1 rank x B rows.
mode 'rccl1': the same HIP path with backend "nccl" (= RCCL) and a ONE-rank group: every gradient / SyncBN / scalar all-reduce
              and the weight broadcast really go through RCCL (communicator set-up, stream ordering against the HIP kernels,
              fp32 and fp64 buffers); must reproduce the plain single-process run bit for bit.
Writes a pickle with losses and final weights to argv[2].<rank>.
"""
import os
import pickle
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


def run_cpu(out):
    from gennet_amd import bbh, dist
    from oracle import keras_ref as K
    from oracle import nets_ref as N
    dp = dist.init('gloo')
    rank, world = (dp.rank, dp.world_size) if dp else (0, 1)
    rng = np.random.RandomState(0)
    B, n_pix = 8, 64
    lo, hi = rank * B // world, (rank + 1) * B // world
    res = {}
    # (1) gradients of the two-branch CNN: local rows, loss normalised by the GLOBAL batch, SUM all-reduce
    pe = N.PENet(n_pix, rng)
    pe.mc.params[-1][...] = 25.0; pe.q.params[-1][...] = 0.6
    x = rng.randn(B, n_pix, 1); ymc = rng.uniform(20, 35, B); yq = rng.uniform(0.5, 1, B)
    pm = pe.mc.forward(x[lo:hi], True); pq = pe.q.forward(x[lo:hi], True)
    _, dm = K.mse_loss(pm, ymc[lo:hi].reshape(-1, 1)); _, dq = K.mse_loss(pq, yq[lo:hi].reshape(-1, 1))
    scale = (hi - lo) / float(B)
    _, gm = pe.mc.backward(dm * scale); _, gq = pe.q.backward(dq * scale)
    flat = torch.from_numpy(np.concatenate([g.ravel() for g in gm + gq]))
    if dp:
        dp.all_reduce_sum(flat)
    res['pe_grads'] = flat.numpy().copy()
    # (2) SyncBN: forward statistics and backward sums through the all-reduce reproduce full-batch BatchNorm
    xb = rng.randn(B, 12, 5); gamma = rng.rand(5) + 0.5; beta = rng.randn(5); dy = rng.randn(B, 12, 5)
    xl = xb[lo:hi].reshape(-1, 5); dyl = dy[lo:hi].reshape(-1, 5)
    sums = torch.from_numpy(np.concatenate([xl.sum(0), (xl * xl).sum(0)]))
    if dp:
        dp.all_reduce_sum(sums)
    n = B * 12
    mean = sums.numpy()[:5] / n
    var = sums.numpy()[5:] / n - mean * mean
    inv = 1.0 / np.sqrt(var + K.BN_EPS)
    xhat = (xl - mean) * inv
    dsum = torch.from_numpy(np.concatenate([dyl.sum(0), (dyl * xhat).sum(0)]))
    local = dsum.clone()
    if dp:
        dp.all_reduce_sum(dsum)
    ds = dsum.numpy()
    dx = gamma * inv * (dyl - ds[:5] / n - xhat * ds[5:] / n)
    res['bn'] = {'rows': (lo, hi), 'y': gamma * xhat + beta, 'dx': dx, 'mean': mean, 'var': var, 'dgamma_local': local.numpy()[5:], 'dsum': ds}
    # (3) rank-sliced host sampling: BASELINE configs[3]'s global batch of 4096 rows out of a 100 000-template bank, 4096 / world rows per rank
    random.seed(1)
    res['idx'] = [bbh.sample_indices(100000, 4096 // world, random, rank, world) for _ in range(3)]
    res['next'] = random.random()
    # (4) device random draws tile ONE Philox stream (engine.PhiloxStream.take_rows; host-only bookkeeping, so it runs here): the counter ranges this
    # rank takes for the draws of one CNN step + one GAN iteration at BASELINE configs[3]'s global batch 4096 (CNN noise rows, latents, noise column,
    # the discriminator's and the generator's dropout masks) and where the stream stands afterwards
    from gennet_amd import engine
    st = engine.PhiloxStream(1000)
    Bg, n_pix = 4096, 2048
    b = Bg // world
    rngs = []

    def take(row_len, blocks, grows):
        seed, offs = st.take_rows(row_len, blocks, grows)
        rngs.append([(o, o + (nr * row_len + 3) // 4) for (g0, nr), o in zip(blocks, offs)])
    n_noisy_g = Bg // 8
    n_noisy = max(0, min(b, n_noisy_g - rank * b))
    take(n_pix, [(rank * b, n_noisy)] if n_noisy else [], n_noisy_g)                         # CNN noise rows
    take(100, [(rank * b, b)], Bg)                                                            # latents
    take(n_pix, [(rank * b, b)], Bg)                                                          # noise column
    take(n_pix // 2 * 2 * 256, [(rank * b, b), (Bg + (world - 1 - rank) * b, b)], 2 * Bg)     # D conv1 dropout on [real | fake reversed]
    take(100, [(rank * b, b)], Bg)                                                            # second latent draw
    take(256 * (n_pix // 2), [(rank * b, b)], Bg)                                             # G's first dropout
    res['philox'] = {'ranges': rngs, 'end': st.offset}
    pickle.dump(res, open('%s.%d' % (out, rank), 'wb'))
    if dp:
        torch.distributed.barrier()


def run_gpu(out, backend='gloo'):
    from gennet_amd import bbh, dist, engine
    from gennet_amd.layers import Dropout
    dp = dist.init(backend, allow_single=(backend == 'nccl'))
    rank, world = (dp.rank, dp.world_size) if dp else (0, 1)
    engine.set_init_seed(3 + int(os.environ.get('GN_TEST_SEED', '0')))
    n_pix, B = 64, 8
    lo, hi = rank * B // world, (rank + 1) * B // world
    rng = np.random.RandomState(11 + int(os.environ.get('GN_TEST_SEED', '0')))
    event = f32(rng.randn(n_pix, 1))
    nets = bbh.build_and_compile(event, n_pix, data_parallel=dp)
    G, D, DG, PE = nets.generator, nets.signal_discriminator, nets.signal_discriminator_on_generator, nets.signal_pe
    res = {'losses': []}
    for it in range(2):
        x = f32(rng.randn(B, n_pix, 1)); ymc = f32(rng.uniform(20, 35, B)); yq = f32(rng.uniform(0.5, 1, B))
        res['losses'].append(PE.train_on_batch(x[lo:hi], [ymc[lo:hi], yq[lo:hi]]))
        # discriminator step on 2B rows [real | fake]: rank r takes its slice of EACH half
        sX = f32(rng.randn(2 * B, n_pix, 2, 1)); sy = np.array([1.0] * B + [0.0] * B)
        rows = np.r_[lo:hi, B + lo:B + hi]
        dmask = {}
        h = (2 * B, n_pix // 2, 2, 256)
        for l, shp in zip([l for l in D.layers if isinstance(l, Dropout)], (h, (2 * B, n_pix // 4, 2, 512))):
            dmask[l.name] = (rng.rand(*shp) >= 0.4).astype(np.uint8)
        res['losses'].append(D.train_on_batch(sX[rows], sy[rows], dropout_masks={k: v[rows] for k, v in dmask.items()}))
        # generator step through the frozen discriminator (SyncBN inside G)
        z = f32(rng.uniform(-1, 1, (B, 100)))
        masks = {}
        shapes = [(B, 256 * (n_pix // 2)), (B, n_pix // 2, 64), (B, n_pix, 128), (B, n_pix, 256), (B, n_pix, 512), (B, n_pix, 1024)]
        for l, shp in zip([l for l in G.layers if isinstance(l, Dropout)], shapes):
            masks[l.name] = (rng.rand(*shp) >= 0.2).astype(np.uint8)
        for l, shp in zip([l for l in D.layers if isinstance(l, Dropout)], ((B, n_pix // 2, 2, 256), (B, n_pix // 4, 2, 512))):
            masks[l.name] = (rng.rand(*shp) >= 0.4).astype(np.uint8)
        res['losses'].append(DG.train_on_batch(z[lo:hi], np.ones(hi - lo), dropout_masks={k: v[lo:hi] for k, v in masks.items()}))
    # the hipGraph loop body under data parallelism runs its eager body (the collectives are not captured) and still tiles the global batch
    bank = bbh.DeviceBank(f32(rng.randn(32, n_pix)), np.stack([rng.uniform(20, 35, 32), rng.uniform(0.5, 1, 32)], 1))
    random.seed(5)
    step = bbh.GraphedPEStep(PE, bank, B // world, cnn_noise_frac=0.0, rank=rank, world=world)
    for _ in range(3):
        res['losses'].append(step())
    assert step.eager_only == (dp is not None) and (step.sg is None) == (dp is not None)
    res['weights'] = {'G': G.get_weights(), 'D': D.get_weights(), 'PE': PE.get_weights()}
    if dp:
        for m in (G, D, PE):
            dp.sync_model(m)               # broadcast from rank 0 (a no-op in value; exercises the collective)
        res['backend'] = torch.distributed.get_backend()
    pickle.dump(res, open('%s.%d' % (out, rank), 'wb'))
    if dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def run_gpu_public(out):
    """The PUBLIC loop bodies (bbh.pe_train_step, bbh.gan_train_step: nothing injected -- host index stream, device latents, noise and dropout masks all
    drawn inside) with N ranks x B / N rows against one rank x B rows: the draws of the ranks tile the single-process draws."""
    from gennet_amd import bbh, dist, engine
    dp = dist.init('gloo')
    rank, world = (dp.rank, dp.world_size) if dp else (0, 1)
    engine.set_init_seed(3)
    engine.set_device_seed(77)                       # the same device stream on every rank
    random.seed(5); np.random.seed(6)                # the same host streams on every rank
    n_pix, B = 64, 8
    rng = np.random.RandomState(11)
    event = f32(rng.randn(n_pix, 1))
    nets = bbh.build_and_compile(event, n_pix, data_parallel=dp)
    bank = bbh.DeviceBank(f32(rng.randn(40, n_pix)), np.stack([rng.uniform(20, 35, 40), rng.uniform(0.5, 1, 40)], 1))
    ev = engine.to_device(event.reshape(-1))
    res = {'losses': []}
    for it in range(2):
        res['losses'].append(bbh.pe_train_step(nets.signal_pe, bank, B // world, cnn_noise_frac=0.5, rank=rank, world=world))
        res['losses'].append(bbh.gan_train_step(nets, bank, ev, B // world, rank=rank, world=world, predict_batch=4))
    res['rng_end'] = engine.device_rng().offset
    res['weights'] = {'G': nets.generator.get_weights(), 'D': nets.signal_discriminator.get_weights(), 'PE': nets.signal_pe.get_weights()}
    pickle.dump(res, open('%s.%d' % (out, rank), 'wb'))
    if dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    {'cpu': run_cpu, 'gpu': run_gpu, 'gpu_public': run_gpu_public, 'rccl1': lambda out: run_gpu(out, 'nccl')}[sys.argv[1]](sys.argv[2])
