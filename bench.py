#!/usr/bin/env python
"""waveforms/sec of the BBH training hot path (CNN point-estimator step + GAN iteration, 2048-sample segments) on MI355X.

One "step" pushes WAVES = 512 synthetic waveforms per GPU through BOTH training loops of bbhMahoGANy.py:
  * CNN point-estimator: two train_on_batch calls at batch 256  (BASELINE.json configs[1]; bbhMahoGANy.py:1153-1168)
  * GAN: one full iteration at batch 512 (G.predict -> D step on 2B -> G step through frozen D;
    BASELINE.json configs[2]; bbhMahoGANy.py:1241-1299)
so value = N_gpus * 512 * steps / wall_time = B / (t_CNN + t_GAN) of SURVEY section 8d.  Inputs (template bank, labels, event)
are resident in HBM before the timed region; every optimizer step, BatchNorm update and dropout draw is inside it.

python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS a
rank; started plainly it launches `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a CHILD process
before anything here touches the GPU, relays the child's output and exits with its code.  A rank whose world size differs from
--gpus fails.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PIX = 2048
CNN_BATCH = 256
GAN_BATCH = 512
WAVES = 512
# algorithmic work per waveform at n_pix = 2048: CNN 3*P_f MACs, GAN 4*G_f + 8*D_f(effective) MACs
GFLOP_PER_WAVE_CNN = 15.36
GFLOP_PER_WAVE_GAN = 79.6
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = 1/16 of the bf16 peak


def pmc_traffic_per_launch():
    """HBM-side bytes per conv_mfma_kernel launch from the committed PMC passes of this same command
    (profiles/r01_pmc_traffic.json, written by scripts/pmc_traffic.py from `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
    runs; counters cannot be read from inside the timed run).  None when the file is absent."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))['kernels']
    sel = [v for k, v in ks.items() if 'conv_mfma_kernel' in k or 'conv_mfma_dma_kernel' in k]
    n = sum(v['launches'] for v in sel)
    b = sum(v['launches'] * v['hbm_bytes_per_launch'] for v in sel)
    return b / n if n else None


def host_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota) -- the GPU box shows 256 logical CPUs
    but grants a 16-CPU quota per GPU; oversubscribing oneDNN with 256 threads there is ~10x slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (IOError, OSError, ValueError):
        pass
    return n


def cpu_baseline(seconds_budget=25.0):
    """The torch-CPU port of the same two steps (oracle/torch_ref.py) on this box's host cores, bounded sample:
    CNN train steps at batch 32 and GAN iterations at batch 8 on 2048-sample segments (1 warm-up + timed repeats)."""
    from oracle import torch_ref as T
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    bc, bg = 32, 8
    pe = T.PENet(N_PIX)
    x = torch.randn(bc, N_PIX, 1); ymc = torch.rand(bc) * 15 + 20; yq = torch.rand(bc) * 0.5 + 0.5
    pe.train_on_batch(x, ymc, yq)
    t0 = time.time(); n = 0
    while n < 2 or (time.time() - t0 < seconds_budget * 0.4 and n < 20):
        pe.train_on_batch(x, ymc, yq); n += 1
    t_cnn = (time.time() - t0) / n / bc
    gan = T.GAN(N_PIX, np.random.RandomState(0).randn(N_PIX))
    real = torch.randn(bg, N_PIX)
    gan.iteration(real, bg)
    t0 = time.time(); m = 0
    while m < 2 or (time.time() - t0 < seconds_budget * 0.6 and m < 20):
        gan.iteration(real, bg); m += 1
    t_gan = (time.time() - t0) / m / bg
    return {'value': 1.0 / (t_cnn + t_gan), 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port',
            'sample': 'torch-CPU fp32 port (oracle/torch_ref.py): %d CNN train steps at batch %d + %d GAN iterations at batch %d, n_pix=%d' % (n, bc, m, bg, N_PIX),
            'cnn_waveforms_per_s': 1.0 / t_cnn, 'gan_waveforms_per_s': 1.0 / t_gan}


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """python bench.py --gpus N started plainly: run the N ranks as a child torch.distributed.run job (this parent has not imported
    torch, let alone touched the GPU), pass its stdout/stderr through and return its exit code."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--bank', type=int, default=100000, help='synthetic template bank size (BASELINE configs[1]: 100k segments)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    global np, torch
    import numpy as np
    import torch
    from gennet_amd import bbh, dist, engine, ops
    # RCCL (backend "nccl") over xGMI; GENNET_DIST_BACKEND=gloo only exists to rehearse the N>1 code path on a one-GPU box
    backend = os.environ.get('GENNET_DIST_BACKEND', 'nccl')
    dp = dist.init(backend) if args.gpus > 1 else None
    rank = dp.rank if dp else 0
    world = dp.world_size if dp else 1
    if world != args.gpus or int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the job has WORLD_SIZE=%s (world size %d)\n' % (args.gpus, os.environ.get('WORLD_SIZE'), world))
        sys.exit(2)
    dev = engine.device()
    engine.set_init_seed(1)                     # identical initial weights on every rank
    engine.set_device_seed(1000 + rank)         # per-rank dropout / latent / noise streams
    random.seed(1); np.random.seed(1)           # host index stream identical on all ranks; rank r keeps its slice

    # synthetic template bank in HBM: noise-free whitened chirps are replaced by unit-variance Gaussian rows of the same shape
    # and labels drawn from the hunt_constrain prior box (mc in [20,35], q in [0.5,1]); arithmetic work is identical.
    bank_n = args.bank
    images = ops.fill_normal((bank_n, N_PIX), 0.0, 1.0, 77, 0, dev)
    pars = torch.stack([ops.fill_uniform((bank_n,), 20.0, 35.0, 78, 0, dev), ops.fill_uniform((bank_n,), 0.5, 1.0, 79, 0, dev)], dim=1).contiguous()
    bank = bbh.DeviceBank(images, pars)
    event_host = np.random.RandomState(5).randn(N_PIX, 1).astype(np.float32)
    nets = bbh.build_and_compile(event_host, N_PIX, data_parallel=dp)
    event = engine.to_device(event_host.reshape(-1))
    if dp:
        for m in (nets.generator, nets.signal_discriminator, nets.signal_pe):
            dp.sync_model(m)

    def step():
        for _ in range(WAVES // CNN_BATCH):
            bbh.pe_train_step(nets.signal_pe, bank, CNN_BATCH, rank=rank, world=world)
        bbh.gan_train_step(nets, bank, event, GAN_BATCH, rank=rank, world=world, predict_batch=GAN_BATCH)

    def barrier():
        torch.cuda.synchronize()
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.prof_enable(True); ops.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ops.prof_enable(False)
    if dp:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    conv = ops.prof_collect(0); wgrad = ops.prof_collect(1)

    # SURVEY 8d also asks for the two loops separately: timed AFTER the K steps (not part of `value`)
    def timed(fn, reps):
        barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return (time.perf_counter() - t) / reps
    t_cnn = timed(lambda: bbh.pe_train_step(nets.signal_pe, bank, CNN_BATCH, rank=rank, world=world), 4)
    t_gan = timed(lambda: bbh.gan_train_step(nets, bank, event, GAN_BATCH, rank=rank, world=world, predict_batch=GAN_BATCH), 2)
    # ... and for the synthesiser (metric iii; BASELINE configs[4] fuses it into the loop): templates/s of OnlineBank.draw -- chirp ->
    # whiten -> irFFT x2 -> align/crop + white noise, everything in HBM -- with a smooth analytic PSD (no LAL here)
    from gennet_amd import templates as T
    f = np.arange(N_PIX * 2 + 1) * 0.25
    psd = 1e-46 * ((np.maximum(f, 10.0) / 150.0) ** -4.0 + 2.0 + 2.0 * (f / 150.0) ** 2.0)
    psd[f < 10.0] = 0.0
    synth = T.OnlineBank(N_PIX, 4, psd, seed=1000 + rank, noise='white')
    synth.draw(1024)
    t_syn = timed(lambda: synth.draw(4096), 3)

    if rank == 0:
        value = world * WAVES * args.steps / dt
        ach = conv['flop'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0
        out = {
            'metric': 'waveforms/sec (CNN+GAN step, 2048-sample BBH)', 'value': value, 'unit': 'waveforms/s',
            'n_gpus': world, 'ranks': world, 'collective_backend': ('rccl' if backend == 'nccl' else backend) if dp else None, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'per GPU and step: 2 x CNN point-estimator train_on_batch(batch=256) + 1 GAN iteration(batch=512) '
                                   '(G.predict, D step on 2B, G step through frozen D); n_pix=2048; %d-row synthetic template bank in HBM' % bank_n,
                       'n_pix': N_PIX, 'cnn_batch': CNN_BATCH, 'gan_batch': GAN_BATCH, 'waveforms_per_step_per_gpu': WAVES,
                       'parallelism': 'dp%d' % world},
            'roofline': {'bound': 'mfma', 'kernel': 'conv_mfma_dma_kernel + conv_mfma_kernel (implicit-GEMM Conv1D forward + data gradient, v_mfma_f32_32x32x2_f32; the DMA variant runs every tile without ragged channel edges)',
                         'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': pmc_traffic_per_launch(),
                         'traffic_note': 'fabric-side bytes per conv_mfma_kernel launch (2*FETCH_SIZE + WRITE_SIZE KiB, gfx950 FETCH correction) '
                                         'from the separate PMC passes of this command summarised in profiles/r01_pmc_traffic.json',
                         'launches': conv['launches'], 'avg_launch_ms': conv['ms'] / max(conv['launches'], 1),
                         'algorithmic_flop_per_launch': conv['flop'] / max(conv['launches'], 1),
                         'algorithmic_bytes_per_launch': conv['bytes'] / max(conv['launches'], 1),
                         'wgrad_mfma_kernel': {'achieved': wgrad['flop'] / (wgrad['ms'] * 1e-3) / 1e12 if wgrad['ms'] > 0 else 0.0,
                                               'launches': wgrad['launches'], 'avg_launch_ms': wgrad['ms'] / max(wgrad['launches'], 1)},
                         'mfma_kernel_time_share': (conv['ms'] + wgrad['ms']) * 1e-3 / dt,
                         'step_algorithmic_tflops': world * WAVES * args.steps * (GFLOP_PER_WAVE_CNN + GFLOP_PER_WAVE_GAN) * 1e-3 / dt},
            'breakdown': {'cnn_train_waveforms_per_s': world * CNN_BATCH / t_cnn, 'gan_iteration_waveforms_per_s': world * GAN_BATCH / t_gan,
                          'cnn_ms_per_batch': 1e3 * t_cnn, 'gan_ms_per_iteration': 1e3 * t_gan,
                          'synth_templates_per_s': world * 4096 / t_syn,
                          'note': 'rank-0 clock, measured after the timed steps; value = B / (t_CNN + t_GAN) comes from the K timed steps only'},
        }
        conv_math = os.environ.get('GENNET_CONV_MATH', 'fp32')
        if conv_math != 'fp32':      # the opt-in experiment: say so in the line; never the default configuration
            x3 = ops.prof_collect(2)
            out['dtype'] = 'f32 operands split into 3 bf16 pieces on the large unit-stride conv launches (opt-in GENNET_CONV_MATH=%s), f32 elsewhere' % conv_math
            out['config']['conv_math'] = conv_math
            out['roofline']['bf16x3_launches'] = {'launches': x3['launches'], 'avg_launch_ms': x3['ms'] / max(x3['launches'], 1),
                                                  'fp32_equivalent_tflops': x3['flop'] / (x3['ms'] * 1e-3) / 1e12 if x3['ms'] > 0 else 0.0}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
