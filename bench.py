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

# per workload: n_pix, batch sizes, algorithmic GFLOP per waveform (CNN 3*P_f MACs, GAN 4*G_f + 8*D_f(effective) MACs)
WORKLOADS = {
    'default': {'n_pix': 2048, 'cnn_batch': 256, 'gan_batch': 512, 'waves': 512, 'gflop_cnn': 15.36, 'gflop_gan': 79.6, 'online': False},
    # BASELINE configs[4]: srate 4096, every batch synthesised on the GPU inside the loop (no stored bank)
    'cfg5': {'n_pix': 4096, 'cnn_batch': 256, 'gan_batch': 512, 'waves': 512, 'gflop_cnn': 30.85, 'gflop_gan': 159.1, 'online': True},
}
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = 1/16 of the bf16 peak
PEAK_HBM_TBS = 8.0                 # MI355X_MICROARCH.md: HBM3E spec peak


def pmc_traffic_per_launch():
    """Fabric-side bytes per conv_mfma launch from the committed PMC passes of this same command (profiles/r0N_pmc_traffic.json,
    written by scripts/pmc_traffic.py from `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs; counters cannot be read from
    inside the timed run).  Newest round first; (None, None) when no file is there."""
    for name in ('r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            ks = json.load(open(path))['kernels']
            sel = [v for k, v in ks.items() if 'conv_mfma' in k]          # conv_mfma_pipe_kernel, conv_mfma_dma_kernel, conv_mfma_kernel
            n = sum(v['launches'] for v in sel)
            b = sum(v['launches'] * v['hbm_bytes_per_launch'] for v in sel)
            return (b / n if n else None), 'profiles/' + name
    return None, None


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


def cpu_baseline(n_pix, seconds_budget=30.0):
    """The torch-CPU port of the same two steps (oracle/torch_ref.py) on this box's host cores, bounded sample: CNN train steps at
    batch 32 (repeated) plus ONE at the benchmark's batch 256, GAN iterations at batch 8 (a batch-512 iteration is ~40 TFLOP: minutes
    on the host).  `value` combines the batch-256 CNN rate with the batch-8 GAN rate; every batch size is a field of the object."""
    from oracle import torch_ref as T
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    bc, bg, bc_big = 32, 8, 256
    pe = T.PENet(n_pix)

    def cnn_batch(b):
        return torch.randn(b, n_pix, 1), torch.rand(b) * 15 + 20, torch.rand(b) * 0.5 + 0.5
    x, ymc, yq = cnn_batch(bc)
    pe.train_on_batch(x, ymc, yq)
    t0 = time.time(); n = 0
    while n < 2 or (time.time() - t0 < seconds_budget * 0.25 and n < 20):
        pe.train_on_batch(x, ymc, yq); n += 1
    t_cnn_small = (time.time() - t0) / n / bc
    xb, ymcb, yqb = cnn_batch(bc_big)
    t0 = time.time()
    pe.train_on_batch(xb, ymcb, yqb)
    t_cnn = (time.time() - t0) / bc_big
    del xb
    gan = T.GAN(n_pix, np.random.RandomState(0).randn(n_pix))
    real = torch.randn(bg, n_pix)
    gan.iteration(real, bg)
    t0 = time.time(); m = 0
    while m < 2 or (time.time() - t0 < seconds_budget * 0.4 and m < 20):
        gan.iteration(real, bg); m += 1
    t_gan = (time.time() - t0) / m / bg
    return {'value': 1.0 / (t_cnn + t_gan), 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port',
            'sample': 'torch-CPU fp32 port (oracle/torch_ref.py), n_pix=%d: 1 CNN train step at batch %d (+ %d at batch %d) and %d GAN iterations at batch %d'
                      % (n_pix, bc_big, n, bc, m, bg),
            'cnn_batch': bc_big, 'cnn_batch_small': bc, 'gan_batch': bg,
            'cnn_waveforms_per_s': 1.0 / t_cnn, 'cnn_waveforms_per_s_small_batch': 1.0 / t_cnn_small, 'gan_waveforms_per_s': 1.0 / t_gan,
            'note': 'the GPU line runs CNN batch 256 and GAN batch 512; the CPU GAN leg is timed at batch %d (baseline only)' % bg}


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
    ap.add_argument('--config', choices=sorted(WORKLOADS), default='default',
                    help="default: BASELINE configs[1]+[2] (n_pix 2048, stored bank in HBM); cfg5: configs[4] (srate 4096, templates synthesised in the loop)")
    ap.add_argument('--predict-batch', type=int, default=0, help='chunk size of generator.predict for the fake half (0: the GAN batch)')
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
    from gennet_amd import templates as T
    # RCCL (backend "nccl") over xGMI; GENNET_DIST_BACKEND=gloo only exists to rehearse the N>1 code path on a one-GPU box
    backend = os.environ.get('GENNET_DIST_BACKEND', 'nccl')
    dp = dist.init(backend) if args.gpus > 1 else None
    rank = dp.rank if dp else 0
    world = dp.world_size if dp else 1
    if world != args.gpus or int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the job has WORLD_SIZE=%s (world size %d)\n' % (args.gpus, os.environ.get('WORLD_SIZE'), world))
        sys.exit(2)
    wl = WORKLOADS[args.config]
    N_PIX, CNN_BATCH, GAN_BATCH, WAVES = wl['n_pix'], wl['cnn_batch'], wl['gan_batch'], wl['waves']
    predict_batch = args.predict_batch or GAN_BATCH
    dev = engine.device()
    engine.set_init_seed(1)                     # identical initial weights on every rank
    engine.set_device_seed(1000 + rank)         # per-rank dropout / latent / noise streams
    random.seed(1); np.random.seed(1)           # host index stream identical on all ranks; rank r keeps its slice

    # synthetic 2048-/4096-sample BBH segments: this project's FD chirp through an analytic aLIGO-like PSD (no LAL, no lalinference
    # PSD file here), whitened, aligned and cropped by the fused synthesiser; scaled to unit variance (the role of gw_norm_constant,
    # gw_template_maker.py:782).  Rank-offset stream: no exchange between ranks.
    f = np.arange(N_PIX * 2 + 1) * 0.25
    psd = 1e-46 * ((np.maximum(f, 10.0) / 150.0) ** -4.0 + 2.0 + 2.0 * (f / 150.0) ** 2.0)
    psd[f < 10.0] = 0.0
    synth = T.OnlineBank(N_PIX, 4, psd, seed=1000 + rank, noise=None)
    probe, _ = synth.draw(4096)
    synth.g = 1.0 / float(probe.std())
    del probe
    bank_n = 0 if wl['online'] else args.bank
    if wl['online']:
        bank = None
        online = T.OnlineBank(N_PIX, 4, psd, gw_norm_constant=synth.g, seed=2000 + rank, noise='white')
    else:
        images, pars = synth.draw(bank_n)
        bank = bbh.DeviceBank(images, pars)
        del images, pars
    event_host = np.random.RandomState(5).randn(N_PIX, 1).astype(np.float32)
    nets = bbh.build_and_compile(event_host, N_PIX, data_parallel=dp)
    event = engine.to_device(event_host.reshape(-1))
    if dp:
        for m_ in (nets.generator, nets.signal_discriminator, nets.signal_pe):
            dp.sync_model(m_)

    if wl['online']:
        def cnn_step():
            return bbh.pe_train_step_online(nets.signal_pe, online, CNN_BATCH)

        def gan_step():
            return bbh.gan_train_step_online(nets, online, event, GAN_BATCH, predict_batch=predict_batch)
    else:
        def cnn_step():
            return bbh.pe_train_step(nets.signal_pe, bank, CNN_BATCH, rank=rank, world=world)

        def gan_step():
            return bbh.gan_train_step(nets, bank, event, GAN_BATCH, rank=rank, world=world, predict_batch=predict_batch)

    def step():
        for _ in range(WAVES // CNN_BATCH):
            cnn_step()
        gan_step()

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
    conv = ops.prof_collect(0); wgrad = ops.prof_collect(1); syn_in_step = ops.prof_collect(3)

    # SURVEY 8d also asks for the two loops and the synthesiser separately: timed AFTER the K steps (not part of `value`)
    def timed(fn, reps):
        barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return (time.perf_counter() - t) / reps
    t_cnn = timed(cnn_step, 4)
    t_gan = timed(gan_step, 2)
    # metric iii: templates/s of the fused synthesiser (chirp -> whiten -> both inverse FFTs -> arg-max -> slide -> crop, one kernel),
    # kernel time from HIP events on the launch stream; algorithmic bytes per template = 2*Nf*16 + Nf*8 read, n_pix*4 written
    SYN_NB = 16384
    synth.draw(1024)
    ops.prof_enable(True); ops.prof_reset()
    t_syn = timed(lambda: synth.draw(SYN_NB), 3)
    ops.prof_enable(False)
    syn = ops.prof_collect(3)

    if rank == 0:
        value = world * WAVES * args.steps / dt
        ach = conv['flop'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0
        traffic, traffic_src = pmc_traffic_per_launch()
        syn_tbs = syn['bytes'] / (syn['ms'] * 1e-3) / 1e12 if syn['ms'] > 0 else 0.0
        out = {
            'metric': 'waveforms/sec (CNN+GAN step, %d-sample BBH)' % N_PIX, 'value': value, 'unit': 'waveforms/s',
            'n_gpus': world, 'ranks': world, 'collective_backend': ('rccl' if backend == 'nccl' else backend) if dp else None, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('BASELINE %s; per GPU and step: 2 x CNN point-estimator train_on_batch(batch=%d) + 1 GAN iteration(batch=%d) '
                                    '(G.predict, D step on 2B, G step through frozen D); n_pix=%d; %s')
                                   % ('configs[4] (cfg5)' if wl['online'] else 'configs[1]+[2]', CNN_BATCH, GAN_BATCH, N_PIX,
                                      'every batch synthesised on the GPU inside the step (fused chirp->irFFT->align->crop kernel + white noise)' if wl['online']
                                      else '%d whitened BBH templates synthesised on the GPU into an HBM-resident bank before the timed region' % bank_n),
                       'name': args.config, 'n_pix': N_PIX, 'cnn_batch': CNN_BATCH, 'gan_batch': GAN_BATCH, 'predict_batch': predict_batch,
                       'waveforms_per_step_per_gpu': WAVES, 'bank_rows': bank_n, 'parallelism': 'dp%d' % world},
            'roofline': {'bound': 'mfma', 'kernel': 'conv_mfma_pipe_kernel (+ conv_mfma_dma_kernel / conv_mfma_kernel for ragged or 1-tap shapes): implicit-GEMM Conv1D forward + data gradient, v_mfma_f32_32x32x2_f32',
                         'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                         'traffic_note': 'fabric-side bytes per conv_mfma launch (2*FETCH_SIZE + WRITE_SIZE KiB, gfx950 FETCH correction) from the separate PMC '
                                         'passes of the default-config command summarised in %s' % traffic_src,
                         'launches': conv['launches'], 'avg_launch_ms': conv['ms'] / max(conv['launches'], 1),
                         'algorithmic_flop_per_launch': conv['flop'] / max(conv['launches'], 1),
                         'algorithmic_bytes_per_launch': conv['bytes'] / max(conv['launches'], 1),
                         'wgrad_mfma_kernel': {'achieved': wgrad['flop'] / (wgrad['ms'] * 1e-3) / 1e12 if wgrad['ms'] > 0 else 0.0,
                                               'frac': (wgrad['flop'] / (wgrad['ms'] * 1e-3) / 1e12 if wgrad['ms'] > 0 else 0.0) / PEAK_F32_MFMA_TFLOPS,
                                               'launches': wgrad['launches'], 'avg_launch_ms': wgrad['ms'] / max(wgrad['launches'], 1),
                                               'algorithmic_flop_per_launch': wgrad['flop'] / max(wgrad['launches'], 1),
                                               'algorithmic_bytes_per_launch': wgrad['bytes'] / max(wgrad['launches'], 1)},
                         'mfma_kernel_time_share': (conv['ms'] + wgrad['ms']) * 1e-3 / dt,
                         'step_algorithmic_tflops': world * WAVES * args.steps * (wl['gflop_cnn'] + wl['gflop_gan']) * 1e-3 / dt},
            'roofline_synth': {'bound': 'hbm', 'kernel': 'synth_fused_kernel (gn_synth_templates)', 'achieved': syn_tbs, 'peak': PEAK_HBM_TBS, 'unit': 'TB/s',
                               'frac': syn_tbs / PEAK_HBM_TBS, 'traffic': None, 'templates_per_launch': SYN_NB,
                               'launches': syn['launches'], 'avg_launch_ms': syn['ms'] / max(syn['launches'], 1),
                               'algorithmic_bytes_per_template': syn['bytes'] / max(syn['launches'], 1) / SYN_NB,
                               'kernel_templates_per_s': syn['launches'] * SYN_NB / (syn['ms'] * 1e-3) if syn['ms'] > 0 else 0.0,
                               'launches_inside_timed_steps': syn_in_step['launches'], 'ms_inside_timed_steps': syn_in_step['ms'],
                               'note': 'algorithmic bytes = SURVEY 8d per-template figure (2 spectra of Nf complex128 + PSD read, n_pix fp32 written); the fused '
                                       'kernel keeps the spectra in registers/LDS, so its real HBM traffic is ~the output row: the kernel is fp64-VALU/LDS-bound, '
                                       'not HBM-bound'},
            'breakdown': {'cnn_train_waveforms_per_s': world * CNN_BATCH / t_cnn, 'gan_iteration_waveforms_per_s': world * GAN_BATCH / t_gan,
                          'cnn_ms_per_batch': 1e3 * t_cnn, 'gan_ms_per_iteration': 1e3 * t_gan,
                          'synth_templates_per_s': world * SYN_NB / t_syn,
                          'note': 'rank-0 clock, measured after the timed steps; value = B / (t_CNN + t_GAN) comes from the K timed steps only; '
                                  'synth_templates_per_s = OnlineBank.draw wall clock (ONE kernel: prior draw, chirp, both inverse FFTs, arg-max, slide, crop; no host random numbers)'},
        }
        conv_math = os.environ.get('GENNET_CONV_MATH', 'fp32')
        if conv_math != 'fp32':      # the opt-in experiment: say so in the line; never the default configuration
            x3 = ops.prof_collect(2)
            out['dtype'] = 'f32 operands split into 3 bf16 pieces on the large unit-stride conv launches (opt-in GENNET_CONV_MATH=%s), f32 elsewhere' % conv_math
            out['config']['conv_math'] = conv_math
            out['roofline']['bf16x3_launches'] = {'launches': x3['launches'], 'avg_launch_ms': x3['ms'] / max(x3['launches'], 1),
                                                  'fp32_equivalent_tflops': x3['flop'] / (x3['ms'] * 1e-3) / 1e12 if x3['ms'] > 0 else 0.0}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(N_PIX)
        print(json.dumps(out), flush=True)
    if dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
