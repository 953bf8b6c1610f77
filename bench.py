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
    # NOT a BASELINE config and never the headline: the reference script's own operating point (batch_size = pe_batch_size = 8, n_pix = 1024,
    # bbhMahoGANy.py:84-89), what a user who drops the package into BBH_version/ unchanged runs.  Both loop bodies replay as captured hipGraphs.
    'refdefaults': {'n_pix': 1024, 'cnn_batch': 8, 'gan_batch': 8, 'waves': 8, 'gflop_cnn': 7.62, 'gflop_gan': 39.8, 'online': False, 'graph': True},
}
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = 1/16 of the bf16 peak
PEAK_HBM_TBS = 8.0                 # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_F64_VALU_TFLOPS = 78.6        # fp64 vector: half the guide's FP32 vector peak (157.3; the guide has no FP64 row; MI355X datasheet 78.6)


def pmc_traffic_per_launch(run_mix):
    """Fabric-side bytes per MFMA conv launch (direct conv_mfma* kernels and the transform-domain conv_wino_kernel together) from the committed PMC
    passes of this same command (profiles/r0N_pmc_traffic.json, written by scripts/pmc_traffic.py from `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` runs; counters cannot be read from inside the timed run).  A file is only believed when its launch mix is this run's:
    run_mix = (direct conv launches, transform-domain conv launches) of the timed steps; the file's ratio of the two kernel families must agree
    (a stale file from another kernel generation does not).  Returns (bytes per launch | None, source, note)."""
    for name in ('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        ks = json.load(open(path))['kernels']
        direct = [v for k, v in ks.items() if 'conv_mfma' in k]          # conv_mfma_pipe_kernel, conv_mfma_dma_kernel, conv_mfma_kernel
        wino = [v for k, v in ks.items() if 'conv_wino_kernel' in k or 'conv_wino_s2_kernel' in k]
        nd, nw = sum(v['launches'] for v in direct), sum(v['launches'] for v in wino)
        rd, rw = run_mix
        if nd + nw == 0 or rd + rw == 0 or abs(nw / float(nd + nw) - rw / float(rd + rw)) > 0.02:
            return None, 'profiles/' + name, ('the newest committed PMC file has %d direct and %d transform-domain conv launches, this run %d and %d per '
                                              'timed region: not the same launch mix, traffic withheld' % (nd, nw, rd, rw))
        b = sum(v['launches'] * v['hbm_bytes_per_launch'] for v in direct + wino)
        return b / (nd + nw), 'profiles/' + name, None
    return None, None, 'no PMC traffic file under profiles/'


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


def cpu_baseline(n_pix, cnn_batch=256, gan_batch_max=512, seconds_budget=30.0):
    """The torch-CPU port of the same two steps (oracle/torch_ref.py) on this box's host cores.  Protocol of BASELINE.md section 2 / SURVEY 8d:
    same batch as the GPU line where the budget allows, >= 3 warm-up + >= 5 timed iterations, MEDIAN.
      CNN leg: train_on_batch at the workload's own CNN batch (256 on the headline: 3 + 5 steps, ~4 s each on 16 cores).
      GAN leg: a batch-512 iteration is ~40 TFLOP (minutes on the host), so the leg runs at the largest power-of-two batch whose 3 + 5
               iterations fit `seconds_budget`, chosen from one probe iteration at batch 8; the batch is a field of the object.
    value = 1 / (t_CNN + t_GAN) per waveform, the CPU counterpart of the GPU line's B / (t_CNN + t_GAN)."""
    from oracle import torch_ref as T
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    WARM, TIMED = 3, 5

    def median_time(fn):
        for _ in range(WARM):
            fn()
        ts = []
        for _ in range(TIMED):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts

    bc = cnn_batch
    pe = T.PENet(n_pix)
    x, ymc, yq = torch.randn(bc, n_pix, 1), torch.rand(bc) * 15 + 20, torch.rand(bc) * 0.5 + 0.5
    t_cnn_batch, cnn_ts = median_time(lambda: pe.train_on_batch(x, ymc, yq))
    t_cnn = t_cnn_batch / bc
    del pe, x
    gan = T.GAN(n_pix, np.random.RandomState(0).randn(n_pix))
    probe = torch.randn(8, n_pix)
    gan.iteration(probe, 8)                                        # first call: allocations, oneDNN primitive creation
    t0 = time.perf_counter(); gan.iteration(probe, 8); per_wave = (time.perf_counter() - t0) / 8
    bg = 8
    while bg < gan_batch_max and (WARM + TIMED) * per_wave * (2 * bg) <= seconds_budget:
        bg *= 2
    real = torch.randn(bg, n_pix)
    t_gan_batch, gan_ts = median_time(lambda: gan.iteration(real, bg))
    t_gan = t_gan_batch / bg
    return {'value': 1.0 / (t_cnn + t_gan), 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port',
            'sample': 'torch-CPU fp32 port (oracle/torch_ref.py), n_pix=%d: median of %d CNN train steps at batch %d and of %d GAN iterations at batch %d, '
                      'each after %d warm-up iterations' % (n_pix, TIMED, bc, TIMED, bg, WARM),
            'protocol': {'warmup': WARM, 'timed': TIMED, 'statistic': 'median'},
            'cnn_batch': bc, 'gan_batch': bg, 'cnn_waveforms_per_s': 1.0 / t_cnn, 'gan_waveforms_per_s': 1.0 / t_gan,
            'cnn_step_seconds': [round(t, 3) for t in cnn_ts], 'gan_iteration_seconds': [round(t, 3) for t in gan_ts],
            'note': 'the GPU line runs CNN batch %d and GAN batch %d; the CPU GAN leg runs at batch %d, the largest power of two (up to the GPU line\'s) whose %d '
                    'iterations fit %.0f s on this host (baseline only)' % (cnn_batch, gan_batch_max, bg, WARM + TIMED, seconds_budget)}


def run_opt_in_child(cmd, env, timeout_s):
    """The opt-in bf16-split leg: the same bench command in a CHILD process (fresh GPU context), started only after the headline numbers of this
    process are final.  Whatever happens to it -- non-zero exit, device fault, hang (killed at timeout_s), garbage on stdout -- comes back as
    {'error': ...}; it can never cost the parent its line.  Returns the child's parsed JSON line otherwise."""
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {'error': 'child killed after %d s (hang or too slow)' % timeout_s}
    except OSError as e:
        return {'error': 'child could not be started: %s' % e}
    if r.returncode != 0:
        tail = r.stderr.decode('utf-8', 'replace').strip().splitlines()[-3:]
        return {'error': 'child exited with code %d: %s' % (r.returncode, ' | '.join(tail))}
    for line in reversed(r.stdout.decode('utf-8', 'replace').strip().splitlines()):
        if line.startswith('{'):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {'error': 'child printed no JSON line'}


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
                    help="default: BASELINE configs[1]+[2] (n_pix 2048, stored bank in HBM); cfg5: configs[4] (srate 4096, templates synthesised in the loop); "
                         "refdefaults: the reference script's own batch 8 / n_pix 1024 (not a BASELINE config), loop bodies replayed as hipGraphs")
    ap.add_argument('--no-graph', action='store_true', help='refdefaults: run the loop bodies eagerly instead of replaying captured hipGraphs')
    ap.add_argument('--predict-batch', type=int, default=0, help='chunk size of generator.predict for the fake half (0: the GAN batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--batch-scale', type=int, default=1, help='multiply the per-GPU batch sizes: `--gpus 1 --batch-scale N` runs the GLOBAL batch of an N-GPU '
                                                                'job on one GPU -- its last_losses are what `--gpus N` prints (same draws, rank-tiled)')
    ap.add_argument('--opt-in', action='store_true', help='after the headline numbers are final, time the same K steps under the opt-in bf16-split conv math in a '
                                                          'CHILD process (its failure, hang or fault becomes opt_in.error; never part of value)')
    ap.add_argument('--no-opt-in', action='store_true', help='(accepted for old command lines; the extra leg is off unless --opt-in is given)')
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
    N_PIX, CNN_BATCH, GAN_BATCH, WAVES = wl['n_pix'], wl['cnn_batch'] * args.batch_scale, wl['gan_batch'] * args.batch_scale, wl['waves'] * args.batch_scale
    predict_batch = args.predict_batch or GAN_BATCH
    graphed = bool(wl.get('graph')) and not args.no_graph and args.gpus == 1
    if args.config == 'refdefaults' and args.bank == 100000:
        args.bank = 50000                        # sample_num of gw_template_maker.py:60
    dev = engine.device()
    engine.set_init_seed(1)                     # identical initial weights on every rank
    engine.set_device_seed(1000)                # ONE device stream: every rank takes the counters of its rows of the global draw (latents, noise, dropout
                                                # masks: engine.PhiloxStream.take_rows), so N ranks x B rows draw what one rank draws for N x B rows
    random.seed(1); np.random.seed(1)           # host index stream identical on all ranks; rank r keeps its slice

    # synthetic 2048-/4096-sample BBH segments: this project's FD chirp through an analytic aLIGO-like PSD (no LAL, no lalinference
    # PSD file here), whitened, aligned and cropped by the fused synthesiser; scaled to unit variance (the role of gw_norm_constant,
    # gw_template_maker.py:782).  Rank-offset stream: no exchange between ranks.
    f = np.arange(N_PIX * 2 + 1) * 0.25
    psd = 1e-46 * ((np.maximum(f, 10.0) / 150.0) ** -4.0 + 2.0 + 2.0 * (f / 150.0) ** 2.0)
    psd[f < 10.0] = 0.0
    synth = T.OnlineBank(N_PIX, 4, psd, seed=1000, noise=None)      # the stored bank: the SAME templates on every rank (ranks slice one index list)
    probe, _ = synth.draw(4096)
    synth.g = 1.0 / float(probe.std())
    del probe
    bank_n = 0 if wl['online'] else args.bank
    if wl['online']:
        bank = None
        online = T.OnlineBank(N_PIX, 4, psd, gw_norm_constant=synth.g, seed=2000 + rank, noise='coloured')
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
    elif graphed:
        cnn_step = bbh.GraphedPEStep(nets.signal_pe, bank, CNN_BATCH)
        gan_step = bbh.GraphedGANStep(nets, bank, event, GAN_BATCH, predict_batch=predict_batch)
    else:
        def cnn_step():
            return bbh.pe_train_step(nets.signal_pe, bank, CNN_BATCH, rank=rank, world=world)

        def gan_step():
            return bbh.gan_train_step(nets, bank, event, GAN_BATCH, rank=rank, world=world, predict_batch=predict_batch)

    last = {}

    def step():
        # (refdefaults replays hipGraphs: every replay returns the step's losses through one device -> host read, as the eager loop does)
        # the return values of the three train_on_batch calls the loop prints (bbhMahoGANy.py:1165, :1292, :1296) are kept and checked after
        # the timed region: a launch that wrote garbage at these batch sizes must not score
        for _ in range(WAVES // CNN_BATCH):
            last['cnn'] = cnn_step()
        last['gan'] = gan_step()

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
    conv = ops.prof_collect(0); wgrad = ops.prof_collect(1); syn_in_step = ops.prof_collect(3); x3 = ops.prof_collect(2); wino = ops.prof_collect(5); wgwino = ops.prof_collect(6); wino2 = ops.prof_collect(7); wgwino2 = ops.prof_collect(8)
    last_losses = {'cnn [total, mc_loss, q_loss, mc_acc, q_acc]': [float(v) for v in last['cnn']],
                   'gan [sg_loss, sg_acc, sd_loss, sd_acc]': [float(v) for v in last['gan']]}
    bad = [k for k, v in last_losses.items() if not np.all(np.isfinite(v))]
    sg_loss, sg_acc, sd_loss, sd_acc = last['gan']
    if not (0.0 <= sg_acc <= 1.0 and 0.0 <= sd_acc <= 1.0 and sg_loss >= 0.0 and sd_loss >= 0.0 and all(v >= 0.0 for v in last['cnn'][:3])):
        bad.append('range')
    if bad:
        sys.stderr.write('bench.py: rank %d: the last step returned non-finite or out-of-range losses (%s): %r\n' % (rank, bad, last_losses))
        sys.exit(3)

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
    # N > 1: the exchange step measured, not estimated -- one more step with every all-reduce bracketed by HIP events on the launch stream
    coll = None
    if dp:
        dp.reset_counters(); dp.timing = True
        barrier()
        t = time.perf_counter()
        step()
        barrier()
        t_coll_step = time.perf_counter() - t
        dp.timing = False
        coll = dp.collect()
        # ... and checked against what the models' shapes say one step must exchange (dist.expected_collectives): the first real N > 1 run
        # validates its own exchange step; a mismatch is an error, not a footnote
        expect = dist.expected_collectives(nets, WAVES // CNN_BATCH)
        coll['expected'] = expect
        coll['matches_expected'] = (coll['calls'], coll['bytes']) == (expect['calls'], expect['bytes'])
        if not coll['matches_expected']:      # the line is still printed (the measurement is not thrown away); the process then exits 4
            sys.stderr.write('bench.py: rank %d: one step issued %d all-reduces / %d bytes, the models\' shapes say %d / %d (%r)\n'
                             % (rank, coll['calls'], coll['bytes'], expect['calls'], expect['bytes'], expect['parts']))
        coll.update(step_ms=1e3 * t_coll_step, share_of_step=(coll['ms'] * 1e-3 / t_coll_step) if coll['ms'] is not None else None,
                    note='all-reduces of ONE step on rank 0 (flat gradient buffers of the CNN x2, D, G; SyncBN sums; loss scalars), event-bracketed on the '
                         'launch stream; they are not overlapped with compute, so ms is their whole cost to the step (it includes waiting for the slowest rank)')
    # metric iii: rows/s of the fused synthesiser, kernel time from HIP events on the launch stream.  default: templates only (prior -> chirp ->
    # whiten -> both inverse FFTs -> arg-max -> slide -> crop, one kernel); cfg5: the launch the CNN loop makes -- the same plus gen_noise ->
    # whiten_data('td') -> crop -> add in the same workgroup.  Priced twice: fp64 VALU operations (what bounds it) and SURVEY 8d's byte figure.
    SYN_NB = 16384
    syn_bank = online if wl['online'] else synth
    syn_bank.draw(1024)
    ops.prof_enable(True); ops.prof_reset()
    t_syn = timed(lambda: syn_bank.draw(SYN_NB), 3)
    ops.prof_enable(False)
    syn = ops.prof_collect(3)
    nz = None
    if wl['online']:
        ops.prof_enable(True); ops.prof_reset()
        timed(lambda: online.draw_noise(SYN_NB), 3)
        ops.prof_enable(False)
        nz = ops.prof_collect(4)

    if rank == 0:
        value = world * WAVES * args.steps / dt
        # Forward / data-gradient convolutions = two kernel families: the direct implicit-GEMM kernels (executed flop = algorithmic flop) and the
        # transform-domain kernel of the unit-stride 5-tap layers (conv_wino.hip: 6 multiplies per two outputs instead of 10 -- it EXECUTES 0.6 of the
        # algorithmic count, which is what the library reports for it).  achieved / frac price EXECUTED flop against the matrix peak (so frac <= 1);
        # the algorithmic rate is reported beside it.
        WINO_RATIO, WINO2_RATIO = 0.6, 0.7     # F(2,5): 6 of 10 multiplies; stride-2 layers, F(2,3) + F(2,2): 7 of 10
        fam_ms = conv['ms'] + wino['ms'] + wino2['ms']
        fam_exec = conv['flop'] + wino['flop'] + wino2['flop']
        fam_alg = conv['flop'] + wino['flop'] / WINO_RATIO + wino2['flop'] / WINO2_RATIO
        n_td = wino['launches'] + wino2['launches']
        wg_td = {'launches': wgwino['launches'] + wgwino2['launches'], 'ms': wgwino['ms'] + wgwino2['ms'], 'flop': wgwino['flop'] + wgwino2['flop'],
                 'alg': wgwino['flop'] / WINO_RATIO + wgwino2['flop'] / WINO2_RATIO, 'bytes': wgwino['bytes'] + wgwino2['bytes']}
        ach = fam_exec / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
        traffic, traffic_src, traffic_why = pmc_traffic_per_launch((conv['launches'], n_td))
        syn_tbs = syn['bytes'] / (syn['ms'] * 1e-3) / 1e12 if syn['ms'] > 0 else 0.0
        out = {
            'metric': 'waveforms/sec (CNN+GAN step, %d-sample BBH)' % N_PIX, 'value': value, 'unit': 'waveforms/s',
            'n_gpus': world, 'ranks': world, 'collective_backend': ('rccl' if backend == 'nccl' else backend) if dp else None, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'last_losses': last_losses,
            'collectives': coll,
            'config': {'workload': ('BASELINE %s; per GPU and step: %d x CNN point-estimator train_on_batch(batch=%d) + 1 GAN iteration(batch=%d) '
                                    '(G.predict, D step on 2B, G step through frozen D); n_pix=%d; %s')
                                   % ('configs[4] (cfg5)' if wl['online'] else ('-- NOT a BASELINE config: the reference script\'s own defaults (bbhMahoGANy.py:84-89), loop bodies %s'
                                                                                 % ('replayed as captured hipGraphs' if graphed else 'run eagerly')
                                                                                 if args.config == 'refdefaults' else 'configs[1]+[2]'), WAVES // CNN_BATCH, CNN_BATCH, GAN_BATCH, N_PIX,
                                      'every batch synthesised on the GPU inside the step: CNN rows = template + PSD-coloured noise whitened with the same PSD, one launch (prior -> chirp -> irFFT -> align -> crop -> gen_noise -> whiten_data(td) -> add); GAN real images = [noise-free template | coloured whitened noise], two launches' if wl['online']
                                      else '%d whitened BBH templates synthesised on the GPU into an HBM-resident bank before the timed region' % bank_n),
                       'name': args.config, 'n_pix': N_PIX, 'cnn_batch': CNN_BATCH, 'gan_batch': GAN_BATCH, 'predict_batch': predict_batch,
                       'waveforms_per_step_per_gpu': WAVES, 'bank_rows': bank_n, 'parallelism': 'dp%d' % world, 'hipgraph_replay': graphed,
                       'global_cnn_batch': CNN_BATCH * world, 'global_gan_batch': GAN_BATCH * world, 'batch_scale': args.batch_scale,
                       'equivalence': 'losses are normalised by the GLOBAL batch and every random draw (host index list, device latents / noise / dropout masks) is '
                                      'the rank slice of ONE global draw: `--gpus N` and `--gpus 1 --batch-scale N` print the same last_losses up to fp32 summation order'},
            'roofline': {'bound': 'mfma', 'kernel': 'Conv1D forward + data gradient on the fp32 matrix instructions: conv_wino_kernel (transform-domain F(2,5), unit-stride 5-tap '
                                                    'layers, v_mfma_f32_16x16x4_f32) + conv_mfma_pipe_kernel (direct implicit GEMM, v_mfma_f32_32x32x2_f32; + conv_mfma_dma_kernel / '
                                                    'conv_mfma_kernel for ragged or 1-tap shapes)',
                         'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                         'achieved_note': 'EXECUTED matrix-pipe flop of both kernel families / their summed launch time (HIP events on the launch stream)',
                         'algorithmic_tflops': fam_alg / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0,
                         'traffic_note': traffic_why or ('fabric-side bytes per conv launch of both families (2*FETCH_SIZE + WRITE_SIZE KiB, gfx950 FETCH correction) from the '
                                                         'separate PMC passes of the default-config command summarised in %s (launch mix checked against this run)' % traffic_src),
                         'launches': conv['launches'] + n_td, 'avg_launch_ms': fam_ms / max(conv['launches'] + n_td, 1),
                         'executed_flop_per_launch': fam_exec / max(conv['launches'] + n_td, 1),
                         'algorithmic_flop_per_launch': fam_alg / max(conv['launches'] + n_td, 1),
                         'algorithmic_bytes_per_launch': (conv['bytes'] + wino['bytes'] + wino2['bytes']) / max(conv['launches'] + n_td, 1),
                         'direct_kernels': {'launches': conv['launches'], 'avg_launch_ms': conv['ms'] / max(conv['launches'], 1),
                                            'achieved': conv['flop'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0,
                                            'frac': (conv['flop'] / (conv['ms'] * 1e-3) / 1e12 if conv['ms'] > 0 else 0.0) / PEAK_F32_MFMA_TFLOPS},
                         'transform_domain_kernel': {'launches': wino['launches'], 'avg_launch_ms': wino['ms'] / max(wino['launches'], 1),
                                                     'multiplies_executed_per_algorithmic': WINO_RATIO,
                                                     'achieved_executed': wino['flop'] / (wino['ms'] * 1e-3) / 1e12 if wino['ms'] > 0 else 0.0,
                                                     'frac': (wino['flop'] / (wino['ms'] * 1e-3) / 1e12 if wino['ms'] > 0 else 0.0) / PEAK_F32_MFMA_TFLOPS,
                                                     'algorithmic_tflops': wino['flop'] / WINO_RATIO / (wino['ms'] * 1e-3) / 1e12 if wino['ms'] > 0 else 0.0,
                                                     'stride2_kernel': {'kernel': 'conv_wino_s2_kernel: F(2,3) + F(2,2) on the even / odd rows of the stride-2 5-tap layers (forward and merged data gradient)',
                                                                        'launches': wino2['launches'], 'avg_launch_ms': wino2['ms'] / max(wino2['launches'], 1),
                                                                        'multiplies_executed_per_algorithmic': WINO2_RATIO,
                                                                        'achieved_executed': wino2['flop'] / (wino2['ms'] * 1e-3) / 1e12 if wino2['ms'] > 0 else 0.0,
                                                                        'algorithmic_tflops': wino2['flop'] / WINO2_RATIO / (wino2['ms'] * 1e-3) / 1e12 if wino2['ms'] > 0 else 0.0},
                                                     'note': 'F(2,5) on points {0, 1, -1, 1/2, -2, inf}: fp32 operands, fp32 products, 6 per output pair instead of 10; error against '
                                                             'fp64 1.2-1.4x the direct fp32 chain\'s (profiles/r05_winograd_gate1.txt); GENNET_CONV_MATH=fp32 runs the direct kernels everywhere'},
                         'wgrad_mfma_kernel': {'kernel': 'weight gradient: wgrad_wino_kernel / wgrad_wino_s2_kernel (transform domain: the unit-stride and the stride-2 5-tap layers) + wgrad_pipe_kernel / wgrad_mfma_kernel (direct)',
                                               'achieved': (wgrad['flop'] + wg_td['flop']) / ((wgrad['ms'] + wg_td['ms']) * 1e-3) / 1e12 if wgrad['ms'] + wg_td['ms'] > 0 else 0.0,
                                               'frac': ((wgrad['flop'] + wg_td['flop']) / ((wgrad['ms'] + wg_td['ms']) * 1e-3) / 1e12 if wgrad['ms'] + wg_td['ms'] > 0 else 0.0) / PEAK_F32_MFMA_TFLOPS,
                                               'achieved_note': 'EXECUTED flop of both families / their summed launch time',
                                               'algorithmic_tflops': (wgrad['flop'] + wg_td['alg']) / ((wgrad['ms'] + wg_td['ms']) * 1e-3) / 1e12 if wgrad['ms'] + wg_td['ms'] > 0 else 0.0,
                                               'launches': wgrad['launches'] + wg_td['launches'], 'avg_launch_ms': (wgrad['ms'] + wg_td['ms']) / max(wgrad['launches'] + wg_td['launches'], 1),
                                               'direct_kernels': {'launches': wgrad['launches'], 'avg_launch_ms': wgrad['ms'] / max(wgrad['launches'], 1),
                                                                  'achieved': wgrad['flop'] / (wgrad['ms'] * 1e-3) / 1e12 if wgrad['ms'] > 0 else 0.0},
                                               'transform_domain_kernels': {'launches': wg_td['launches'], 'avg_launch_ms': wg_td['ms'] / max(wg_td['launches'], 1),
                                                                            'achieved_executed': wg_td['flop'] / (wg_td['ms'] * 1e-3) / 1e12 if wg_td['ms'] > 0 else 0.0,
                                                                            'algorithmic_tflops': wg_td['alg'] / (wg_td['ms'] * 1e-3) / 1e12 if wg_td['ms'] > 0 else 0.0,
                                                                            'stride2_launches': wgwino2['launches']},
                                               'algorithmic_bytes_per_launch': (wgrad['bytes'] + wg_td['bytes']) / max(wgrad['launches'] + wg_td['launches'], 1)},
                         'mfma_kernel_time_share': (conv['ms'] + wino['ms'] + wino2['ms'] + wgrad['ms'] + wg_td['ms']) * 1e-3 / dt,
                         'profiler_note': 'the per-launch figures come from HIP events the library records on the launch stream around every MFMA launch INSIDE the '
                                          'timed region (two hipEventRecord per launch, ~%d launches per step): their cost is included in value, i.e. counts '
                                          'against this line' % ((conv['launches'] + n_td + wgrad['launches'] + wg_td['launches']) // max(args.steps, 1)),
                         'step_algorithmic_tflops': world * WAVES * args.steps * (wl['gflop_cnn'] + wl['gflop_gan']) * 1e-3 / dt},
            'roofline_synth': {'bound': 'valu_f64', 'kernel': 'synth_fused_kernel<.., NOISE=%s> (%s)' % (('true', 'gn_synth_templates_noise: prior + template + coloured whitened noise')
                                                                                                         if wl['online'] else ('false', 'gn_synth_templates_prior')),
                               'achieved': syn['flop'] / (syn['ms'] * 1e-3) / 1e12 if syn['ms'] > 0 else 0.0, 'peak': PEAK_F64_VALU_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': (syn['flop'] / (syn['ms'] * 1e-3) / 1e12 if syn['ms'] > 0 else 0.0) / PEAK_F64_VALU_TFLOPS, 'traffic': None,
                               'rows_per_launch': SYN_NB, 'launches': syn['launches'], 'avg_launch_ms': syn['ms'] / max(syn['launches'], 1),
                               'algorithmic_f64_flop_per_row': syn['flop'] / max(syn['launches'], 1) / SYN_NB,
                               'algorithmic_bytes_per_row': syn['bytes'] / max(syn['launches'], 1) / SYN_NB,
                               'algorithmic_equivalent_TBps': syn_tbs, 'algorithmic_equivalent_frac_of_hbm_peak': syn_tbs / PEAK_HBM_TBS,
                               'kernel_rows_per_s': syn['launches'] * SYN_NB / (syn['ms'] * 1e-3) if syn['ms'] > 0 else 0.0,
                               'launches_inside_timed_steps': syn_in_step['launches'], 'ms_inside_timed_steps': syn_in_step['ms'],
                               'noise_kernel': None if nz is None else {
                                   'kernel': 'noise_whitened_kernel (gn_noise_whitened: gen_noise -> whiten_data(td) -> crop)', 'launches': nz['launches'],
                                   'avg_launch_ms': nz['ms'] / max(nz['launches'], 1), 'achieved': nz['flop'] / (nz['ms'] * 1e-3) / 1e12 if nz['ms'] > 0 else 0.0,
                                   'frac': (nz['flop'] / (nz['ms'] * 1e-3) / 1e12 if nz['ms'] > 0 else 0.0) / PEAK_F64_VALU_TFLOPS,
                                   'kernel_rows_per_s': nz['launches'] * SYN_NB / (nz['ms'] * 1e-3) if nz['ms'] > 0 else 0.0},
                               'note': 'the kernel keeps spectra and series in registers / LDS and writes only the output row, so fp64 vector arithmetic (and LDS traffic) '
                                       'bounds it, not HBM: achieved = counted fp64 operations (transforms 5 M log2 M each, ~150 per spectrum bin, Box-Muller ~120 per noise '
                                       'bin; csrc/synth_fused.hip, noise_fused.hip) / kernel time against the fp64 vector peak.  algorithmic_equivalent_TBps prices the '
                                       'same launches with the SURVEY 8d byte figure (spectra + PSD + window read, row written) -- traffic the fused kernel never issues'},
            'breakdown': {'cnn_train_waveforms_per_s': world * CNN_BATCH / t_cnn, 'gan_iteration_waveforms_per_s': world * GAN_BATCH / t_gan,
                          'cnn_ms_per_batch': 1e3 * t_cnn, 'gan_ms_per_iteration': 1e3 * t_gan,
                          'synth_templates_per_s': world * SYN_NB / t_syn,
                          'note': 'rank-0 clock, measured after the timed steps; value = B / (t_CNN + t_GAN) comes from the K timed steps only; '
                                  'synth_templates_per_s = OnlineBank.draw wall clock of the launch priced in roofline_synth (no host random numbers)'},
        }
        conv_math = ops.default_conv_math()
        out['config']['conv_math'] = conv_math      # 'wino' (default: transform-domain fp32 on the unit-stride 5-tap layers) | 'fp32' (direct kernels only) | 'bf16x3'
        if conv_math == 'bf16x3':      # the opt-in experiment: say so in the line; never the default configuration
            out['dtype'] = 'f32 operands split into 3 bf16 pieces on the large unit-stride conv launches (opt-in GENNET_CONV_MATH=%s), f32 elsewhere' % conv_math
            out['roofline']['bf16x3_launches'] = {'launches': x3['launches'], 'avg_launch_ms': x3['ms'] / max(x3['launches'], 1),
                                                  'fp32_equivalent_tflops': x3['flop'] / (x3['ms'] * 1e-3) / 1e12 if x3['ms'] > 0 else 0.0,
                                                  'ceiling_fp32_equivalent_tflops': 310.0,
                                                  'note': 'six bf16 products per fp32 product; the bare v_mfma_f32_32x32x16_bf16 stream sustains 1842-1881 TFLOP/s on live '
                                                          'operands (scripts/mfma_bf16_peak.hip, profiles/r04_mfma_bf16_peak.txt): 307-313 fp32-equivalent is the ceiling; the '
                                                          'split passes run before every such launch and are NOT inside this figure (they are inside value)'}
        if graphed:          # HIP events are not part of the captured graph: the per-kernel figures are not measured on this line
            for k in ('achieved', 'frac', 'traffic', 'avg_launch_ms'):
                out['roofline'][k] = None
            out['roofline']['wgrad_mfma_kernel'] = None
            out['roofline']['mfma_kernel_time_share'] = None
            out['roofline']['note'] = ('loop bodies replayed as hipGraphs: the launch-stream HIP events of the eager path are not captured, so per-kernel '
                                       'figures are not measured here; the same command with --no-graph measures them (eager, same losses bit for bit)')
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(N_PIX, CNN_BATCH, GAN_BATCH)
        print(json.dumps(out), flush=True)           # the ONE line the driver parses: printed before anything experimental runs
        if args.opt_in and world == 1 and not graphed and conv_math != 'bf16x3':
            # the headline line is out; the child gets its own process, its own GPU context and a deadline.  Its result is a SECOND line.
            env = dict(os.environ)
            env['GENNET_CONV_MATH'] = 'bf16x3'
            env.setdefault('GENNET_CONV_WS_GB', str(12 * max(1, N_PIX // 2048)))      # the split operands of the largest launch: 12 GB at n_pix 2048
            cmd = os.environ.get('GENNET_BENCH_OPT_IN_CMD')                            # (tests: a child that fails)
            cmd = cmd.split() if cmd else [sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(max(1, min(args.warmup, 2))),
                                           '--config', args.config, '--bank', str(args.bank), '--no-cpu-baseline']
            child = run_opt_in_child(cmd, env, 600)
            if 'error' in child:
                leg = {'conv_math': 'bf16x3', 'value': None, 'error': child['error']}
            else:
                leg = {'conv_math': 'bf16x3', 'value': child.get('value'), 'unit': child.get('unit'), 'ms_per_step': child.get('ms_per_step'),
                                 'steps': child.get('steps'), 'ratio_to_value': (child['value'] / out['value']) if child.get('value') else None,
                                 'dtype': child.get('dtype'), 'last_losses': child.get('last_losses'),
                                 'split_launches': (child.get('roofline') or {}).get('bf16x3_launches'),
                                 'note': 'NOT the headline and never the default (narrower operand arithmetic than the reference\'s fp32): the same command in a child '
                                         'process under GENNET_CONV_MATH=bf16x3, started after every number of this line was final; DESIGN.md section 6c'}
            print(json.dumps({'opt_in': leg}), flush=True)
    if dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        if coll is not None and not coll['matches_expected']:
            sys.exit(4)


if __name__ == '__main__':
    main()
