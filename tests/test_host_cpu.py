"""CPU-only checks (no compute calls): the C-ABI library loads and exports every symbol include/gennet_hip.h declares, host logic of
the Keras-style engine (graph construction, peephole fusion plan, collect-at-compile trainability, flat parameter segments),
the host helpers of the synthesiser against the reference's golden vectors, and the ts/pars file layout."""
import ctypes
import os
import pickle
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'synth_golden.npz'))


def test_library_exports_every_declared_symbol():
    from gennet_amd import _lib, build
    build.build(verbose=False)
    hdr = open(os.path.join(ROOT, 'include', 'gennet_hip.h')).read()
    declared = set(re.findall(r'\b(gn_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 45
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    L.gn_version.restype = ctypes.c_int
    assert L.gn_version() >= 100


def test_no_cpu_fallback_in_product_path():
    """ops refuse CPU tensors, and nothing under gennet_amd/ or scripts/ imports the oracle (only tests/, smoke() and bench.py's cpu_baseline may)."""
    import torch
    from gennet_amd import _lib, ops
    with pytest.raises(_lib.GennetHipError):
        ops.act_fwd(torch.zeros(8), 'relu')
    for top in ('gennet_amd', 'scripts'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith('.py'):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    assert [m.start() > bench.index('def cpu_baseline(') and m.start() < bench.index('def free_port(') for m in re.finditer(r'(from|import)\s+oracle', bench)] == [True]


def test_graphs_fusion_plan_and_param_counts():
    from gennet_amd import bbh
    n_pix = 2048
    G_ = bbh.generator_model(n_pix); D = bbh.signal_discriminator_model(n_pix); P = bbh.signal_pe_model(n_pix)
    # SURVEY Appendix A parameter counts @2048
    assert sum(p.size for l in P.layers for p in l.params) == 4928514
    assert sum(p.size for l in D.layers for p in l.params) == 3808257
    assert sum(p.size for p in G_.weights) == 31095745 + 7936             # incl. BN moving statistics (Appendix A generator table)
    assert sum(p.size for l in G_.layers for p in l.params) == 30575425   # trainable ("G ~30.6 M trainable")
    assert P.output_shape == [(None, 1), (None, 1)] and G_.output_shape == (None, n_pix, 1) and D.output_shape == (None, 1)
    shapes = [n.out_shape for n in P.nodes if n.layer.__class__.__name__ == 'Conv1D']
    assert shapes == [(1024, 64), (510, 128), (253, 256), (125, 512), (2048, 64), (2044, 128), (2040, 256), (1018, 512), (507, 1024)]
    G_._plan()
    bn = [n for n in G_.nodes if n.layer.__class__.__name__ == 'BatchNormalization']
    assert len(bn) == 6 and all(n.fused_act == ('tanh', 0.0) and n.fused_drop[0] == 0.2 for n in bn)
    assert sum(n.absorbed for n in G_.nodes) == 15                     # 6 x (Activation + Dropout) + the final linear Activation + 2 UpSampling1D
    ups = [n for n in G_.nodes if n.layer.__class__.__name__ == 'UpSampling1D']
    folded = [n for n in G_.nodes if n.fold_up is not None]
    assert len(ups) == 2 and all(u.absorbed for u in ups) and [n.fold_up for n in folded] == ups      # never materialised
    assert [(n.layer.filters, n.layer.stride) for n in folded] == [(64, 2), (128, 1)]
    D._plan()
    convs = [n for n in D.nodes if n.layer.__class__.__name__ == 'Conv2D']
    assert all(n.fused_act == ('leaky', float(np.float32(0.2))) for n in convs)      # K.cast_to_floatx(alpha)


def test_collect_at_compile_trainability():
    """bbhMahoGANy.py:1104-1115: the combined model is compiled while D is frozen -> trains G only; D compiled after unfreezing."""
    from gennet_amd import bbh
    nets = bbh.build_and_compile(np.zeros((64, 1), np.float32), 64)
    g_ids = set(id(p) for l in nets.generator.layers for p in l.params)
    d_ids = set(id(p) for l in nets.signal_discriminator.layers for p in l.params)
    assert set(id(p) for p in nets.signal_discriminator_on_generator._train_params) == g_ids
    assert set(id(p) for p in nets.data_subtraction_on_generator._train_params) == g_ids
    assert set(id(p) for p in nets.signal_discriminator._train_params) == d_ids
    assert all(l.trainable for l in nets.signal_discriminator.layers)
    # three compiled models share the generator's weight objects but own separate optimizers
    assert nets.signal_discriminator_on_generator.optimizer is not nets.data_subtraction_on_generator.optimizer
    # freezing the nested model object alone (without touching its layers) also counts
    g = bbh.generator_model(64); d = bbh.signal_discriminator_model(64)
    comb = bbh.generator_containing_signal_discriminator(bbh.generator_after_subtracting_noise(g, bbh.data_subtraction_model(np.zeros(64), 64)), d)
    d.trainable = False
    comb.compile(loss='binary_crossentropy', optimizer='adam')
    assert set(id(p) for p in comb._train_params) == set(id(p) for l in g.layers for p in l.params)


def test_segments_merge_adjacent_params():
    from gennet_amd.engine import ParamGroup, segments

    class P(object):
        def __init__(self, size):
            self.size, self.group, self.offset = size, None, 0

    ps = [P(10), P(64), P(65)]
    grp = object()
    off = 0
    for p in ps:
        p.group, p.offset = grp, off
        off += -(-p.size // ParamGroup.ALIGN) * ParamGroup.ALIGN
    assert segments(ps) == [(grp, 0, 64 + 64 + 128)]
    assert segments([ps[0], ps[2]]) == [(grp, 0, 64), (grp, 128, 256)]


def test_conv_geometry_tf_rules():
    from gennet_amd import ops
    assert ops.conv_geometry(2048, 5, 1, 'same') == (2048, 2)
    assert ops.conv_geometry(2048, 5, 2, 'same') == (1024, 1)
    assert ops.conv_geometry(2044, 5, 1, 'valid') == (2040, 0)
    assert ops.conv_geometry(1018, 5, 2, 'valid') == (507, 0)


def test_unsupported_configurations_raise():
    from gennet_amd.layers import Conv2D, Dense, UpSampling1D
    with pytest.raises(NotImplementedError):
        Conv2D(8, (3, 3), padding='same')
    with pytest.raises(NotImplementedError):
        UpSampling1D(size=3)
    with pytest.raises(NotImplementedError):
        Dense(4, kernel_initializer='he_normal')


def test_conv1d_strides_and_weight_files_fail_loudly(tmp_path):
    """strides the kernels do not implement are refused when the layer is built (not at launch time); a weights file that is not
    HDF5 is a format error (it is never unpickled)."""
    from gennet_amd import h5lite
    from gennet_amd.engine import Sequential
    from gennet_amd.layers import Conv1D
    with pytest.raises(NotImplementedError):
        Sequential().add(Conv1D(16, 5, strides=3, input_shape=(64, 8)))
    Sequential().add(Conv1D(16, 5, strides=3, input_shape=(64, 4)))          # small-Cin kernel: any stride
    import pickle
    bad = tmp_path / 'w.h5'
    bad.write_bytes(pickle.dumps({'weights': []}))
    m = Sequential(); m.add(Conv1D(16, 5, input_shape=(64, 4)))
    with pytest.raises(h5lite.H5Error):
        m.load_weights(str(bad))


def test_bench_gpus_n_spawns_n_ranks(monkeypatch):
    """`python bench.py --gpus N` started plainly launches N ranks itself (child torch.distributed.run, before torch is imported in
    the parent) and exits with the child's code; a rank whose WORLD_SIZE differs from --gpus exits non-zero."""
    import importlib
    import subprocess
    bench = importlib.import_module('bench')
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'] = cmd; seen['env'] = env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    monkeypatch.delenv('RANK', raising=False); monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # inside a rank: world size must equal --gpus
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, RANK='0', WORLD_SIZE='1'))
    assert r.returncode == 2 and 'WORLD_SIZE' in r.stderr


def test_synth_host_helpers_match_reference_golden():
    from gennet_amd import templates as T
    for key, (M, alpha) in {'tukey_2184': (2184, 1 / 8.), 'tukey_4369': (4369, 1 / 8.), 'tukey_8738': (8738, 1 / 8.), 'tukey_64_half': (64, 0.5)}.items():
        assert np.array_equal(T.tukey(M, alpha), G[key])
    for fs, b0, b1, lo, hi in G['convert_beta']:
        assert T.convert_beta([b0, b1], int(fs), 4) == (int(lo), int(hi))
    np.random.seed(1)
    acc = np.array([np.concatenate([m12, [mc, eta]]) for m12, mc, eta in (T.gen_masses(5.0, 100.0, 'hunt_constrain') for _ in range(200))])
    assert np.array_equal(acc, G['hunt_seed1'])
    assert np.array_equal(np.random.uniform(0, 1, 3), G['hunt_seed1_next_uniform'])
    s = T._whiten_scale(G['wh_psd'], 256)
    assert np.array_equal(G['wh_fd_in'] * s, G['wh_fd_out'])


def test_gen_par_matches_reference_execution():
    """templates.gen_par (host numpy, legacy MT19937 stream) against what gw_template_maker.py:372-460 produced when executed
    (tests/golden/indexing_golden.npz): every field bit-exact, idx exact, same stream position afterwards."""
    from gennet_amd import templates as T
    IG = np.load(os.path.join(ROOT, 'tests', 'golden', 'indexing_golden.npz'))
    rows = IG['gen_par_rows']
    k = 0
    for fs, seed in ((1024, 1), (2048, 2), (4096, 3), (256, 4)):
        np.random.seed(seed)
        for j in range(40):
            p = T.gen_par(fs, 4, mdist='hunt_constrain', beta=[0.45, 0.55], gw_tmp=(j % 13 == 12))
            assert [p.mc, p.M, p.eta, p.m1, p.m2, p.ra, p.dec, p.iota, p.phi, p.psi, p.idx] == list(rows[k][2:13])
            k += 1
        assert np.array_equal(np.random.uniform(0, 1, 3), rows[k][3:6])
        k += 1
    np.random.seed(5)
    assert [T.gen_par(1024, 4, mdist='hunt_constrain', beta=[0.75, 0.95]).idx for _ in range(20)] == list(IG['gen_par_beta_75_95'])


def test_ts_pars_file_layout_roundtrip(tmp_path):
    """[ts (Ns,1,fs) f64, yval], list of __main__.bbhparams, pickle protocol 2, reference file names."""
    from gennet_amd import templates as T
    rng = np.random.RandomState(0)
    ts = [rng.randn(5, 1, 64), np.ones(5, dtype=int)]
    pars = [T.bbhparams(30.0 + i, 60.0, 0.25, 33.0 + i, 27.0, T.RA, T.DEC, T.IOTA, T.PHI, T.PSI, 100 + i, None, None) for i in range(5)]
    base = str(tmp_path) + '/'
    tp, pp = T.save_ts_pars(base, 'gw150914', 0, 50000, '_srate-1024hz_oversamp', ts, pars)
    assert tp.endswith('gw150914_ts_0_50000Samp_srate-1024hz_oversamp.sav') and pp.endswith('gw150914_params_0_50000Samp_srate-1024hz_oversamp.sav')
    raw = open(pp, 'rb').read()
    assert raw[:2] == b'\x80\x02' and b'__main__' in raw and b'bbhparams' in raw
    ts2, pars2 = T.load_ts_pars(tp, pp)
    assert np.array_equal(ts2[0], ts[0]) and ts2[0].dtype == np.float64 and ts2[0].shape == (5, 1, 64)
    assert [(p.mc, p.m1, p.m2, p.idx) for p in pars2] == [(p.mc, p.m1, p.m2, p.idx) for p in pars]
    images, labels, ev, evl = T.training_arrays(ts2, pars2)
    assert images.shape == (4, 64) and labels.shape == (4, 2) and np.array_equal(ev, ts[0][-1, 0])
    assert np.allclose(labels[:, 1], [27.0 / (33.0 + i) for i in range(4)])
    assert T.bbhparams.__module__ == 'gennet_amd.templates'


def test_background_writer_orders_jobs_and_reports_errors(tmp_path):
    """hostio.BackgroundWriter: jobs run in submission order on one thread, flush() waits, a failing job's exception
    surfaces in the caller's thread at the next call, close() is idempotent and refuses later submissions."""
    import pickle
    import threading
    import time
    from gennet_amd import hostio
    seen = []
    gate = threading.Event()
    with hostio.BackgroundWriter() as bg:
        bg.submit(lambda: (gate.wait(5), seen.append('a')))
        bg.submit(seen.append, 'b')
        p = str(tmp_path / 'x.sav')
        bg.pickle({'k': [1, 2, 3]}, p)
        bg.pickle({'k': 'later'}, p)
        assert seen == []                        # the caller is not blocked by the first job
        gate.set()
        bg.flush()
        assert seen == ['a', 'b']
        with open(p, 'rb') as f:
            assert pickle.load(f) == {'k': 'later'}
    with pytest.raises(RuntimeError):
        bg.submit(seen.append, 'c')
    bg.close()

    bg = hostio.BackgroundWriter()
    bg.pickle(1, str(tmp_path / 'no_such_dir' / 'y.sav'))
    t0 = time.time()
    with pytest.raises(IOError):
        while time.time() - t0 < 5:
            bg.submit(seen.append, 'd')
            time.sleep(0.01)
    bg.close()                                   # the error was reported once; close() is clean


def test_save_weights_through_the_background_writer_equals_the_inline_write(tmp_path):
    from gennet_amd import bbh, hostio
    m = bbh.signal_discriminator_model(64)
    a, b = str(tmp_path / 'a.h5'), str(tmp_path / 'b.h5')
    m.save_weights(a, True)
    with hostio.BackgroundWriter() as bg:
        m.save_weights(b, True, writer=bg)
    assert open(a, 'rb').read() == open(b, 'rb').read()


def test_lalinf_posterior_conversion_matches_the_reference_executed_fixture(tmp_path):
    """data/get_lalinf_pars.py (in front of row n3): templates.lalinf_pars' closed forms against the reference's own sympy loops executed
    on supplied columns (tests/golden/lalinf_pars_golden.npz), and scripts/get_lalinf_pars.py writing the reference's three files."""
    import pickle
    import subprocess
    from gennet_amd import templates as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lalinf_pars_golden.npz'))
    pars = T.lalinf_pars(g['post_mc'], g['post_q'])
    assert np.abs(pars['m1_m2'] - g['m1_m2']).max() <= 1e-11 * np.abs(g['m1_m2']).max()
    assert np.abs(pars['mc_M'] - g['mc_M']).max() <= 1e-11 * np.abs(g['mc_M']).max()
    assert np.array_equal(pars['mc_q'], np.array([g['post_mc'], g['post_q']]))
    # the pair the posterior-driven maker forms from the file (lalinf_post_waveform_maker.py:385: [column 1, column 0]) is (heavier, lighter)
    m1, m2 = T.m1m2_from_mc_q(g['post_mc'], g['post_q'])
    assert np.allclose(m1, g['m1_m2'][1], rtol=1e-11) and np.allclose(m2, g['m1_m2'][0], rtol=1e-11)
    with pytest.raises(ValueError):
        T.lalinf_pars([30.0], [0.0])
    post = str(tmp_path / 'post.npz')
    np.savez(post, mc=g['post_mc'], q=g['post_q'])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts/get_lalinf_pars.py'), '--posterior', post, '--tag', 'srate-2048', '--mc-M', '1',
                        '--out', str(tmp_path / 'data')], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for stem, key in (('gw150914_m1_m2_lainf_post_srate-2048.sav', 'm1_m2'), ('gw150914_mc_M_lainf_post_srate-2048.sav', 'mc_M'),
                      ('gw150914_mc_q_lalinf_post_srate-2048.sav', 'mc_q')):
        with open(str(tmp_path / 'data' / stem), 'rb') as f:
            a = pickle.load(f)
        assert a.shape == (2, 6) and np.array_equal(a, pars[key])


def test_gen_masses_all_four_distributions_match_the_reference_executed_fixture():
    """gw_template_maker.py:289-370 ('astro', 'hunt_constrain', 'gh', 'metric'): values bit for bit and the same stream position afterwards."""
    from gennet_amd import templates as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'masses_golden.npz'))
    for k, mdist in enumerate(('astro', 'hunt_constrain', 'gh', 'metric')):
        np.random.seed(40 + k)
        rows = []
        for _ in range(25):
            m12, mc, eta = T.gen_masses(5.0, 100.0, mdist)
            rows.append([m12[0], m12[1], mc, eta])
        assert np.array_equal(np.array(rows, np.float64), g[mdist]), mdist
        assert np.array_equal(np.random.uniform(0, 1, 2), g[mdist + '_next']), mdist
    with pytest.raises(ValueError):
        T.gen_masses(5.0, 100.0, 'flat')


@pytest.mark.parametrize("fs", [256, 512])
def test_posterior_mode_draws_match_the_reference_executed_fixture(fs):
    """templates.posterior_block_pars (the host half of sim_data_posterior: every draw from the numpy legacy stream) against
    lalinf_post_waveform_maker.py:356-475 / :649-746 executed over two consecutive blocks (tests/golden/posterior_mode_golden.npz):
    parameters, shuffle, event-like row, and the stream position after each block -- including the randint the event-like row's gen_par
    draws and discards (:440-444 before :460-461)."""
    from gennet_amd import templates as T
    PM = np.load(os.path.join(ROOT, 'tests', 'golden', 'posterior_mode_golden.npz'))
    key = 'pm_%d_' % fs
    _, size, batch_size, seed = [int(v) for v in PM[key + 'meta']]
    f = PM[key + 'm1_m2_file']
    np.random.seed(seed)
    for blk in range(2):
        pars, perm, ev = T.posterior_block_pars(fs, 4, f[1], f[0], PM[key + 'post_mc'], size=size, batch_size=batch_size)
        rows = [pars[i] for i in perm] + [ev]
        got = np.array([[p.mc, p.M, p.eta, p.m1, p.m2, p.ra, p.dec, p.iota, p.phi, p.psi, p.idx] for p in rows])
        assert np.array_equal(got, PM[key + 'pars_%d' % blk])
        st = np.random.get_state()
        assert np.array_equal(np.random.uniform(0, 1, 3), PM[key + 'next_uniform_%d' % blk])
        np.random.set_state(st)
    with pytest.raises(IndexError):
        T.posterior_block_pars(fs, 4, f[1][:3], f[0][:3], None, size=5)      # the event-like row reads posterior row `cnt`: must exist


def test_interrupted_writes_never_truncate_the_previous_file(tmp_path):
    """files are written to path + '.tmp' and moved into place, so a job that dies half-way (here: an unpicklable object, and an
    h5 tree whose write fails) leaves the previous content under the final name; the `with` form closes the writer when the loop raises."""
    from gennet_amd import h5lite, hostio
    path = str(tmp_path / 'samples.sav')
    with hostio.BackgroundWriter() as bg:
        bg.pickle([1, 2, 3], path)
    with pytest.raises(Exception):
        with hostio.BackgroundWriter() as bg:
            bg.pickle([lambda: 0], path)                         # pickling fails after the temporary file was opened
    with open(path, 'rb') as f:
        assert pickle.load(f) == [1, 2, 3]
    assert not os.path.exists(path + '.tmp')                     # the failed job removed its temporary file
    done = []
    with pytest.raises(KeyboardInterrupt):
        with hostio.BackgroundWriter() as bg:                    # the loop dies: queued jobs are still completed by __exit__
            bg.submit(lambda: done.append(1))
            raise KeyboardInterrupt
    assert done == [1]
    w = h5lite.Writer()
    w.root.attrs['a'] = np.int32(1)
    h5 = str(tmp_path / 'w.h5')
    w.save(h5)
    first = open(h5, 'rb').read()
    w2 = h5lite.Writer()
    w2.tobytes = lambda: (_ for _ in ()).throw(IOError('disk full'))
    with pytest.raises(IOError):
        w2.save(h5)
    assert open(h5, 'rb').read() == first and not os.path.exists(h5 + '.tmp') or open(h5, 'rb').read() == first


def test_measurement_tools_build_and_parse(tmp_path):
    """scripts/mfma_peak.hip (the bare-MFMA-pipe / chunk-loop microbenchmark behind DESIGN section 6) cross-compiles for gfx950, and the counter
    summariser reads a rocprofv3 counter CSV of the shape it documents."""
    import json
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'scripts', 'mfma_peak.hip')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-c', '--cuda-device-only', src, '-o', str(tmp_path / 'mfma_peak.o')], check=True, timeout=600,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = tmp_path / 'pmc' / 'run'
    d.mkdir(parents=True)
    rows = ['Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value']
    for disp in (1, 2):
        rows += ['%d,k_conv,SQ_LDS_BANK_CONFLICT,%d' % (disp, 780 * disp), '%d,k_conv,SQ_LDS_IDX_ACTIVE,%d' % (disp, 1000 * disp),
                 '%d,k_conv,GRBM_GUI_ACTIVE,%d' % (disp, 500 * disp)]
    (d / '1_counter_collection.csv').write_text('\n'.join(rows) + '\n')
    out = tmp_path / 'lds.json'
    subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'pmc_lds.py'), str(tmp_path / 'pmc'), str(out), '256'], check=True, timeout=60, stdout=subprocess.DEVNULL)
    k = json.load(open(out))['kernels']['k_conv']
    assert k['launches'] == 2 and abs(k['conflict_share_of_lds_cycles'] - 0.78) < 1e-12 and abs(k['lds_busy_share_of_launch'] - 3000.0 / (1500.0 * 256)) < 1e-12


def test_expected_collectives_of_one_bench_step():
    """dist.expected_collectives: what bench.py --gpus N checks DataParallel's counters against after its timed region (exit 4 on a mismatch) -- the
    per-step exchange of SURVEY 8e at the headline size, from the models' shapes alone: the CNN's flat gradient twice (19.7 MB each), D's (15.2 MB),
    G's (122 MB), twelve SyncBN sums (fp64), four loss-scalar blocks = 20 all-reduces, 185.4 MB."""
    from gennet_amd import bbh, dist
    nets = bbh.build_and_compile(np.zeros((2048, 1), np.float32), 2048)
    e = dist.expected_collectives(nets, 2)
    assert e['calls'] == 20
    p = e['parts']
    assert 2 * 19.7e6 < p['cnn_gradients'] < 2 * 19.75e6 and 15.2e6 < p['discriminator_gradients'] < 15.3e6
    assert 122e6 < p['generator_through_frozen_discriminator_gradients'] < 122.6e6
    assert p['syncbn_sums'] == 2 * 2 * 8 * (256 * 1024 + 64 + 128 + 256 + 512 + 1024)
    assert p['cnn_loss_scalars'] == 32 and p['discriminator_loss_scalars'] == 8
    assert e['bytes'] == sum(p.values()) and 185.3e6 < e['bytes'] < 185.5e6
    # unpadded parameter counts of SURVEY Appendix A: 4 928 514 (CNN), 3 808 257 (D), 31 095 745 + 7 936 conv-BN (G)
    assert sum(q.size for q in nets.signal_pe._train_params) == 4928514
    assert sum(q.size for q in nets.signal_discriminator._train_params) == 3808257
    # ... of which the moving statistics (half of every BatchNormalization's four vectors) are not trainable and are never exchanged
    assert sum(q.size for q in nets.signal_discriminator_on_generator._train_params) == 31095745 + 7936 - 2 * (256 * 1024 + 1984)


def test_committed_bench_line_keeps_the_driver_contract():
    """The driver parses ONE JSON line from bench.py; the newest committed line (profiles/r04_bench_default.json, printed by `python bench.py` on an
    MI355X) must carry every field of the contract, with the units and meanings the task fixes: whole-job waveforms/s, dtype = the arithmetic type,
    vs_baseline null (BASELINE.md holds no published number), roofline with achieved / peak / frac / traffic, cpu_baseline with cores / kind / sample."""
    import json
    line = open(os.path.join(ROOT, 'profiles', 'r04_bench_default.json')).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline', 'last_losses'):
        assert k in d, k
    assert d['unit'] == 'waveforms/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['n_gpus'] == 1
    assert 'workload' in d['config'] and 'configs[1]+[2]' in d['config']['workload'] and 'model' not in d['config']
    assert abs(d['value'] - d['config']['waveforms_per_step_per_gpu'] * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0.5 < r['frac'] < 1.0
    assert r['traffic'] > r['algorithmic_bytes_per_launch']                    # fabric-side bytes can only exceed the algorithmic ones
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['unit'] == 'waveforms/s'
    assert 1426.0 < d['value'] < 1427.0 and c['value'] < d['value'] / 100.0      # the north-star's >= 100x over the CPU path


def test_bench_line_with_the_opt_in_leg_keeps_the_headline_exact():
    """The opt-in conv math is reported beside the headline, never as it: the committed default line computes `value` in f32 and carries the
    bf16-split figure in its own object."""
    import json
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench_default_with_opt_in.json')))
    assert d['dtype'] == 'f32' and 'conv_math' not in d['config'] and d['vs_baseline'] is None
    o = d['opt_in']
    assert o['conv_math'] == 'bf16x3' and o['unit'] == d['unit'] and o['steps'] == d['steps']
    assert abs(o['ratio_to_value'] - o['value'] / d['value']) < 1e-12 and 1.1 < o['ratio_to_value'] < 1.5
    assert all(np.isfinite(v) for v in o['last_losses']['cnn'] + o['last_losses']['gan'])


def test_every_profile_the_readme_lists_is_committed():
    import re
    txt = open(os.path.join(ROOT, 'profiles', 'README.md')).read()
    names = set(re.findall(r'`(r0[1-5]_[A-Za-z0-9_.]+\.(?:json|csv|txt|log))`', txt))
    assert len(names) > 40
    missing = sorted(n for n in names if not os.path.exists(os.path.join(ROOT, 'profiles', n)))
    assert not missing, missing


def test_opt_in_leg_is_a_child_that_cannot_cost_the_headline():
    """bench.py --opt-in (VERDICT r4 #4 / ADVICE r4): the experimental leg runs in a child process AFTER the headline line has been printed; a child that
    fails, prints garbage or hangs comes back as an error string."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    env = dict(os.environ)
    r = bench.run_opt_in_child([sys.executable, '-c', 'import sys; sys.stderr.write("boom\\n"); sys.exit(3)'], env, 30)
    assert 'error' in r and 'code 3' in r['error'] and 'boom' in r['error']
    r = bench.run_opt_in_child([sys.executable, '-c', 'import time; time.sleep(30)'], env, 1)
    assert 'error' in r and 'killed' in r['error']
    r = bench.run_opt_in_child([sys.executable, '-c', 'print("not json")'], env, 30)
    assert 'error' in r
    r = bench.run_opt_in_child(['/nonexistent/binary'], env, 5)
    assert 'error' in r
    r = bench.run_opt_in_child([sys.executable, '-c', 'print("noise"); print(\'{"value": 2.5, "unit": "waveforms/s"}\')'], env, 30)
    assert r == {'value': 2.5, 'unit': 'waveforms/s'}
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert src.index("print(json.dumps(out), flush=True)") < src.index("child = run_opt_in_child(")      # the line is out before the child starts
    assert 'set_conv_math(' not in src                                                                      # no experimental kernels in the bench process


def test_pmc_traffic_file_is_refused_when_its_launch_mix_is_not_the_runs(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC file; a file from another kernel generation (different ratio of direct to transform-domain
    conv launches) must give traffic: null and a note, not a stale number."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / 'profiles'
    prof.mkdir()
    ks = {'void gn::conv_mfma_pipe_kernel<4, 1, 5, 2>(gn::ConvArgs, int, int, int)': {'launches': 80, 'hbm_bytes_per_launch': 3.0e9},
          'void gn::conv_wino_kernel<4, 0>(gn::ConvArgs, float const*, int, int, int, int)': {'launches': 40, 'hbm_bytes_per_launch': 2.0e9},
          'gn::adam_kernel': {'launches': 16, 'hbm_bytes_per_launch': 1.0e8}}
    json.dump({'kernels': ks}, open(str(prof / 'r05_pmc_traffic.json'), 'w'))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    t, src, why = bench.pmc_traffic_per_launch((160, 80))            # same 2 : 1 mix, other step count
    assert why is None and src == 'profiles/r05_pmc_traffic.json' and abs(t - (80 * 3.0e9 + 40 * 2.0e9) / 120) < 1.0
    t, src, why = bench.pmc_traffic_per_launch((244, 0))             # a run without the transform-domain kernel: not this file's mix
    assert t is None and 'launch mix' in why
    t, src, why = bench.pmc_traffic_per_launch((0, 0))
    assert t is None
