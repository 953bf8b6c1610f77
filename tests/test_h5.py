"""Keras `.h5` layout, CPU only: the dependency-free HDF5 subset (gennet_amd/h5lite.py) and the
Keras 2.x model / weight file layout on top of it (gennet_amd/keras_io.py).

The reader is pinned on the four real Keras files the reference ships (2_model_version/weight_version/*.hdf5) through the
committed structure fixture tests/golden/keras_h5_golden.json (names, shapes, CRC32s -- made by tests/golden/make_h5_golden.py);
those tests skip where /root/reference is absent (the GPU box).  Everything else runs anywhere.
"""
import json
import os
import struct

import numpy as np
import pytest

from gennet_amd import h5lite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/2_model_version/weight_version'
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'keras_h5_golden.json')))
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference data files are not on this machine')


# ---------------------------------------------------------------------------------------------------------------------
# h5lite
# ---------------------------------------------------------------------------------------------------------------------
def test_h5lite_roundtrip_types_and_nesting():
    rng = np.random.RandomState(0)
    w = h5lite.Writer()
    w.root.attrs['keras_version'] = b'2.2.4'
    w.root.attrs['text'] = 'café'                               # str -> utf-8 bytes, as keras encodes its own attrs
    w.root.attrs['names'] = np.array([b'a', b'bcd', b'ef'], dtype='S')
    w.root.attrs['empty'] = np.zeros((0,), np.float64)
    w.root.attrs['i64'] = np.int64(-7)
    w.root.attrs['f32v'] = np.arange(5, dtype=np.float32)
    arrays = {'g1/kernel:0': rng.randn(5, 3, 7).astype(np.float32), 'g1/sub/deeper/bias:0': rng.randn(7).astype(np.float32),
              'scalar': np.asarray(2532, np.int64), 'f64': rng.randn(4, 4), 'u8': rng.randint(0, 255, (3, 9)).astype(np.uint8),
              'empty': np.zeros((0, 4), np.float32)}
    for k, v in arrays.items():
        w.root.create_dataset(k, v)
    w.root.children['g1'].attrs['weight_names'] = [b'g1/kernel:0', b'g1/sub/deeper/bias:0']
    f = h5lite.File(w.tobytes())
    assert f.attrs['keras_version'] == b'2.2.4'
    assert f.attrs['text'].decode('utf-8') == 'café'
    assert f.attrs['names'].tolist() == [b'a', b'bcd', b'ef']
    assert f.attrs['empty'].shape == (0,)
    assert f.attrs['i64'] == -7 and f.attrs['i64'].dtype == np.int64
    assert np.array_equal(f.attrs['f32v'], np.arange(5, dtype=np.float32))
    for k, v in arrays.items():
        d = f[k]
        assert d.shape == v.shape and d.dtype == v.dtype, k
        assert np.array_equal(d.value, v), k
    assert f['g1'].attrs['weight_names'].tolist() == [b'g1/kernel:0', b'g1/sub/deeper/bias:0']
    assert sorted(f.keys()) == sorted(['g1', 'scalar', 'f64', 'u8', 'empty'])
    assert 'g1/sub/deeper' in f and 'g1/nope' not in f
    with pytest.raises(KeyError):
        f['g1/kernel:0/x']


@pytest.mark.parametrize('n', [0, 1, 8, 9, 64, 257, 700])
def test_h5lite_group_btree_sizes(n):
    """0..700 links in one group: one symbol-table node, several nodes under one B-tree node (> 8), two B-tree levels (> 256)."""
    w = h5lite.Writer()
    g = w.root.create_group('layers')
    names = ['layer_%d' % ((i * 7919) % 1000003) for i in range(n)]      # unsorted insertion order
    for i, nm in enumerate(names):
        g.create_dataset(nm, np.full((2,), i, np.float32))
    f = h5lite.File(w.tobytes())
    assert f['layers'].keys() == sorted(names, key=lambda s: s.encode())
    for i, nm in enumerate(names):
        assert f['layers'][nm].value[0] == i
    # B-tree invariants libhdf5 relies on: keys ascend (by the name they point to), every child's names fall in (key_i, key_i+1]
    b = f._buf
    btree, heap = f['layers']._stab
    data = struct.unpack_from('<Q', b, heap + 24)[0]

    def name_at(off):
        return b[data + off: b.index(b'\x00', data + off)]

    def check(addr, lo, hi):
        assert b[addr:addr + 4] in (b'TREE', b'SNOD')
        if b[addr:addr + 4] == b'SNOD':
            cnt = struct.unpack_from('<H', b, addr + 6)[0]
            assert cnt <= 2 * f.leaf_k
            ns = [name_at(struct.unpack_from('<Q', b, addr + 8 + 40 * i)[0]) for i in range(cnt)]
            assert ns == sorted(ns)
            for x in ns:
                assert lo < x <= hi or (lo == b'' and x <= hi)
            return
        _t, level, used = struct.unpack_from('<BBH', b, addr + 4)
        assert used <= 2 * f.internal_k
        keys = [name_at(struct.unpack_from('<Q', b, addr + 24 + 16 * i)[0]) for i in range(used + 1)]
        assert keys == sorted(keys)
        for i in range(used):
            check(struct.unpack_from('<Q', b, addr + 32 + 16 * i)[0], keys[i], keys[i + 1])
    if n:
        check(btree, b'', b'\xff')


def test_h5lite_writes_the_header_forms_h5py_writes():
    """Superblock v0 with libhdf5's default K values, version-1 object headers, and the dataset header message set
    (type, version, flags) found in the real Keras files: dataspace v1 with max dims, datatype/fill/layout marked constant,
    fill value v2 {2,2,1,size 0}, layout v3 contiguous."""
    w = h5lite.Writer()
    w.root.create_dataset('kernel:0', np.ones((16, 1, 50), np.float32))
    raw = w.tobytes()
    assert raw[:8] == b'\x89HDF\r\n\x1a\n' and raw[8:16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])
    assert struct.unpack_from('<HH', raw, 16) == (4, 16)
    assert struct.unpack_from('<Q', raw, 40)[0] == len(raw)                       # end-of-file address
    f = h5lite.File(raw)
    addr = f._load()['kernel:0']
    assert raw[addr] == 1 and addr % 8 == 0
    nmsg, _rc, hsize = struct.unpack_from('<HII', raw, addr + 2)
    p, seen = addr + 16, []
    for _ in range(nmsg):
        t, sz, fl = struct.unpack_from('<HHB', raw, p)
        assert sz % 8 == 0
        seen.append((t, raw[p + 8], fl))
        if t == 0x05:
            assert raw[p + 8:p + 16] == bytes([2, 2, 2, 1, 0, 0, 0, 0])
        if t == 0x01:
            assert raw[p + 8:p + 12] == bytes([1, 3, 1, 0]) and sz == 8 + 2 * 3 * 8
        if t == 0x03:
            assert raw[p + 8:p + 8 + 20] == bytes.fromhex('11201f000400000000002000170800177f000000')   # the real files' float32 message
        p += 8 + sz
    assert p == addr + 16 + hsize
    assert seen == [(0x01, 1, 0), (0x03, 0x11, 1), (0x05, 2, 1), (0x08, 3, 1)]


def test_h5lite_rejects_what_it_does_not_implement():
    with pytest.raises(h5lite.H5Error):
        h5lite.File(b'not an hdf5 file at all')
    w = h5lite.Writer()
    w.root.attrs['big'] = b'x' * 70000
    with pytest.raises(h5lite.H5Error):
        w.tobytes()
    w2 = h5lite.Writer()
    w2.root.create_dataset('c', np.zeros(3, np.complex64))
    with pytest.raises(h5lite.H5Error):
        w2.tobytes()


# ---------------------------------------------------------------------------------------------------------------------
# the reference's real Keras files
# ---------------------------------------------------------------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize('fn', sorted(GOLD))
def test_reader_on_real_keras_files(fn):
    got = h5lite.dump_structure(os.path.join(REF, fn))
    want = GOLD[fn]
    assert got['groups'] == want['groups']
    assert {k: list(v) for k, v in got['datasets'].items()} == want['datasets']
    assert json.loads(json.dumps(got['attrs'])) == want['attrs']


@needs_ref
def test_load_model_on_a_real_keras_file():
    """keras.models.load_model on the reference's d_model.hdf5 (Keras 2.1.6, functional Model): same layers, every weight
    equal to the file's dataset, trainable flags, compile settings and Adam iteration count restored."""
    from gennet_amd.keras.models import load_model
    from gennet_amd import keras_io
    path = os.path.join(REF, 'd_model.hdf5')
    m = load_model(path)
    want = GOLD['d_model.hdf5']
    layers = keras_io.top_layers(m)
    assert [[('InputLayer' if isinstance(l, keras_io.InputLayer) else l.__class__.__name__), l.name] for l in layers] == want['model']['layers']
    f = h5lite.File(path)
    n = 0
    for l in layers:
        for p in keras_io.keras_weights(l):
            assert np.array_equal(p.numpy(), f['model_weights'][l.name][p.name + ':0'].value), p.name
            n += 1
    assert n == 6
    assert m.output_shape == (None, 2) and m.count_params() == 16 * 50 + 50 + 1750 * 50 + 50 + 50 * 2 + 2
    assert all(not l.trainable for l in m.layers)                                 # the file was saved with D frozen
    assert m.loss == 'binary_crossentropy' and abs(m.optimizer.lr - 0.004) < 1e-8 and m.optimizer.beta_1 == 0.5
    assert int(np.asarray(m._pending_optimizer_weights[0])) == 2532
    # and the weights-only sibling file loads into the same architecture
    m.load_weights(os.path.join(REF, 'best_d_weights.hdf5'))


def test_oracle_and_layer_defaults_match_what_keras_itself_recorded():
    """The only Keras-held facts the reference ships: model_config / training_config that Keras 2.1.x wrote into d_model.hdf5 / g_model.hdf5
    (fixture: tests/golden/keras_h5_golden.json, full JSON).  They pin the defaults the oracle restates from memory (SURVEY Appendix B.3,
    B.4, B.7, B.11) and the product's layer / optimizer defaults: BatchNormalization epsilon / momentum / axis / initialisers, LeakyReLU
    alpha as a float32, the glorot VarianceScaling spec, zero biases, Flatten's channels_last order, Adam's beta_2 / epsilon / float32
    hyper-parameter variables."""
    from oracle import keras_ref as K
    from gennet_amd import engine, layers as L
    lays = {fn: GOLD[fn]['model_config']['config']['layers'] for fn in ('d_model.hdf5', 'g_model.hdf5')}
    bns = [l['config'] for l in lays['g_model.hdf5'] if l['class_name'] == 'BatchNormalization']
    assert len(bns) >= 5
    mine = L.BatchNormalization()
    for c in bns:
        assert c['epsilon'] == K.BN_EPS == mine.epsilon == 1e-3
        assert c['momentum'] == mine.momentum == 0.99                                           # Keras' default = what bbhMahoGANy.py:223 passes
        assert c['axis'] == -1 and c['center'] and c['scale']
        assert [c[k]['class_name'] for k in ('gamma_initializer', 'beta_initializer', 'moving_mean_initializer', 'moving_variance_initializer')] \
            == ['Ones', 'Zeros', 'Zeros', 'Ones']
    g = bbh_generator_bn_initial_values()
    assert g == {'gamma': 1.0, 'beta': 0.0, 'moving_mean': 0.0, 'moving_variance': 1.0}
    # LeakyReLU(alpha=0.2): a float32 in the graph
    lk = [l['config'] for l in lays['d_model.hdf5'] if l['class_name'] == 'LeakyReLU']
    assert lk and all(c['alpha'] == float(np.float32(0.2)) for c in lk)
    assert L.LeakyReLU(alpha=0.2).alpha == float(np.float32(0.2))
    # kernels: VarianceScaling(scale 1, fan_avg, uniform) = glorot_uniform; biases Zeros; conv / dense carry a bias
    kinds = [l for fn in lays for l in lays[fn] if 'kernel_initializer' in l['config']]
    assert len(kinds) >= 8
    for l in kinds:
        ki = l['config']['kernel_initializer']
        assert ki == {'class_name': 'VarianceScaling', 'config': {'distribution': 'uniform', 'scale': 1.0, 'seed': None, 'mode': 'fan_avg'}}
        assert l['config']['bias_initializer'] == {'class_name': 'Zeros', 'config': {}} and l['config']['use_bias'] is True
    for shape in ((16, 1, 50), (1750, 50), (5, 512, 1024), (5, 5, 256, 512)):
        rec = int(np.prod(shape[:-2]))
        fan_avg = 0.5 * rec * (shape[-2] + shape[-1])
        lim = np.sqrt(3.0 * 1.0 / fan_avg)                                   # VarianceScaling: uniform limit sqrt(3 scale / n), n = fan_avg
        for w in (K.glorot_uniform(np.random.RandomState(0), shape), engine.glorot_uniform(shape)):
            assert w.dtype == np.float32 and np.abs(w).max() <= lim and np.abs(w).max() > 0.97 * lim or w.size < 1000
            assert abs(w.std() - lim / np.sqrt(3.0)) < 0.08 * lim
    fl = [l['config'] for l in lays['d_model.hdf5'] if l['class_name'] == 'Flatten']
    assert fl and fl[0]['data_format'] == 'channels_last'
    conv = [l['config'] for l in lays['d_model.hdf5'] if l['class_name'] == 'Conv1D'][0]
    assert conv['dilation_rate'] == [1] and conv['padding'] == 'valid' and conv['activation'] == 'linear'
    # Adam as Keras records it: float32 variables for lr / beta_1 / beta_2, epsilon = K.epsilon() = 1e-7, no amsgrad, no decay
    oc = GOLD['d_model.hdf5']['training']['optimizer_config']
    assert oc['class_name'] == 'Adam'
    c = oc['config']
    assert c['epsilon'] == K.K_EPS == engine.Adam().epsilon == 1e-7
    assert c['beta_2'] == float(np.float32(0.999)) == engine.Adam().beta_2 and c['beta_2'] != 0.999
    assert c['beta_1'] == 0.5 and c['lr'] == float(np.float32(0.004)) == engine.Adam(lr=0.004).lr and c['decay'] == 0.0 and c['amsgrad'] is False
    # ... and the oracle's step uses exactly those float32 values
    p, g_, m, v = np.ones(3), np.full(3, 0.25), np.zeros(3), np.zeros(3)
    _, m1, v1 = K.adam_step(p, g_, m, v, 1)
    assert np.array_equal(v1, (1 - float(np.float32(0.999))) * g_ * g_) and not np.array_equal(v1, (1 - 0.999) * g_ * g_)
    assert GOLD['d_model.hdf5']['training']['loss'] == 'binary_crossentropy'


def bbh_generator_bn_initial_values():
    from gennet_amd import bbh
    g = bbh.generator_model(64)
    bn = [l for l in g._top if l.__class__.__name__ == 'BatchNormalization'][0]
    return {k: float(np.unique(getattr(bn, k).numpy())[0]) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')}


# ---------------------------------------------------------------------------------------------------------------------
# Keras layout written by this package
# ---------------------------------------------------------------------------------------------------------------------
def _weights(m):
    return [w.copy() for w in m.get_weights()]


def test_save_weights_layout_generator(tmp_path):
    from gennet_amd import bbh, keras_io
    g = bbh.generator_model(64)
    path = str(tmp_path / 'generator.h5')
    g.save_weights(path, True)
    f = h5lite.File(path)
    assert f.attrs['backend'] == b'tensorflow' and f.attrs['keras_version'] == b'2.2.4'
    names = [n.decode() for n in f.attrs['layer_names'].tolist()]
    assert names == [l.name for l in g._top] and len(names) == len(set(names))
    assert any(n.startswith('batch_normalization_') for n in names) and any(n.startswith('up_sampling1d_') for n in names)
    bn = [l for l in g._top if l.__class__.__name__ == 'BatchNormalization'][0]
    assert [n.decode() for n in f[bn.name].attrs['weight_names'].tolist()] == [bn.name + s for s in ('/gamma:0', '/beta:0', '/moving_mean:0', '/moving_variance:0')]
    conv = [l for l in g._top if l.__class__.__name__ == 'Conv1D'][0]
    k = f[conv.name][conv.name]['kernel:0']
    assert k.shape == (5, 256, 64) and k.dtype == np.float32                      # keras' (k, Cin, Cout)
    assert np.array_equal(k.value, conv.kernel.numpy())
    drop = [l for l in g._top if l.__class__.__name__ == 'Dropout'][0]
    assert f[drop.name].attrs['weight_names'].shape == (0,)
    g2 = bbh.generator_model(64)
    assert not all(np.array_equal(a, b) for a, b in zip(_weights(g), _weights(g2)))
    g2.load_weights(path)
    assert all(np.array_equal(a, b) for a, b in zip(_weights(g), _weights(g2)))
    with pytest.raises(ValueError):
        bbh.signal_discriminator_model(64).load_weights(path)
    with pytest.raises(ValueError):
        bbh.generator_model(128).load_weights(path)                               # same layers, different shapes


def test_nested_models_are_one_layer_with_keras_weight_order(tmp_path):
    """signal_dis_on_gen.h5 (bbhMahoGANy.py:1375): Sequential(Sequential(G, MyLayer), D).  Keras stores a nested model as ONE
    layer whose weights are trainable_weights + non_trainable_weights; a frozen sub-model's weights all count as non-trainable."""
    from gennet_amd import bbh, keras_io
    n = 64
    ev = np.random.RandomState(1).randn(n, 1).astype(np.float32)
    G, D = bbh.generator_model(n), bbh.signal_discriminator_model(n)
    sub = bbh.data_subtraction_model(ev, n)
    gs = bbh.generator_after_subtracting_noise(G, sub)
    bbh.set_trainable(D, False)
    full = bbh.generator_containing_signal_discriminator(gs, D)
    path = str(tmp_path / 'signal_dis_on_gen.h5')
    full.save_weights(path, True)
    f = h5lite.File(path)
    names = [x.decode() for x in f.attrs['layer_names'].tolist()]
    assert names == [gs.name, D.name]
    wn = [x.decode() for x in f[gs.name].attrs['weight_names'].tolist()]
    tw = [p.name + ':0' for l in G._top for p in l.params]
    ntw = [p.name + ':0' for l in G._top for p in l.buffers]
    assert wn == tw + ntw and len(ntw) == 12                                       # all gammas/betas/kernels first, then the 6 BN moving pairs
    dn = [x.decode() for x in f[D.name].attrs['weight_names'].tolist()]
    assert dn == [p.name + ':0' for l in D._top for p in l.params]                 # frozen: everything non-trainable, layer order
    G2, D2 = bbh.generator_model(n), bbh.signal_discriminator_model(n)
    full2 = bbh.generator_containing_signal_discriminator(bbh.generator_after_subtracting_noise(G2, bbh.data_subtraction_model(ev, n)), D2)
    full2.load_weights(path)
    assert all(np.array_equal(a, b) for a, b in zip(_weights(G), _weights(G2)))
    assert all(np.array_equal(a, b) for a, b in zip(_weights(D), _weights(D2)))
    my = sub._top[0]
    assert my.__class__.__name__ == 'MyLayer' and keras_io.keras_weights(my) == [] and my.name.startswith('my_layer_')


def test_functional_model_layer_order_is_keras_depth_order():
    """Network._init_graph_network: layers by decreasing depth (longest distance to an output), ties by DFS post-order from
    the outputs.  Two branches of different length, as in signal_pe_model (bbhMahoGANy.py:356-404)."""
    from gennet_amd.keras.layers import Input, Conv1D, Flatten, Dense
    from gennet_amd.keras.models import Model
    from gennet_amd import keras_io
    inp = Input(shape=(32, 1), name='in')
    a1 = Conv1D(4, 5, padding='same', name='a1')(inp)
    a2 = Flatten(name='a2')(a1)
    a3 = Dense(1, name='a3')(a2)
    b1 = Conv1D(4, 5, padding='same', name='b1')(inp)
    b2 = Conv1D(4, 5, padding='same', name='b2')(b1)
    b3 = Flatten(name='b3')(b2)
    b4 = Dense(1, name='b4')(b3)
    m = Model(inputs=inp, outputs=[a3, b4], name='two')
    # depths: in 4 | b1 3 | a1 2, b2 2 | a2 1, b3 1 | a3 0, b4 0
    assert [l.name for l in keras_io.top_layers(m)] == ['in', 'b1', 'a1', 'b2', 'a2', 'b3', 'a3', 'b4']
    cfg = keras_io.model_config(m)
    assert cfg['class_name'] == 'Model' and cfg['config']['input_layers'] == [['in', 0, 0]] and cfg['config']['output_layers'] == [['a3', 0, 0], ['b4', 0, 0]]
    assert cfg['config']['layers'][2]['inbound_nodes'] == [[['in', 0, 0, {}]]]
    m2 = keras_io.model_from_config(json.loads(json.dumps(cfg)))
    assert [l.name for l in keras_io.top_layers(m2)] == ['in', 'b1', 'a1', 'b2', 'a2', 'b3', 'a3', 'b4']
    assert m2.output_shape == m.output_shape


def test_save_and_load_model_roundtrip_on_host(tmp_path):
    """model.save / load_model without a device: model_config (keras JSON), training_config, model_weights."""
    from gennet_amd import bbh, keras_io
    from gennet_amd.engine import Adam
    from gennet_amd.keras.models import load_model
    for build, loss in ((lambda: bbh.signal_pe_model(128), 'mean_squared_error'), (lambda: bbh.signal_discriminator_model(64), 'binary_crossentropy'),
                        (lambda: bbh.generator_model(64), 'binary_crossentropy'),
                        # the reference's edit-the-file knobs (bbhMahoGANy.py:228, :424-426): a 10-tap generator, a 4-layer discriminator with
                        # BatchNormalization and MaxPooling2D((2,1)) -- kernel_size / pool_size go through the Keras JSON
                        (lambda: bbh.generator_model(64, 10), 'binary_crossentropy'),
                        (lambda: bbh.signal_discriminator_model(64, num_lays=4, batchnorm=True, maxpool=True), 'binary_crossentropy')):
        m = build()
        m.compile(loss=loss, optimizer=Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])
        path = str(tmp_path / (m.name + '.h5'))
        m.save(path, True)
        f = h5lite.File(path)
        cfg = json.loads(f.attrs['model_config'].decode())
        assert cfg['class_name'] in ('Sequential', 'Model') and cfg['config']['name'] == m.name
        tc = json.loads(f.attrs['training_config'].decode())
        assert tc['loss'] == loss and tc['optimizer_config']['class_name'] == 'Adam' and tc['optimizer_config']['config']['beta_1'] == 0.5
        assert 'optimizer_weights' not in f                                        # never stepped: no optimizer state yet
        m2 = load_model(path)
        assert [(l.__class__.__name__, l.name) for l in m2.layers] == [(l.__class__.__name__, l.name) for l in m.layers]
        assert all(np.array_equal(a, b) for a, b in zip(_weights(m), _weights(m2)))
        assert m2.loss == loss and m2.optimizer.lr == float(np.float32(9e-5)) and m2.metrics == ['accuracy']
        lines1, lines2 = [], []
        m.summary(print_fn=lines1.append); m2.summary(print_fn=lines2.append)
        assert lines1 == lines2
    with pytest.raises(IOError):
        m.save(path, False)


def test_custom_layer_needs_custom_objects(tmp_path):
    from gennet_amd import bbh
    from gennet_amd.keras.models import load_model
    from gennet_amd.layers import MyLayer
    ev = np.arange(64, dtype=np.float32).reshape(64, 1)
    sub = bbh.data_subtraction_model(ev, 64)
    path = str(tmp_path / 'sub.h5')
    sub.save(path, True)
    with pytest.raises(ValueError):
        load_model(path)
    m = load_model(path, custom_objects={'MyLayer': lambda cfg: MyLayer(ev, name=cfg['name'])})
    assert m.output_shape == (None, 64, 2, 1)
