"""Keras 2.x HDF5 persistence for gennet_amd models: the byte layout `model.save`,
`save_weights`, `load_weights` and `keras.models.load_model` use in bbhMahoGANy.py:1135-1142, :1171-1173, :1372-1375, so
that `.h5` files interchange with real Keras.  HDF5 itself comes from gennet_amd/h5lite.py (h5py is not in the image).

Layout (keras/engine/saving.py of Keras 2.2.4, restated; confirmed on the real Keras files the reference ships under
2_model_version/weight_version/):
  weights file : root attrs layer_names [S], backend, keras_version; one group per layer of `model.layers` with attr
                 weight_names [S] and one dataset per weight named by the weight ('dense_1/kernel:0' -> nested group);
  model file   : root attrs keras_version, backend, model_config (JSON), training_config (JSON);
                 group model_weights (as above); group optimizer_weights (attr weight_names + datasets:
                 iterations, then Adam's m per trainable weight, v per trainable weight, then (1,)-shaped vhat stubs).
`model.layers` order: Sequential = layers as added (no InputLayer); functional Model = Keras' depth order (Network
._init_graph_network): depth = longest distance to an output, ties by post-order index of a DFS from the outputs.
A nested model is ONE layer whose weights are trainable_weights + non_trainable_weights (Keras' Layer.weights).
"""
import json

import numpy as np

from . import h5lite
from .engine import Adam, Input, Layer, Model, Sequential, to_snake_case  # noqa: F401

KERAS_VERSION = '2.2.4'
BACKEND = 'tensorflow'


# ---------------------------------------------------------------------------------------------------------------------
# model.layers and layer.weights in Keras' order
# ---------------------------------------------------------------------------------------------------------------------
class InputLayer(object):
    """Stand-in for keras.layers.InputLayer in `model.layers` / model_config of functional models (it has no weights)."""

    def __init__(self, sym):
        self.sym = sym
        self.name = sym.name
        self.trainable = False

    weights = ()


def _sym_graph(model):
    """Post-order DFS over the symbolic tensors from the outputs: [(tensor, layer-or-InputLayer)], each layer call once."""
    order, seen = [], set()

    def visit(t):
        if id(t) in seen:
            return
        seen.add(id(t))
        for s in t.inbound:
            visit(s)
        order.append(t)
    for t in model._sym_outputs:
        visit(t)
    return order


def top_layers(model):
    """`model.layers` as Keras orders them."""
    if isinstance(model, Sequential):
        return list(model._top)
    if getattr(model, '_sym_outputs', None) is None:
        return list(model.layers)
    order = _sym_graph(model)                       # nodes_in_decreasing_depth of Keras' build_map
    index = {id(t): i for i, t in enumerate(order)}
    depth = {}
    for t in reversed(order):
        d = depth.setdefault(id(t), 0)
        for s in t.inbound:
            depth[id(s)] = max(depth.get(id(s), 0), d + 1)
    # a layer called once = one node; depth of the layer = depth of its node
    items = sorted(order, key=lambda t: (-depth[id(t)], index[id(t)]))
    return [InputLayer(t) if t.layer is None else t.layer for t in items]


def _tw(layer):
    if isinstance(layer, InputLayer) or not layer.trainable:
        return []
    if isinstance(layer, Model):
        out = []
        for l in top_layers(layer):
            out += _tw(l)
        return out
    return list(layer.params)


def _ntw(layer):
    if isinstance(layer, InputLayer):
        return []
    if isinstance(layer, Model):
        w = []
        for l in top_layers(layer):
            w += _ntw(l)
        if not layer.trainable:
            t = []
            for l in top_layers(layer):
                t += _tw(l)
            return t + w
        return w
    return list(layer.buffers) if layer.trainable else list(layer.params) + list(layer.buffers)


def keras_weights(layer):
    """`layer.weights` of Keras: trainable_weights + non_trainable_weights (differs from per-layer order for nested models)."""
    return _tw(layer) + _ntw(layer)


def _wname(p):
    return p.name + ':0'


# ---------------------------------------------------------------------------------------------------------------------
# weights <-> HDF5 groups
# ---------------------------------------------------------------------------------------------------------------------
def save_weights_to_group(g, layers):
    g.attrs['layer_names'] = np.array([l.name.encode('utf-8') for l in layers], dtype='S') if layers else np.zeros((0,), 'S1')
    g.attrs['backend'] = BACKEND.encode('utf-8')
    g.attrs['keras_version'] = KERAS_VERSION.encode('utf-8')
    for l in layers:
        lg = g.create_group(l.name)
        ws = keras_weights(l)
        names = [_wname(p) for p in ws]
        if len(set(names)) != len(names):
            raise ValueError('layer %s has duplicate weight names' % l.name)
        lg.attrs['weight_names'] = np.array([n.encode('utf-8') for n in names], dtype='S') if names else np.zeros((0,), np.float64)
        for p, n in zip(ws, names):
            lg.create_dataset(n, np.ascontiguousarray(p.numpy(), dtype=np.float32))


def _decode_list(a):
    if a is None:
        return []
    return [x.decode('utf-8') if isinstance(x, bytes) else str(x) for x in np.asarray(a).ravel().tolist()]


def load_weights_from_group(g, layers):
    """keras.engine.saving.load_weights_from_hdf5_group: match the layers that HAVE weights, in order; check counts and shapes."""
    layer_names = _decode_list(g.attrs.get('layer_names'))
    file_layers = []
    for name in layer_names:
        wn = _decode_list(g[name].attrs.get('weight_names'))
        if wn:
            file_layers.append((name, wn))
    model_layers = [l for l in layers if keras_weights(l)]
    if len(file_layers) != len(model_layers):
        raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.' % (len(file_layers), len(model_layers)))
    todo = []
    for (name, wn), l in zip(file_layers, model_layers):
        ws = keras_weights(l)
        if len(wn) != len(ws):
            raise ValueError('Layer %s (in the file: %s) expects %d weight(s), but the saved weights have %d element(s).' % (l.name, name, len(ws), len(wn)))
        for p, n in zip(ws, wn):
            v = g[name][n].value
            if tuple(v.shape) != tuple(p.shape):
                raise ValueError('Layer %s: weight %s has shape %s in the file, %s in the model' % (l.name, n, tuple(v.shape), tuple(p.shape)))
            todo.append((p, v))
    for p, v in todo:        # nothing is assigned unless everything matched
        p.assign(np.asarray(v, np.float32))


# ---------------------------------------------------------------------------------------------------------------------
# layer / model configs (keras get_config / from_config)
# ---------------------------------------------------------------------------------------------------------------------
_GLOROT = {'class_name': 'VarianceScaling', 'config': {'scale': 1.0, 'mode': 'fan_avg', 'distribution': 'uniform', 'seed': None}}
_ZEROS = {'class_name': 'Zeros', 'config': {}}
_ONES = {'class_name': 'Ones', 'config': {}}
_ACT_OF_SPEC = {'relu': 'relu', 'tanh': 'tanh', 'sigmoid': 'sigmoid', 'linear': 'linear'}


def _kernel_part():
    return {'use_bias': True, 'kernel_initializer': _GLOROT, 'bias_initializer': _ZEROS, 'kernel_regularizer': None, 'bias_regularizer': None,
            'activity_regularizer': None, 'kernel_constraint': None, 'bias_constraint': None}


def layer_config(layer):
    from . import layers as L
    cfg = {'name': layer.name, 'trainable': bool(layer.trainable)}
    if layer.input_shape_arg is not None:
        cfg['batch_input_shape'] = [None] + list(layer.input_shape_arg)
        cfg['dtype'] = 'float32'
    if isinstance(layer, L.Dense):
        cfg.update(units=layer.units, activation=_ACT_OF_SPEC[layer.activation[0]], **_kernel_part())
    elif isinstance(layer, L.Conv1D):
        cfg.update(filters=layer.filters, kernel_size=[layer.k], strides=[layer.stride], padding=layer.padding, data_format='channels_last',
                   dilation_rate=[1], activation=_ACT_OF_SPEC[layer.activation[0]], **_kernel_part())
    elif isinstance(layer, L.Conv2D):
        cfg.update(filters=layer.filters, kernel_size=[layer.kh, layer.kw], strides=[layer.sh, layer.sw], padding=layer.padding,
                   data_format='channels_last', dilation_rate=[1, 1], activation=_ACT_OF_SPEC[layer.activation[0]], **_kernel_part())
    elif isinstance(layer, L.BatchNormalization):
        cfg.update(axis=-1, momentum=layer.momentum, epsilon=layer.epsilon, center=True, scale=True, beta_initializer=_ZEROS, gamma_initializer=_ONES,
                   moving_mean_initializer=_ZEROS, moving_variance_initializer=_ONES, beta_regularizer=None, gamma_regularizer=None,
                   beta_constraint=None, gamma_constraint=None)
    elif isinstance(layer, L.PReLU):
        cfg.update(alpha_initializer=_ZEROS, alpha_regularizer=None, alpha_constraint=None, shared_axes=None)
    elif isinstance(layer, L.LeakyReLU):
        cfg.update(alpha=layer.act_spec[1])
    elif isinstance(layer, L.ReLU):
        cfg.update(max_value=(layer.act_spec[1] if layer.act_spec[0] == 'relu_max' else None), negative_slope=0.0, threshold=0.0)
    elif isinstance(layer, L.Activation):
        cfg.update(activation=_ACT_OF_SPEC[layer.act_spec[0]])
    elif isinstance(layer, L.Dropout):
        cfg.update(rate=layer.rate, noise_shape=None, seed=None)
    elif isinstance(layer, L.Reshape):
        cfg.update(target_shape=list(getattr(layer, 'target_shape_arg', layer.target_shape)))
    elif isinstance(layer, L.Flatten):
        cfg.update(data_format='channels_last')
    elif isinstance(layer, L.UpSampling1D):
        cfg.update(size=2)
    elif isinstance(layer, L.MaxPooling2D):
        cfg.update(pool_size=[2, 1], padding='valid', strides=[2, 1], data_format='channels_last')
    # custom layers (MyLayer, bbhMahoGANy.py:164-188, defines no get_config): base config only, like Keras writes for them
    return cfg


def model_config(model):
    """{'class_name': 'Sequential' | 'Model', 'config': ...} in the Keras 2.2.4 form."""
    def entry(l):
        if isinstance(l, Model):
            return model_config(l)
        return {'class_name': l.__class__.__name__, 'config': layer_config(l)}

    if isinstance(model, Sequential):
        return {'class_name': 'Sequential', 'config': {'name': model.name, 'layers': [entry(l) for l in model._top]}}
    if getattr(model, '_sym_outputs', None) is None:
        raise ValueError('model %s was not built through Sequential.add or Model(inputs=, outputs=)' % model.name)
    order = _sym_graph(model)
    name_of = {id(t): (t.name if t.layer is None else t.layer.name) for t in order}
    layers = []
    for l in top_layers(model):
        if isinstance(l, InputLayer):
            layers.append({'name': l.name, 'class_name': 'InputLayer', 'inbound_nodes': [],
                           'config': {'batch_input_shape': [None] + list(l.sym.shape), 'dtype': 'float32', 'sparse': False, 'name': l.name}})
            continue
        t = [s for s in order if s.layer is l][0]
        e = entry(l)
        e['name'] = l.name
        e['inbound_nodes'] = [[[name_of[id(s)], 0, 0, {}] for s in t.inbound]]
        layers.append(e)
    return {'class_name': 'Model', 'config': {'name': model.name, 'layers': layers,
                                              'input_layers': [[t.name, 0, 0] for t in model._sym_inputs],
                                              'output_layers': [[name_of[id(t)], 0, 0] for t in model._sym_outputs]}}


def _layer_from_config(class_name, cfg, custom_objects):
    from . import layers as L
    custom_objects = custom_objects or {}
    kw = {'name': cfg.get('name'), 'trainable': cfg.get('trainable', True)}
    if cfg.get('batch_input_shape') is not None:
        kw['input_shape'] = tuple(cfg['batch_input_shape'][1:])
    if class_name in custom_objects:
        obj = custom_objects[class_name]
        if isinstance(obj, Layer):
            layer = obj
        else:
            layer = obj(**{k: v for k, v in cfg.items() if k not in ('batch_input_shape', 'dtype')}) if isinstance(obj, type) else obj(cfg)
        if layer.input_shape_arg is None and 'input_shape' in kw:
            layer.input_shape_arg = kw['input_shape']
        return layer

    def init_ok(c):
        if c is None or c == 'glorot_uniform':
            return
        if isinstance(c, dict) and (c.get('class_name') == 'GlorotUniform' or (c.get('class_name') == 'VarianceScaling' and c['config'].get('mode') == 'fan_avg'
                                                                                and c['config'].get('distribution') == 'uniform')):
            return
        raise NotImplementedError('kernel_initializer %r' % (c,))

    if class_name == 'Dense':
        init_ok(cfg.get('kernel_initializer'))
        return L.Dense(cfg['units'], activation=cfg.get('activation'), use_bias=cfg.get('use_bias', True), **kw)
    if class_name == 'Conv1D':
        init_ok(cfg.get('kernel_initializer'))
        return L.Conv1D(cfg['filters'], cfg['kernel_size'], strides=cfg.get('strides', 1), padding=cfg.get('padding', 'valid'), activation=cfg.get('activation'),
                        use_bias=cfg.get('use_bias', True), **kw)
    if class_name == 'Conv2D':
        init_ok(cfg.get('kernel_initializer'))
        return L.Conv2D(cfg['filters'], tuple(cfg['kernel_size']), strides=tuple(cfg.get('strides', (1, 1))), padding=cfg.get('padding', 'valid'),
                        activation=cfg.get('activation'), use_bias=cfg.get('use_bias', True), **kw)
    if class_name == 'BatchNormalization':
        return L.BatchNormalization(axis=cfg.get('axis', -1), momentum=cfg.get('momentum', 0.99), epsilon=cfg.get('epsilon', 1e-3), **kw)
    if class_name == 'PReLU':
        return L.PReLU(alpha_initializer=cfg.get('alpha_initializer', 'zeros'), shared_axes=cfg.get('shared_axes'), **kw)
    if class_name == 'LeakyReLU':
        return L.LeakyReLU(alpha=cfg.get('alpha', 0.3), **kw)
    if class_name == 'ReLU':
        return L.ReLU(max_value=cfg.get('max_value'), negative_slope=cfg.get('negative_slope', 0.0), threshold=cfg.get('threshold', 0.0), **kw)
    if class_name == 'Activation':
        return L.Activation(cfg['activation'], **kw)
    if class_name == 'Dropout':
        return L.Dropout(cfg['rate'], **kw)
    if class_name == 'Reshape':
        return L.Reshape(tuple(cfg['target_shape']), **kw)
    if class_name == 'Flatten':
        return L.Flatten(**kw)
    if class_name == 'UpSampling1D':
        return L.UpSampling1D(size=cfg.get('size', 2), **kw)
    if class_name == 'MaxPooling2D':
        return L.MaxPooling2D(pool_size=tuple(cfg.get('pool_size', (2, 2))), strides=cfg.get('strides'), padding=cfg.get('padding', 'valid'), **kw)
    raise ValueError('Unknown layer: %s (pass it through custom_objects={%r: ...})' % (class_name, class_name))


def model_from_config(config, custom_objects=None):
    """keras.models.model_from_config for Sequential / functional Model configs of Keras 2.0 - 2.2 (both Sequential forms)."""
    cls, cfg = config['class_name'], config['config']
    if cls == 'Sequential':
        entries = cfg if isinstance(cfg, list) else cfg['layers']
        m = Sequential(name=None if isinstance(cfg, list) else cfg.get('name'))
        for e in entries:
            if e['class_name'] in ('Sequential', 'Model') and e['class_name'] not in (custom_objects or {}):
                m.add(model_from_config(e, custom_objects))
            elif e['class_name'] == 'InputLayer':
                continue
            else:
                m.add(_layer_from_config(e['class_name'], e['config'], custom_objects))
        return m
    if cls == 'Model':
        tensors = {}
        pending = list(cfg['layers'])
        while pending:
            progressed = False
            for e in list(pending):
                if e['class_name'] == 'InputLayer':
                    tensors[e['name']] = Input(shape=tuple(e['config']['batch_input_shape'][1:]), name=e['name'])
                elif len(e['inbound_nodes']) != 1:
                    raise NotImplementedError('layer %s is called %d times; shared layers are not supported' % (e['name'], len(e['inbound_nodes'])))
                elif all(ref[0] in tensors for ref in e['inbound_nodes'][0]):
                    ins = [tensors[ref[0]] for ref in e['inbound_nodes'][0]]
                    if len(ins) != 1:
                        raise NotImplementedError('layer %s has %d inputs; merge layers are not on the BBH path' % (e['name'], len(ins)))
                    layer = model_from_config(e, custom_objects) if e['class_name'] in ('Sequential', 'Model') else _layer_from_config(e['class_name'], e['config'], custom_objects)
                    tensors[e['name']] = layer(ins[0])
                else:
                    continue
                pending.remove(e)
                progressed = True
            if not progressed:
                raise ValueError('model config has a cycle or a dangling inbound reference: %s' % [e['name'] for e in pending])
        ins = [tensors[r[0]] for r in cfg['input_layers']]
        outs = [tensors[r[0]] for r in cfg['output_layers']]
        return Model(inputs=ins if len(ins) > 1 else ins[0], outputs=outs if len(outs) > 1 else outs[0], name=cfg.get('name'))
    raise ValueError('cannot build a model from class_name %r' % cls)


# ---------------------------------------------------------------------------------------------------------------------
# files
# ---------------------------------------------------------------------------------------------------------------------
def is_hdf5(path):
    with open(path, 'rb') as fh:
        return fh.read(8) == h5lite.SIGNATURE


STATE_GROUP = 'gennet_amd_state'      # private root group (not a keras layer group: keras' loaders only visit attrs['layer_names'])


def _bn_layers(model):
    return [l for l in model.layers if getattr(l, 'is_batchnorm', False)]


def save_private_state(root, model):
    """BatchNormalization's zero-debias shadow variables (layers.BatchNormalization): TF graph variables that keras itself never saves.
    Written to <STATE_GROUP>/<layer>/<training model>/{biased_mean, biased_var, local_step} so that a run resumed from a file written
    HERE continues the same moving averages; a file from real keras has no such group and the state restarts at zero, as in keras."""
    bns = [l for l in _bn_layers(model) if l.zero_debias]
    if not bns:
        return
    g = root.create_group(STATE_GROUP)
    for l in bns:
        lg = g.create_group(l.name)
        for site, (bm, bv, step) in sorted(l.zero_debias.items()):
            sg = lg.create_group(site)
            sg.create_dataset('biased_mean', np.ascontiguousarray(bm.cpu().numpy(), np.float32))
            sg.create_dataset('biased_var', np.ascontiguousarray(bv.cpu().numpy(), np.float32))
            sg.create_dataset('local_step', np.asarray([step], np.int64))


def load_private_state(f, model):
    from .engine import to_device
    for l in _bn_layers(model):
        l.zero_debias = {}                       # keras semantics for files without the private group: shadow variables restart at zero
    if STATE_GROUP not in f:
        return
    g = f[STATE_GROUP]
    for l in _bn_layers(model):
        if l.name not in g:
            continue
        for site, sg in g[l.name].items():
            l.zero_debias[site] = [to_device(sg['biased_mean'].value), to_device(sg['biased_var'].value), int(np.asarray(sg['local_step'].value).reshape(-1)[0])]


def _write(w, path, writer):
    """The tree `w` holds host copies only: with a hostio.BackgroundWriter the serialisation + file write leaves the caller's thread."""
    if writer is None:
        w.save(path)
    else:
        writer.submit(w.save, path)


def save_weights(model, path, writer=None):
    w = h5lite.Writer()
    save_weights_to_group(w.root, top_layers(model))
    save_private_state(w.root, model)
    _write(w, path, writer)


def load_weights(model, path):
    f = h5lite.File(path)
    g = f['model_weights'] if 'layer_names' not in f.attrs and 'model_weights' in f else f
    load_weights_from_group(g, top_layers(model))
    load_private_state(f, model)


def _loss_json(loss):
    if isinstance(loss, (list, tuple)):
        return [_loss_json(l) for l in loss]
    return loss if isinstance(loss, str) else getattr(loss, '__name__', str(loss))


def save_model(model, path, include_optimizer=True, writer=None):
    w = h5lite.Writer()
    w.root.attrs['keras_version'] = KERAS_VERSION.encode('utf-8')
    w.root.attrs['backend'] = BACKEND.encode('utf-8')
    w.root.attrs['model_config'] = json.dumps(model_config(model)).encode('utf-8')
    save_weights_to_group(w.root.create_group('model_weights'), top_layers(model))
    opt = model.optimizer
    if include_optimizer and opt is not None:
        w.root.attrs['training_config'] = json.dumps({
            'optimizer_config': {'class_name': 'Adam', 'config': {'lr': opt.lr, 'beta_1': opt.beta_1, 'beta_2': opt.beta_2, 'decay': 0.0,
                                                                  'epsilon': opt.epsilon, 'amsgrad': False}},
            'loss': _loss_json(model.loss), 'metrics': list(model.metrics or []), 'sample_weight_mode': None, 'loss_weights': None}).encode('utf-8')
        if opt.state is not None and model._train_params:
            og = w.root.create_group('optimizer_weights')
            mv = opt.param_moments(model._keras_train_order())      # keras: model.trainable_weights order
            names = ['Adam/iterations:0']
            og.create_dataset(names[0], np.asarray(opt.iterations, np.int64))
            k = 0
            for arrs in ([m for m, _ in mv], [v for _, v in mv], [np.zeros((1,), np.float32) for _ in mv]):
                for a in arrs:
                    n = 'training/Adam/Variable%s:0' % ('' if k == 0 else '_%d' % k)
                    og.create_dataset(n, np.asarray(a, np.float32))
                    names.append(n)
                    k += 1
            og.attrs['weight_names'] = np.array([n.encode('utf-8') for n in names], dtype='S')
    save_private_state(w.root, model)
    _write(w, path, writer)


def load_model(path, custom_objects=None, compile=True):
    f = h5lite.File(path)
    mc = f.attrs.get('model_config')
    if mc is None:
        raise ValueError('No model found in config file.')
    model = model_from_config(json.loads(mc.decode('utf-8') if isinstance(mc, bytes) else mc), custom_objects)
    load_weights_from_group(f['model_weights'], top_layers(model))
    load_private_state(f, model)
    tc = f.attrs.get('training_config')
    if compile and tc is not None:
        tc = json.loads(tc.decode('utf-8') if isinstance(tc, bytes) else tc)
        oc = tc['optimizer_config']
        if oc['class_name'] != 'Adam':
            raise NotImplementedError('optimizer %s' % oc['class_name'])
        c = oc['config']
        model.compile(loss=tc['loss'], optimizer=Adam(lr=c['lr'], beta_1=c['beta_1'], beta_2=c['beta_2'], epsilon=c.get('epsilon'), decay=c.get('decay', 0.0)),
                      metrics=tc.get('metrics') or [])
        if 'optimizer_weights' in f:
            og = f['optimizer_weights']
            names = _decode_list(og.attrs.get('weight_names'))
            vals = [og[n].value for n in names]
            model._pending_optimizer_weights = vals      # applied when the optimizer state is bound (first device use)
    return model
