#!/usr/bin/env python
"""Writes tests/golden/keras_h5_golden.json: names, shapes, dtypes and CRC32s of everything in the four real Keras 2.1.6
files the reference ships (2_model_version/weight_version/*.hdf5), read with gennet_amd/h5lite.py.  The fixture holds no
weights and no reference text; model_config / training_config -- the JSON Keras ITSELF wrote into the two full-model files, i.e. Keras'
own record of its layer and optimizer defaults (BatchNormalization epsilon / momentum / initialisers, LeakyReLU alpha, the glorot
VarianceScaling spec, Adam beta_2 / epsilon stored as float32 variables) -- are recorded in full as parsed JSON ('model_config',
'training') next to the structure summary ('model').  Run where /root/reference exists.
"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/2_model_version/weight_version'
spec = importlib.util.spec_from_file_location('h5lite', os.path.join(HERE, '..', '..', 'gennet_amd', 'h5lite.py'))
h5lite = importlib.util.module_from_spec(spec)
spec.loader.exec_module(h5lite)

out = {}
for fn in ('d_model.hdf5', 'g_model.hdf5', 'best_d_weights.hdf5', 'best_g_weights.hdf5'):
    path = os.path.join(REF, fn)
    d = h5lite.dump_structure(path)
    f = h5lite.File(path)
    d['file_bytes'] = os.path.getsize(path)
    mc = f.attrs.get('model_config')
    if mc is not None:
        cfg = json.loads(mc)
        d['model'] = {'class_name': cfg['class_name'], 'name': cfg['config']['name'],
                      'layers': [[l['class_name'], l['name']] for l in cfg['config']['layers']]}
        d['model_config'] = cfg
        d['training'] = json.loads(f.attrs['training_config'])
    out[fn] = d
json.dump(out, open(os.path.join(HERE, 'keras_h5_golden.json'), 'w'), indent=1, sort_keys=True)
print('wrote', {k: (len(v['datasets']), len(v['groups'])) for k, v in out.items()})
