"""Import-compatible facade for the subset of Keras 2.2.4 that BBH_version/bbhMahoGANy.py imports (bbhMahoGANy.py:32-43, :65):

    from gennet_amd.keras.models import Sequential, Model            # was: from keras.models import ...
    from gennet_amd.keras.layers import Dense, Input, Reshape, Dropout
    from gennet_amd.keras.layers.core import Activation, Flatten
    from gennet_amd.keras.layers.normalization import BatchNormalization
    from gennet_amd.keras.layers.convolutional import UpSampling1D, Conv2D, Conv1D
    from gennet_amd.keras.layers.advanced_activations import LeakyReLU, ReLU
    from gennet_amd.keras.engine.topology import Layer
    from gennet_amd.keras.optimizers import Adam
    from gennet_amd.keras import backend as K
Every class executes on the HIP kernel library; see INTEGRATION.md.

The sub-module tree of the import lines is a name space over gennet_amd.engine / gennet_amd.layers and nothing more, so it is built here,
in one table, and registered in sys.modules (importing gennet_amd.keras makes every dotted path above resolvable); only `backend` has code
of its own.  Names the script imports but never instantiates on the BBH path (:33-43) resolve to placeholders whose constructor raises,
instead of silently running something else.
"""
import sys
import types

from .. import engine as _engine
from .. import layers as _layers
from . import backend  # noqa: F401


def _placeholder(name, where):
    def __init__(self, *a, **k):
        raise NotImplementedError('%s is imported by bbhMahoGANy.py (%s) but never used on the BBH hot path; gennet_amd does not '
                                  'provide it' % (name, where))
    return type(name, (object,), {'__init__': __init__, '__doc__': 'placeholder for keras %s (not on the hot path)' % name})


def _unused(where, *names):
    return dict((n, _placeholder(n, 'bbhMahoGANy.py:' + where)) for n in names)


def _pick(mod, *names):
    return dict((n, getattr(mod, n)) for n in names)


_core = _pick(_layers, 'Activation', 'Dense', 'Dropout', 'Flatten', 'Reshape')
_norm = _pick(_layers, 'BatchNormalization')
_conv = dict(_pick(_layers, 'Conv1D', 'Conv2D', 'UpSampling1D', 'MaxPooling2D'),
             **_unused('37-38', 'UpSampling2D', 'Conv2DTranspose', 'AveragePooling1D', 'MaxPooling1D'))
_act = dict(_pick(_layers, 'LeakyReLU', 'PReLU', 'ReLU'), **_unused('39', 'ThresholdedReLU'))
_top = dict(_pick(_engine, 'Input'), **_pick(_layers, 'MyLayer'))
_top.update(_unused('33-34', 'GlobalAveragePooling1D', 'AlphaDropout', 'GaussianDropout', 'GaussianNoise'))
for _d in (_core, _norm, _conv, _act):
    _top.update((k, v) for k, v in _d.items() if k in _layers.__dict__)

_TREE = {
    'models': _pick(_engine, 'Model', 'Sequential', 'load_model', 'model_from_json'),
    'optimizers': dict(_pick(_engine, 'Adam'), **_unused('43', 'RMSprop', 'Adagrad', 'Adadelta', 'Adamax', 'Nadam')),
    'engine': {},
    'engine.topology': _pick(_engine, 'Layer'),
    'layers': _top,
    'layers.core': _core,
    'layers.normalization': _norm,
    'layers.convolutional': _conv,
    'layers.advanced_activations': _act,
}


def _register():
    for dotted in sorted(_TREE):            # parents sort before their children
        m = types.ModuleType(__name__ + '.' + dotted, 'gennet_amd.keras facade: see gennet_amd/keras/__init__.py')
        m.__dict__.update(_TREE[dotted])
        m.__package__ = __name__ + '.' + dotted if any(k.startswith(dotted + '.') for k in _TREE) else (__name__ + '.' + dotted).rpartition('.')[0]
        if m.__package__ == m.__name__:
            m.__path__ = []                 # a package: `from gennet_amd.keras.layers.core import ...` walks through it
        sys.modules[m.__name__] = m
        parent, _, leaf = m.__name__.rpartition('.')
        setattr(sys.modules[parent], leaf, m)


_register()
