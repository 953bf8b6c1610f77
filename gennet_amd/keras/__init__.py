"""Import-compatible facade for the subset of Keras 2.2.4 that BBH_version/bbhMahoGANy.py imports (bbhMahoGANy.py:32-43,65):

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
"""
from . import backend, layers, models, optimizers  # noqa: F401
