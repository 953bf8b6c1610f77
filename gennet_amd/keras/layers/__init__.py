from ...engine import Input  # noqa: F401
from ...layers import (Activation, BatchNormalization, Conv1D, Conv2D, Dense, Dropout, Flatten, LeakyReLU, MyLayer, PReLU, ReLU,  # noqa: F401
                       Reshape, UpSampling1D)
from .._unused import AlphaDropout, GaussianDropout, GaussianNoise, GlobalAveragePooling1D  # noqa: F401
from . import advanced_activations, convolutional, core, normalization  # noqa: F401
