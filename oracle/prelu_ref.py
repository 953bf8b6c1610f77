"""TEST INFRASTRUCTURE ONLY (never imported by gennet_amd/): numpy restatement of keras.layers.PReLU as Keras 2.2.4 computes it
(advanced_activations.py: `pos = K.relu(x); neg = -alpha * K.relu(-x); return pos + neg`, one alpha per feature of a sample,
zeros at initialisation), the layer bbhMahoGANy.py:39 imports and its act = 'prelu' branches use (:237-286, :315-325).
PARITY UNPINNED against real Keras (absent from the image); pinned against torch-CPU autograd in tests/test_prelu.py."""
import numpy as np


def prelu_fwd(x, alpha):
    return np.maximum(x, 0.0) - alpha * np.maximum(-x, 0.0)


def prelu_bwd(dy, x, alpha):
    """-> (dx, dalpha): d relu is 1 for a positive argument and 0 otherwise (TF), so both branches vanish at x == 0."""
    dx = dy * ((x > 0) + alpha * (x < 0))
    dalpha = (dy * np.minimum(x, 0.0)).sum(axis=0)
    return dx, dalpha
