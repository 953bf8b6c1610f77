"""The `keras.backend` names bbhMahoGANy.py touches (:81, :162, :172, :184).  The reference uses K.constant / K.stack only inside
MyLayer.call, which gennet_amd provides as the built-in layers.MyLayer (one fused HIP kernel), and K.sum / K.square only in
chisquare_Loss, which is dead code under the default chi_loss = False (:97); they are therefore not graph-building ops here."""


def epsilon():
    return 1e-7


def floatx():
    return 'float32'


def set_session(session):
    """No-op: there is no TF session; device memory comes from the HIP allocator of the process (one process per GPU)."""
    return None


def constant(value, dtype=None, shape=None, name=None):
    import numpy as np
    return np.asarray(value, np.float32)


def _unsupported(name):
    def fn(*a, **k):
        raise NotImplementedError('K.%s as a graph op is not provided: use gennet_amd.layers.MyLayer for the subtract/stack layer '
                                  '(bbhMahoGANy.py:164-188); custom losses are not on the hot path' % name)
    return fn


stack, sum, square = _unsupported('stack'), _unsupported('sum'), _unsupported('square')
