"""The `keras.backend` names bbhMahoGANy.py touches (:81, :162, :172, :184): K.set_session, K.constant, K.stack, K.sum, K.square.

There is no tensor runtime behind these names.  A user-defined `Layer.call` (the script's own MyLayer, :164-188) or a callable loss
(chisquare_Loss, :146-162) is executed ONCE on symbolic operands when the graph is built; the small expression tree it produces is
matched against the forms the HIP library implements and lowered to one fused kernel:

  * layer:  K.stack([a0*x + b0, a1*x + b1], axis=2) on x of shape (batch, n, 1), every entry affine in x with constant vectors
            (MyLayer: diff = self.const - x; K.stack([x, diff], axis=2))            -> gn_affine_stack_fwd / _bwd
  * loss:   K.sum | K.mean (K.square(yTrue - yPred) [* or / scalar], axis=-1)         -> gn_mse_loss with a scale factor

Anything else raises NotImplementedError naming the expression: nothing is ever evaluated on the host, and nothing silently runs
something different from what the user wrote.
"""
import numpy as np


def epsilon():
    return 1e-7


def floatx():
    return 'float32'


def set_session(session):
    """No-op: there is no TF session; device memory comes from the HIP allocator of the process (one process per GPU)."""
    return None


class Sym(object):
    """Node of a traced expression.  op: 'input' | 'const' | 'add' | 'sub' | 'mul' | 'div' | 'neg' | 'square' | 'stack' | 'sum' | 'mean'."""

    def __init__(self, op, args=(), value=None, name=None, axis=None):
        self.op, self.args, self.value, self.name, self.axis = op, tuple(args), value, name, axis

    # arithmetic builds the tree; python / numpy scalars and arrays become constants
    def __add__(self, o): return Sym('add', (self, _wrap(o)))
    def __radd__(self, o): return Sym('add', (_wrap(o), self))
    def __sub__(self, o): return Sym('sub', (self, _wrap(o)))
    def __rsub__(self, o): return Sym('sub', (_wrap(o), self))
    def __mul__(self, o): return Sym('mul', (self, _wrap(o)))
    def __rmul__(self, o): return Sym('mul', (_wrap(o), self))
    def __truediv__(self, o): return Sym('div', (self, _wrap(o)))
    def __rtruediv__(self, o): return Sym('div', (_wrap(o), self))
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __neg__(self): return Sym('neg', (self,))
    def __pow__(self, o):
        if isinstance(o, (int, float)) and o == 2:
            return Sym('square', (self,))
        raise NotImplementedError('K tensor ** %r' % (o,))
    __array_priority__ = 1000            # numpy arrays defer to these operators instead of broadcasting over an object array

    def __array__(self, dtype=None, copy=None):
        if self.op != 'const':
            raise TypeError('a symbolic keras tensor has no value')
        return np.asarray(self.value, dtype)

    def __repr__(self):
        if self.op == 'input':
            return '<%s>' % self.name
        if self.op == 'const':
            v = np.asarray(self.value)
            return 'const%s' % (tuple(v.shape),) if v.size > 1 else repr(float(v.reshape(-1)[0]))
        extra = ', axis=%r' % (self.axis,) if self.axis is not None else ''
        return '%s(%s%s)' % (self.op, ', '.join(repr(a) for a in self.args), extra)


def _wrap(v):
    if isinstance(v, Sym):
        return v
    return Sym('const', value=np.asarray(v, np.float64))


def constant(value, dtype=None, shape=None, name=None):
    """K.constant: a constant operand of a traced expression (np.asarray(K.constant(v)) gives the values back)."""
    v = np.asarray(value, np.float64)
    if shape is not None:
        v = np.broadcast_to(v, shape).copy()
    return Sym('const', value=v, name=name)


def stack(x, axis=0):
    return Sym('stack', [_wrap(t) for t in x], axis=axis)


def sum(x, axis=None, keepdims=False):      # noqa: A001 (keras' name)
    return Sym('sum', (_wrap(x),), axis=axis)


def mean(x, axis=None, keepdims=False):
    return Sym('mean', (_wrap(x),), axis=axis)


def square(x):
    return Sym('square', (_wrap(x),))


# ---------------------------------------------------------------------------------------------------------------------
# lowering
# ---------------------------------------------------------------------------------------------------------------------
def _scalar(e):
    """The python float of a constant-scalar expression, else None."""
    if e.op == 'const' and np.asarray(e.value).size == 1:
        return float(np.asarray(e.value).reshape(-1)[0])
    if e.op == 'neg':
        s = _scalar(e.args[0])
        return None if s is None else -s
    if e.op in ('mul', 'div', 'add', 'sub'):
        a, b = _scalar(e.args[0]), _scalar(e.args[1])
        if a is None or b is None:
            return None
        return {'mul': a * b, 'div': a / b, 'add': a + b, 'sub': a - b}[e.op]
    return None


def affine_in(e, var):
    """e == alpha * var + beta with a python-float alpha and a constant beta (numpy array or 0.0): returns (alpha, beta), else raises."""
    if e is var:
        return 1.0, 0.0
    if e.op == 'const':
        return 0.0, np.asarray(e.value, np.float64)
    if e.op == 'neg':
        a, b = affine_in(e.args[0], var)
        return -a, -b if isinstance(b, np.ndarray) else -b
    if e.op in ('add', 'sub'):
        a0, b0 = affine_in(e.args[0], var)
        a1, b1 = affine_in(e.args[1], var)
        sg = 1.0 if e.op == 'add' else -1.0
        return a0 + sg * a1, b0 + sg * b1
    if e.op in ('mul', 'div'):
        s1 = _scalar(e.args[1])
        if s1 is not None:
            a, b = affine_in(e.args[0], var)
            return (a * s1, b * s1) if e.op == 'mul' else (a / s1, b / s1)
        s0 = _scalar(e.args[0])
        if s0 is not None and e.op == 'mul':
            a, b = affine_in(e.args[1], var)
            return a * s0, b * s0
    raise NotImplementedError('expression %r is not affine in %r with constant coefficients' % (e, var))


class AffineStack(object):
    """Lowered form of K.stack([a0*x + b0, a1*x + b1], axis=2)."""

    def __init__(self, a0, b0, a1, b1, n):
        def vec(b):
            if not isinstance(b, np.ndarray) and b == 0.0:
                return None
            v = np.asarray(b, np.float32).reshape(-1)
            if v.size == 1:
                v = np.full(n, v[0], np.float32)
            if v.size != n:
                raise NotImplementedError('constant of %d values in a layer over %d samples' % (v.size, n))
            return np.ascontiguousarray(v)
        self.a0, self.a1 = float(a0), float(a1)
        self.b0_host, self.b1_host = vec(b0), vec(b1)
        self._dev = None

    def betas(self):
        if self._dev is None:
            from ..engine import to_device
            self._dev = tuple(None if b is None else to_device(b) for b in (self.b0_host, self.b1_host))
        return self._dev


def lower_layer_call(layer, input_shape):
    """Trace layer.call on a symbolic input of shape (batch,) + input_shape and lower the result; see the module docstring."""
    if len(input_shape) != 2 or input_shape[1] != 1:
        raise NotImplementedError('user-defined Layer.call is lowered for inputs of shape (batch, n, 1); got %r' % ((None,) + tuple(input_shape),))
    x = Sym('input', name='x')
    out = layer.call(x)
    if not isinstance(out, Sym) or out.op != 'stack' or out.axis != 2 or len(out.args) != 2:
        raise NotImplementedError('user-defined Layer.call must return K.stack([e0, e1], axis=2) with e0, e1 affine in the input; got %r' % (out,))
    (a0, b0), (a1, b1) = affine_in(out.args[0], x), affine_in(out.args[1], x)
    return AffineStack(a0, b0, a1, b1, int(input_shape[0]))


def lower_loss(fn):
    """Trace a callable loss(yTrue, yPred): returns ('mean_squared_error', scale) for sum|mean(square(yTrue - yPred) [*|/ scalar], axis=-1)
    on outputs with a last dimension of 1 (every output of the BBH models); else raises."""
    yt, yp = Sym('input', name='yTrue'), Sym('input', name='yPred')
    e = fn(yt, yp)
    if not isinstance(e, Sym) or e.op not in ('sum', 'mean') or e.axis not in (-1, 1):
        raise NotImplementedError('callable loss must be K.sum / K.mean(..., axis=-1) of a squared error; got %r' % (e,))
    scale, body = 1.0, e.args[0]
    while body.op in ('mul', 'div'):
        s1 = _scalar(body.args[1])
        s0 = _scalar(body.args[0]) if body.op == 'mul' else None
        if s1 is not None:
            scale, body = (scale * s1 if body.op == 'mul' else scale / s1), body.args[0]
        elif s0 is not None:
            scale, body = scale * s0, body.args[1]
        else:
            break
    if body.op != 'square' or body.args[0].op != 'sub' or set(id(a) for a in body.args[0].args) != {id(yt), id(yp)}:
        raise NotImplementedError('callable loss must be a (scaled) squared error of yTrue - yPred; got %r' % (e,))
    return 'mean_squared_error', float(scale)
