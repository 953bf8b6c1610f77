from ..engine import Adam  # noqa: F401
from ._unused import Adadelta, Adagrad, Adamax, Nadam, RMSprop  # noqa: F401
