from ...layers import LeakyReLU, PReLU, ReLU  # noqa: F401
from .._unused import ThresholdedReLU  # noqa: F401
