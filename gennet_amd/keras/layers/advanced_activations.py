from ...layers import LeakyReLU, ReLU  # noqa: F401
