from ...layers import BatchNormalization  # noqa: F401
