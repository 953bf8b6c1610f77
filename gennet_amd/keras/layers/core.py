from ...layers import Activation, Dense, Dropout, Flatten, Reshape  # noqa: F401
