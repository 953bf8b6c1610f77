from ..engine import Adam  # noqa: F401
