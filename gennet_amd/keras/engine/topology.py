from ...engine import Layer  # noqa: F401
