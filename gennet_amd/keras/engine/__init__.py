from . import topology  # noqa: F401
