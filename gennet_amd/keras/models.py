from ..engine import Model, Sequential, load_model  # noqa: F401
