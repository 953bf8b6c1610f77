from ..engine import Model, Sequential, load_model, model_from_json  # noqa: F401
