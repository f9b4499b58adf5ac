from .params import Params, reset_defaults  # noqa: F401
