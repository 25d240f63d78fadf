from . import register_model  # noqa: F401
