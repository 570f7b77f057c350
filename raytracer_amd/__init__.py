from . import capi  # noqa: F401
