from .base import Normalize  # noqa: F401
