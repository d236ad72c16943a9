from .networks import HuPRNet  # noqa: F401
