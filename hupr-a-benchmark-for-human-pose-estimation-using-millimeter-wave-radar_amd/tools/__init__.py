from .optim import FusedAdam  # noqa: F401


def __getattr__(name):          # lazy: Runner pulls in datasets/models
    if name == "Runner":
        from .run import Runner
        return Runner
    raise AttributeError(name)
