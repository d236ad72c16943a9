from .base import Normalize  # noqa: F401
from .dataset import HuPR3D_horivert, SequenceFFTCache, SyntheticHuPR, getDataset, window_indices  # noqa: F401
