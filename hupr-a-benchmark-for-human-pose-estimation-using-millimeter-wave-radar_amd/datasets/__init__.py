from .base import Normalize, generateGTAnnot  # noqa: F401
from .dataset import HuPR3D_horivert, HuPRRawADC, SequenceFFTCache, SyntheticHuPR, getDataset, window_indices  # noqa: F401
