from .utils import generateTarget  # noqa: F401
from .metrics import get_max_preds  # noqa: F401
from .losses import LossComputer  # noqa: F401
from .plot import plotHumanPose  # noqa: F401
