"""MI355X-native radar -> pose hot path (HuPR), gfx950 only.

Sub-packages mirror the reference's top-level modules so call sites read the
same (reference file in brackets):

  preprocessing.process_iwr1843.RadarObject   [preprocessing/process_iwr1843.py]
  datasets.base.Normalize, datasets.dataset    [datasets/base.py, datasets/dataset.py]
  models.HuPRNet                               [models/networks.py]
  misc.losses.LossComputer                     [misc/losses.py]
  tools.Runner                                 [tools/run.py]

All compute goes through ``runtime`` (ctypes over the C ABI in ``include/hupr.h``,
built from ``csrc/*.hip`` for gfx950).  There is no CPU fallback: importing
``runtime`` without the built library raises.
"""
__version__ = "0.1.0"
