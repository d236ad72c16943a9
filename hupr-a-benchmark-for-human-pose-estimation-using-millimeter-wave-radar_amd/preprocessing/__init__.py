from .process_iwr1843 import (RadarObject, dca1000_frames, fft_chain, fft_chain_loader, fft_chain_loader_means,  # noqa: F401
                              loader_normalize)
