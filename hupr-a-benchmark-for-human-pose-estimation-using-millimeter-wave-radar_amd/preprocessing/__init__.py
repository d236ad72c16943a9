from .process_iwr1843 import RadarObject, dca1000_frames, fft_chain, fft_chain_loader, loader_normalize  # noqa: F401
