"""esvo_amd — MI355X-native ESVO hot path (Time-Surface raster + stereo mapper).

The compute lives in esvo_amd/csrc (hand-written HIP for gfx950 behind the C-ABI of
include/esvo_hip.h); this package is the Python host harness over that C-ABI.
"""
__version__ = "0.1.0"
