"""amcpy_amd -- MI355X-native drop-in for the `amcpy extract` hot path.

    from amcpy_amd.config import Config
    from amcpy_amd.feature_extraction import run_extraction
    from amcpy_amd.features import calculate_features, features18

The arithmetic lives in hand-written HIP kernels behind a C ABI
(include/amcx.h, amcpy_amd/csrc); there is no CPU fallback.
"""
__version__ = "0.1.0"
