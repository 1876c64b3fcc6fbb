"""cstp_amd -- MI355X-native CSTP (R(2+1)D-BYOL) pre-training step.

Host side mirrors the reference's interface for this path (model_name=r21d_byol, main_byol.py,
opts.py); compute runs in hand-written HIP kernels behind the C ABI in include/cstp_hip.h.
"""
__version__ = "0.1.0"
