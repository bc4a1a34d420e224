"""convdr_amd -- MI355X-native engine for ConvDR's data-parallel hot path.

Python host code mirrors the reference's operator surface (``model.models``,
the FAISS flat-IP index calls and the driver loops around them); all device
arithmetic runs in hand-written gfx950 HIP kernels reached through the C ABI
of ``libconvdr_hip.so`` (include/convdr_hip.h).  PyTorch tensors are containers
for device memory / streams only.  There is no CPU fallback: using a compute
entry point without the built library or without a GPU raises.
"""
__version__ = "0.1.0"
