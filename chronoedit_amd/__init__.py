"""chronoedit_amd — MI355X-native ChronoEdit denoising engine.

Hand-written HIP (gfx950) kernels behind the reference's own interfaces
(`ChronoEditTransformer3DModel.forward`, `ChronoEditPipeline.__call__`); the host side stays
Python on PyTorch-ROCm (device memory, streams, torch.distributed) exactly like the reference.
The compute path is `lib/libchronoedit_hip.so` (C ABI in include/chronoedit_hip.h); there is
no CPU or eager-PyTorch fallback: importing `chronoedit_amd.ops` without the library raises.
"""
__version__ = "0.1.0"
