"""theia_amd -- MI355X-native (gfx950 / CDNA4) implementation of the Theia distillation hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed); all math runs in hand-written HIP
kernels behind the C ABI of ``include/theia_hip.h`` (``theia_amd/lib/libtheia_hip.so``).
"""
__version__ = "0.1.0"
