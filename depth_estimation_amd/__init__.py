"""depth_estimation_amd -- MI355X-native view-synthesis / photometric-loss ops.

The package holds only the hot path: hand-written HIP kernels for gfx950 behind
a C ABI (`include/depthwarp.h`, `csrc/`), torch custom ops over it (`ops`),
drop-in modules with the usual call signatures (`layers`), the reference's own
loss function (`loss.compute_loss`, /root/reference/train.py:272-309), synthetic
inputs (`synth`) and a per-GPU data-parallel training loop (`train`).
"""
from . import synth  # noqa: F401

__version__ = "0.1.0"
