"""Host-side Python over the libr2s_hip C ABI (MI355X / gfx950).

torch supplies device memory, streams and ``torch.distributed`` only; all compute is in the
hand-written HIP kernels of ``csrc/``.
"""
from ._lib import LIB_PATH, R2SError, lib  # noqa: F401
