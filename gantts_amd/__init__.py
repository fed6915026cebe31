"""gantts_amd -- MI355X-native engine behind the Python entry points of r9y9/gantts.

Importing this package loads libgantts_hip.so (hand-written HIP kernels for gfx950); it raises
if the library is missing -- there is no CPU / PyTorch fallback path.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is absent)
from . import hparams, models, multistream, optim, paramgen, seqloss, train  # noqa: F401

__version__ = "0.1.0"
