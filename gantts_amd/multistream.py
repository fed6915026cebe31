"""Multi-stream (mgc / lf0 / vuv / bap) utilities with the reference's signatures
(gantts/multistream.py:33-123).  Index arithmetic is integer and host-side; the data movement
(bit-exact column gathers) and the per-stream MLPG run as HIP kernels."""
import numpy as np
import torch

from . import _lib as L
from ._lib import check, lib, ptr


def get_static_stream_sizes(stream_sizes, has_dynamic_features, num_windows):
    """Static width of each stream: size/num_windows for streams with dynamic features
    (multistream.py:46-53).  Returns an int64 numpy array like the reference."""
    out = np.array(stream_sizes, dtype=np.int64)
    dyn = np.array(has_dynamic_features, dtype=bool)
    out[dyn] = out[dyn] // int(num_windows)
    return out


def _gather(inputs, cols):
    if not inputs.is_cuda:
        raise RuntimeError("inputs are on %s; gantts_amd runs on the GPU only" % inputs.device)
    if inputs.dtype != torch.float32:
        raise TypeError("inputs must be float32")
    x = inputs.contiguous()
    B, T, D = x.shape
    idx = torch.tensor(cols, dtype=torch.int32, device=x.device)
    out = torch.empty(B, T, len(cols), device=x.device, dtype=torch.float32)
    check(lib.gt_op_gather_cols(ptr(x), D, ptr(idx), len(cols), ptr(out), len(cols), 0, B * T, L.current_stream()))
    return out


def select_streams(inputs, stream_sizes=[60, 1, 1, 1], streams=[True, True, True, True]):
    """Concatenation of the enabled streams (multistream.py:33-43)."""
    cols, start = [], 0
    for size, enabled in zip(stream_sizes, streams):
        if enabled:
            cols.extend(range(start, start + int(size)))
        start += int(size)
    return _gather(inputs, cols)


def static_columns(num_windows, stream_sizes, has_dynamic_features, streams=None, D=None):
    """Columns of the static components in the static+delta layout (multistream.py:56-79)."""
    if stream_sizes is None or (len(stream_sizes) == 1 and has_dynamic_features[0]):
        width = D if D is not None else stream_sizes[0]
        return list(range(width // num_windows))
    if len(stream_sizes) == 1 and not has_dynamic_features[0]:
        return list(range(D if D is not None else stream_sizes[0]))
    if streams is None:
        streams = [True] * len(stream_sizes)
    cols, start = [], 0
    for size, dyn, enabled in zip(stream_sizes, has_dynamic_features, streams):
        if enabled:
            w = size // num_windows if dyn else size
            cols.extend(range(start, start + w))
        start += size
    return cols


def get_static_features(inputs, num_windows, stream_sizes=[180, 3, 1, 3],
                        has_dynamic_features=[True, True, False, True],
                        streams=[True, True, True, True]):
    """Static features from static+dynamic features (multistream.py:56-79)."""
    return _gather(inputs, static_columns(num_windows, stream_sizes, has_dynamic_features, streams, inputs.size(-1)))


class _HP(object):
    def __init__(self, stream_sizes, has_dynamic_features, num_windows):
        self.stream_sizes, self.has_dynamic_features = list(stream_sizes), list(has_dynamic_features)
        self.windows = [None] * num_windows
        self.adversarial_streams = None
        self.mask_nth_mgc_for_adv_loss = 0
        self.discriminator_linguistic_condition = False


def multi_stream_mlpg(inputs, R, stream_sizes=[180, 3, 1, 3],
                      has_dynamic_features=[True, True, False, True],
                      streams=[True, True, True, True]):
    """Split streams and apply MLPG to those with dynamic features (multistream.py:82-123)."""
    from .engine import engine_for
    B, T, D = inputs.size()
    if D != sum(stream_sizes):
        raise RuntimeError("You probably have specified wrong dimention params.")
    if R is None:
        out = inputs.contiguous().clone()     # num_windows = 1: every stream passes through
    else:
        num_windows = R.size(1) // R.size(0)
        out = engine_for(_HP(stream_sizes, has_dynamic_features, num_windows)).mlpg_forward(inputs, R)
    if not all(streams):
        nW = 1 if R is None else R.size(1) // R.size(0)
        out = select_streams(out, get_static_stream_sizes(stream_sizes, has_dynamic_features, nW), streams)
    return out
