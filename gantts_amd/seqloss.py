"""``sequence_mask`` and ``MaskedMSELoss`` with the reference's signatures
(gantts/seqloss.py:9-43), computed by HIP kernels (``gt_op_sequence_mask``, ``gt_op_masked_mse``)."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import check, lib, ptr


def sequence_mask(sequence_length, max_len=None):
    """(B,) int lengths -> (B, max_len) float32 mask, 1.0 where t < length (seqloss.py:9-20)."""
    if not isinstance(sequence_length, torch.Tensor):
        sequence_length = torch.as_tensor(sequence_length)
    if not sequence_length.is_cuda:
        raise RuntimeError("sequence_mask: lengths are on %s; gantts_amd runs on the GPU only" % sequence_length.device)
    lengths = sequence_length.long().contiguous()
    if max_len is None:
        max_len = int(lengths.max().item())
    B = lengths.size(0)
    mask = torch.empty(B, int(max_len), device=lengths.device, dtype=torch.float32)
    check(lib.gt_op_sequence_mask(ptr(lengths), B, int(max_len), ptr(mask), L.current_stream()))
    return mask


class MaskedMSELoss(object):
    """``sum((input*mask - target*mask)^2) / mask.sum()`` -- normalised by valid FRAMES, not
    frames x dims (seqloss.py:41-43).  Returns a 0-dim tensor; ``grad_input`` of the last call is
    kept in ``self.grad_input`` when ``compute_grad=True`` (the engine's update_generator fuses
    this gradient instead)."""

    def __init__(self, compute_grad=False):
        self.compute_grad = compute_grad
        self.grad_input = None

    def forward(self, input, target, lengths=None, mask=None, max_len=None):
        if lengths is None and mask is None:
            raise RuntimeError("Should provide either lengths or mask")
        if mask is None:
            mask = sequence_mask(lengths, max_len).unsqueeze(-1)
        if not input.is_cuda:
            raise RuntimeError("MaskedMSELoss: input is on %s; gantts_amd runs on the GPU only" % input.device)
        inp, tgt = input.float().contiguous(), target.float().contiguous()
        B, T, D = inp.shape
        m = mask.float().contiguous()
        if m.numel() != B * T:
            raise RuntimeError("mask must be (B, T, 1)")
        grad = torch.empty_like(inp) if self.compute_grad else None
        out = C.c_float()
        check(lib.gt_op_masked_mse(ptr(inp), ptr(tgt), ptr(m), B, T, D, C.byref(out), ptr(grad), L.current_stream()))
        self.grad_input = grad
        return torch.tensor(out.value, device=inp.device)

    __call__ = forward
