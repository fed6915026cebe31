#!/usr/bin/env python
"""Micro-benchmark of the f32 MFMA GEMM family at the cfg2 layer shapes (GPU box only).
Times gt_op_linear_forward / gt_op_linear_backward with HIP events via the engine's profiler hooks."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantts_amd import _lib as L  # noqa: E402
from gantts_amd._lib import check, lib, ptr  # noqa: E402

SHAPES = [  # (rows, in, out, label)
    (16384, 425, 512, "G L1"), (16384, 512, 512, "G L2/3"), (16384, 512, 187, "G L4"),
    (32768, 483, 256, "D L1 (2N)"), (32768, 256, 256, "D L2/3 (2N)"), (16384, 256, 256, "D L2/3 (N)"),
]
VARIANTS = ["fwd BN64", "fwd BN128", "bwdD BN64", "bwdD BN128", "bwdW BN64", "bwdW BN128"]


def read():
    ms, fl, cnt = (C.c_double * 6)(), (C.c_double * 6)(), (C.c_int64 * 6)()
    check(lib.gt_profile_read(ms, fl, cnt))
    return [(VARIANTS[v], ms[v] / cnt[v] * 1e3, fl[v] / (ms[v] * 1e-3) / 1e12) for v in range(6) if cnt[v]]


def main(iters=10):
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, din, dout, label in SHAPES:
        X = torch.randn(rows, din, device="cuda")
        W = torch.randn(dout, din, device="cuda") / din ** 0.5
        b = torch.randn(dout, device="cuda")
        Y = torch.empty(rows, dout, device="cuda")
        dY = torch.randn(rows, dout, device="cuda")
        dX = torch.empty(rows, din, device="cuda")
        dW = torch.empty(dout, din, device="cuda")
        db = torch.empty(dout, device="cuda")
        for phase in ("warm", "time"):
            check(lib.gt_profile_enable(1 if phase == "time" else 0))
            for _ in range(2 if phase == "warm" else iters):
                check(lib.gt_op_linear_forward(ptr(X), din, ptr(W), ptr(b), ptr(Y), dout, rows, din, dout, 1, None, 0.0, s))
                check(lib.gt_op_linear_backward(ptr(dY), dout, ptr(X), din, ptr(W), rows, din, dout, ptr(dX), din, ptr(X), 1,
                                                None, 0.0, None, None, s))
                check(lib.gt_op_linear_backward(ptr(dY), dout, ptr(X), din, None, rows, din, dout, None, 0, None, 0, None, 0.0,
                                                ptr(dW), ptr(db), s))
            torch.cuda.synchronize()
        check(lib.gt_profile_enable(0))
        print("%-14s rows=%5d in=%3d out=%3d : " % (label, rows, din, dout) +
              " | ".join("%s %6.1f us %5.1f TF" % r for r in read()))


if __name__ == "__main__":
    main()
