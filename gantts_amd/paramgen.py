"""Host-side construction of the MLPG matrix ``R = (W^T W)^-1 W^T`` -- the piece of
``nnmnkwii.paramgen`` the step path needs (reference train.py:49, 510-515 calls
``unit_variance_mlpg_matrix(hp.windows, max_len)`` for every batch).

Differences from the reference call pattern (results identical): W^T W is banded, so it is
solved with a banded Cholesky in float64 (O(T^2) instead of a dense inverse), and the result is
cached per (windows, T) together with its device copy -- the reference rebuilds it on the host
and uploads T x 3T floats every batch.
"""
import numpy as np
import scipy.linalg

_cache = {}
_dev_cache = {}


def _signature(windows, T):
    return (int(T),) + tuple((int(l), int(u), tuple(float(c) for c in np.asarray(w).ravel())) for (l, u, w) in windows)


def unit_variance_mlpg_matrix(windows, T):
    """(T, len(windows)*T) float32; column block w holds window w (window-major)."""
    T = int(T)
    key = _signature(windows, T)
    hit = _cache.get(key)
    if hit is not None:
        return hit
    nW = len(windows)
    Wt = np.zeros((T, nW * T), dtype=np.float64)            # W^T, block w = W_w^T
    bw = max(max(int(l), int(u)) for (l, u, _) in windows)
    for w, (l, u, coef) in enumerate(windows):
        coef = np.asarray(coef, dtype=np.float64)
        for k in range(-int(l), int(u) + 1):
            c = coef[k + int(l)]
            if c == 0.0:
                continue
            t = np.arange(max(0, -k), min(T, T - k))          # (W_w x)[t] += c * x[t + k]
            Wt[t + k, w * T + t] = c
    P = Wt @ Wt.T                                             # W^T W, bandwidth 2*bw
    ub = min(2 * bw, T - 1)
    ab = np.zeros((ub + 1, T), dtype=np.float64)              # upper banded storage for solveh_banded
    for d in range(ub + 1):
        ab[ub - d, d:] = np.diagonal(P, d)
    R = scipy.linalg.solveh_banded(ab, Wt, lower=False, check_finite=False)
    R = np.ascontiguousarray(R.astype(np.float32))
    R.setflags(write=False)
    _cache[key] = R
    return R


def unit_variance_mlpg_matrix_cuda(windows, T, device="cuda"):
    """Device-resident copy, cached: one upload per distinct T instead of one per batch."""
    import torch
    key = (_signature(windows, T), str(device))
    hit = _dev_cache.get(key)
    if hit is None:
        hit = _dev_cache[key] = torch.from_numpy(np.array(unit_variance_mlpg_matrix(windows, T))).to(device)
    return hit
