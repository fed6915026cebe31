"""CPU-runnable checks of the drop-in boundary: the C-ABI shared library loads, exports every
symbol include/gantts_hip.h declares, and fails loudly (no fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gantts_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gantts_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libgantts_hip.so does not export %s" % n
    # and the ctypes binding table covers the whole header (nothing bound by accident only)
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_channel():
    from gantts_amd import _lib
    assert b"gfx950" in _lib.lib.gt_version()
    assert isinstance(_lib.lib.gt_last_error(), bytes)


def test_ctypes_structs_match_header_layout():
    from gantts_amd import _lib
    assert C.sizeof(_lib.StreamConfig) == 4 * (1 + 8 + 8 + 1 + 8 + 3)
    assert C.sizeof(_lib.DResult) == 24 and C.sizeof(_lib.GResult) == 20
    assert _lib.ModelDesc.params.offset == 48 and C.sizeof(_lib.ModelDesc) == 72
    assert _lib.OptimDesc.step.offset == 32 and C.sizeof(_lib.OptimDesc) == 56


def test_invalid_arguments_are_reported_not_crashing():
    from gantts_amd import _lib
    cfg = _lib.StreamConfig()
    cfg.n_streams = 0
    h = C.c_void_p()
    rc = _lib.lib.gt_engine_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.GT_ERR_INVALID and b"n_streams" in _lib.lib.gt_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    assert _lib.lib.gt_zero_grad(None, 0) == _lib.GT_ERR_INVALID


@pytest.mark.skipif(torch.cuda.is_available(), reason="this is the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_fallback():
    import gantts_amd.train as T
    from gantts_amd import hparams, models
    from gantts_amd.engine import StepEngine
    m = models.MLP(in_dim=8, out_dim=3, num_hidden=1, hidden_dim=4, last_sigmoid=False)
    with pytest.raises(RuntimeError, match="GPU only|no CPU|no ROCm-capable device"):
        m(torch.rand(1, 5, 8))
    T.hp = hparams.tts_acoustic
    with pytest.raises(RuntimeError):
        T.apply_generator(m, torch.rand(1, 5, 8), None, [5])
    with pytest.raises(Exception) as ei:      # engine creation needs device memory
        StepEngine(hparams.tts_acoustic)
    assert "HIP" in str(ei.value) or "hip" in str(ei.value)
