"""ctypes binding of libgantts_hip.so (C ABI declared in include/gantts_hip.h).

The HIP library is the product: if it is missing or fails to load this module raises --
there is no CPU or PyTorch fallback.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C gantts_amd/csrc``.
"""
import ctypes as C
import os

# PyTorch-ROCm first: it carries its own HIP runtime (libamdhip64 of its wheel).  Loaded first, the dynamic loader resolves this library's
# libamdhip64 dependency to that SAME copy; loaded second it maps a second runtime beside the system one and every HIP call of the process
# that lands in the late copy fails with "no ROCm-capable device is detected" (seen in round 6 when build() imported the package before
# smoke() imported torch, in one process).  The tensors the step functions take are torch tensors: torch is a dependency either way.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GT_HIP_LIB") or os.path.join(_HERE, "libgantts_hip.so")      # GT_HIP_LIB: A/B of two builds (tools/)

GT_OK, GT_ERR_INVALID, GT_ERR_HIP, GT_ERR_STATE, GT_ERR_DIM = 0, 1, 2, 3, 4
ROLE_G, ROLE_D = 0, 1
OPT_LSTM_PERSISTENT, OPT_LSTM_FWD_UNITS, OPT_LSTM_XCD_LOCAL, OPT_MATMUL_BF16, OPT_SPLIT_FIRST_LAYER, OPT_FUSED_OPTIMIZER = 2, 3, 4, 5, 6, 7
OPT_SIDE_OVERLAP, OPT_LSTM_SIDE, OPT_COMM_D_ONE_MSG, OPT_COMM_EARLY_G, OPT_COMM_GROUP, OPT_COMM_FORCE = 8, 9, 10, 11, 12, 13
OPT_LAUNCH_RIDERS, OPT_COMM_CLOSE_INLINE, OPT_POLL_RESULTS = 14, 15, 16
OPT_COMM_TV_IN_SUMS, OPT_COMM_IPC, OPT_FUSED_DSTACK = 17, 18, 19
IPC_HANDLE_BYTES, IPC_MAX_WORLD = 64, 8
PROFILE_SLOTS = 16
ARCH_MLP, ARCH_IN2OUT, ARCH_LSTM, ARCH_SRU, ARCH_IN2OUT_RNN = 0, 1, 2, 3, 4
OPT_ADAGRAD, OPT_ADAM = 0, 1
MAX_STREAMS = 8
COMM_ID_BYTES = 128


class StreamConfig(C.Structure):
    _fields_ = [("n_streams", C.c_int32),
                ("stream_sizes", C.c_int32 * MAX_STREAMS),
                ("has_dynamic_features", C.c_int32 * MAX_STREAMS),
                ("num_windows", C.c_int32),
                ("adversarial_streams", C.c_int32 * MAX_STREAMS),
                ("mask_nth_mgc_for_adv_loss", C.c_int32),
                ("discriminator_linguistic_condition", C.c_int32),
                ("cond_dim", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("arch", C.c_int32), ("in_dim", C.c_int32), ("out_dim", C.c_int32),
                ("num_hidden", C.c_int32), ("hidden_dim", C.c_int32), ("static_dim", C.c_int32),
                ("dropout", C.c_float), ("last_sigmoid", C.c_int32),
                ("bidirectional", C.c_int32), ("use_relu", C.c_int32),
                ("rnn_dropout", C.c_float), ("reserved_", C.c_int32),
                ("params", C.c_void_p), ("grads", C.c_void_p), ("n_params", C.c_int64)]


class OptimDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("weight_decay", C.c_float), ("eps", C.c_float),
                ("lr_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("max_grad_norm", C.c_float), ("step", C.c_int64),
                ("state0", C.c_void_p), ("state1", C.c_void_p)]


class DResult(C.Structure):
    _fields_ = [("loss_d", C.c_float), ("loss_fake_d", C.c_float), ("loss_real_d", C.c_float),
                ("real_correct_count", C.c_float), ("fake_correct_count", C.c_float), ("grad_norm", C.c_float)]


class GResult(C.Structure):
    _fields_ = [("loss_mse", C.c_float), ("loss_mge", C.c_float), ("loss_adv", C.c_float),
                ("loss_g", C.c_float), ("grad_norm", C.c_float)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes); every symbol declared in include/gantts_hip.h
class DistortionSums(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("s_mcd", "s_bap", "s_f0", "n_voiced", "n_vuv_err", "s_mse", "n_frames")]


SIGNATURES = {
    "gt_last_error": (C.c_char_p, []),
    "gt_version": (C.c_char_p, []),
    "gt_engine_create": (_I, [C.POINTER(StreamConfig), C.POINTER(_P)]),
    "gt_engine_destroy": (None, [_P]),
    "gt_bind_model": (_I, [_P, _I, C.POINTER(ModelDesc)]),
    "gt_bind_optimizer": (_I, [_P, _I, C.POINTER(OptimDesc)]),
    "gt_set_training": (_I, [_P, _I, _I]),
    "gt_set_lr": (_I, [_P, _I, _F]),
    "gt_get_optimizer_step": (_I, [_P, _I, C.POINTER(_L)]),
    "gt_set_seed": (_I, [_P, C.c_uint64]),
    "gt_set_dropout_mask": (_I, [_P, _I, _I, _I, _P]),
    "gt_op_philox_mask": (_I, [_P, _I, _I, _I, _L, _F, _L, _I, _P, _P]),
    "gt_set_lengths": (_I, [_P, C.POINTER(_L), _I, _P]),
    "gt_invalidate_mlpg_cache": (_I, [_P]),
    "gt_zero_grad": (_I, [_P, _I]),
    "gt_apply_generator": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "gt_update_discriminator": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, C.POINTER(DResult), _P]),
    "gt_update_generator": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _F, _F, _F, C.POINTER(GResult), _P]),
    "gt_set_loss_normalizer": (_I, [_P, _F]),
    "gt_set_loss_normalizer_device": (_I, [_P, _P]),
    "gt_set_option": (_I, [_P, _I, _I]),
    "gt_set_x_pitch": (_I, [_P, _I, _I]),
    "gt_set_tuning": (_I, [C.c_char_p, _I]),
    "gt_check_faults": (_I, [_P, _P]),
    "gt_clear_faults": (_I, [_P, _P]),
    "gt_comm_unique_id": (_I, [_P]),
    "gt_comm_init": (_I, [_P, _I, _I, _P]),
    "gt_comm_destroy": (_I, [_P]),
    "gt_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "gt_comm_trace": (_I, [_P, _I]),
    "gt_comm_trace_read": (_I, [_P, C.POINTER(C.c_double), _I, C.POINTER(_I)]),
    "gt_set_shard": (_I, [_P, _I, _I]),
    "gt_comm_ipc_export": (_I, [_P, _P]),
    "gt_comm_ipc_attach": (_I, [_P, _I, _I, _P]),
    "gt_comm_ipc_messages": (_I, [_P, C.POINTER(C.c_longlong)]),
    "gt_update_discriminator_begin": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "gt_update_discriminator_end": (_I, [_P, _I, C.POINTER(DResult), _P]),
    "gt_update_generator_begin": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _F, _F, _F, _P]),
    "gt_update_generator_end": (_I, [_P, _I, _F, _F, _F, C.POINTER(GResult), _P]),
    "gt_update_discriminator_result": (_I, [_P, C.POINTER(DResult)]),
    "gt_update_generator_result": (_I, [_P, C.POINTER(GResult)]),
    "gt_scalar_buffer": (_I, [_P, C.POINTER(_P), C.POINTER(_I)]),
    "gt_model_forward": (_I, [_P, _I, _P, _P, _I, _I, _P, _P, _P]),
    "gt_flush_generator_grads": (_I, [_P, _P]),
    "gt_op_sequence_mask": (_I, [_P, _I, _I, _P, _P]),
    "gt_op_masked_mse": (_I, [_P, _P, _P, _I, _I, _I, C.POINTER(_F), _P, _P]),
    "gt_compute_distortions": (_I, [_P, _P, _I, _P, _P, _I, C.POINTER(_I), C.POINTER(_I), _I, C.POINTER(_L), _I, _I,
                                    C.POINTER(DistortionSums), _P]),
    "gt_op_gather_cols": (_I, [_P, _I, _P, _I, _P, _I, _I, _L, _P]),
    "gt_op_pad_sequences": (_I, [_P, _I, _P, _P, _I, _I, _P, _I, _P]),
    "gt_op_mlpg_forward": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "gt_op_mlpg_backward": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "gt_op_linear_forward": (_I, [_P, _I, _P, _P, _P, _I, _L, _I, _I, _I, _P, _F, _P]),
    "gt_op_linear_bf16": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _F, _P, _P, _P, _I, _P, _F, _P, _P, _P, _P, _P, _P]),
    "gt_profile_enable": (_I, [_I]),
    "gt_profile_read": (_I, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_L)]),
    "gt_profile_bytes": (_I, [C.POINTER(C.c_double)]),
    "gt_op_linear_backward": (_I, [_P, _I, _P, _I, _P, _L, _I, _I, _P, _I, _P, _I, _P, _F, _P, _P, _P]),
}


def _build_if_missing():
    """The in-tree library is normally built by ``__graft_entry__.build()``.  If it is MISSING and a HIP compiler is
    at hand, build it now (one process at a time: ranks of a multi-GPU job import concurrently) -- this is still the
    HIP product, not a fallback.  An existing library is never rebuilt here."""
    import fcntl
    import shutil
    import subprocess
    if os.path.isfile(LIB_PATH):
        return
    if shutil.which("hipcc") is None and not os.path.isfile("/opt/rocm/bin/hipcc"):
        return
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.isfile(LIB_PATH):       # another rank may have built it while we waited
                subprocess.call(["make", "-j8", "-C", os.path.join(_HERE, "csrc")], stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _load():
    _build_if_missing()
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "gantts_amd: %s not found -- the HIP engine is the product and has no fallback. "
            "Build it: make -C %s" % (LIB_PATH, os.path.join(_HERE, "csrc")))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class GanttsHipError(RuntimeError):
    pass


def check(rc):
    """Maps a C status to the exception the reference raises at the same place."""
    if rc == GT_OK:
        return
    msg = lib.gt_last_error().decode("utf-8", "replace")
    if rc == GT_ERR_DIM:
        raise RuntimeError(msg)            # multistream.py:93-94 raises RuntimeError
    if rc == GT_ERR_INVALID:
        raise ValueError(msg)
    if rc == GT_ERR_STATE:
        raise RuntimeError(msg)
    raise GanttsHipError("HIP error: " + msg)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
