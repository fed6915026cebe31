"""Python handle on one ``gt_engine`` (libgantts_hip.so): binds networks / optimizers by raw
device pointer and forwards the step functions.  PyTorch is used for device memory and the
current HIP stream only."""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib as L
from ._lib import check, lib, ptr


def stream_config_from_hp(hp):
    """gt_stream_config from the hp.* fields read on the step path (reference train.py:61,
    233-241, 248, 254, 299, 304, 352-353)."""
    cfg = L.StreamConfig()
    ss = list(hp.stream_sizes)
    hd = list(hp.has_dynamic_features) if hp.has_dynamic_features is not None else [False] * len(ss)
    if len(ss) > L.MAX_STREAMS:
        raise ValueError("at most %d streams are supported" % L.MAX_STREAMS)
    cfg.n_streams = len(ss)
    for i, (s, d) in enumerate(zip(ss, hd)):
        cfg.stream_sizes[i] = int(s)
        cfg.has_dynamic_features[i] = int(bool(d))
    cfg.num_windows = len(hp.windows)
    adv = getattr(hp, "adversarial_streams", None)
    if adv is None:
        cfg.adversarial_streams[0] = -1
    else:
        for i, a in enumerate(adv):
            cfg.adversarial_streams[i] = int(bool(a))
    cfg.mask_nth_mgc_for_adv_loss = int(getattr(hp, "mask_nth_mgc_for_adv_loss", 0))
    cfg.discriminator_linguistic_condition = int(bool(getattr(hp, "discriminator_linguistic_condition", False)))
    cfg.cond_dim = 0   # derived from the bound discriminator's in_dim
    return cfg


def _hp_signature(hp):
    adv = getattr(hp, "adversarial_streams", None)
    return (tuple(hp.stream_sizes), tuple(bool(b) for b in hp.has_dynamic_features), len(hp.windows),
            None if adv is None else tuple(bool(a) for a in adv),
            int(getattr(hp, "mask_nth_mgc_for_adv_loss", 0)),
            bool(getattr(hp, "discriminator_linguistic_condition", False)))


def _same_storage(t, c, name):
    """``t`` came from apply_generator (it carries the engine's graph bookkeeping) but is not contiguous: the copy the
    kernels would read is a different tensor, and the engine would silently treat it as detached (no D->G gradient)."""
    if c is not t and getattr(t, "_gt_engine", None) is not None:
        raise RuntimeError("%s must be passed as returned by apply_generator (contiguous); a strided view would "
                           "drop the gradient path the reference keeps (train.py:265)" % name)
    return c


def _check_frames(t, name, last_dim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: gantts_amd runs on the GPU only (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32" % name)
    if last_dim is not None and t.size(-1) != last_dim:
        raise RuntimeError("%s: expected last dimension %d, got %d" % (name, last_dim, t.size(-1)))
    return t if t.is_contiguous() else t.contiguous()


def pitched_empty(B, T, D, device="cuda", align=4):
    """A (B, T, D) float32 view whose rows have a pitch that is a multiple of ``align`` floats (16 bytes): the layout
    ``gt_set_x_pitch`` lets the engine read with 16-byte loads in every product (``DevicePrefetcher(pitch_x=True)`` stages x so)."""
    P = (D + align - 1) // align * align
    return torch.zeros(B, T, P, device=device, dtype=torch.float32)[:, :, :D]


def _check_x(t, name, last_dim):
    """x of the step functions: (tensor to pass, row pitch in floats or 0 for dense rows).  A (B, T, D) view over a (B, T, P)
    buffer (``pitched_empty``) is taken as it is; anything else goes through ``_check_frames`` (dense, contiguous)."""
    if isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and not t.is_contiguous() \
            and t.size(-1) == last_dim and t.stride(2) == 1 and t.stride(1) >= last_dim and t.stride(1) % 4 == 0 \
            and t.stride(0) == t.size(1) * t.stride(1) and t.data_ptr() % 16 == 0:
        return t, int(t.stride(1))
    return _check_frames(t, name, last_dim), 0


class _SingleStreamHP(object):
    """hp stand-in for forward-only engines: one stream, no dynamic features."""

    def __init__(self, width, num_windows=1, dynamic=False):
        self.stream_sizes = [width]
        self.has_dynamic_features = [dynamic]
        self.windows = [None] * num_windows
        self.adversarial_streams = None
        self.mask_nth_mgc_for_adv_loss = 0
        self.discriminator_linguistic_condition = False


class StepEngine(object):
    def __init__(self, hp):
        self.signature = _hp_signature(hp)
        cfg = stream_config_from_hp(hp)
        h = C.c_void_p()
        check(lib.gt_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._finalizer = weakref.finalize(self, lib.gt_engine_destroy, h)
        self._bound = {L.ROLE_G: (None, -1, False), L.ROLE_D: (None, -1, False)}   # (model ref, version, with_grads)
        self._bound_opt = {L.ROLE_G: (None, -1, None), L.ROLE_D: (None, -1, None)}
        self._keep = {}
        self._ld_gx = self._ld_cx = 0            # row pitches last announced through gt_set_x_pitch (0 = dense)
        self._pitch_sent = (0, 0)
        self._opt_bf16 = False
        nW = len(hp.windows)
        ss = [s // nW if d else s for s, d in zip(hp.stream_sizes, hp.has_dynamic_features)]
        self.static_dim = int(sum(ss))
        self.num_windows = nW

    @classmethod
    def for_forward_only(cls, model):
        if model.include_parameter_generation():
            return cls(_SingleStreamHP(model.out_dim, model.out_dim // model.static_dim, True))
        return cls(_SingleStreamHP(model.out_dim))

    # ---- binding --------------------------------------------------------------------------
    def bind_model(self, role, model, with_grads=True):
        ref, ver, wg = self._bound[role]
        if ref is not None and ref() is model and ver == model._version and (wg or not with_grads):
            check(lib.gt_set_training(self._h, role, int(model.training)))
            return
        desc = model._desc(with_grads)
        check(lib.gt_bind_model(self._h, role, C.byref(desc)))
        for (pass_idx, layer), m in model._masks.items():
            check(lib.gt_set_dropout_mask(self._h, role, pass_idx, layer, ptr(m)))
        check(lib.gt_set_training(self._h, role, int(model.training)))
        self._bound[role] = (weakref.ref(model), model._version, with_grads)
        self._bound_opt[role] = (None, -1, None)
        model._bound_engines[id(self)] = (weakref.ref(self), role)

    def bind_optimizer(self, role, optimizer):
        """torch.optim reads ``param_groups`` on every step: any hyper-parameter edited since the last bind (weight_decay,
        eps, betas, lr_decay, ``max_grad_norm``) re-binds; a change of ``lr`` alone (exp_lr_scheduler, train.py:323-333)
        takes the cheap ``gt_set_lr`` path."""
        ref, ver, hyper = self._bound_opt[role]
        now = optimizer._hyper()
        if ref is not None and ref() is optimizer and ver == optimizer._version and hyper[1:] == now[1:]:
            if hyper[0] != now[0]:
                check(lib.gt_set_lr(self._h, role, now[0]))
                self._bound_opt[role] = (ref, ver, now)
            return
        desc = optimizer._desc()
        check(lib.gt_bind_optimizer(self._h, role, C.byref(desc)))
        self._bound_opt[role] = (weakref.ref(optimizer), optimizer._version, now)
        optimizer._engines[id(self)] = (weakref.ref(self), role)

    # ---- data-parallel communicator (RCCL inside the engine; include/gantts_hip.h gt_comm_*) -------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes identifying a new communicator: rank 0 creates them, every rank passes them to ``comm_init``."""
        buf = (C.c_char * L.COMM_ID_BYTES)()
        check(lib.gt_comm_unique_id(buf))
        return bytes(buf.raw)

    def comm_init(self, rank, world, unique_id):
        """Collective over all ranks.  From here on the fused step functions of this engine are data-parallel: gradients
        (bucketed per layer, overlapped with backward), loss sums and the valid-frame count are summed over the ranks
        inside ``update_discriminator`` / ``update_generator``, which return the GLOBAL scalars on every rank."""
        if len(unique_id) != L.COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % L.COMM_ID_BYTES)
        buf = (C.c_char * L.COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(lib.gt_comm_init(self._h, int(rank), int(world), buf))
        self._dp_world = int(world)

    def comm_destroy(self):
        check(lib.gt_comm_destroy(self._h))

    def comm_ipc_export(self):
        """This rank's interprocess arena for the two-shot all-reduce (``gt_comm_ipc_export``): the handle to ship to every rank."""
        buf = (C.c_char * L.IPC_HANDLE_BYTES)()
        check(lib.gt_comm_ipc_export(self._h, buf))
        return bytes(buf.raw)

    def comm_ipc_attach(self, rank, world, handles):
        """``handles``: the ``world`` exported handles in rank order (bytes, ``world * IPC_HANDLE_BYTES``).  From here on every
        message of the step that fits a slot is reduced over the peers' arenas instead of by RCCL (``set_option("comm_ipc", 0)``
        switches back)."""
        if len(handles) != world * L.IPC_HANDLE_BYTES:
            raise ValueError("handles must be world * %d bytes" % L.IPC_HANDLE_BYTES)
        buf = (C.c_char * len(handles)).from_buffer_copy(handles)
        check(lib.gt_comm_ipc_attach(self._h, int(rank), int(world), buf))

    def comm_ipc_messages(self):
        n = C.c_longlong()
        check(lib.gt_comm_ipc_messages(self._h, C.byref(n)))
        return n.value

    def set_shard(self, rank, world):
        """This engine holds sequences rank, rank + world, ... of the minibatch (no communicator: the host all-reduces between
        the split-phase calls).  Keys the dropout streams globally, so a world-k run reproduces the one-process masks."""
        check(lib.gt_set_shard(self._h, int(rank), int(world)))
        self._dp_world = int(world)

    def comm_trace(self, enable):
        """Starts (clearing the records) or stops the schedule trace of the data-parallel step (gt_comm_trace)."""
        check(lib.gt_comm_trace(self._h, 1 if enable else 0))

    def comm_trace_read(self):
        """Records of the schedule trace as an (n, 5) float64 array: kind (0 message, 1 wait of the step stream), bytes, issued on
        the step stream itself, start us, end us."""
        import numpy as np
        n = C.c_int()
        check(lib.gt_comm_trace_read(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 5), dtype=np.float64)
        check(lib.gt_comm_trace_read(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
        return out[:n.value]

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        check(lib.gt_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def check_faults(self):
        """Synchronises the current stream and raises if a persistent kernel gave up waiting for a peer."""
        check(lib.gt_check_faults(self._h, L.current_stream()))

    def clear_faults(self):
        """Re-arms the engine after a reported fault (gt_clear_faults): the faulted step's optimizer updates were skipped
        on the device, the step counters are put back, the next call must be apply_generator."""
        check(lib.gt_clear_faults(self._h, L.current_stream()))

    def invalidate_mlpg_cache(self):
        check(lib.gt_invalidate_mlpg_cache(self._h))

    def optimizer_step_count(self, role):
        n = C.c_int64()
        check(lib.gt_get_optimizer_step(self._h, role, C.byref(n)))
        return n.value

    def zero_grad(self, role):
        check(lib.gt_zero_grad(self._h, role))

    def set_seed(self, seed):
        check(lib.gt_set_seed(self._h, C.c_uint64(int(seed))))

    def philox_mask(self, role, pass_index, layer, p, rows, cols, steps_ahead=1):
        """Parity hook: the (rows, cols) 0/1 keep mask of Philox dropout site (role, pass, layer) of the step that
        starts ``steps_ahead`` apply_generator calls from now (gt_op_philox_mask)."""
        m = torch.empty(rows, cols, device="cuda", dtype=torch.float32)
        check(lib.gt_op_philox_mask(self._h, role, pass_index, layer, int(steps_ahead), float(p), int(rows), int(cols),
                                    ptr(m), L.current_stream()))
        return m

    def set_option(self, name, value):
        """Engine switches (results unchanged up to fp32 summation order): ``"lstm_persistent"``,
        ``"lstm_fwd_units"``, ``"lstm_xcd_local"``, ``"split_first_layer"`` (conditioned D: x product once per D step),
        ``"fused_optimizer"`` (combines + norm + clip + step in one launch); ``"matmul_bf16"`` switches the GEMMs to bf16 products
        with float32 accumulation."""
        opts = {"lstm_persistent": L.OPT_LSTM_PERSISTENT,
                "lstm_fwd_units": L.OPT_LSTM_FWD_UNITS, "lstm_xcd_local": L.OPT_LSTM_XCD_LOCAL,
                "matmul_bf16": L.OPT_MATMUL_BF16, "split_first_layer": L.OPT_SPLIT_FIRST_LAYER,
                "fused_optimizer": L.OPT_FUSED_OPTIMIZER, "side_overlap": L.OPT_SIDE_OVERLAP, "lstm_side": L.OPT_LSTM_SIDE,
                "comm_d_one_msg": L.OPT_COMM_D_ONE_MSG, "comm_early_g": L.OPT_COMM_EARLY_G, "comm_group": L.OPT_COMM_GROUP,
                "comm_force": L.OPT_COMM_FORCE, "launch_riders": L.OPT_LAUNCH_RIDERS,
                "comm_close_inline": L.OPT_COMM_CLOSE_INLINE, "poll_results": L.OPT_POLL_RESULTS,
                "comm_tv_in_sums": L.OPT_COMM_TV_IN_SUMS, "comm_ipc": L.OPT_COMM_IPC, "fused_dstack": L.OPT_FUSED_DSTACK}
        if name not in opts:
            raise ValueError("unknown engine option %r" % (name,))
        check(lib.gt_set_option(self._h, opts[name], int(value)))
        if name == "matmul_bf16":
            self._opt_bf16 = bool(value)

    def set_loss_normalizer(self, tv):
        """``tv``: python number, or a 1-element CUDA float64 tensor (kept alive here; read in stream order --
        the data-parallel path hands over the all-reduced ``sum(mask)`` without a host sync)."""
        if isinstance(tv, torch.Tensor) and tv.is_cuda:
            if tv.dtype != torch.float64 or tv.numel() != 1:
                raise ValueError("device loss normaliser must be a 1-element float64 tensor")
            self._keep["tv"] = tv
            check(lib.gt_set_loss_normalizer_device(self._h, ptr(tv)))
        else:
            self._keep.pop("tv", None)
            check(lib.gt_set_loss_normalizer(self._h, float(tv)))

    # ---- step functions -------------------------------------------------------------------
    def set_lengths(self, lengths, B, T):
        """`lengths` as the reference passes them (list of ints / 0-dim tensors, LongTensor, numpy)."""
        if lengths is None:
            vals = [T] * B
        elif isinstance(lengths, torch.Tensor):
            vals = [int(v) for v in lengths.detach().cpu().view(-1).tolist()]
        else:
            vals = [int(v) for v in lengths]
        if len(vals) != B:
            raise RuntimeError("lengths has %d entries for a batch of %d sequences" % (len(vals), B))
        arr = (C.c_int64 * B)(*vals)
        check(lib.gt_set_lengths(self._h, arr, B, L.current_stream()))

    def apply_generator(self, model_g, x, R, lengths=None):
        x, ld = _check_x(x, "x", model_g.in_dim)
        # pitched rows (gt_set_x_pitch) are read in place by the float32 MLP generator; for every other network the ENGINE makes the
        # dense copy its kernels read (eng_step.hip: dense_gx), so a pitched batch is valid for any model
        self._set_pitch(gx=ld)
        B, T, _ = x.shape
        model_g._check_masks(B, T)
        self.bind_model(L.ROLE_G, model_g, with_grads=True)
        if getattr(model_g, "needs_lengths", False):
            self.set_lengths(lengths, B, T)
        static_w = model_g.static_dim if model_g.include_parameter_generation() else self.static_dim
        y_hat = torch.empty(B, T, model_g.out_dim, device=x.device, dtype=torch.float32)
        y_hat_static = torch.empty(B, T, static_w, device=x.device, dtype=torch.float32)
        if R is not None:
            R = _check_frames(R, "R")
            if R.dim() != 2 or R.size(0) != T or R.size(1) != self.num_windows * T:
                raise RuntimeError("R must be (T, num_windows*T) = (%d, %d), got %s" % (T, self.num_windows * T, tuple(R.shape)))
        check(lib.gt_apply_generator(self._h, ptr(x), ptr(R), B, T, ptr(y_hat), ptr(y_hat_static), L.current_stream()))
        self._keep["g"] = (x, R, y_hat, y_hat_static)
        y_hat._gt_engine = y_hat_static._gt_engine = self
        return y_hat, y_hat_static

    def _set_pitch(self, gx=None, cx=None):
        if gx is not None:
            self._ld_gx = gx
        if cx is not None:
            self._ld_cx = cx
        if (self._ld_gx, self._ld_cx) != self._pitch_sent:
            check(lib.gt_set_x_pitch(self._h, self._ld_gx, self._ld_cx))
            self._pitch_sent = (self._ld_gx, self._ld_cx)

    def _cond_x(self, x, model_d):
        """The conditioning x of D (train.py:254-256): dense or pitched rows.  The split first layer of the conditioned float32 MLP
        discriminator reads pitched rows in place; for every other discriminator (recurrent, bf16 storage, split switched off) the
        engine densifies them once per step (eng_step.hip: dense_cx) -- the decision lives in ONE place, next to d_split_ok()."""
        x, ld = _check_x(x, "x", model_d.in_dim - self._adv_width())
        self._set_pitch(cx=ld)
        return x

    def _mask2d(self, mask, B, T):
        mask = _check_frames(mask, "mask")
        if mask.numel() != B * T:
            raise RuntimeError("mask must have B*T = %d elements, got %s" % (B * T, tuple(mask.shape)))
        return mask

    def _d_lengths(self, model_d, lengths, B, T):
        """A recurrent discriminator (LSTMRNN in the discriminator slot, train.py:773-774) packs by `lengths` (train.py:262, 268, 307)."""
        if getattr(model_d, "needs_lengths", False):
            self.set_lengths(lengths, B, T)

    def update_discriminator(self, model_d, optimizer_d, x, y_static, y_hat_static, mask, phase, eps=1e-20, lengths=None):
        y_static = _check_frames(y_static, "y_static")
        y_hat_static_c = _same_storage(y_hat_static, _check_frames(y_hat_static, "y_hat_static", y_static.size(-1)), "y_hat_static")
        B, T, _ = y_static.shape
        mask = self._mask2d(mask, B, T)
        train = phase == "train"
        model_d._check_masks(B, T)
        self._d_lengths(model_d, lengths, B, T)
        self.bind_model(L.ROLE_D, model_d, with_grads=True)
        if train:
            self.bind_optimizer(L.ROLE_D, optimizer_d)
        if self.signature[5]:
            x = self._cond_x(x, model_d)
        else:
            x = None
        res = L.DResult()
        check(lib.gt_update_discriminator(self._h, ptr(x), ptr(y_static), ptr(y_hat_static_c), ptr(mask), B, T,
                                          int(train), float(eps), C.byref(res), L.current_stream()))
        if train:
            optimizer_d._note_step(self, L.ROLE_D)
        return res.loss_d, res.loss_fake_d, res.loss_real_d, res.real_correct_count, res.fake_correct_count

    def update_generator(self, model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static,
                         adv_w, mask, phase, mse_w, mge_w, eps=1e-20, lengths=None):
        y = _check_frames(y, "y", model_g.out_dim)
        y_hat_c = _check_frames(y_hat, "y_hat", model_g.out_dim)
        y_static = _check_frames(y_static, "y_static")
        y_hat_static_c = _same_storage(y_hat_static, _check_frames(y_hat_static, "y_hat_static", y_static.size(-1)), "y_hat_static")
        B, T, _ = y.shape
        mask = self._mask2d(mask, B, T)
        train = phase == "train"
        self.bind_model(L.ROLE_G, model_g, with_grads=True)
        if train:
            self.bind_optimizer(L.ROLE_G, optimizer_g)
        if adv_w > 0:
            model_d._check_masks(B, T)
            self._d_lengths(model_d, lengths, B, T)
            self.bind_model(L.ROLE_D, model_d, with_grads=False)
            x = self._cond_x(x, model_d) if self.signature[5] else None
        else:
            x = None
        res = L.GResult()
        check(lib.gt_update_generator(self._h, ptr(x), ptr(y), ptr(y_hat_c), ptr(y_static), ptr(y_hat_static_c),
                                      float(adv_w), ptr(mask), B, T, int(train), float(mse_w), float(mge_w),
                                      float(eps), C.byref(res), L.current_stream()))
        if train:
            optimizer_g._note_step(self, L.ROLE_G)
        return res.loss_mse, res.loss_mge, res.loss_adv, res.loss_g

    # ---- split-phase forms (data parallelism, gantts_amd/parallel.py) ------------------------
    def scalar_sums(self, which=None):
        """Zero-copy torch view of the additive float64 loss / count sums of the current step
        (D step: [0:4], G step: [4:7])."""
        p, n = C.c_void_p(), C.c_int()
        check(lib.gt_scalar_buffer(self._h, C.byref(p), C.byref(n)))

        class _Arr(object):
            __cuda_array_interface__ = {"shape": (n.value,), "typestr": "<f8", "data": (p.value, False), "version": 2}
        t = torch.as_tensor(_Arr(), device="cuda")
        return t if which is None else (t[0:4] if which == "D" else t[4:7])

    def update_discriminator_begin(self, model_d, optimizer_d, x, y_static, y_hat_static, mask, phase, eps=1e-20, lengths=None):
        y_static = _check_frames(y_static, "y_static")
        y_hat_static_c = _same_storage(y_hat_static, _check_frames(y_hat_static, "y_hat_static", y_static.size(-1)), "y_hat_static")
        B, T, _ = y_static.shape
        mask = self._mask2d(mask, B, T)
        train = phase == "train"
        model_d._check_masks(B, T)
        self._d_lengths(model_d, lengths, B, T)
        self.bind_model(L.ROLE_D, model_d, with_grads=True)
        if train:
            self.bind_optimizer(L.ROLE_D, optimizer_d)
        x = self._cond_x(x, model_d) if self.signature[5] else None
        check(lib.gt_update_discriminator_begin(self._h, ptr(x), ptr(y_static), ptr(y_hat_static_c), ptr(mask), B, T,
                                                int(train), float(eps), L.current_stream()))

    def update_discriminator_end(self, optimizer_d, phase, defer=False):
        """``defer=True``: enqueue only (optimizer step, finalisation, D2H of the scalars) and return None;
        ``update_discriminator_result()`` blocks on just that copy later."""
        train = phase == "train"
        res = L.DResult()
        check(lib.gt_update_discriminator_end(self._h, int(train), None if defer else C.byref(res), L.current_stream()))
        if train:
            optimizer_d._note_step(self, L.ROLE_D)
        if defer:
            return None
        return res.loss_d, res.loss_fake_d, res.loss_real_d, res.real_correct_count, res.fake_correct_count

    def update_discriminator_result(self):
        res = L.DResult()
        check(lib.gt_update_discriminator_result(self._h, C.byref(res)))
        return res.loss_d, res.loss_fake_d, res.loss_real_d, res.real_correct_count, res.fake_correct_count

    def update_generator_begin(self, model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static,
                               adv_w, mask, phase, mse_w, mge_w, eps=1e-20, lengths=None):
        y = _check_frames(y, "y", model_g.out_dim)
        y_hat_c = _check_frames(y_hat, "y_hat", model_g.out_dim)
        y_static = _check_frames(y_static, "y_static")
        y_hat_static_c = _same_storage(y_hat_static, _check_frames(y_hat_static, "y_hat_static", y_static.size(-1)), "y_hat_static")
        B, T, _ = y.shape
        mask = self._mask2d(mask, B, T)
        train = phase == "train"
        self.bind_model(L.ROLE_G, model_g, with_grads=True)
        if train:
            self.bind_optimizer(L.ROLE_G, optimizer_g)
        if adv_w > 0:
            model_d._check_masks(B, T)
            self._d_lengths(model_d, lengths, B, T)
            self.bind_model(L.ROLE_D, model_d, with_grads=False)
            x = self._cond_x(x, model_d) if self.signature[5] else None
        else:
            x = None
        check(lib.gt_update_generator_begin(self._h, ptr(x), ptr(y), ptr(y_hat_c), ptr(y_static), ptr(y_hat_static_c),
                                            float(adv_w), ptr(mask), B, T, int(train), float(mse_w), float(mge_w),
                                            float(eps), L.current_stream()))

    def update_generator_end(self, optimizer_g, adv_w, mse_w, mge_w, phase, defer=False):
        train = phase == "train"
        res = L.GResult()
        check(lib.gt_update_generator_end(self._h, int(train), float(adv_w), float(mse_w), float(mge_w),
                                          None if defer else C.byref(res), L.current_stream()))
        if train:
            optimizer_g._note_step(self, L.ROLE_G)
        if defer:
            return None
        return res.loss_mse, res.loss_mge, res.loss_adv, res.loss_g

    def update_generator_result(self):
        res = L.GResult()
        check(lib.gt_update_generator_result(self._h, C.byref(res)))
        return res.loss_mse, res.loss_mge, res.loss_adv, res.loss_g

    def _adv_width(self):
        ss, hd, nW, adv, nmask, _ = self.signature
        static = [s // nW if d else s for s, d in zip(ss, hd)]
        if adv is None:
            return sum(static)
        return sum(s for s, a in zip(static, adv) if a) - (nmask if nmask > 0 else 0)

    def flush_generator_grads(self):
        check(lib.gt_flush_generator_grads(self._h, L.current_stream()))

    def model_forward(self, model, x, R=None, lengths=None):
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        x = _check_frames(x, "x", model.in_dim)
        B, T, _ = x.shape
        model._check_masks(B, T)
        self.bind_model(L.ROLE_G, model, with_grads=False)
        if getattr(model, "needs_lengths", False):
            self.set_lengths(lengths, B, T)
        out = torch.empty(B, T, model.out_dim, device=x.device, dtype=torch.float32)
        out2 = None
        if model.include_parameter_generation():
            if R is None:
                raise RuntimeError("In2OutHighwayNet.forward needs R")
            R = _check_frames(R, "R")
            out2 = torch.empty(B, T, model.static_dim, device=x.device, dtype=torch.float32)
        check(lib.gt_model_forward(self._h, L.ROLE_G, ptr(x), ptr(R), B, T, ptr(out), ptr(out2), L.current_stream()))
        if out2 is not None:
            return out, out2
        return out.squeeze(0) if squeeze else out

    # ---- stand-alone ops that need the stream config --------------------------------------
    def mlpg_forward(self, y, R):
        y = _check_frames(y, "inputs")
        R = _check_frames(R, "R")
        B, T, _ = y.shape
        out = torch.empty(B, T, self.static_dim, device=y.device, dtype=torch.float32)
        check(lib.gt_op_mlpg_forward(self._h, ptr(y), ptr(R), B, T, ptr(out), L.current_stream()))
        return out

    def mlpg_backward(self, g_static, R, full_dim):
        g = _check_frames(g_static, "grad")
        R = _check_frames(R, "R")
        B, T, _ = g.shape
        out = torch.empty(B, T, full_dim, device=g.device, dtype=torch.float32)
        check(lib.gt_op_mlpg_backward(self._h, ptr(g), ptr(R), B, T, ptr(out), L.current_stream()))
        return out


_engines_by_sig = {}


def engine_for(hp, model_g=None):
    """One engine per (generator, hp signature); generators are the anchor because every step
    starts with apply_generator (train.py:543)."""
    sig = _hp_signature(hp)
    if model_g is not None:
        eng = getattr(model_g, "_step_engine", None)
        if eng is None or eng.signature != sig:
            eng = StepEngine(hp)
            model_g._step_engine = eng
        return eng
    eng = _engines_by_sig.get(sig)
    if eng is None:
        eng = _engines_by_sig[sig] = StepEngine(hp)
    return eng


class HipStepBackend(object):
    """Adapter of one (G, D, optimizers, hp) set to the protocol ``DataParallelStep`` drives.
    A batch is ``dict(x=, y=, y_static=, mask=, R=)`` of this rank's CUDA tensors."""

    def __init__(self, hp, model_g, model_d, optimizer_g, optimizer_d):
        self.hp, self.mg, self.md, self.og, self.od = hp, model_g, model_d, optimizer_g, optimizer_d
        self.engine = engine_for(hp, model_g)
        self._out = None

    def mask_of(self, batch):
        return batch["mask"]

    def philox_mask(self, role, pass_index, layer, p, rows, cols, steps_ahead=1):
        """Parity hook: the (rows, cols) 0/1 keep mask of Philox dropout site (role, pass, layer) of the step that
        starts ``steps_ahead`` apply_generator calls from now (gt_op_philox_mask)."""
        return self.engine.philox_mask(role, pass_index, layer, p, rows, cols, steps_ahead)

    def set_option(self, name, value):
        self.engine.set_option(name, value)

    def set_shard(self, rank, world):
        self.engine.set_shard(rank, world)

    device_normalizer = True      # set_loss_normalizer accepts a CUDA float64 tensor (no host sync)

    def set_loss_normalizer(self, tv):
        self.engine.set_loss_normalizer(tv)

    def zero_grad(self):
        self.og.zero_grad()
        self.od.zero_grad()

    def flat_grads(self, which):
        return (self.mg if which == "G" else self.md).flat_grads()

    def scalar_sums(self, which):
        return self.engine.scalar_sums(which)

    def apply_generator(self, batch):
        self._out = self.engine.apply_generator(self.mg, batch.get("g_in", batch["x"]), batch.get("R"), batch.get("lengths"))
        return self._out

    def update_discriminator_begin(self, batch, phase):
        self.engine.update_discriminator_begin(self.md, self.od, batch["x"], batch["y_static"], self._out[1],
                                               batch["mask"], phase)

    deferred_results = True       # *_end(defer=True) + *_result(): the host does not have to wait inside *_end

    def update_discriminator_end(self, batch, phase, defer=False):
        return self.engine.update_discriminator_end(self.od, phase, defer=defer)

    def update_discriminator_result(self):
        return self.engine.update_discriminator_result()

    def update_generator_begin(self, batch, adv_w, mse_w, mge_w, phase):
        self.engine.update_generator_begin(self.mg, self.md, self.og, batch["x"], batch["y"], self._out[0],
                                           batch["y_static"], self._out[1], adv_w, batch["mask"], phase, mse_w, mge_w)

    def update_generator_end(self, batch, adv_w, mse_w, mge_w, phase, defer=False):
        return self.engine.update_generator_end(self.og, adv_w, mse_w, mge_w, phase, defer=defer)

    def update_generator_result(self):
        return self.engine.update_generator_result()
