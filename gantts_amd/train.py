"""The step functions of the reference's ``train.py`` with identical signatures and return
values, executed by the HIP engine:

    apply_generator(model_g, x, R, lengths)                          train.py:336-355
    update_discriminator(model_d, optimizer_d, x, y_static, y_hat_static, lengths, mask, phase, eps)
                                                                      train.py:245-279
    update_generator(model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static,
                     adv_w, lengths, mask, phase, mse_w, mge_w, eps)  train.py:282-320
    get_selected_static_stream, exp_lr_scheduler, save_checkpoint, load_checkpoint

As in the reference, the module-global ``hp`` supplies stream_sizes / has_dynamic_features /
windows / adversarial_streams / mask_nth_mgc_for_adv_loss / discriminator_linguistic_condition
(train.py:61).  Set ``gantts_amd.train.hp = hparams.tts_acoustic`` before calling.

Autograd is replaced by the engine's own graph bookkeeping: ``y_hat``/``y_hat_static`` returned
by ``apply_generator`` remember their engine, so ``update_discriminator`` can keep
d loss_d / d y_hat_static (which the reference leaks into G's .grad, train.py:265,274) for the
following ``update_generator``.
"""
import math
from os.path import join

import torch

from . import hparams
from .engine import engine_for
from .multistream import get_static_stream_sizes, select_streams

hp = None  # to be set by the caller (train.py:61)
global_epoch = 0


def get_selected_static_stream(y_hat_static):
    """Adversarial-loss input: selected static streams minus the first n mgc columns (train.py:232-242)."""
    static_stream_sizes = get_static_stream_sizes(hp.stream_sizes, hp.has_dynamic_features, len(hp.windows))
    y_hat_selected = select_streams(y_hat_static, static_stream_sizes, streams=hp.adversarial_streams)
    if hp.mask_nth_mgc_for_adv_loss > 0:
        assert hp.name == "acoustic"
        y_hat_selected = y_hat_selected[:, :, hp.mask_nth_mgc_for_adv_loss:]
    return y_hat_selected


def apply_generator(model_g, x, R, lengths):
    """Returns ``(y_hat, y_hat_static)``; for generic models y_hat_static = multi_stream_mlpg(y_hat)."""
    if not model_g.include_parameter_generation():
        assert hp.has_dynamic_features is not None
        # Deliberate deviation: a generic packed-sequence generator (LSTMRNN / GRURNN) whose longest sequence is shorter than the
        # batch's T returns max(lengths) frames in the reference (pad_packed_sequence), which train.py:347-350 then zero-pads ON THE LEFT to T --
        # every frame moves by T - max(lengths) against x and y.  train_loop cannot reach it (collate_fn pads to the batch's longest sequence);
        # the engine does not reproduce the shift and refuses the call instead of returning differently aligned frames.  (A data-parallel rank
        # holds a SHARD of a batch that was padded to ITS longest sequence, which may live on another rank: not this case.)
        if getattr(model_g, "needs_lengths", False) and lengths is not None:
            longest = max(int(v) for v in (lengths.view(-1).tolist() if hasattr(lengths, "view") else lengths))
            sharded = False
            if longest < x.size(1):
                try:
                    sharded = getattr(engine_for(hp, model_g), "_dp_world", 1) > 1
                except Exception:      # noqa: BLE001 -- no engine (no device): the argument check stands on its own
                    sharded = False
            if longest < x.size(1) and not sharded:
                raise ValueError("apply_generator: the longest sequence has %d frames in a batch padded to %d -- the reference left-pads the "
                                 "generator's output by the difference here (train.py:347-350), which this engine does not reproduce; "
                                 "pad the batch to its longest sequence as train.py's collate_fn does" % (longest, x.size(1)))
    return engine_for(hp, model_g).apply_generator(model_g, x, R, lengths)


def _engine_of(t, model_g=None):
    eng = getattr(t, "_gt_engine", None)
    if eng is not None:
        return eng
    return engine_for(hp, model_g)


def update_discriminator(model_d, optimizer_d, x, y_static, y_hat_static, lengths,
                         mask, phase, eps=1e-20):
    """Returns ``(loss_d, loss_fake_d, loss_real_d, real_correct_count, fake_correct_count)``."""
    eng = _engine_of(y_hat_static)
    return eng.update_discriminator(model_d, optimizer_d, x, y_static, y_hat_static, mask, phase, eps, lengths=lengths)


def update_generator(model_g, model_d, optimizer_g,
                     x, y, y_hat, y_static, y_hat_static,
                     adv_w, lengths, mask, phase,
                     mse_w=None, mge_w=None, eps=1e-20):
    """Returns ``(loss_mse, loss_mge, loss_adv, loss_g)``."""
    eng = _engine_of(y_hat_static, model_g)
    return eng.update_generator(model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static,
                                adv_w, mask, phase, mse_w, mge_w, eps, lengths=lengths)


# ---- distortion metrics of the training loop (train.py:358-432) --------------------------------
_LOGDB_CONST = 10.0 / math.log(10.0) * math.sqrt(2.0)      # nnmnkwii.metrics.melcd
ROLE_MCD, ROLE_BAP, ROLE_LF0, ROLE_VUV, ROLE_MSE, ROLE_NONE = 0, 1, 2, 3, 4, -1


def _acoustic_columns():
    """Per static column of the acoustic layout: (role, index of its statistics).  The statistics
    are indexed in the static+dynamic domain: mgc at 0, lf0 at mgc_dim, vuv after lf0, bap after
    vuv, each for the first ``dim // num_windows`` entries of its stream (train.py:358-372)."""
    mgc_dim, lf0_dim, vuv_dim, bap_dim = hp.stream_sizes
    nw = len(hp.windows)
    smgc, slf0, svuv, sbap = [int(v) for v in
                              get_static_stream_sizes(hp.stream_sizes, hp.has_dynamic_features, nw)]
    lf0_0, vuv_0, bap_0 = mgc_dim, mgc_dim + lf0_dim, mgc_dim + lf0_dim + vuv_dim
    if mgc_dim // nw != smgc or lf0_dim // nw != slf0 or bap_dim // nw != sbap:
        # the reference's broadcast of Y_mean[:dim // len(windows)] against the static slice fails here too
        raise RuntimeError("stream statistics do not match the static stream sizes")
    roles, stats = [], []
    for j in range(smgc):                       # "mcd" skips the 0-th (energy) coefficient: mgc[:, :, 1:]
        roles.append(ROLE_NONE if j == 0 else ROLE_MCD)
        stats.append(j)
    for j in range(slf0):
        roles.append(ROLE_LF0)
        stats.append(lf0_0 + j)
    roles.append(ROLE_VUV)                      # vuv = y_static[:, :, vuv_start_idx]: ONE column
    stats.append(vuv_0)
    for j in range(svuv - 1):                   # never read by the reference
        roles.append(ROLE_NONE)
        stats.append(vuv_0)
    nbap = smgc + slf0 + svuv
    for j in range(sbap):
        roles.append(ROLE_BAP)
        stats.append(bap_0 + j)
    return roles, stats, smgc + slf0, nbap


def _distortion_sums(y_static, y_hat_static, Y_data_mean, Y_data_std, roles, stats, vuv_col, lengths):
    from . import _lib as L
    import ctypes as C
    if y_static.dim() == 2:
        y_static, y_hat_static = y_static.unsqueeze(0), y_hat_static.unsqueeze(0)
    y = y_static.detach().to(torch.float32).contiguous()
    yh = y_hat_static.detach().to(torch.float32).contiguous()
    if not (y.is_cuda and yh.is_cuda):
        raise RuntimeError("gantts_amd: compute_distortions needs CUDA(HIP) tensors (the HIP engine has no CPU path)")
    if y.shape != yh.shape or y.size(-1) != len(roles):
        raise RuntimeError("You probably have specified wrong dimention params.")
    B, T, Ds = y.shape
    f64 = Y_data_mean.dtype == torch.float64 or Y_data_std.dtype == torch.float64
    dt = torch.float64 if f64 else torch.float32
    mean = Y_data_mean.detach().to(y.device, dt).contiguous().view(-1)
    std = Y_data_std.detach().to(y.device, dt).contiguous().view(-1)
    if max(stats) >= mean.numel() or mean.numel() != std.numel():
        raise RuntimeError("Y_data_mean / Y_data_std are shorter than the feature layout")
    lens = None
    if lengths is not None:
        vals = [int(v) for v in (lengths.detach().cpu().view(-1).tolist() if isinstance(lengths, torch.Tensor) else lengths)]
        if len(vals) != B:
            raise RuntimeError("lengths has %d entries for a batch of %d sequences" % (len(vals), B))
        lens = (C.c_int64 * B)(*vals)
    out = L.DistortionSums()
    L.check(L.lib.gt_compute_distortions(L.ptr(y), L.ptr(yh), Ds, L.ptr(mean), L.ptr(std), int(f64),
                                         (C.c_int * Ds)(*stats), (C.c_int * Ds)(*roles), vuv_col, lens, B, T,
                                         C.byref(out), L.current_stream()))
    return out


def inv_scale(mgc, lf0, vuv, bap, Y_mean, Y_std, binalize_vuv=True):
    """train.py:358-380 on device tensors (plain torch indexing; used by the evaluation scripts --
    the per-step metric path is the fused kernel behind ``compute_distortions``)."""
    mgc_dim, lf0_dim, vuv_dim, bap_dim = hp.stream_sizes
    nw = len(hp.windows)
    lf0_0, vuv_0, bap_0 = mgc_dim, mgc_dim + lf0_dim, mgc_dim + lf0_dim + vuv_dim
    mgc = mgc * Y_std[:mgc_dim // nw] + Y_mean[:mgc_dim // nw]
    lf0 = lf0 * Y_std[lf0_0:lf0_0 + lf0_dim // nw] + Y_mean[lf0_0:lf0_0 + lf0_dim // nw]
    bap = bap * Y_std[bap_0:bap_0 + bap_dim // nw] + Y_mean[bap_0:bap_0 + bap_dim // nw]
    vuv = vuv * Y_std[vuv_0] + Y_mean[vuv_0]
    if binalize_vuv:
        vuv = (vuv > 0.5).long()
    return mgc, lf0, vuv, bap


def split_streams(y_static, Y_data_mean, Y_data_std):
    """train.py:383-396"""
    smgc, slf0, svuv, sbap = [int(v) for v in
                              get_static_stream_sizes(hp.stream_sizes, hp.has_dynamic_features, len(hp.windows))]
    lf0_0, vuv_0, bap_0 = smgc, smgc + slf0, smgc + slf0 + svuv
    return inv_scale(y_static[:, :, :lf0_0], y_static[:, :, lf0_0:vuv_0], y_static[:, :, vuv_0],
                     y_static[:, :, bap_0:], Y_data_mean, Y_data_std)


def compute_distortions(y_static, y_hat_static, Y_data_mean, Y_data_std, lengths=None):
    """train.py:399-432: {"mcd", "bap_mcd", "f0_rmse", "vuv_err"} (acoustic), {"dur_rmse"} (duration) or
    {"mcd"} (vc), from ONE fused masked reduction on the device and one D2H of 7 doubles instead of
    the reference's per-sequence python loops."""
    Ds = y_static.size(-1)
    if hp.name == "acoustic":
        roles, stats, vuv_col, _ = _acoustic_columns()
        r = _distortion_sums(y_static, y_hat_static, Y_data_mean, Y_data_std, roles, stats, vuv_col, lengths)
        f0_mse = r.s_f0 / r.n_voiced if r.n_voiced > 0 else float("nan")     # ZeroDivisionError -> nan (:408-409)
        return {"mcd": _LOGDB_CONST * r.s_mcd / r.n_frames,
                "bap_mcd": _LOGDB_CONST * r.s_bap / r.n_frames / 10.0,
                "f0_rmse": math.sqrt(f0_mse) if f0_mse == f0_mse else float("nan"),
                "vuv_err": r.n_vuv_err / r.n_frames}
    if hp.name == "duration":
        r = _distortion_sums(y_static, y_hat_static, Y_data_mean, Y_data_std, [ROLE_MSE] * Ds, list(range(Ds)), -1, lengths)
        return {"dur_rmse": math.sqrt(r.s_mse / (r.n_frames * Ds))}
    if hp.name == "vc":
        if Ds != hp.order:
            raise RuntimeError("vc distortion expects static_dim == hp.order (train.py:423)")
        r = _distortion_sums(y_static, y_hat_static, Y_data_mean, Y_data_std, [ROLE_MCD] * Ds, list(range(Ds)), -1, lengths)
        return {"mcd": _LOGDB_CONST * r.s_mcd / r.n_frames}
    assert False


def exp_lr_scheduler(optimizer, epoch, nepoch, init_lr=0.0001, lr_decay_epoch=100):
    """lr = init_lr * 0.1 ** (epoch // lr_decay_epoch), written into param_groups (train.py:323-333)."""
    lr = init_lr * (0.1 ** (epoch // lr_decay_epoch))
    if epoch % lr_decay_epoch == 0:
        print("LR is set to {} at epoch {}".format(lr, epoch))
    for group in optimizer.param_groups:
        group["lr"] = lr
    return optimizer


def save_checkpoint(model, optimizer, epoch, checkpoint_dir, name):
    """{"state_dict", "optimizer", "global_epoch"} -> checkpoint_epoch{N}_{name}.pth (train.py:162-171)."""
    path = join(checkpoint_dir, "checkpoint_epoch{}_{}.pth".format(epoch, name))
    torch.save({"state_dict": model.state_dict(), "optimizer": optimizer.state_dict(), "global_epoch": epoch}, path)
    print("Saved checkpoint:", path)
    return path


def load_checkpoint(model, optimizer, checkpoint_path):
    """Restores model, optionally optimizer, and global_epoch (train.py:651-658)."""
    global global_epoch
    print("Load checkpoint from: {}".format(checkpoint_path))
    ckpt = torch.load(checkpoint_path, map_location="cpu")
    model.load_state_dict(ckpt["state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    global_epoch = ckpt["global_epoch"]
    return global_epoch


# ---- the training loop (train.py:435-643) --------------------------------------------------------
checkpoint_dir = "checkpoints"     # train.py global set from --checkpoint-dir
checkpoint_interval = 10           # train.py:66


def log_value(name, value, step):
    """tensorboard_logger.log_value stand-in (train.py:45); rebind ``gantts_amd.train.log_value`` to log."""


def train_loop(models, optimizers, dataset_loaders, w_d=0.0, mse_w=0.0, mge_w=1.0,
               update_d=True, update_g=True, reference_discriminator=None):
    """train.py:435-643 with the same arguments, curriculum (``adv_w = w_d * clip(E_mge / E_adv, 0, 1e3)``),
    logged names/values and checkpoint cadence.  Differences are all on the data path: batches arrive
    through ``DevicePrefetcher`` (pinned, double-buffered H2D overlapped with the previous step), the
    MLPG matrix is cached per T on the device instead of being rebuilt and uploaded every batch
    (train.py:509-513), and the distortion metrics come from one fused reduction."""
    from .data import DevicePrefetcher
    from .multistream import get_static_features
    from .paramgen import unit_variance_mlpg_matrix_cuda
    from .seqloss import sequence_mask
    import numpy as np
    global global_epoch
    model_g, model_d = models
    optimizer_g, optimizer_d = optimizers
    model_g, model_d = model_g.cuda(), model_d.cuda()
    if reference_discriminator is not None:
        reference_discriminator = reference_discriminator.cuda()
        reference_discriminator.eval()

    ds = dataset_loaders["train"].dataset
    if hp.name == "vc":
        Y_data_mean, Y_data_std = ds.data_mean, ds.data_std
    else:
        Y_data_mean, Y_data_std = ds.Y_data_mean, ds.Y_data_std
    Y_data_mean = torch.from_numpy(np.asarray(Y_data_mean)).cuda()
    Y_data_std = torch.from_numpy(np.asarray(Y_data_std)).cuda()

    E_loss_mge, E_loss_adv = 1, 1
    has_dynamic = bool(np.any(hp.has_dynamic_features))
    noise_gen = torch.Generator(device="cuda")
    noise_gen.manual_seed(int(getattr(hp, "generator_noise_seed", 0)))

    for global_epoch in range(global_epoch + 1, hp.nepoch + 1):
        if hp.lr_decay_schedule and update_g:
            optimizer_g = exp_lr_scheduler(optimizer_g, global_epoch - 1, hp.nepoch,
                                           init_lr=hp.optimizer_g_params["lr"], lr_decay_epoch=hp.lr_decay_epoch)
        if hp.lr_decay_schedule and update_d:
            optimizer_d = exp_lr_scheduler(optimizer_d, global_epoch - 1, hp.nepoch,
                                           init_lr=hp.optimizer_d_params["lr"], lr_decay_epoch=hp.lr_decay_epoch)
        for phase in ["train", "test"]:
            running_loss = {"generator": 0.0, "mse": 0.0, "mge": 0.0, "loss_real_d": 0.0, "loss_fake_d": 0.0,
                            "loss_adv": 0.0, "discriminator": 0.0}
            if phase == "train":
                model_g.train(), model_d.train()
            else:
                model_g.eval(), model_d.eval()
            running_metrics = {}
            real_correct_count, fake_correct_count = 0, 0
            regard_fake_as_natural = 0
            N = len(dataset_loaders[phase])
            total_num_frames = 0
            # x is staged on a 16-byte row pitch (gt_set_x_pitch): the layout the engine reads with 16-byte loads in every product --
            # free here, the batch is being copied anyway, and every network accepts it (others get a dense copy inside the engine);
            # hp.pitch_x = False restores dense rows
            for batch in DevicePrefetcher(dataset_loaders[phase], pitch_x=bool(getattr(hp, "pitch_x", True))):
                x, y, sorted_lengths, cpu_sorted_lengths = batch.x, batch.y, batch.lengths, batch.cpu_lengths
                max_len = batch.max_len
                # generator noise z ~ U[0,1) (train.py:504-506), drawn on the device
                z = torch.rand(x.size(0), max_len, hp.generator_noise_dim, device=x.device, generator=noise_gen) \
                    if hp.generator_add_noise else None
                R = unit_variance_mlpg_matrix_cuda(hp.windows, max_len) if has_dynamic else None
                y_static = get_static_features(y, len(hp.windows), hp.stream_sizes, hp.has_dynamic_features)
                total_num_frames += float(sum(cpu_sorted_lengths))
                mask = sequence_mask(sorted_lengths).unsqueeze(-1)
                optimizer_g.zero_grad()
                optimizer_d.zero_grad()

                generator_input = torch.cat((x, z), -1) if z is not None else x
                y_hat, y_hat_static = apply_generator(model_g, generator_input, R, cpu_sorted_lengths)
                assert x.size(1) == y_hat.size(1)

                # spoofing rate against a frozen reference discriminator (train.py:549-558)
                if reference_discriminator is not None:
                    y_hat_static_ref = get_selected_static_stream(y_hat_static) \
                        if hp.adversarial_streams is not None else y_hat_static
                    target = reference_discriminator(y_hat_static_ref, lengths=cpu_sorted_lengths)
                    # accumulated on the device (float64 count); read once per epoch, no per-batch host synchronisation
                    regard_fake_as_natural = regard_fake_as_natural + ((target > 0.5).float() * mask).sum(dtype=torch.float64)

                if update_d:
                    loss_d, loss_fake_d, loss_real_d, _real, _fake = update_discriminator(
                        model_d, optimizer_d, x, y_static, y_hat_static, cpu_sorted_lengths, mask, phase)
                    running_loss["discriminator"] += loss_d
                    running_loss["loss_fake_d"] += loss_fake_d
                    running_loss["loss_real_d"] += loss_real_d
                    real_correct_count += _real
                    fake_correct_count += _fake

                if update_g:
                    adv_w = w_d * float(np.clip(E_loss_mge / E_loss_adv, 0, 1e+3))
                    loss_mse, loss_mge, loss_adv, loss_g = update_generator(
                        model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static,
                        adv_w, cpu_sorted_lengths, mask, phase, mse_w=mse_w, mge_w=mge_w)
                    running_loss["mse"] += loss_mse
                    running_loss["mge"] += loss_mge
                    running_loss["loss_adv"] += loss_adv
                    running_loss["generator"] += loss_g
                    distortions = compute_distortions(y_static, y_hat_static, Y_data_mean, Y_data_std, cpu_sorted_lengths)
                    for k, v in distortions.items():
                        running_metrics[k] = running_metrics.get(k, 0.0) + float(v)

            if update_d and update_g and phase == "train":
                E_loss_mge = (mse_w * running_loss["mse"] + mge_w * running_loss["mge"]) / N
                E_loss_adv = running_loss["loss_adv"] / N
                log_value("E(mge)", E_loss_mge, global_epoch)
                log_value("E(adv)", E_loss_adv, global_epoch)
                log_value("MGE/ADV loss weight", E_loss_mge / E_loss_adv, global_epoch)

            for ty, enabled in [("mse", update_g), ("mge", update_g), ("discriminator", update_d),
                                ("loss_real_d", update_d), ("loss_fake_d", update_d),
                                ("loss_adv", update_g and update_d), ("generator", update_g)]:
                if enabled:
                    log_value("{} {} loss".format(phase, ty), running_loss[ty] / N, global_epoch)
            for k, v in running_metrics.items():
                log_value("{} {} metric".format(phase, k), v / N, global_epoch)
            if update_d:
                log_value("Real {} acc".format(phase), real_correct_count / total_num_frames, global_epoch)
                log_value("Fake {} acc".format(phase), fake_correct_count / total_num_frames, global_epoch)
            if reference_discriminator is not None:
                log_value("{} spoofing rate".format(phase), float(regard_fake_as_natural) / total_num_frames, global_epoch)

        if global_epoch % checkpoint_interval == 0:
            for model, optimizer, enabled, name in [(model_g, optimizer_g, update_g, "Generator"),
                                                    (model_d, optimizer_d, update_d, "Discriminator")]:
                if enabled:
                    save_checkpoint(model, optimizer, global_epoch, checkpoint_dir, name)
    return 0
