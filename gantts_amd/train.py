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
    return eng.update_discriminator(model_d, optimizer_d, x, y_static, y_hat_static, mask, phase, eps)


def update_generator(model_g, model_d, optimizer_g,
                     x, y, y_hat, y_static, y_hat_static,
                     adv_w, lengths, mask, phase,
                     mse_w=None, mge_w=None, eps=1e-20):
    """Returns ``(loss_mse, loss_mge, loss_adv, loss_g)``."""
    eng = _engine_of(y_hat_static, model_g)
    return eng.update_generator(model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static,
                                adv_w, mask, phase, mse_w, mge_w, eps)


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
