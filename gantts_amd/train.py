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
