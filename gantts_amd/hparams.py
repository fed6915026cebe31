"""Hyper-parameter sets ``vc``, ``tts_duration`` and ``tts_acoustic``.

Same names, fields and default values as the reference's ``hparams.py`` (:16-83, :87-164,
:167-258) so ``getattr(hparams, name)`` / ``hp.parse("k=v,...")`` / ``hparams_debug_string(hp)``
(train.py:665-669) keep working, but without TensorFlow: ``HParams`` here is a small pure-Python
attribute bag with the subset of ``tf.contrib.training.HParams`` the reference touches
(``values()``, ``parse()``, attribute access, identity comparison ``hp == hparams.vc``).
"""
import ast
from os.path import dirname, join

import numpy as np


class HParams(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, "_names", [])
        for k, v in kwargs.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if name in self._names:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        self._names.append(name)
        object.__setattr__(self, name, value)

    def values(self):
        return {k: getattr(self, k) for k in self._names}

    def parse(self, spec):
        """``"batch_size=16,nepoch=50"`` -> in-place override; unknown names raise ValueError."""
        if not spec:
            return self
        for item in _split_top_level(spec):
            if "=" not in item:
                raise ValueError("Could not parse hparam assignment: %r" % item)
            name, raw = item.split("=", 1)
            name = name.strip()
            if name not in self._names:
                raise ValueError("Unknown hyperparameter: %s" % name)
            try:
                value = ast.literal_eval(raw.strip())
            except (ValueError, SyntaxError):
                value = raw.strip()
            cur = getattr(self, name)
            if isinstance(cur, bool) and not isinstance(value, bool):
                value = str(value).lower() in ("1", "true", "yes")
            elif isinstance(cur, float) and isinstance(value, int):
                value = float(value)
            object.__setattr__(self, name, value)
        return self

    def __repr__(self):
        return "HParams(%s)" % ", ".join("%s=%r" % (k, getattr(self, k)) for k in self._names)


def _split_top_level(spec):
    out, depth, cur = [], 0, ""
    for ch in spec:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def hparams_debug_string(params):
    vals = params.values()
    return "Hyperparameters:\n" + "\n".join("  %s: %s" % (k, vals[k]) for k in sorted(vals))


def _delta_windows(n):
    """static, delta, delta-delta windows as (left, right, coefficients)."""
    w = [(0, 0, np.array([1.0])),
         (1, 1, np.array([-0.5, 0.0, 0.5])),
         (1, 1, np.array([1.0, -2.0, 1.0]))]
    return w[:n]


_QUESTIONS = join(dirname(__file__), "nnmnkwii_gallery", "data", "questions-radio_dnn_416.hed")


def _adagrad(wd):
    return {"lr": 0.01, "weight_decay": wd}


def _adam():
    return {"lr": 0.001, "betas": (0.5, 0.9), "weight_decay": 0}


def _sru_generator(dropout):
    return {"in_dim": None, "out_dim": None, "num_hidden": 6, "hidden_dim": 512, "bidirectional": True,
            "dropout": dropout, "use_relu": 1, "rnn_dropout": 0.2, "last_sigmoid": False}


def _mlp_discriminator(in_dim, num_hidden, dropout):
    return {"in_dim": in_dim, "out_dim": 1, "num_hidden": num_hidden, "hidden_dim": 256,
            "dropout": dropout, "last_sigmoid": True}


_loader = dict(num_workers=1, pin_memory=True, cache_size=1200)

# voice conversion (reference hparams.py:16-83)
vc = HParams(
    name="vc",
    order=59, frame_period=5,
    windows=_delta_windows(3),
    stream_sizes=[59 * 3], has_dynamic_features=[True],
    adversarial_streams=[True], mask_nth_mgc_for_adv_loss=0,
    generator_add_noise=False, generator_noise_dim=200,
    generator="In2OutHighwayNet",
    generator_params={"in_dim": None, "out_dim": None, "num_hidden": 3, "hidden_dim": 512,
                      "static_dim": 59, "dropout": 0.5},
    optimizer_g="Adagrad", optimizer_g_params=_adagrad(0),
    discriminator_linguistic_condition=False,
    discriminator="MLP", discriminator_params=_mlp_discriminator(59, 2, 0.5),
    optimizer_d="Adagrad", optimizer_d_params=_adagrad(0),
    nepoch=200, lr_decay_schedule=False, lr_decay_epoch=10,
    batch_size=20, **_loader)

# TTS duration model (reference hparams.py:87-164)
tts_duration = HParams(
    name="duration",
    use_phone_alignment=False, subphone_features=None, add_frame_features=False,
    question_path=_QUESTIONS,
    windows=_delta_windows(1),
    stream_sizes=[5], has_dynamic_features=[False],
    recompute_delta_features=False,
    adversarial_streams=[True], mask_nth_mgc_for_adv_loss=0,
    generator="SRURNN", generator_add_noise=False, generator_noise_dim=200,
    generator_params=_sru_generator(0.0),
    optimizer_g="Adam", optimizer_g_params=_adam(),
    discriminator_linguistic_condition=True,
    discriminator="MLP", discriminator_params=_mlp_discriminator(None, 3, 0.0),
    optimizer_d="Adam", optimizer_d_params=_adam(),
    nepoch=200, lr_decay_schedule=False, lr_decay_epoch=25,
    batch_size=32, **_loader)

# TTS acoustic model (reference hparams.py:167-258)
tts_acoustic = HParams(
    name="acoustic",
    use_phone_alignment=False, subphone_features="full", add_frame_features=True,
    question_path=_QUESTIONS,
    order=59, frame_period=5, f0_floor=71.0, f0_ceil=700, use_harvest=True,
    windows=_delta_windows(3),
    f0_interpolation_kind="quadratic", mod_spec_smoothing=True, mod_spec_smoothing_cutoff=50,
    recompute_delta_features=False,
    stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],   # mgc, lf0, vuv, bap
    adversarial_streams=[True, False, False, False], mask_nth_mgc_for_adv_loss=2,
    generator_add_noise=False, generator_noise_dim=200,
    generator="SRURNN", generator_params=_sru_generator(0.2),
    optimizer_g="Adagrad", optimizer_g_params=_adagrad(1e-7),
    discriminator_linguistic_condition=True,
    discriminator="MLP", discriminator_params=_mlp_discriminator(None, 3, 0.5),
    optimizer_d="Adagrad", optimizer_d_params=_adagrad(1e-7),
    nepoch=200, lr_decay_schedule=False, lr_decay_epoch=25,
    batch_size=20, **_loader)
