"""Inference path of the reference's evaluation scripts, on the HIP forward kernels.

Only the model-side arithmetic is here; text front end (HTS labels, Merlin question sets) and the
vocoder (pyworld / pysptk) stay what they are in the reference -- CPU third-party code either side
of these calls:

    _generator_input(hp, x, seed)                     evaluation_tts.py:133-140
    predict_duration(model, feats, X_min, X_max, Y_mean, Y_std)
                                                      evaluation_tts.py:153-178 (gen_duration minus the label I/O)
    predict_acoustic(model, feats, X_min, X_max)      evaluation_tts.py:205-219
    gen_parameters(y_predicted, Y_mean, Y_std)        evaluation_tts.py:47-100 (MGE branch)
    vc_convert(model, mc, data_mean, data_std)        evaluation_vc.py:56-92 (between analysis and synthesis)

``hp_acoustic`` / ``hp_duration`` / ``hp_vc`` are module globals like in the reference scripts.
Checkpoints written by the reference load unchanged (``gantts_amd.train.load_checkpoint``): the
model classes keep torch's state_dict keys.
"""
import numpy as np
import torch

from . import hparams
from .data import minmax_scale
from .engine import StepEngine
from .multistream import get_static_stream_sizes
from .paramgen import unit_variance_mlpg_matrix_cuda

hp_acoustic = hparams.tts_acoustic
hp_duration = hparams.tts_duration
hp_vc = hparams.vc

_engines = {}


def _engine(hp):
    key = (tuple(hp.stream_sizes), tuple(hp.has_dynamic_features), len(hp.windows))
    eng = _engines.get(key)
    if eng is None:
        eng = _engines[key] = StepEngine(hp)
    return eng


def _generator_input(hp, x, seed=None):
    """Appends z ~ U[0,1) of width hp.generator_noise_dim when the generator was trained with noise."""
    if seed is not None:
        torch.manual_seed(seed)
    if hp.generator_add_noise:
        z = torch.rand(x.size(0), x.size(1), hp.generator_noise_dim).to(x.device)
        return torch.cat((x, z), -1)
    return x


def _forward(model, hp, feats):
    model.eval()
    x = torch.from_numpy(np.ascontiguousarray(feats)).float()
    xl = len(x)
    x = x.view(1, -1, x.size(-1))
    x = _generator_input(hp, x).cuda()
    out = model(x, [xl])
    return out.detach().cpu().numpy().reshape(-1, out.shape[-1])


def predict_duration(duration_model, linguistic_features, X_min, X_max, Y_mean, Y_std):
    """Rounded state durations (>= 1 frame) for one utterance; the statistics are the per-type dicts
    of the reference (``X_min["duration"]`` ...)."""
    ty = "duration"
    feats = minmax_scale(np.asarray(linguistic_features, dtype=np.float32), X_min[ty], X_max[ty], feature_range=(0.01, 0.99))
    pred = _forward(duration_model.cuda(), hp_duration, feats)
    pred = np.round(Y_std[ty] * pred + Y_mean[ty])
    pred[pred <= 0] = 1            # minimum state duration
    return pred


def predict_acoustic(acoustic_model, linguistic_features, X_min, X_max):
    """Normalised acoustic features (T, sum(stream_sizes)) for one utterance.  The reference passes
    ``hp_duration`` to _generator_input here as well (evaluation_tts.py:216); so does this."""
    ty = "acoustic"
    feats = minmax_scale(np.asarray(linguistic_features), X_min[ty], X_max[ty], feature_range=(0.01, 0.99))
    return _forward(acoustic_model.cuda(), hp_duration, feats)


def gen_parameters(y_predicted, Y_mean, Y_std, mge_training=True):
    """(mgc, lf0, vuv, bap): multi-stream MLPG on the normalised features (the banded device kernel,
    unit variance) followed by inverse scaling with statistics indexed in the static+dynamic domain.
    ``Y_mean`` / ``Y_std`` are the reference's dicts (``["acoustic"]``) or plain arrays."""
    if not mge_training:
        raise NotImplementedError("the reference's non-MGE branch multiplies a dict (evaluation_tts.py:86) and "
                                  "cannot run; GAN generators are MGE-trained")
    hp = hp_acoustic
    mean = Y_mean["acoustic"] if isinstance(Y_mean, dict) else Y_mean
    std = Y_std["acoustic"] if isinstance(Y_std, dict) else Y_std
    mgc_dim, lf0_dim, vuv_dim, bap_dim = hp.stream_sizes
    nw = len(hp.windows)
    lf0_0, vuv_0, bap_0 = mgc_dim, mgc_dim + lf0_dim, mgc_dim + lf0_dim + vuv_dim
    y = y_predicted if isinstance(y_predicted, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y_predicted))
    y = y.float().cuda().view(1, -1, y.shape[-1])
    T = y.size(1)
    static = _engine(hp).mlpg_forward(y, unit_variance_mlpg_matrix_cuda(hp.windows, T))[0].double().cpu().numpy()
    smgc, slf0, svuv, sbap = [int(v) for v in get_static_stream_sizes(hp.stream_sizes, hp.has_dynamic_features, nw)]
    mean, std = np.asarray(mean), np.asarray(std)
    mgc = static[:, :smgc] * std[:mgc_dim // nw] + mean[:mgc_dim // nw]
    lf0 = static[:, smgc:smgc + slf0] * std[lf0_0:lf0_0 + lf0_dim // nw] + mean[lf0_0:lf0_0 + lf0_dim // nw]
    bap = static[:, smgc + slf0 + svuv:] * std[bap_0:bap_0 + bap_dim // nw] + mean[bap_0:bap_0 + bap_dim // nw]
    vuv = y[0, :, vuv_0].cpu().numpy() * std[vuv_0] + mean[vuv_0]
    return mgc, lf0, vuv, bap


def vc_convert(model, mc, data_mean, data_std, diffvc=True):
    """Static mel-cepstrum conversion of one utterance.  ``mc`` (T, 3*order) = delta_features of the
    smoothed source mel-cepstrum (without c0).  Returns ``(inputs, outputs, mc_static_for_synthesis)``:
    the source statics, the converted statics, and what the reference hands to the synthesis filter
    (the difference to the source when ``diffvc``)."""
    hp = hp_vc
    model = model.cuda()
    model.eval()
    mc = np.asarray(mc, dtype=np.float32)
    T = mc.shape[0]
    static_dim = mc.shape[-1] // len(hp.windows)
    inputs = mc[:, :static_dim].copy()
    mc_scaled = torch.from_numpy(((mc - data_mean) / data_std).astype(np.float32)).cuda().view(1, T, -1)
    R = unit_variance_mlpg_matrix_cuda(hp.windows, T)
    if model.include_parameter_generation():
        _, y_hat_static = model(mc_scaled, R, lengths=[T])
    else:
        assert hp.has_dynamic_features is not None
        y_hat = model(mc_scaled, lengths=[T])
        y_hat_static = _engine(hp).mlpg_forward(y_hat, R)
    pred = y_hat_static.detach().cpu().numpy().reshape(-1, static_dim)
    pred = data_std[:static_dim] * pred + data_mean[:static_dim]
    outputs = pred.copy()
    if diffvc:
        pred = pred - mc[:, :static_dim]
    return inputs, outputs, pred
