"""Seeded, toolchain-independent definitions of the golden cases.

Shared by ``make_golden.py`` (drives the REAL reference in the build container),
``tests/test_oracle.py`` (pins the CPU restatement) and the ``-m gpu`` parity tests
(HIP engine vs oracle vs fixtures).  Everything random comes from
``numpy.random.RandomState`` so inputs/weights/masks are bit-identical everywhere.
"""
import math

import numpy as np

WINDOWS = [
    (0, 0, np.array([1.0])),
    (1, 1, np.array([-0.5, 0.0, 0.5])),
    (1, 1, np.array([1.0, -2.0, 1.0])),
]

# name -> case description.  "hp" mirrors the hp.* fields read on the step path.
CASES = {
    # cfg2 (headline) shape family at reduced size: TTS acoustic MLP G/D, conditioned D.
    "acoustic_mlp": dict(
        hp="tts_acoustic", B=4, T=40, din=425, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="MLP", in_dim=425, out_dim=187, num_hidden=3, hidden_dim=64,
               dropout=0.5, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=483, out_dim=1, num_hidden=3, hidden_dim=32,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=3, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # same, with dropout masks injected and a non-zero MSE weight
    "acoustic_mlp_dropout": dict(
        hp="tts_acoustic", B=3, T=33, din=425, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="MLP", in_dim=425, out_dim=187, num_hidden=3, hidden_dim=64,
               dropout=0.5, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=483, out_dim=1, num_hidden=3, hidden_dim=32,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=2, adv_w=0.7, mse_w=0.25, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # multi-stream adversarial selection (mgc + bap), unconditioned D, Adam
    "acoustic_multistream_adam": dict(
        hp="tts_acoustic", B=3, T=24, din=40, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, True], mask_nth_mgc=0, cond=False,
        g=dict(kind="MLP", in_dim=40, out_dim=187, num_hidden=2, hidden_dim=48,
               dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=61, out_dim=1, num_hidden=2, hidden_dim=24,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=3, steps=3, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # cfg1: VC In2OutHighwayNet (order 25) + MLP D, CPU plumbing config at reduced size
    "vc_in2out": dict(
        hp="vc", B=3, T=48, din=75, dout=75,
        stream_sizes=[75], has_dynamic_features=[True],
        adversarial_streams=[True], mask_nth_mgc=0, cond=False,
        g=dict(kind="In2OutHighwayNet", in_dim=75, out_dim=75, static_dim=25,
               num_hidden=3, hidden_dim=64, dropout=0.5),
        d=dict(kind="MLP", in_dim=25, out_dim=1, num_hidden=2, hidden_dim=32,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=0)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=0)),
        windows=3, steps=3, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # VC recurrent highway generator (gantts/models.py:72-118): returns its input as y_hat, BiLSTM body
    "vc_in2out_rnn": dict(
        hp="vc", B=3, T=40, din=75, dout=75,
        stream_sizes=[75], has_dynamic_features=[True],
        adversarial_streams=[True], mask_nth_mgc=0, cond=False,
        g=dict(kind="In2OutRNNHighwayNet", in_dim=75, out_dim=75, static_dim=25,
               num_hidden=2, hidden_dim=24, bidirectional=True, dropout=0.5),
        d=dict(kind="MLP", in_dim=25, out_dim=1, num_hidden=2, hidden_dim=32,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=0)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=0)),
        windows=3, steps=3, adv_w=1.0, mse_w=1.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # wider discriminator (hidden 128, 3 layers; fixture name kept from round 1): injected dropout masks,
    # conditioned D, ragged panel tail (2*B*T = 138 rows), real/fake halves sharing the x part of layer 0
    "acoustic_chain_d": dict(
        hp="tts_acoustic", B=3, T=23, din=30, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="MLP", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=32,
               dropout=0.5, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=88, out_dim=1, num_hidden=3, hidden_dim=128,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=3, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # same without conditioning and without dropout (eval-mode modules), Adam
    "acoustic_chain_d_uncond": dict(
        hp="tts_acoustic", B=2, T=40, din=30, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=False,
        g=dict(kind="MLP", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=32,
               dropout=0.5, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=58, out_dim=1, num_hidden=2, hidden_dim=128,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # duration model: no dynamic features (R=None), Adam, conditioned D
    "duration_mlp": dict(
        hp="tts_duration", B=5, T=17, din=60, dout=5,
        stream_sizes=[5], has_dynamic_features=[False],
        adversarial_streams=[True], mask_nth_mgc=0, cond=True,
        g=dict(kind="MLP", in_dim=60, out_dim=5, num_hidden=2, hidden_dim=32,
               dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=65, out_dim=1, num_hidden=3, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=1, steps=3, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # cfg3 family at reduced size: BiLSTM generator (packed variable lengths) + conditioned MLP D
    "acoustic_lstm": dict(
        hp="tts_acoustic", B=4, T=21, din=30, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="LSTMRNN", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=16,
               bidirectional=True, dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=88, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # a RECURRENT discriminator (train.py:773-774: getattr(gantts.models, hp.discriminator) -- any model class may sit there): bidirectional
    # two-layer LSTMRNN(out_dim=1, last_sigmoid=True) scoring [x | static adversarial features] frame by frame, packed by `lengths`
    "acoustic_lstm_d": dict(
        hp="tts_acoustic", B=4, T=21, din=30, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="MLP", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=32,
               dropout=0.0, last_sigmoid=False),
        d=dict(kind="LSTMRNN", in_dim=88, out_dim=1, num_hidden=2, hidden_dim=12,
               bidirectional=True, dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # unidirectional single-layer variant under the GRURNN name (an nn.LSTM as attribute `gru`)
    "acoustic_grurnn_uni": dict(
        hp="tts_acoustic", B=3, T=17, din=20, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="GRURNN", in_dim=20, out_dim=187, num_hidden=1, hidden_dim=24,
               bidirectional=False, dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=78, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # curriculum stages (train_gan.sh): D warm-up only / G only with w_d=0
    "acoustic_d_warmup": dict(
        hp="tts_acoustic", B=3, T=20, din=30, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="MLP", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=32,
               dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=88, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=False),
    "acoustic_g_only": dict(
        hp="tts_acoustic", B=3, T=20, din=30, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="MLP", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=32,
               dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=88, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=2, adv_w=0.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=False, update_g=True),
}

# generators that end in a sigmoid (models.py:141, 213: `last_sigmoid=True`; train.py:773-774 builds whatever hparams name):
# the gradient at y_hat passes through s (1 - s) before the last layer's products
CASES["acoustic_mlp_sigmoid_g"] = dict(
    CASES["acoustic_mlp"], B=3, T=29, steps=2, mse_w=0.5,
    g=dict(kind="MLP", in_dim=425, out_dim=187, num_hidden=2, hidden_dim=48, dropout=0.0, last_sigmoid=True),
    d=dict(kind="MLP", in_dim=483, out_dim=1, num_hidden=2, hidden_dim=32, dropout=0.0, last_sigmoid=True))
CASES["acoustic_lstm_sigmoid_g"] = dict(
    CASES["acoustic_lstm"], g=dict(CASES["acoustic_lstm"]["g"], last_sigmoid=True))


# Cases with NO reference-generated fixture: the SRU cell is third-party code that is neither vendored in
# the reference nor runnable here (CUDA-only), so these are checked HIP-vs-oracle only (parity unpinned).
ORACLE_ONLY_CASES = {
    # nn.LSTM inter-layer dropout (training mode) cannot be mask-injected into the real reference
    # (it lives inside _VF.lstm): HIP vs oracle with injected masks; eval-mode parity is in CASES.
    "acoustic_lstm_dropout": dict(
        hp="tts_acoustic", B=4, T=18, din=22, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="LSTMRNN", in_dim=22, out_dim=187, num_hidden=3, hidden_dim=12,
               bidirectional=True, dropout=0.3, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=80, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # recurrent generator AND recurrent discriminator, both with nn.LSTM inter-layer dropout in training mode (three injected D passes:
    # D(real), D(fake) of the D step -- run as ONE batch of 2B sequences by the engine, each half with its own masks -- and D(fake) of the
    # G step); unidirectional 3-layer D
    "acoustic_lstm_d_dropout": dict(
        hp="tts_acoustic", B=4, T=18, din=22, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="LSTMRNN", in_dim=22, out_dim=187, num_hidden=2, hidden_dim=12,
               bidirectional=True, dropout=0.3, last_sigmoid=False),
        d=dict(kind="LSTMRNN", in_dim=80, out_dim=1, num_hidden=3, hidden_dim=20,
               bidirectional=False, dropout=0.4, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.5, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # cfg3 (BASELINE.json configs[2]) at its real widths -- BiLSTM 3 x 256 generator, 425 -> 187, conditioned MLP D
    # 483 -> 256 x 3 -> 1, B = 32 (two 16-sequence batch tiles), variable lengths -- with T cut to what the CPU oracle's
    # python time loop finishes in seconds: every code path of the persistent recurrence kernels that depends on the
    # widths (64 workgroups per group, K split over the waves, the 4H-wide backward exchange) is the full-size one
    "acoustic_lstm_at_size": dict(
        hp="tts_acoustic", B=32, T=96, din=425, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="LSTMRNN", in_dim=425, out_dim=187, num_hidden=3, hidden_dim=256,
               bidirectional=True, dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=483, out_dim=1, num_hidden=3, hidden_dim=256,
               dropout=0.0, last_sigmoid=True),
        # warm accumulators (as after some training): the first step of a COLD Adagrad is lr * g / (|g| + 1e-10), which
        # turns the relative rounding error of every near-zero gradient entry into an absolute parameter error
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # hidden width in (256, 512]: the 512-unit instances of the persistent recurrence kernels (38 / 19 workgroups per
    # group, units padded to 304), three 8-sequence batch tiles per direction with a ragged last one
    "acoustic_lstm_wide": dict(
        hp="tts_acoustic", B=20, T=24, din=60, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="LSTMRNN", in_dim=60, out_dim=187, num_hidden=2, hidden_dim=300,
               bidirectional=True, dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=118, out_dim=1, num_hidden=2, hidden_dim=32,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    "vc_in2out_rnn_dropout": dict(
        hp="vc", B=3, T=33, din=75, dout=75,
        stream_sizes=[75], has_dynamic_features=[True],
        adversarial_streams=[True], mask_nth_mgc=0, cond=False,
        g=dict(kind="In2OutRNNHighwayNet", in_dim=75, out_dim=75, static_dim=25,
               num_hidden=3, hidden_dim=20, bidirectional=False, dropout=0.5),
        d=dict(kind="MLP", in_dim=25, out_dim=1, num_hidden=2, hidden_dim=32,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # cfg4 family at reduced size: bidirectional SRU generator (default hparams generator), 3 dynamic streams
    "vc_sru_multistream": dict(
        hp="tts_acoustic", B=3, T=19, din=20, dout=183,
        stream_sizes=[177, 3, 3], has_dynamic_features=[True, True, True],
        adversarial_streams=[True, False, False], mask_nth_mgc=0, cond=False,
        g=dict(kind="SRURNN", in_dim=20, out_dim=183, num_hidden=3, hidden_dim=12,
               bidirectional=True, dropout=0.0, last_sigmoid=False, use_relu=1, rnn_dropout=0.0),
        d=dict(kind="MLP", in_dim=59, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
    # the hparams-default generator family (hparams.py:111-124, 211-222) with BOTH variational dropouts on, masks
    # injected: bidirectional, first layer k = 4 (n_in != ncols), upper layers k = 3 (highway gradient by-passes the
    # input dropout), relu
    "acoustic_sru_dropout": dict(
        hp="tts_acoustic", B=5, T=26, din=30, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="SRURNN", in_dim=30, out_dim=187, num_hidden=3, hidden_dim=20,
               bidirectional=True, dropout=0.2, last_sigmoid=False, use_relu=1, rnn_dropout=0.2),
        d=dict(kind="MLP", in_dim=88, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.3, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # cfg4 / hparams-default generator at its real widths (hparams.py:211-222: SRU 6 x 512 bidirectional, relu,
    # dropout 0.2, rnn_dropout 0.2; 425 -> 187) with the conditioned 3 x 256 discriminator, B = 16; T cut to what
    # the oracle's python time loop finishes in seconds.  Parity unpinned (un-vendored SRU), masks injected.
    "acoustic_sru_at_size": dict(
        hp="tts_acoustic", B=16, T=64, din=425, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="SRURNN", in_dim=425, out_dim=187, num_hidden=6, hidden_dim=512,
               bidirectional=True, dropout=0.2, last_sigmoid=False, use_relu=1, rnn_dropout=0.2),
        d=dict(kind="MLP", in_dim=483, out_dim=1, num_hidden=3, hidden_dim=256,
               dropout=0.5, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        windows=3, steps=1, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # unidirectional tanh SRU with both dropouts, k = 3 in every layer (n_in == ncols from the start)
    "acoustic_sru_uni_k3_dropout": dict(
        hp="tts_acoustic", B=3, T=19, din=16, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="SRURNN", in_dim=16, out_dim=187, num_hidden=3, hidden_dim=16,
               bidirectional=False, dropout=0.3, last_sigmoid=False, use_relu=0, rnn_dropout=0.25),
        d=dict(kind="MLP", in_dim=74, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.0, mge_w=1.0, dropout_on=True,
        update_d=True, update_g=True),
    # unidirectional tanh SRU whose first layer already has n_in == ncols (k = 3 everywhere)
    "acoustic_sru_uni_k3": dict(
        hp="tts_acoustic", B=2, T=23, din=16, dout=187,
        stream_sizes=[180, 3, 1, 3], has_dynamic_features=[True, True, False, True],
        adversarial_streams=[True, False, False, False], mask_nth_mgc=2, cond=True,
        g=dict(kind="SRURNN", in_dim=16, out_dim=187, num_hidden=2, hidden_dim=16,
               bidirectional=False, dropout=0.0, last_sigmoid=False, use_relu=0, rnn_dropout=0.0),
        d=dict(kind="MLP", in_dim=74, out_dim=1, num_hidden=2, hidden_dim=16,
               dropout=0.0, last_sigmoid=True),
        opt_g=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        windows=3, steps=2, adv_w=1.0, mse_w=0.5, mge_w=1.0, dropout_on=False,
        update_d=True, update_g=True),
}
# gate pre-activations of +-100 (fast_exp must saturate, not produce inf * 0): the persistent LSTM forward
# kernel's fast gate functions and the SRU scans' fast sigmoids against the oracle's torch.sigmoid / tanh
ORACLE_ONLY_CASES["acoustic_lstm_saturated"] = dict(CASES["acoustic_lstm"], saturate_gates=True)
ORACLE_ONLY_CASES["acoustic_sru_saturated"] = dict(ORACLE_ONLY_CASES["acoustic_sru_uni_k3"], saturate_gates=True)
ORACLE_ONLY_CASES["acoustic_sru_bi_saturated"] = dict(ORACLE_ONLY_CASES["acoustic_sru_dropout"], saturate_gates=True)


def param_shapes(spec):
    """(name, shape) in state_dict / parameters() order for a model spec."""
    kind = spec["kind"]
    out = []
    if kind == "MLP":
        ins = [spec["in_dim"]] + [spec["hidden_dim"]] * (spec["num_hidden"] - 1)
        for i, n_in in enumerate(ins):
            out += [("layers.%d.weight" % i, (spec["hidden_dim"], n_in)),
                    ("layers.%d.bias" % i, (spec["hidden_dim"],))]
        out += [("last_linear.weight", (spec["out_dim"], spec["hidden_dim"])),
                ("last_linear.bias", (spec["out_dim"],))]
    elif kind == "In2OutHighwayNet":
        sd = spec["static_dim"]
        out += [("T.weight", (sd, sd)), ("T.bias", (sd,))]
        ins = [spec["in_dim"]] + [spec["hidden_dim"]] * (spec["num_hidden"] - 1)
        for i, n_in in enumerate(ins):
            out += [("H.%d.weight" % i, (spec["hidden_dim"], n_in)),
                    ("H.%d.bias" % i, (spec["hidden_dim"],))]
        out += [("last_linear.weight", (spec["out_dim"], spec["hidden_dim"])),
                ("last_linear.bias", (spec["out_dim"],))]
    elif kind in ("LSTMRNN", "GRURNN", "In2OutRNNHighwayNet"):
        prefix = "gru" if kind == "GRURNN" else "lstm"
        if kind == "In2OutRNNHighwayNet":
            out += [("T.weight", (spec["static_dim"],) * 2), ("T.bias", (spec["static_dim"],))]
        H, dirs = spec["hidden_dim"], 2 if spec["bidirectional"] else 1
        for l in range(spec["num_hidden"]):
            n_in = spec["in_dim"] if l == 0 else H * dirs
            for d in range(dirs):
                sfx = "_l%d%s" % (l, "_reverse" if d else "")
                out += [("%s.weight_ih%s" % (prefix, sfx), (4 * H, n_in)), ("%s.weight_hh%s" % (prefix, sfx), (4 * H, H)),
                        ("%s.bias_ih%s" % (prefix, sfx), (4 * H,)), ("%s.bias_hh%s" % (prefix, sfx), (4 * H,))]
        out += [("hidden2out.weight", (spec["out_dim"], H * dirs)), ("hidden2out.bias", (spec["out_dim"],))]
    elif kind == "SRURNN":
        H, dirs = spec["hidden_dim"], 2 if spec["bidirectional"] else 1
        ncols = H * dirs
        for l in range(spec["num_hidden"]):
            n_in = spec["in_dim"] if l == 0 else ncols
            k = 3 if n_in == ncols else 4
            out += [("gru.rnn_lst.%d.weight" % l, (n_in, ncols * k)), ("gru.rnn_lst.%d.bias" % l, (2 * ncols,))]
        out += [("hidden2out.weight", (spec["out_dim"], ncols)), ("hidden2out.bias", (spec["out_dim"],))]
    else:
        raise ValueError(kind)
    return out


def make_weights(spec, seed, saturate_gates=False):
    """U(+-1/sqrt(fan_in)) like nn.Linear's default init (U(+-1/sqrt(H)) for LSTM tensors), from numpy RandomState.
    saturate_gates: every third element of the recurrent cells' gate biases becomes +-100, so gate pre-activations of
    about +-100 occur on every layer (sigmoid / tanh must saturate to 0 / 1 / +-1 like torch's, not overflow)."""
    rs = np.random.RandomState(seed)
    sd = {}
    shapes = param_shapes(spec)
    fan_in = None
    for name, shape in shapes:
        if name.endswith("weight"):
            fan_in = shape[1]
        if ".weight_ih" in name or ".weight_hh" in name or ".bias_ih" in name or ".bias_hh" in name:
            fan_in = spec["hidden_dim"]
        k = 1.0 / math.sqrt(fan_in)
        if "rnn_lst" in name:        # SRU: U(+-sqrt(3/n_in)) weights, small non-zero biases for the test
            k = math.sqrt(3.0 / shape[0]) if name.endswith("weight") else 0.5
        sd[name] = ((rs.rand(*shape) * 2 - 1) * k).astype(np.float32)
        if saturate_gates and (".bias_ih" in name or ("rnn_lst" in name and name.endswith("bias"))):
            idx = np.arange(0, shape[0], 3)
            sd[name][idx] = np.where((idx // 3) % 2 == 0, -100.0, 100.0).astype(np.float32)
    return sd


def make_lengths(B, T, seed):
    rs = np.random.RandomState(seed)
    lens = np.sort(rs.randint(max(1, T // 2), T + 1, size=B))[::-1].copy()
    lens[0] = T
    return lens.astype(np.int64)


def make_batch(case, seed=1234):
    """x in [0.01,0.99] (min-max scaled linguistic features), y ~ N(0,1) with a
    standardised binary vuv column; zero beyond each length (train.py:139-142)."""
    B, T, din, dout = case["B"], case["T"], case["din"], case["dout"]
    rs = np.random.RandomState(seed)
    x = (0.01 + 0.98 * rs.rand(B, T, din)).astype(np.float32)
    y = rs.randn(B, T, dout).astype(np.float32)
    if case["stream_sizes"] == [180, 3, 1, 3]:
        v = (rs.rand(B, T) > 0.5).astype(np.float32)
        y[:, :, 183] = (v - 0.5) / 0.5
    lengths = make_lengths(B, T, seed + 1)
    for b, n in enumerate(lengths):
        x[b, n:] = 0
        y[b, n:] = 0
    return x, y, lengths


def hidden_sites(spec):
    """Widths of the dropout sites of one forward pass."""
    if spec["kind"] in ("LSTMRNN", "GRURNN", "In2OutRNNHighwayNet"):   # nn.LSTM: between layers only
        return [spec["hidden_dim"] * (2 if spec["bidirectional"] else 1)] * (spec["num_hidden"] - 1)
    return [spec["hidden_dim"]] * spec["num_hidden"]


def make_dropout_masks(case, step, seed=99):
    """Per step: masks for G forward (1 pass), D real, D fake (D step), D fake (G step).
    Returned in the order the reference consumes nn.Dropout calls."""
    rs = np.random.RandomState(seed + 1000 * step)
    B, T = case["B"], case["T"]

    def draw(spec):
        if spec["kind"] == "SRURNN":      # variational masks, one per sequence, in the order the cell draws them
            ncols = spec["hidden_dim"] * (2 if spec["bidirectional"] else 1)
            out = []
            for l in range(spec["num_hidden"]):
                n_in = spec["in_dim"] if l == 0 else ncols
                if spec["rnn_dropout"] > 0:
                    out.append((rs.rand(B, n_in) >= spec["rnn_dropout"]).astype(np.float32))
                if spec["dropout"] > 0 and l + 1 < spec["num_hidden"]:
                    out.append((rs.rand(B, ncols) >= spec["dropout"]).astype(np.float32))
            return out
        if spec["dropout"] <= 0:
            return []
        return [(rs.rand(B, T, h) >= spec["dropout"]).astype(np.float32) for h in hidden_sites(spec)]

    g = draw(case["g"])
    d = draw(case["d"]) + draw(case["d"]) + draw(case["d"])
    return g, d


# ---------------------------------------------------------------------------------------------
# Distortion metrics (reference train.py:358-432): inputs for tests/golden/distortions.npz
# ---------------------------------------------------------------------------------------------
DISTORTION_CASES = {
    "acoustic_f32": dict(name="acoustic", hp="tts_acoustic", B=4, T=37, stream_sizes=[180, 3, 1, 3],
                         has_dynamic_features=[True, True, False, True], windows=3, stats="float32", voiced=True),
    "acoustic_f64": dict(name="acoustic", hp="tts_acoustic", B=3, T=29, stream_sizes=[180, 3, 1, 3],
                         has_dynamic_features=[True, True, False, True], windows=3, stats="float64", voiced=True),
    "acoustic_unvoiced": dict(name="acoustic", hp="tts_acoustic", B=2, T=16, stream_sizes=[180, 3, 1, 3],
                              has_dynamic_features=[True, True, False, True], windows=3, stats="float32", voiced=False),
    "acoustic_small_streams": dict(name="acoustic", hp="tts_acoustic", B=3, T=23, stream_sizes=[30, 3, 1, 15],
                                   has_dynamic_features=[True, True, False, True], windows=3, stats="float32", voiced=True),
    "duration": dict(name="duration", hp="tts_duration", B=5, T=21, stream_sizes=[5],
                     has_dynamic_features=[False], windows=1, stats="float32", voiced=True),
    "vc": dict(name="vc", hp="vc", B=3, T=41, stream_sizes=[75], has_dynamic_features=[True],
               windows=3, stats="float64", voiced=True),
}


def static_sizes(case):
    nw = case["windows"]
    return [s // nw if d else s for s, d in zip(case["stream_sizes"], case["has_dynamic_features"])]


def make_distortion_inputs(case, seed=4321):
    """(y_static, y_hat_static, Y_mean, Y_std, lengths): normalised features, statistics in the
    static+dynamic domain; the vuv column inverse-scales to values scattered around 0.5."""
    rs = np.random.RandomState(seed)
    B, T = case["B"], case["T"]
    Ds, D = sum(static_sizes(case)), sum(case["stream_sizes"])
    y = rs.randn(B, T, Ds).astype(np.float32)
    yh = (y + 0.4 * rs.randn(B, T, Ds)).astype(np.float32)
    mean = rs.randn(D) * 0.5
    std = 0.5 + 1.5 * rs.rand(D)
    if case["name"] == "acoustic":
        mgc, lf0, vuv, bap = case["stream_sizes"]
        smgc, slf0 = static_sizes(case)[:2]
        mean[mgc:mgc + lf0] = 5.0 + 0.1 * rs.randn(lf0)      # log-F0
        std[mgc:mgc + lf0] = 0.25
        mean[mgc + lf0], std[mgc + lf0] = 0.55, 0.45            # vuv
        if not case["voiced"]:
            y[:, :, smgc + slf0] = -2.0                         # reference: every frame unvoiced
        # a few values that land exactly on / next to the 0.5 threshold after inverse scaling
        yh[0, :4, smgc + slf0] = np.array([(0.5 - 0.55) / 0.45, -0.1111111, -0.1111112, -0.111111], np.float32)
    lengths = make_lengths(B, T, seed + 1)
    dt = np.float32 if case["stats"] == "float32" else np.float64
    return y, yh, mean.astype(dt), std.astype(dt), lengths


# ---------------------------------------------------------------------------------------------
# Whole training loop (reference train.py:435-643) on a tiny in-memory dataset: tests/golden/train_loop_*.npz
# Dropout is 0 everywhere (train_loop switches the models to .train()) and no generator noise, so the
# run is deterministic; everything else (length sort, curriculum weight, lr schedule, test phase,
# distortion metrics, accuracies, spoofing rate) is exercised as the reference does it.
# ---------------------------------------------------------------------------------------------
TRAIN_LOOP_CASES = {
    "train_loop_acoustic": dict(
        hp="tts_acoustic", din=30, dout=187, stream_sizes=[180, 3, 1, 3],
        has_dynamic_features=[True, True, False, True], adversarial_streams=[True, False, False, False],
        mask_nth_mgc=2, cond=False, windows=3, nepoch=3, lr_decay_schedule=True, lr_decay_epoch=2,
        g=dict(kind="MLP", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=32, dropout=0.0, last_sigmoid=False),
        # the spoofing-rate branch feeds the reference D without the linguistic condition (train.py:551-555):
        # it only runs with discriminator_linguistic_condition=False
        d=dict(kind="MLP", in_dim=58, out_dim=1, num_hidden=2, hidden_dim=16, dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)), opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
        batches=dict(train=[(4, 24), (4, 19), (3, 22)], test=[(3, 17), (2, 21)]),
        w_d=1.0, mse_w=0.0, mge_w=1.0, update_d=True, update_g=True, reference_d=True, stats="float32"),
    "train_loop_vc": dict(
        hp="vc", din=75, dout=75, stream_sizes=[75], has_dynamic_features=[True], adversarial_streams=None,
        mask_nth_mgc=0, cond=False, windows=3, nepoch=2, lr_decay_schedule=False, lr_decay_epoch=10, order=25,
        g=dict(kind="In2OutHighwayNet", in_dim=75, out_dim=75, static_dim=25, num_hidden=2, hidden_dim=32, dropout=0.0),
        d=dict(kind="MLP", in_dim=25, out_dim=1, num_hidden=2, hidden_dim=16, dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=0)), opt_d=("Adagrad", dict(lr=0.01, weight_decay=0)),
        batches=dict(train=[(3, 30), (3, 26)], test=[(2, 23)]),
        w_d=1.0, mse_w=0.0, mge_w=1.0, update_d=True, update_g=True, reference_d=False, stats="float64"),
    "train_loop_d_warmup": dict(
        hp="tts_acoustic", din=30, dout=187, stream_sizes=[180, 3, 1, 3],
        has_dynamic_features=[True, True, False, True], adversarial_streams=[True, False, False, False],
        mask_nth_mgc=2, cond=True, windows=3, nepoch=2, lr_decay_schedule=False, lr_decay_epoch=25,
        g=dict(kind="MLP", in_dim=30, out_dim=187, num_hidden=2, hidden_dim=32, dropout=0.0, last_sigmoid=False),
        d=dict(kind="MLP", in_dim=88, out_dim=1, num_hidden=2, hidden_dim=16, dropout=0.0, last_sigmoid=True),
        opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7)), opt_d=("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0)),
        batches=dict(train=[(4, 20), (2, 25)], test=[(3, 18)]),
        w_d=1.0, mse_w=0.0, mge_w=1.0, update_d=True, update_g=False, reference_d=False, stats="float32"),
}


def make_train_loop_data(case, seed=777):
    """{"train": [(x, y, lengths), ...], "test": [...]} (collated, UNsorted lengths, zero padded) and the
    output statistics in the static+dynamic domain."""
    rs = np.random.RandomState(seed)
    din, dout = case["din"], case["dout"]
    out = {}
    for phase in ("train", "test"):
        out[phase] = []
        for (B, T) in case["batches"][phase]:
            lens = rs.randint(max(1, T // 2), T + 1, size=B)
            lens[rs.randint(B)] = T
            x = (0.01 + 0.98 * rs.rand(B, T, din)).astype(np.float32)
            y = rs.randn(B, T, dout).astype(np.float32)
            if case["stream_sizes"] == [180, 3, 1, 3]:
                y[:, :, 183] = ((rs.rand(B, T) > 0.5).astype(np.float32) - 0.5) / 0.5
            for b, n in enumerate(lens):
                x[b, n:] = 0
                y[b, n:] = 0
            out[phase].append((x, y, lens.astype(np.int64)))
    dt = np.float32 if case["stats"] == "float32" else np.float64
    mean = (rs.randn(dout) * 0.3).astype(dt)
    std = (0.5 + rs.rand(dout)).astype(dt)
    if case["stream_sizes"] == [180, 3, 1, 3]:
        mean[180], std[180], mean[183], std[183] = 5.0, 0.2, 0.5, 0.5
    return out, mean, std


# ---------------------------------------------------------------------------------------------
# Inference path (reference evaluation_tts.py / evaluation_vc.py): tests/golden/inference.npz
# ---------------------------------------------------------------------------------------------
INFERENCE = dict(
    T_acoustic=57, T_states=23, din=40, T_vc=44,
    acoustic=dict(kind="LSTMRNN", in_dim=40, out_dim=187, num_hidden=2, hidden_dim=24, bidirectional=True,
                  dropout=0.0, last_sigmoid=False),
    acoustic_mlp=dict(kind="MLP", in_dim=40, out_dim=187, num_hidden=2, hidden_dim=48, dropout=0.5, last_sigmoid=False),
    duration=dict(kind="MLP", in_dim=40, out_dim=5, num_hidden=2, hidden_dim=32, dropout=0.5, last_sigmoid=False),
    vc=dict(kind="In2OutHighwayNet", in_dim=75, out_dim=75, static_dim=25, num_hidden=2, hidden_dim=40, dropout=0.5),
)


def make_inference_inputs(seed=2468):
    rs = np.random.RandomState(seed)
    I = INFERENCE
    out = {}
    for ty, T in (("acoustic", I["T_acoustic"]), ("duration", I["T_states"])):
        lo = rs.randn(I["din"]) - 2.0
        hi = lo + 0.5 + 3.0 * rs.rand(I["din"])
        out["X_min_" + ty], out["X_max_" + ty] = lo, hi
        out["feats_" + ty] = (lo + (hi - lo) * rs.rand(T, I["din"])).astype(np.float32)
    out["Y_mean_acoustic"] = rs.randn(187) * 0.4
    out["Y_std_acoustic"] = 0.5 + rs.rand(187)
    out["Y_mean_duration"] = 4.0 + rs.rand(5) * 6.0
    out["Y_std_duration"] = 1.0 + 4.0 * rs.rand(5)
    c = np.cumsum(rs.randn(I["T_vc"], 25) * 0.2, axis=0)               # smooth-ish static mel-cepstrum
    out["vc_static"] = c.astype(np.float32)
    out["vc_mean"] = rs.randn(75) * 0.3
    out["vc_std"] = 0.5 + rs.rand(75)
    return out
