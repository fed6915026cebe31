"""Pins the CPU oracle (oracle/gantts_oracle.py): against the golden fixtures generated
from the REAL reference (tests/golden/*.npz), against torch.optim, against the
self-consistency properties of MLPG, and -- when the reference tree is present (build
container only) -- live against the reference's own functions."""
import os

import numpy as np
import pytest
import torch

import cases as C
import gantts_oracle as O
from oracle_runner import run_oracle_case

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(C.CASES))
def test_oracle_matches_reference_golden(name):
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = run_oracle_case(C.CASES[name])
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        g, o = gold[k], np.asarray(got[k])
        if "scalars" in k:
            np.testing.assert_allclose(o, g, rtol=2e-6, atol=1e-7, err_msg=k)
            if k.startswith("d_scalars"):
                assert o[3] == g[3] and o[4] == g[4], k  # counts exact
        else:
            np.testing.assert_allclose(o, g, rtol=1e-5, atol=2e-6, err_msg=k)


def test_mlpg_matrix_is_left_inverse_of_window_stack():
    for T in (5, 17, 64):
        W = np.vstack([O._window_matrix(l, u, c, T) for (l, u, c) in C.WINDOWS])
        R = O.unit_variance_mlpg_matrix(C.WINDOWS, T).astype(np.float64)
        np.testing.assert_allclose(R @ W, np.eye(T), atol=1e-5)


def test_mlpg_recovers_static_from_consistent_deltas():
    T, d = 50, 4
    rs = np.random.RandomState(0)
    c = rs.randn(T, d)
    feats = np.concatenate([O._window_matrix(l, u, w, T) @ c for (l, u, w) in C.WINDOWS], axis=1)
    R = torch.from_numpy(O.unit_variance_mlpg_matrix(C.WINDOWS, T))
    out = O.unit_variance_mlpg(R, torch.from_numpy(feats.astype(np.float32)))
    np.testing.assert_allclose(out.numpy(), c, atol=1e-4)


def test_mlpg_matrix_is_numerically_banded():
    R = O.unit_variance_mlpg_matrix(C.WINDOWS, 256)
    T = 256
    t = np.arange(T)
    for w in range(3):
        blk = np.abs(R[:, w * T:(w + 1) * T])
        far = np.abs(t[:, None] - t[None, :]) > 24
        assert blk[far].max() < 1e-10


def test_stream_index_arithmetic_reference_cases():
    # mirrors reference tests/test_gantts.py:60-129 (exact equality of selected columns)
    ss = [60, 1, 1, 1]
    x = torch.arange(0, 63).expand(2, 5, 63)
    assert O.select_streams(x, ss, [True, True, True, True]).shape == (2, 5, 63)
    assert O.select_streams(x, ss, [True, False, False, True]).shape == (2, 5, 61)
    assert (O.select_streams(x, ss, [False, False, False, True]).squeeze(-1) == x[:, :, -1]).all()
    assert (O.select_streams(x, ss, [False, True, False, False]).squeeze(-1) == x[:, :, -3]).all()
    assert list(O.get_static_stream_sizes([180, 3, 1, 3], [True, True, False, True], 3)) == [60, 1, 1, 1]
    y = torch.rand(2, 5, 187)
    s = O.get_static_features(y, 3, [180, 3, 1, 3], [True, True, False, True])
    assert s.shape == (2, 5, 63)
    assert torch.equal(s[:, :, :60], y[:, :, :60]) and torch.equal(s[:, :, 61], y[:, :, 183])
    assert O.adversarial_columns([180, 3, 1, 3], [True, True, False, True], 3,
                                 [True, False, False, False], 2) == list(range(2, 60))
    assert O.adversarial_columns([180, 3, 1, 3], [True, True, False, True], 3,
                                 [True, False, False, True], 0) == list(range(60)) + [62]


def test_multi_stream_mlpg_matches_per_stream_and_vuv_passthrough():
    # reference tests/test_gantts.py:132-163
    T = 30
    R = torch.from_numpy(O.unit_variance_mlpg_matrix(C.WINDOWS, T))
    x = torch.rand(4, T, 187)
    y = O.multi_stream_mlpg(x, R, [180, 3, 1, 3], [True, True, False, True])
    assert y.shape == (4, T, 63)
    assert torch.equal(O.unit_variance_mlpg(R, x[:, :, :180]), y[:, :, :60])
    assert torch.equal(O.unit_variance_mlpg(R, x[:, :, 180:183]).squeeze(-1), y[:, :, 60])
    assert torch.equal(x[:, :, 183], y[:, :, 61])
    assert torch.equal(O.unit_variance_mlpg(R, x[:, :, 184:187]).squeeze(-1), y[:, :, 62])
    with pytest.raises(RuntimeError):
        O.multi_stream_mlpg(x[:, :, :100], R, [180, 3, 1, 3], [True, True, False, True])


@pytest.mark.parametrize("kind,kw", [("Adagrad", dict(lr=0.01, weight_decay=1e-7)),
                                     ("Adam", dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0))])
def test_optimizer_restatement_matches_torch_optim(kind, kw):
    torch.manual_seed(0)
    p0 = [torch.randn(7, 5), torch.randn(5)]
    a = [p.clone().requires_grad_(True) for p in p0]
    b = [p.clone().requires_grad_(True) for p in p0]
    oa = O.make_optimizer(kind, a, **kw)
    ob = getattr(torch.optim, kind)(b, **kw)
    for _ in range(4):
        gs = [torch.randn_like(p) * 3 for p in p0]
        for p, q, g in zip(a, b, gs):
            p.grad, q.grad = g.clone(), g.clone()
        O.clip_grad_norm_(a, 1.0)
        torch.nn.utils.clip_grad_norm_(b, 1.0)
        oa.step(), ob.step()
    for p, q in zip(a, b):
        np.testing.assert_allclose(p.detach().numpy(), q.detach().numpy(), rtol=1e-6, atol=1e-8)


def test_masked_mse_divides_by_valid_frames():
    m = O.sequence_mask([3, 1], 4).unsqueeze(-1)
    a, b = torch.ones(2, 4, 5), torch.zeros(2, 4, 5)
    assert float(O.masked_mse(a, b, m)) == pytest.approx(5.0)  # 4 valid frames * 5 dims / 4


@pytest.mark.skipif(not os.path.isfile("/root/reference/train.py"), reason="reference tree absent")
def test_oracle_live_against_reference_models():
    import ref_loader
    train, hparams, gantts = ref_loader.load_reference()
    spec = dict(kind="MLP", in_dim=20, out_dim=7, num_hidden=2, hidden_dim=16, dropout=0.5, last_sigmoid=True)
    ref = gantts.models.MLP(**{k: v for k, v in spec.items() if k != "kind"}).eval()
    sd = C.make_weights(spec, 3)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mine = O.OracleMLP(**{k: v for k, v in spec.items() if k != "kind"})
    mine.load_state_dict(sd)
    mine.training = False
    x = torch.rand(2, 9, 20)
    assert torch.allclose(ref(x), mine(x), atol=1e-7)
    assert list(ref.state_dict().keys()) == mine.names


def test_sru_restatement_gradients_by_finite_differences():
    """The SRU cell is third-party and un-vendored (parity unpinned): the restatement is checked for
    self-consistency -- autograd gradients of the recurrence against central differences in float64."""
    torch.manual_seed(0)
    m = O.OracleSRURNN(in_dim=5, out_dim=4, num_hidden=2, hidden_dim=3, bidirectional=True, use_relu=0)
    m.training = False
    m.params = [p.detach().double().requires_grad_(True) for p in m.params]
    with torch.no_grad():
        m.params[1].uniform_(-0.5, 0.5)
        m.params[3].uniform_(-0.5, 0.5)
    x = torch.randn(2, 6, 5, dtype=torch.float64)
    w = torch.randn(2, 6, 4, dtype=torch.float64)
    loss = (m(x) * w).sum()
    grads = torch.autograd.grad(loss, m.params)
    for p, g in zip(m.params, grads):
        flat = p.detach().view(-1)
        for idx in torch.randperm(flat.numel())[:6].tolist():
            old = float(flat[idx])
            flat[idx] = old + 1e-6
            lp = float((m(x) * w).sum())
            flat[idx] = old - 1e-6
            lm = float((m(x) * w).sum())
            flat[idx] = old
            assert (lp - lm) / 2e-6 == pytest.approx(float(g.view(-1)[idx]), rel=1e-5, abs=1e-8)
    # k = 3 layers use the layer input itself as the highway term, k = 4 a fourth projection
    assert m.ks == [4, 3]
    assert m.names[:2] == ["gru.rnn_lst.0.weight", "gru.rnn_lst.0.bias"]
    assert tuple(m.params[0].shape) == (5, 6 * 4) and tuple(m.params[2].shape) == (6, 6 * 3)


def _dist_cfg(case):
    return O.StreamConfig(case["stream_sizes"], case["has_dynamic_features"], case["windows"])


@pytest.mark.parametrize("name", sorted(C.DISTORTION_CASES))
def test_distortions_match_reference_golden(name):
    """oracle.compute_distortions / split_streams vs the real train.compute_distortions (train.py:358-432)."""
    gold = np.load(os.path.join(GOLDEN, "distortions.npz"))
    case = C.DISTORTION_CASES[name]
    y, yh, mean, std, lengths = C.make_distortion_inputs(case)
    ty, tyh, tm, ts = (torch.from_numpy(a) for a in (y, yh, mean, std))
    got = O.compute_distortions(_dist_cfg(case), case["name"], ty, tyh, tm, ts, list(lengths))
    keys = [k for k in gold.files if k.startswith(name + ".") and ".split." not in k]
    assert sorted(k.split(".", 1)[1] for k in keys) == sorted(got)
    for k in keys:
        g, o = float(gold[k]), got[k.split(".", 1)[1]]
        assert (np.isnan(g) and np.isnan(o)) or abs(o - g) <= 1e-9 * max(1.0, abs(g)), (k, o, g)
    if case["name"] == "acoustic":
        for tag, t in (("y", ty), ("yh", tyh)):
            _, lf0, vuv, _ = O.split_streams(_dist_cfg(case), t, tm, ts)
            assert vuv.dtype == torch.int64
            np.testing.assert_array_equal(vuv.numpy(), gold["%s.split.%s.vuv" % (name, tag)])     # bit-exact
            np.testing.assert_array_equal(lf0.numpy(), gold["%s.split.%s.lf0" % (name, tag)])


def test_distortion_metrics_without_lengths_use_every_frame():
    rs = np.random.RandomState(0)
    X, Y = torch.from_numpy(rs.randn(3, 11, 4)), torch.from_numpy(rs.randn(3, 11, 4))
    assert abs(O.melcd(X, Y) - O.melcd(X, Y, [11, 11, 11])) < 1e-12
    z = (X - Y).numpy()
    assert abs(O.melcd(X, Y) - O._LOGDB_CONST * np.sqrt((z * z).sum(-1)).mean()) < 1e-12
    assert abs(O.mean_squared_error(X, Y) - (z * z).mean()) < 1e-12
    v = torch.from_numpy((rs.rand(3, 11) > 0.5).astype(np.int64))
    w = torch.from_numpy((rs.rand(3, 11) > 0.5).astype(np.int64))
    assert abs(O.vuv_error(v, w) - float((v != w).double().mean())) < 1e-12
    with pytest.raises(ZeroDivisionError):
        O.lf0_mean_squared_error(X[:, :, :1], torch.zeros(3, 11, dtype=torch.long), Y[:, :, :1], w)


def test_inference_restatements_match_reference_golden():
    """oracle.gen_parameters vs the real evaluation_tts.gen_parameters outputs; generic mlpg() with unit
    variance == the unit-variance matrix form; weighted mlpg solves its normal equations."""
    gold = np.load(os.path.join(GOLDEN, "inference.npz"))
    inp = C.make_inference_inputs()
    for tag in ("lstm", "mlp"):
        got = O.gen_parameters([180, 3, 1, 3], C.WINDOWS, gold["acoustic_predicted." + tag],
                               inp["Y_mean_acoustic"], inp["Y_std_acoustic"])
        for n, v in zip(("mgc", "lf0", "vuv", "bap"), got):
            np.testing.assert_allclose(v, gold["%s.%s" % (n, tag)], rtol=1e-10, atol=1e-10, err_msg=n)
    rs = np.random.RandomState(0)
    mu = rs.randn(19, 6)
    R = O.unit_variance_mlpg_matrix(C.WINDOWS, 19).astype(np.float64)
    want = R @ mu.reshape(19, 3, 2).transpose(1, 0, 2).reshape(57, 2)
    np.testing.assert_allclose(O.mlpg(mu, np.ones(6), C.WINDOWS), want, atol=2e-6)
    var = 0.2 + rs.rand(19, 6)
    c = O.mlpg(mu, var, C.WINDOWS)
    Ws = [O._window_matrix(l, u, w, 19) for (l, u, w) in C.WINDOWS]
    for d in range(2):      # gradient of the weighted least-squares objective vanishes at the solution
        g = sum(Ww.T @ ((Ww @ c[:, d] - mu[:, w * 2 + d]) / var[:, w * 2 + d]) for w, Ww in enumerate(Ws))
        assert np.abs(g).max() < 1e-9
    d = O.predict_durations(np.array([[-3.0, 0.2, 0.26]]), np.array([1.0, 1.0, 1.0]), np.array([1.0, 1.0, 2.0]))
    np.testing.assert_array_equal(d, [[1.0, 1.0, 2.0]])
