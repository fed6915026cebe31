"""Host-side batch pipeline (gantts_amd/data.py) -- CPU tests: on-disk format and split rule of the
reference (train.py:64-93), collate/padding (:139-159), the restated nnmnkwii.preprocessing
helpers, and the prefetcher's sort/trim semantics (train.py:494-501) on the CPU device."""
import os

import numpy as np
import pytest
import torch

import cases as C
import gantts_oracle as O


def _data():
    pytest.importorskip("ctypes")
    try:
        from gantts_amd import data
    except ImportError as e:       # libgantts_hip.so not built
        pytest.skip(str(e))
    return data


def test_npy_source_split_matches_reference_rule(tmp_path):
    D = _data()
    from sklearn.model_selection import train_test_split
    rs = np.random.RandomState(0)
    names = ["utt_%03d.npy" % i for i in range(40)]
    for n in names:
        np.save(os.path.join(tmp_path, n), rs.randn(rs.randint(5, 9), 3).astype(np.float32))
    (tmp_path / "notes.txt").write_text("ignored")
    full = sorted(os.path.join(tmp_path, n) for n in names)
    assert D.NPYDataSource(str(tmp_path), test=True).collect_files() == full[-5:]
    tr, te = train_test_split(full[:-5], test_size=0.112, random_state=1234)
    assert D.NPYDataSource(str(tmp_path), train=True).collect_files() == tr
    assert D.NPYDataSource(str(tmp_path), train=False).collect_files() == te
    capped = D.NPYDataSource(str(tmp_path), train=True, max_files=20).collect_files()
    assert set(capped) <= set(full[:20])
    ds = D.MemoryCacheDataset(D.FileSourceDataset(D.NPYDataSource(str(tmp_path), train=True)), cache_size=3)
    assert len(ds) == len(tr)
    np.testing.assert_array_equal(ds[1], np.load(tr[1]))
    for i in range(5):
        ds[i]
    assert len(ds.cached) == 3


def test_collate_pads_with_zeros_and_reports_lengths():
    D = _data()
    rs = np.random.RandomState(1)
    batch = [(rs.rand(n, 4).astype(np.float32), rs.randn(n, 6).astype(np.float32)) for n in (5, 9, 2)]
    x, y, lengths = D.collate_fn(batch)
    assert x.shape == (3, 9, 4) and y.shape == (3, 9, 6) and x.dtype == torch.float32
    assert lengths.dtype == torch.int64 and lengths.tolist() == [5, 9, 2]
    for b, (xs, ys) in enumerate(batch):
        np.testing.assert_array_equal(x[b, :len(xs)].numpy(), xs)
        assert float(x[b, len(xs):].abs().sum()) == 0 and float(y[b, len(ys):].abs().sum()) == 0


def test_ragged_collate_and_the_host_fallback_equal_the_padded_collate():
    """collate_ragged (device-side collate: utterances back to back) through the prefetcher on the CPU device -- where the
    padding falls back to the host -- gives exactly the batches of the reference-shaped collate_fn."""
    D = _data()
    rs = np.random.RandomState(5)
    batch = [(rs.rand(n, 4).astype(np.float32), rs.randn(n, 6).astype(np.float32)) for n in (5, 9, 2, 9, 7)]
    xr, yr, lr = D.collate_ragged(batch)
    assert xr.shape == (32, 4) and yr.shape == (32, 6) and lr.tolist() == [5, 9, 2, 9, 7]
    a = list(D.DevicePrefetcher([D.collate_fn(batch)], device="cpu"))[0]
    b = list(D.DevicePrefetcher([(xr, yr, lr)], device="cpu"))[0]
    assert a.cpu_lengths == b.cpu_lengths == [9, 9, 7, 5, 2]
    assert torch.equal(a.x, b.x) and torch.equal(a.y, b.y) and torch.equal(a.lengths, b.lengths)


def test_preprocessing_restatements():
    D = _data()
    rs = np.random.RandomState(2)
    utts = [rs.randn(rs.randint(3, 30), 5) * 2 + 1 for _ in range(7)]
    allf = np.concatenate(utts)
    mean, var = D.meanvar(utts)
    np.testing.assert_allclose(mean, allf.mean(0), rtol=1e-12)
    np.testing.assert_allclose(var, allf.var(0), rtol=1e-12)
    m1, v1, n1 = D.meanvar(utts[:3], return_last_sample_count=True)
    m2, v2 = D.meanvar(utts[3:], mean_=m1, var_=v1, last_sample_count=n1)       # the VC two-pass use (train.py:722-727)
    np.testing.assert_allclose(m2, mean, rtol=1e-12)
    np.testing.assert_allclose(v2, var, rtol=1e-12)
    lo, hi = D.minmax(utts)
    np.testing.assert_array_equal(lo, allf.min(0))
    np.testing.assert_array_equal(hi, allf.max(0))
    min_, scale_ = D.minmax_scale_params(lo, hi, feature_range=(0.01, 0.99))
    z = D.minmax_scale(allf, min_=min_, scale_=scale_, feature_range=(0.01, 0.99))
    assert abs(z.min() - 0.01) < 1e-12 and abs(z.max() - 0.99) < 1e-12
    np.testing.assert_allclose(D.inv_scale(D.scale(allf, mean, np.sqrt(var)), mean, np.sqrt(var)), allf, atol=1e-12)
    # delta features == the stacked window matrix the MLPG restatement is built from
    c = rs.randn(11, 3)
    W = np.vstack([O._window_matrix(l, u, w, 11) for (l, u, w) in C.WINDOWS])
    got = D.delta_features(c, C.WINDOWS)
    want = (W @ c).reshape(3, 11, 3).transpose(1, 0, 2).reshape(11, 9)
    np.testing.assert_allclose(got, want, atol=1e-12)
    Y = np.hstack([got, rs.randn(11, 1)])
    Y2 = Y.copy()
    Y2[:, 3:9] = 0
    D.recompute_delta_features(Y2, None, None, C.WINDOWS, [9, 1], [True, False])
    np.testing.assert_allclose(Y2, Y, atol=1e-12)


def test_tts_dataset_scaling():
    D = _data()
    rs = np.random.RandomState(3)
    X = [rs.rand(6, 4) * 10 for _ in range(3)]
    Y = [rs.randn(6, 5) for _ in range(3)]
    lo, hi = D.minmax(X)
    mean, var = D.meanvar(Y)
    ds = D.TTSDataset(X, Y, lo, hi, mean, np.sqrt(var))
    x, y = ds[1]
    assert x.min() >= 0.01 - 1e-12 and x.max() <= 0.99 + 1e-12
    np.testing.assert_allclose(y, (Y[1] - mean) / np.sqrt(var))
    vc = D.VCDataset(Y, Y, mean, np.sqrt(var))
    np.testing.assert_allclose(vc[0][0], vc[0][1])


def test_prefetcher_sorts_trims_and_preserves_order_on_cpu():
    D = _data()
    rs = np.random.RandomState(4)
    batches = []
    for B, T in ((4, 12), (3, 7), (5, 9)):
        lens = rs.randint(1, T - 1, size=B)           # padded beyond the longest sequence on purpose
        x = rs.rand(B, T, 3).astype(np.float32)
        for b, n in enumerate(lens):
            x[b, n:] = 0
        batches.append((torch.from_numpy(x), torch.from_numpy(x * 2), torch.from_numpy(lens.astype(np.int64))))
    out = list(D.DevicePrefetcher(batches, device="cpu"))
    assert len(out) == 3
    for (x, y, lens), b in zip(batches, out):
        sl, idx = torch.sort(lens, descending=True)
        assert b.cpu_lengths == sl.tolist() and b.max_len == int(sl[0])
        assert b.x.shape[1] == b.max_len
        torch.testing.assert_close(b.x, x[idx][:, :b.max_len])
        torch.testing.assert_close(b.y, y[idx][:, :b.max_len])
