"""CPU-runnable tests of the host-side mirror of the reference interface (no GPU compute)."""
import numpy as np
import pytest
import torch

import cases as C
import gantts_oracle as O


def test_hparams_sets_and_parse():
    from gantts_amd import hparams as H
    assert H.tts_acoustic.stream_sizes == [180, 3, 1, 3]
    assert H.tts_acoustic.has_dynamic_features == [True, True, False, True]
    assert H.tts_acoustic.adversarial_streams == [True, False, False, False]
    assert H.tts_acoustic.mask_nth_mgc_for_adv_loss == 2
    assert H.tts_acoustic.optimizer_g_params == {"lr": 0.01, "weight_decay": 1e-7}
    assert H.tts_duration.optimizer_d == "Adam" and H.tts_duration.optimizer_d_params["betas"] == (0.5, 0.9)
    assert H.vc.generator == "In2OutHighwayNet" and H.vc.stream_sizes == [177]
    assert len(H.vc.windows) == 3 and len(H.tts_duration.windows) == 1
    hp = H.HParams(batch_size=20, nepoch=200, lr_decay_schedule=False, name="x")
    hp.parse("batch_size=16,lr_decay_schedule=True")
    assert hp.batch_size == 16 and hp.lr_decay_schedule is True
    assert hp.parse("") is hp
    with pytest.raises(ValueError):
        hp.parse("nope=1")
    s = H.hparams_debug_string(hp)
    assert s.startswith("Hyperparameters:") and "  batch_size: 16" in s
    assert H.tts_acoustic == H.tts_acoustic and H.tts_acoustic != H.vc     # identity comparisons (train.py:447)


def test_reference_hparams_values_match_when_reference_present():
    import os
    if not os.path.isfile("/root/reference/hparams.py"):
        pytest.skip("reference tree absent")
    import ref_loader
    _, ref, _ = ref_loader.load_reference()
    from gantts_amd import hparams as H
    for name in ("vc", "tts_duration", "tts_acoustic"):
        a, b = getattr(H, name).values(), getattr(ref, name).values()
        assert set(a) == set(b), name
        for k in a:
            if k == "windows":
                assert len(a[k]) == len(b[k])
                for (l1, u1, c1), (l2, u2, c2) in zip(a[k], b[k]):
                    assert (l1, u1) == (l2, u2) and np.array_equal(c1, c2)
            elif k == "question_path":
                assert a[k].endswith("questions-radio_dnn_416.hed")
            else:
                assert a[k] == b[k], (name, k)


def test_model_state_dict_layout_and_flat_buffer():
    from gantts_amd import models
    spec = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=64, dropout=0.5, last_sigmoid=False)
    m = models.MLP(**spec)
    names = [n for n, _ in C.param_shapes(dict(kind="MLP", **spec))]
    assert list(m.state_dict().keys()) == names
    for (n, shp), p in zip(C.param_shapes(dict(kind="MLP", **spec)), m.parameters()):
        assert tuple(p.shape) == shp
    flat = m.flat_params()
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    off = 0
    for p in m.parameters():           # parameters are views into the flat buffer, in order
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    w = C.make_weights(dict(kind="MLP", **spec), 3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    assert np.array_equal(flat[:64 * 425].view(64, 425).numpy(), w["layers.0.weight"])
    k = 1 / np.sqrt(425)
    m2 = models.MLP(**spec)
    assert float(m2.layers[0].weight.abs().max()) <= k + 1e-7     # nn.Linear default init range
    assert not m.include_parameter_generation()
    i2o = models.In2OutHighwayNet(in_dim=75, out_dim=75, static_dim=25, hidden_dim=32)
    assert i2o.include_parameter_generation()
    assert list(i2o.state_dict().keys())[:4] == ["T.weight", "T.bias", "H.0.weight", "H.0.bias"]
    m.eval()
    assert not m.training
    m.train()
    assert m.training
    with pytest.raises(TypeError):
        models.MLP(nonsense=1)


def test_optimizer_state_dict_is_torch_compatible():
    from gantts_amd import models, optim
    m = models.MLP(in_dim=6, out_dim=2, num_hidden=1, hidden_dim=4, last_sigmoid=False)
    for cls, tcls, kw in ((optim.Adagrad, torch.optim.Adagrad, dict(lr=0.01, weight_decay=1e-7)),
                          (optim.Adam, torch.optim.Adam, dict(lr=1e-3, betas=(0.5, 0.9), weight_decay=0))):
        o = cls(m.parameters(), **kw)
        sd = o.state_dict()
        ref = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2))
        t = tcls(ref.parameters(), **kw)
        assert sd["param_groups"][0]["params"] == t.state_dict()["param_groups"][0]["params"]
        for k in kw:
            assert sd["param_groups"][0][k] == kw[k]
        t.load_state_dict(sd)                     # torch accepts our checkpoint format
        o2 = cls(m.parameters())
        o2.load_state_dict(t.state_dict())        # and we accept torch's
        assert o2.param_groups[0]["lr"] == kw["lr"]
        with pytest.raises(RuntimeError):
            o.step()
    with pytest.raises(TypeError):
        optim.Adagrad(torch.nn.Linear(3, 3).parameters())
    with pytest.raises(ValueError):
        optim.Adagrad(list(m.parameters())[:2])


def test_exp_lr_scheduler_and_stream_index_helpers():
    import gantts_amd.train as T
    from gantts_amd import models, optim
    from gantts_amd.multistream import get_static_stream_sizes, static_columns
    m = models.MLP(in_dim=6, out_dim=2, num_hidden=1, hidden_dim=4, last_sigmoid=False)
    o = optim.Adagrad(m.parameters(), lr=0.01)
    T.exp_lr_scheduler(o, 25, 200, init_lr=0.01, lr_decay_epoch=25)
    assert o.param_groups[0]["lr"] == pytest.approx(0.001)
    assert list(get_static_stream_sizes([180, 3, 1, 3], [True, True, False, True], 3)) == [60, 1, 1, 1]
    ss, hd = [180, 3, 1, 3], [True, True, False, True]
    cols = static_columns(3, ss, hd, None, 187)
    x = torch.arange(187.).expand(1, 2, 187)
    assert torch.equal(x[:, :, cols], O.get_static_features(x, 3, ss, hd))
    assert static_columns(3, [75], [True], None, 75) == list(range(25))
    assert static_columns(1, [5], [False], None, 5) == list(range(5))
    from gantts_amd.engine import _hp_signature
    from gantts_amd import hparams as H
    sig = _hp_signature(H.tts_acoustic)
    assert sig[0] == (180, 3, 1, 3) and sig[2] == 3 and sig[4] == 2 and sig[5] is True


@pytest.mark.parametrize("T_", [1, 2, 7, 64, 300])
def test_paramgen_matches_oracle_dense_inverse(T_):
    from gantts_amd import paramgen
    for n in (1, 2, 3):
        a = paramgen.unit_variance_mlpg_matrix(C.WINDOWS[:n], T_)
        b = O.unit_variance_mlpg_matrix(C.WINDOWS[:n], T_)
        assert a.shape == (T_, n * T_) and a.dtype == np.float32
        np.testing.assert_allclose(a, b, atol=1e-6)
    assert paramgen.unit_variance_mlpg_matrix(C.WINDOWS, T_) is paramgen.unit_variance_mlpg_matrix(C.WINDOWS, T_)


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r*_bench.json is the JSON line `bench.py` printed on the GPU box at the round's final build: it must carry
    every field of the driver's contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
    scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and `cpu_baseline` objects, with
    self-consistent numbers (value = frames / time, frac = achieved / peak, traffic >= algorithmic bytes)."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    d = json.load(open(files[-1]))
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].split(" at ")[0] in base["metric"] and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    frames = d["config"]["frames_per_step"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] >= 0.95 * r["traffic_algorithmic"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def _load_bench():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_condenses_a_schedule_trace_into_exposed_and_hidden_time():
    """bench.py --comm-trace: records of gt_comm_trace_read {kind, bytes, on the step stream, start us, end us} of two identical steps ->
    per message: duration, the part of it during which the step stream stood waiting (exposed) and the rest (hidden); a message issued
    on the step stream itself is exposed for its whole duration."""
    bench = _load_bench()
    step = [
        [0, 40, 0, 0.0, 6.0],            # sums on the communicator's stream, nobody waits: hidden
        [0, 1000000, 1, 10.0, 16.0],     # closing message on the step stream: exposed
        [0, 2000000, 0, 20.0, 60.0],     # under compute ...
        [1, 0, 0, 50.0, 60.0],           # ... until the step stream joins: its last 10 us are exposed
    ]
    rec = np.array(step + [[k, b, i, t0 + 100.0, t1 + 100.0] for k, b, i, t0, t1 in step], dtype=np.float64)

    class A:
        one_device, dp_ipc = False, False
    c = bench.comm_schedule(rec, 2, 1.5, A())
    assert c["messages_per_step"] == 3 and c["waits_per_step"] == 1 and c["bytes_per_step"] == 3000040
    assert c["step_stream_wait_us_per_step"] == pytest.approx(10.0)
    m = c["messages"]
    assert [x["on_step_stream"] for x in m] == [False, True, False]
    assert (m[0]["exposed_us"], m[0]["hidden_us"]) == pytest.approx((0.0, 6.0))
    assert (m[1]["exposed_us"], m[1]["hidden_us"]) == pytest.approx((6.0, 0.0))
    assert (m[2]["exposed_us"], m[2]["hidden_us"]) == pytest.approx((10.0, 30.0))
    assert c["exposed_us_per_step"] == pytest.approx(16.0) and c["hidden_us_per_step"] == pytest.approx(36.0)
    assert c["transport"] == "RCCL"


def test_bench_scaling_model_is_a_labelled_upper_bound():
    """`scaling_model` of the bench line: the measured per-rank proxy, plus the traced schedule's exposed messages at the bandwidth lower
    bound of their ring all-reduce -- the speed-up with communication can only be below the one without, and everything says UNMEASURED."""
    bench = _load_bench()
    m = bench.scaling_model({2: 0.754, 4: 0.5045, 8: 0.3937}, 1.3144)
    assert "UNMEASURED" in m["status"] and m["collectives"]["messages_per_step"] == 5
    assert sum(m["collectives"]["exposed_message_bytes"]) < m["collectives"]["bytes_per_step"]
    for n in ("2", "4", "8"):
        assert m["speedup_over_one_gpu_upper_bound_with_exposed_messages"][n] < m["speedup_over_one_gpu_without_communication"][n] <= int(n)
    w = m["exposed_wire_time_lower_bound_us"]
    assert w["2"] < w["4"] < w["8"] < 2.0 * w["2"]                  # 2 (N - 1) / N of the bytes: saturates at twice the N = 2 time
    assert w["8"] == pytest.approx(1e6 * 2 * 7 / 8 * sum(m["collectives"]["exposed_message_bytes"]) / 153e9)
    # the zero-latency bound is optimistic -- the same with 30 / 60 us assumed per exposed message is carried beside it, and is lower
    lat = m["speedup_over_one_gpu_with_exposed_messages_and_assumed_latency"]
    assert lat["assumed_us_per_exposed_message"] == [30.0, 60.0]
    for n in ("2", "4", "8"):
        assert lat["speedup"][n][1] < lat["speedup"][n][0] < m["speedup_over_one_gpu_upper_bound_with_exposed_messages"][n]
    # the schedule is READ from the newest committed trace (named in the line), not copied into bench.py as constants
    assert m["collectives"]["source"].startswith("profiles/r") and "comm_schedule_dp1_rccl.json" in m["collectives"]["source"]
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    onfile = json.load(open(os.path.join(root, m["collectives"]["source"].split(" ")[0])))["comm_schedule"]
    assert m["collectives"]["bytes_per_step"] == onfile["bytes_per_step"]
    assert m["collectives"]["one_rank_exposed_us"] == pytest.approx(onfile["exposed_us_per_step"])


def test_algorithmic_bytes_recipe_reproduces_the_survey_figure_for_cfg2():
    """tools/bench_configs.py: step_bytes_per_step is SURVEY 8(d)'s recipe for cfg2 ("about 45 KB per frame -> 0.74 GB per step") generalised to
    every configuration (`traffic_algorithmic` of `other_configs`); bf16 storage halves what only feeds products, nothing else."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_configs as bc
    g = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=512)
    d = dict(in_dim=483, out_dim=1, num_hidden=3, hidden_dim=256)
    b32 = bc.step_bytes_per_step("MLP", g, d, 16384, 425, 63, False)
    assert 0.72e9 < b32 < 0.76e9 and 43e3 < b32 / 16384 < 46e3
    b16 = bc.step_bytes_per_step("MLP", g, d, 16384, 425, 63, True)
    hidden = (512 * 3 * 2 + 256 * 3 * 2 * 3) * 16384
    assert b32 - b16 == pytest.approx(2 * hidden)
    assert bc._n_params("MLP", g) == 425 * 512 + 512 + 2 * (512 * 512 + 512) + 512 * 187 + 187
    lstm = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=256, bidirectional=True)
    assert bc._n_params("LSTMRNN", lstm) == 2 * (4 * 256 * 425 + 4 * 256 * 256 + 8 * 256) + 2 * 2 * (4 * 256 * 512 + 4 * 256 * 256 + 8 * 256) + 512 * 187 + 187


def test_apply_generator_refuses_the_reference_left_padding_case():
    """train.py:347-350: a generic packed-sequence generator whose longest sequence is shorter than the batch's T gets its output zero-padded ON
    THE LEFT by the reference.  The engine does not reproduce that shift (deliberate deviations) and says so instead of returning
    frames in another alignment; unreachable from train_loop, whose collate_fn pads to the longest sequence."""
    import gantts_amd.train as T
    from gantts_amd import hparams, models
    T.hp = hparams.tts_acoustic
    g = models.LSTMRNN(in_dim=8, out_dim=4, num_hidden=1, hidden_dim=8, bidirectional=True)
    x = torch.zeros(2, 10, 8)
    with pytest.raises(ValueError, match="left-pads"):
        T.apply_generator(g, x, None, [7, 5])
    with pytest.raises(ValueError, match="left-pads"):
        T.apply_generator(g, x, None, torch.tensor([9, 5]))


def test_committed_schedule_traces_show_one_schedule_on_every_transport():
    """profiles/r*_comm_schedule_*.json (bench.py --comm-trace on the GPU box): one rank over RCCL, two ranks on one device over the RCCL double
    and over the interprocess arenas must show the SAME schedule -- five messages of the same sizes in the same order on the same streams,
    two waits of the step stream -- whose sizes add up to both networks' gradients + the loss sums; exposed + hidden = duration."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*_comm_schedule_dp*.json"))):
        tag = os.path.basename(f).split("_comm_schedule_")[1][:-5]
        runs[tag] = json.load(open(f))["comm_schedule"]
    assert {"dp1_rccl", "dp2_fake", "dp2_ipc"} <= set(runs), sorted(runs)
    ref = [(m["bytes"], m["on_step_stream"]) for m in runs["dp1_rccl"]["messages"]]
    n_g = 425 * 512 + 512 + 2 * (512 * 512 + 512) + 512 * 187 + 187           # cfg2 generator (bench.py G_SPEC)
    n_d = 483 * 256 + 256 + 2 * (256 * 256 + 256) + 256 + 1                   # cfg2 discriminator
    assert sum(b for b, _ in ref) == 4 * (n_g + n_d) + 40 + 24
    for tag in ("dp1_rccl", "dp2_fake", "dp2_ipc"):
        c = runs[tag]
        assert c["messages_per_step"] == 5 and c["waits_per_step"] == 2 and c["bytes_per_step"] == sum(b for b, _ in ref), tag
        assert [(m["bytes"], m["on_step_stream"]) for m in c["messages"]] == ref, tag
        for m in c["messages"]:
            assert m["exposed_us"] >= 0 and m["hidden_us"] >= -1e-6
            assert m["exposed_us"] + m["hidden_us"] == pytest.approx(m["duration_us"], rel=0.35, abs=2.0)      # (medians of three quantities)
            if m["on_step_stream"]:
                assert m["hidden_us"] == pytest.approx(0.0, abs=1e-6)
    two = runs.get("dp1_rccl_d_two_messages")
    if two is not None:       # GT_COMM_D_ONE_MSG=0: the same bytes in six messages, less of them on the step stream
        assert two["messages_per_step"] == 6 and two["bytes_per_step"] == runs["dp1_rccl"]["bytes_per_step"]
        exposed = lambda c: sum(m["bytes"] for m in c["messages"] if m["on_step_stream"])
        assert exposed(two) < exposed(runs["dp1_rccl"])
    bench = _load_bench()
    sched = bench.traced_schedule()              # (what bench.py's scaling_model reads: the newest committed one-rank RCCL trace)
    assert sched["exposed_message_bytes"] == [b for b, on in ref if on]
    assert sched["bytes_per_step"] == runs["dp1_rccl"]["bytes_per_step"]


def test_trace_tools_on_a_synthetic_kernel_trace(tmp_path):
    """tools/step_gaps.py and tools/kernel_avgs.py on a hand-made rocprofv3 kernel trace: three steps of four kernels (the second
    `optim_step_kernel` closes a step), a 5 us hole behind `d_head_finalize_kernel` in every step."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows, t = ["Kernel_Name,Start_Timestamp,End_Timestamp"], 1000
    for _ in range(3):
        for name, dur, gap in (("void gt::gemm_pair_kernel<0, 1>(args)", 100000, 0), ("gt::optim_step_kernel(args)", 8000, 0),
                               ("gt::d_head_finalize_kernel(args)", 10000, 5000), ("gt::optim_step_kernel(args)", 8000, 0)):
            rows.append('"%s",%d,%d' % (name, t, t + dur))
            t += dur + gap
    f = tmp_path / "t_kernel_trace.csv"
    f.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "step_gaps.py"), str(f)], capture_output=True, text=True, check=True).stdout
    assert "2 steps: wall 131.0 us/step, sum of kernels 126.0, union 126.0, idle 5.0, launches 4.0" in out, out
    assert "gt::d_head_finalize_kernel -> gt::optim_step_kernel" in out
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_avgs.py"), str(f), "optim_step", "gemm_pair"],
                         capture_output=True, text=True, check=True).stdout
    assert "optim_step" in out and "avg    8.00 us" in out and "avg  100.00 us" in out, out


def test_fast_gate_functions_stay_at_rounding_level():
    """The fast gate functions of the recurrent kernels (gantts_amd/csrc/fast_math.hip.h: e^x = 2^(x log2 e) with the product's
    rounding error folded back in, sigmoid = rcp(1 + e^-x), tanh by its odd polynomial below 0.3 and (1 - e^-2|x|) / (1 + e^-2|x|)
    above) restated in float32 numpy -- np.exp2 / the division stand in for v_exp_f32 / v_rcp_f32 (1 ulp each on the device) --
    against float64: a few 1e-7 relative everywhere, i.e. what the library functions deliver."""
    f = np.float32

    def fast_exp(x):
        l2e_hi, l2e_lo, ln2 = f(1.44269502162933349609375), f(1.925963033500011e-8), f(0.6931471805599453)
        t = (x * l2e_hi).astype(f)
        r = (x.astype(np.float64) * np.float64(l2e_hi) - t.astype(np.float64)).astype(f)           # fmaf(x, L2E_HI, -t)
        r = (x.astype(np.float64) * np.float64(l2e_lo) + r.astype(np.float64)).astype(f)           # fmaf(x, L2E_LO, r)
        e = np.exp2(t.astype(np.float64)).astype(f)
        return (e.astype(np.float64) * (r * ln2).astype(f).astype(np.float64) + e.astype(np.float64)).astype(f)

    x = np.linspace(-30, 30, 400001).astype(f)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(fast_exp(x).astype(np.float64) - ref) / ref).max() < 3e-7
    sig = (1.0 / (1.0 + fast_exp(-x).astype(np.float64))).astype(f)
    refs = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    assert (np.abs(sig - refs) / refs).max() < 4e-7

    def fast_tanh(x):
        ax, x2 = np.abs(x), (x * x).astype(f)
        p = f(-1382. / 155925.)
        for c in (62. / 2835., -17. / 315., 2. / 15., -1. / 3., 1.):
            p = (x2.astype(np.float64) * np.float64(p) + np.float64(f(c))).astype(f)
        p = (x * p).astype(f)
        t = fast_exp((f(-2.) * ax).astype(f))
        q = np.copysign(((f(1.) - t).astype(f).astype(np.float64) / (1.0 + t.astype(np.float64))).astype(f), x)
        return np.where(ax < f(0.3), p, q)

    x = np.linspace(-12, 12, 400001).astype(f)
    ref = np.tanh(x.astype(np.float64))
    got = fast_tanh(x).astype(np.float64)
    assert np.abs(got - ref).max() < 2e-7
    big = np.abs(x) > 1e-3
    assert (np.abs(got - ref)[big] / np.abs(ref)[big]).max() < 4e-7
