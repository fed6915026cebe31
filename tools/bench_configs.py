#!/usr/bin/env python
"""The BASELINE.json configurations besides the headline (configs[0], configs[2..4]) as timed G+D steps on one MI355X, for
`bench.py`'s `other_configs` object ("let the driver see every config") and for the profile scripts.

    cfg1                   VC In2OutHighwayNet 75 -> 512 x 3 -> 75 + MLP D 25 -> 256 x 2 -> 1, B=8, T=256 (configs[0])
    cfg3_bf16 / cfg3_fp32  BiLSTM 3x256 generator + conditioned MLP D, B=32, T=1024, variable lengths (configs[2])
    cfg4_bf16              VC SRU 6x512 bidirectional generator on mgc/lf0/bap streams + MLP D, B=16, T=2048 (configs[3])
    cfg5                   duration (416+200 -> 5, Adam, R=None) + acoustic (425+200 -> 187, Adagrad) pairs with
                           generator noise and a conditioned D, B=64 (configs[4]); one "step" = one step of each pair
    cfg2_bf16              the headline workload with bf16 storage (GT_OPT_MATMUL_BF16)

Every entry reports ms/step, padded frames/s, the arithmetic type and a STEP-LEVEL roofline: SURVEY 8(d)'s algorithmic
flops per frame (3g - g1 + 8d - d1 MACs, g / d = the networks' multiply-accumulates per frame) x frames / step time against
the dense MFMA peak of the product type.  The recurrent configs are bound by neither peak (exchange latency of the
persistent recurrence); the figure is reported all the same.

    python tools/bench_configs.py [names...] [--steps K] [--warmup W]
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}      # MI355X_MICROARCH.md: dense MFMA peaks (no sparsity)


def _macs_mlp(spec):
    ins = [spec["in_dim"]] + [spec["hidden_dim"]] * (spec["num_hidden"] - 1)
    per = [i * spec["hidden_dim"] for i in ins] + [spec["hidden_dim"] * spec["out_dim"]]
    return sum(per), per[0]


def _macs_lstm(spec):
    H, dirs = spec["hidden_dim"], 2 if spec["bidirectional"] else 1
    tot, first = 0, 0
    for l in range(spec["num_hidden"]):
        n_in = spec["in_dim"] if l == 0 else H * dirs
        tot += dirs * (4 * H * n_in + 4 * H * H)
        if l == 0:
            first = dirs * 4 * H * n_in
    return tot + spec["out_dim"] * H * dirs, first


def _macs_sru(spec):
    ncols = spec["hidden_dim"] * (2 if spec["bidirectional"] else 1)
    tot, first = 0, 0
    for l in range(spec["num_hidden"]):
        n_in = spec["in_dim"] if l == 0 else ncols
        k = 3 if n_in == ncols else 4
        tot += n_in * ncols * k
        if l == 0:
            first = n_in * ncols * k
    return tot + spec["out_dim"] * ncols, first


def _macs_i2o(spec):
    tot, first = _macs_mlp(spec)
    return tot + spec["static_dim"] * spec["static_dim"], first      # + the transform gate T (gantts/models.py:36)


def step_flops_per_frame(kind, g_spec, d_spec):
    """2 * (3g - g1 + 8d - d1) flops per padded frame of one G+D step."""
    g, g1 = {"MLP": _macs_mlp, "LSTMRNN": _macs_lstm, "SRURNN": _macs_sru, "In2OutHighwayNet": _macs_i2o}[kind](g_spec)
    d, d1 = _macs_mlp(d_spec)
    return 2.0 * (3 * g - g1 + 8 * d - d1)


def _n_params(kind, spec):
    if kind in ("MLP", "In2OutHighwayNet"):
        ins = [spec["in_dim"]] + [spec["hidden_dim"]] * (spec["num_hidden"] - 1)
        n = sum(i * spec["hidden_dim"] + spec["hidden_dim"] for i in ins) + spec["hidden_dim"] * spec["out_dim"] + spec["out_dim"]
        return n + (spec["static_dim"] ** 2 + spec["static_dim"] if kind == "In2OutHighwayNet" else 0)
    dirs = 2 if spec["bidirectional"] else 1
    H = spec["hidden_dim"]
    n = 0
    for l in range(spec["num_hidden"]):
        n_in = spec["in_dim"] if l == 0 else H * dirs
        if kind == "LSTMRNN":
            n += dirs * (4 * H * n_in + 4 * H * H + 8 * H)
        else:      # SRU: W (n_in, dirs H k) + 2 dirs H biases
            n += n_in * H * dirs * (3 if n_in == H * dirs else 4) + 2 * H * dirs
    return n + H * dirs * spec["out_dim"] + spec["out_dim"]


def step_bytes_per_step(kind, g_spec, d_spec, frames, cond_dim, static_dim, bf16):
    """ALGORITHMIC HBM bytes of one G+D step (`traffic_algorithmic` for every configuration), by SURVEY 8(d)'s recipe for
    cfg2 ("no cross-layer fusion": every layer's output is written once and read once per pass that needs it) generalised:
      per frame   x: 4 Din x (G forward, G first-layer dW) + 4 cond x 3 D passes;  y: 4 Dout;  y_hat: 4 Dout x 2;  y_hat_static: 4 static x 4;
                  G layer outputs: s x width x 2 (written forward, read backward) + recurrent stashes (float32 in both modes: LSTM gates 4H + c + h
                  per direction, SRU c per direction, each written + read);  D hidden outputs: s x width x 2, for each of its 3 passes;
                  s = 4 (float32) or 2 (bf16 storage of what only feeds products)
      per step    parameters: 4 P_G x 2 (forward, backward) + 4 P_D x 5 (3 forward passes, 2 backward) + 20 (P_G + P_D) for gradient + optimizer
    -> 42.5 KB per frame + 29 MB per step for cfg2 float32 ("about 45 KB per frame").  Measured traffic / this = how much of a
    step's bytes are re-reads, re-formatting or padding."""
    s = 2 if bf16 else 4
    dirs = 2 if g_spec.get("bidirectional") else 1
    H = g_spec["hidden_dim"]
    if kind in ("MLP", "In2OutHighwayNet"):
        g_hidden = s * H * g_spec["num_hidden"] * 2
    elif kind == "LSTMRNN":
        g_hidden = g_spec["num_hidden"] * (s * H * dirs * 2 + 4 * dirs * (4 * H + H + H) * 2)
    else:
        g_hidden = g_spec["num_hidden"] * (s * H * dirs * 2 + 4 * dirs * H * 2 + s * H * dirs * (3 if g_spec["in_dim"] == H * dirs else 4) * 2)     # + U = x W, written and read
    d_hidden = s * d_spec["hidden_dim"] * d_spec["num_hidden"] * 2 * 3
    per_frame = 4 * g_spec["in_dim"] * 2 + 4 * cond_dim * 3 + 4 * g_spec["out_dim"] * 3 + 4 * static_dim * 4 + g_hidden + d_hidden
    pg, pd = _n_params(kind, g_spec), _n_params("MLP", d_spec)
    return per_frame * frames + 4.0 * pg * 2 + 4.0 * pd * 5 + 20.0 * (pg + pd)


def _pair(hp_set, kind, g_spec, d_spec, B, Tn, opt, noise_dim, bf16, seed, variable_lengths=True):
    """One (G, D, optimizers, synthetic batch) set and its step function (train.py:538-585 for one batch)."""
    import torch
    import gantts_amd.train as T
    from gantts_amd import models, optim, paramgen
    from gantts_amd.multistream import get_static_features
    from gantts_amd.seqloss import sequence_mask
    hp = types.SimpleNamespace(**hp_set.values())
    hp.generator_add_noise = noise_dim > 0
    gs = dict(g_spec)
    din = gs["in_dim"]
    gs["in_dim"] = din + noise_dim
    mg = getattr(models, kind)(**gs).cuda().train()
    md = models.MLP(**d_spec).cuda().train()
    if opt == "Adam":
        og, od = optim.Adam(mg.parameters(), lr=1e-3, weight_decay=1e-6), optim.Adam(md.parameters(), lr=1e-3, weight_decay=1e-6)
    else:
        og, od = optim.Adagrad(mg.parameters(), lr=0.01, weight_decay=1e-7), optim.Adagrad(md.parameters(), lr=0.01, weight_decay=1e-7)
    gen = torch.Generator().manual_seed(seed)
    x = torch.rand(B, Tn, din, generator=gen).cuda()
    y = torch.randn(B, Tn, gs["out_dim"], generator=gen).cuda()
    if variable_lengths:
        lengths = torch.sort(torch.randint(Tn // 2, Tn + 1, (B,), generator=torch.Generator().manual_seed(1234)), descending=True)[0]
        lengths[0] = Tn
    else:
        lengths = torch.full((B,), Tn, dtype=torch.long)
    has_dyn = any(hp.has_dynamic_features)
    R = paramgen.unit_variance_mlpg_matrix_cuda(hp.windows, Tn) if has_dyn else None
    ys = get_static_features(y, len(hp.windows), hp.stream_sizes, hp.has_dynamic_features)
    mask = sequence_mask(lengths.cuda()).unsqueeze(-1)
    cl = [int(v) for v in lengths]
    if bf16:
        from gantts_amd.engine import engine_for
        T.hp = hp
        engine_for(hp, mg).set_option("matmul_bf16", 1)
    noise_gen = torch.Generator(device="cuda").manual_seed(seed + 11)

    def step():
        T.hp = hp
        og.zero_grad(), od.zero_grad()
        gin = x
        if noise_dim:
            gin = torch.cat((x, torch.rand(B, Tn, noise_dim, device="cuda", generator=noise_gen)), -1)
        yh, yhs = T.apply_generator(mg, gin, R, cl)
        d = T.update_discriminator(md, od, x, ys, yhs, cl, mask, "train")
        g = T.update_generator(mg, md, og, x, y, yh, ys, yhs, 1.0, cl, mask, "train", mse_w=0.0, mge_w=1.0)
        return d, g

    flops = step_flops_per_frame(kind, gs, d_spec) * B * Tn
    cond = din if getattr(hp, "discriminator_linguistic_condition", False) else 0
    abytes = step_bytes_per_step(kind, gs, d_spec, B * Tn, cond, int(ys.shape[-1]), bf16)
    return step, flops, B * Tn, abytes


_D_ACOUSTIC = dict(in_dim=483, out_dim=1, num_hidden=3, hidden_dim=256, dropout=0.5, last_sigmoid=True)


def _build(name):
    """-> (step function, algorithmic flops per step, headline frames per step, dtype, description, algorithmic HBM bytes per step)"""
    from gantts_amd import hparams
    if name in ("cfg3_bf16", "cfg3_fp32"):
        g = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=256, bidirectional=True, dropout=0.0)
        step, fl, fr, ab = _pair(hparams.tts_acoustic, "LSTMRNN", g, _D_ACOUSTIC, 32, 1024, "Adagrad", 0, name.endswith("bf16"), 0)
        return step, fl, fr, "bf16" if name.endswith("bf16") else "f32", \
            "cfg3: BiLSTM 3x256 G 425->187 + conditioned MLP D 483-256x3-1, B=32 T=1024, variable lengths, Adagrad", ab
    if name == "cfg4_bf16":
        hp = hparams.vc
        vals = dict(hp.values())
        vals.update(stream_sizes=[177, 3, 3], has_dynamic_features=[True, True, True], adversarial_streams=[True, False, False],
                    mask_nth_mgc_for_adv_loss=0, discriminator_linguistic_condition=False)
        hp_set = types.SimpleNamespace(values=lambda: vals)
        g = dict(in_dim=183, out_dim=183, num_hidden=6, hidden_dim=512, bidirectional=True, dropout=0.2, use_relu=1, rnn_dropout=0.2)
        d = dict(in_dim=59, out_dim=1, num_hidden=2, hidden_dim=256, dropout=0.5, last_sigmoid=True)
        step, fl, fr, ab = _pair(hp_set, "SRURNN", g, d, 16, 2048, "Adagrad", 0, True, 0)
        return step, fl, fr, "bf16", "cfg4: VC SRU 6x512 bidirectional G 183->183 (mgc/lf0/bap streams [177,3,3]) + MLP D 59-256x2-1, " \
                                     "B=16 T=2048, both variational dropouts 0.2, Adagrad", ab
    if name == "cfg1":
        hp = hparams.vc
        vals = dict(hp.values())
        vals.update(stream_sizes=[75], has_dynamic_features=[True], adversarial_streams=[True], mask_nth_mgc_for_adv_loss=0,
                    discriminator_linguistic_condition=False)
        hp_set = types.SimpleNamespace(values=lambda: vals)
        g = dict(in_dim=75, out_dim=75, static_dim=25, num_hidden=3, hidden_dim=512, dropout=0.5)
        d = dict(in_dim=25, out_dim=1, num_hidden=2, hidden_dim=256, dropout=0.5, last_sigmoid=True)
        step, fl, fr, ab = _pair(hp_set, "In2OutHighwayNet", g, d, 8, 256, "Adagrad", 0, False, 0, variable_lengths=False)
        return step, fl, fr, "f32", "cfg1 (BASELINE.json configs[0], the reference's CPU plumbing configuration, here on the GPU): VC In2OutHighwayNet " \
                                    "75->512x3->75 (static 25, mgc order 25) + MLP D 25-256x2-1, B=8 T=256, Adagrad", ab
    if name == "cfg2_bf16":
        g = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=512, dropout=0.5, last_sigmoid=False)
        step, fl, fr, ab = _pair(hparams.tts_acoustic, "MLP", g, _D_ACOUSTIC, 32, 512, "Adagrad", 0, True, 0, variable_lengths=False)
        return step, fl, fr, "bf16", "cfg2 with bf16 storage: MLP G 425-512x3-187 + conditioned MLP D, B=32 T=512, Adagrad", ab
    if name == "cfg5":
        ga = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=512, dropout=0.5, last_sigmoid=False)
        gd = dict(in_dim=416, out_dim=5, num_hidden=3, hidden_dim=512, dropout=0.5, last_sigmoid=False)
        dd = dict(in_dim=421, out_dim=1, num_hidden=3, hidden_dim=256, dropout=0.5, last_sigmoid=True)
        s_dur, f_dur, _, ab_dur = _pair(hparams.tts_duration, "MLP", gd, dd, 64, 40, "Adam", 200, False, 5)
        s_ac, f_ac, fr, ab_ac = _pair(hparams.tts_acoustic, "MLP", ga, _D_ACOUSTIC, 64, 512, "Adagrad", 200, False, 6)

        def step():
            return s_dur(), s_ac()
        return step, f_dur + f_ac, fr, "f32", "cfg5: duration pair (416+200 noise -> 5, 40 phones, Adam, R=None) + acoustic pair " \
                                              "(425+200 noise -> 187, T=512, Adagrad), conditioned D, B=64; one step of each pair", ab_dur + ab_ac
    raise KeyError(name)


ALL = ["cfg1", "cfg3_bf16", "cfg3_fp32", "cfg4_bf16", "cfg5", "cfg2_bf16"]


def run_config(name, steps=5, warmup=2):
    """Times `steps` G+D steps of configuration `name` after `warmup` untimed ones (cuda synchronize on both sides)."""
    import torch
    step, flops, frames, dtype, desc, abytes = _build(name)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    peak = PEAK_TFLOPS[dtype]
    ach = flops / dt / 1e12
    out = {"ms_per_step": dt * 1e3, "frames_per_s": frames / dt, "dtype": dtype, "steps": steps, "warmup": warmup, "config": desc,
           "step_algorithmic_gflop": flops / 1e9, "traffic_algorithmic": abytes,
           "roofline": {"bound": "mfma", "level": "step (SURVEY 8(d) algorithmic flops / step time)", "achieved": ach, "peak": peak,
                        "unit": "TFLOP/s", "frac": ach / peak}}
    # HBM bytes per step from the committed PMC passes of the newest round that has this configuration
    for tag in ("r06", "r05", "r04"):
        prof = os.path.join(ROOT, "profiles", "%s_other_configs.json" % tag)
        if not os.path.isfile(prof):
            continue
        try:
            p = json.load(open(prof)).get(name)
            if p and p.get("hbm_bytes_per_step"):
                bw = p["hbm_bytes_per_step"] / dt / 1e12
                out["roofline"]["hbm"] = {"traffic": p["hbm_bytes_per_step"], "achieved": bw * 1e3, "peak": 8000.0, "unit": "GB/s",
                                          "frac": bw / 8.0, "traffic_algorithmic": abytes,
                                          "traffic_over_algorithmic": p["hbm_bytes_per_step"] / abytes,
                                          "source": "profiles/%s_other_configs.json (rocprofv3 --pmc, FETCH x2 corrected); algorithmic bytes: tools/bench_configs.py step_bytes_per_step (SURVEY 8(d) recipe)" % tag}
                break
        except Exception:      # noqa: BLE001
            pass
    del step
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=ALL)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    res = {n: run_config(n, a.steps, a.warmup) for n in a.names}
    print(json.dumps(res))
