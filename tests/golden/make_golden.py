"""Generate golden vectors by running the REAL reference (r9y9/gantts @ /root/reference).

Build-container only (the reference tree does not exist on the GPU box).  Usage:

    python tests/golden/make_golden.py            # writes tests/golden/<case>.npz

For every case in ``cases.py`` this executes the reference's own
``train.apply_generator`` / ``train.update_discriminator`` / ``train.update_generator``
(train.py:336-355, 245-279, 282-320) with ``gantts.models`` modules and
``torch.optim`` optimizers, on seeded numpy inputs/weights, in the order
``train_loop`` uses them (train.py:528-585), and records outputs, the 9 step scalars,
the G gradient norm right after the D update (the D->G leak), and the final
parameters / optimizer state.  Dropout parity uses injected masks: ``nn.Dropout.forward``
is patched to multiply by a supplied mask and 1/(1-p) (what torch's dropout computes).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import cases as C  # noqa: E402
import ref_loader  # noqa: E402


def run_case(name, case):
    train, hparams, gantts = ref_loader.load_reference()
    from gantts.multistream import get_static_features
    from gantts.seqloss import sequence_mask
    from nnmnkwii.paramgen import unit_variance_mlpg_matrix

    hp = getattr(hparams, case["hp"])
    saved = dict(hp.__dict__)
    try:
        windows = C.WINDOWS[:case["windows"]]
        hp.__dict__.update(
            stream_sizes=case["stream_sizes"], has_dynamic_features=case["has_dynamic_features"],
            windows=windows, adversarial_streams=case["adversarial_streams"],
            mask_nth_mgc_for_adv_loss=case["mask_nth_mgc"],
            discriminator_linguistic_condition=case["cond"])
        train.hp = hp

        def build(spec, seed):
            kw = {k: v for k, v in spec.items() if k != "kind"}
            m = getattr(gantts.models, spec["kind"])(**kw)
            sd = {k: torch.from_numpy(v) for k, v in C.make_weights(spec, seed).items()}
            m.load_state_dict(sd)
            return m

        model_g, model_d = build(case["g"], 11), build(case["d"], 22)
        og = getattr(torch.optim, case["opt_g"][0])(model_g.parameters(), **case["opt_g"][1])
        od = getattr(torch.optim, case["opt_d"][0])(model_d.parameters(), **case["opt_d"][1])

        queue = []
        orig_dropout = torch.nn.Dropout.forward

        def patched(self, inp):
            if not self.training or self.p == 0:
                return inp
            m = queue.pop(0)
            assert m.shape == inp.shape, (m.shape, inp.shape)
            return inp * m / (1.0 - self.p)

        torch.nn.Dropout.forward = patched
        if case["dropout_on"]:
            model_g.train(), model_d.train()
        else:
            model_g.eval(), model_d.eval()

        x_np, y_np, lengths = C.make_batch(case)
        x, y = torch.from_numpy(x_np), torch.from_numpy(y_np)
        T = case["T"]
        has_dyn = bool(np.any(case["has_dynamic_features"]))
        R = torch.from_numpy(unit_variance_mlpg_matrix(windows, T)) if has_dyn else None
        sl = torch.from_numpy(lengths)
        cpu_lengths = list(sl)
        out = {}
        try:
            for step in range(case["steps"]):
                if case["dropout_on"]:
                    gm, dm = C.make_dropout_masks(case, step)
                    queue[:] = [torch.from_numpy(m) for m in gm + dm]
                y_static = get_static_features(y, len(windows), hp.stream_sizes, hp.has_dynamic_features)
                mask = sequence_mask(sl).unsqueeze(-1)
                og.zero_grad()
                od.zero_grad()
                y_hat, y_hat_static = train.apply_generator(model_g, x, R, cpu_lengths)
                if step == 0:
                    out["y_hat"] = y_hat.detach().numpy().copy()
                    out["y_hat_static"] = y_hat_static.detach().numpy().copy()
                if case["update_d"]:
                    res = train.update_discriminator(model_d, od, x, y_static, y_hat_static,
                                                     cpu_lengths, mask, "train")
                    out["d_scalars_%d" % step] = np.array(res, dtype=np.float64)
                    gn = [p.grad for p in model_g.parameters() if p.grad is not None]
                    out["g_leak_norm_%d" % step] = np.array(
                        float(torch.sqrt(sum((g ** 2).sum() for g in gn))) if gn else 0.0)
                if case["update_g"]:
                    res = train.update_generator(model_g, model_d, og, x, y, y_hat, y_static,
                                                 y_hat_static, case["adv_w"], cpu_lengths, mask, "train",
                                                 mse_w=case["mse_w"], mge_w=case["mge_w"])
                    out["g_scalars_%d" % step] = np.array(res, dtype=np.float64)
                if case["dropout_on"] and not case["update_g"]:
                    queue[:] = []
                assert not queue, "unused dropout masks: %d" % len(queue)
        finally:
            torch.nn.Dropout.forward = orig_dropout

        for k, v in model_g.state_dict().items():
            out["G." + k] = v.numpy().copy()
        for k, v in model_d.state_dict().items():
            out["D." + k] = v.numpy().copy()
        for tag, opt, model in (("G", og, model_g), ("D", od, model_d)):
            names = list(model.state_dict().keys())
            for i, p in enumerate(opt.param_groups[0]["params"]):
                st = opt.state.get(p, {})
                for key in ("sum", "exp_avg", "exp_avg_sq"):
                    if key in st:
                        out["%s.opt.%s.%s" % (tag, key, names[i])] = st[key].numpy().copy()
        return out
    finally:
        hp.__dict__.clear()
        hp.__dict__.update(saved)


def run_distortions():
    """The real train.compute_distortions / split_streams (train.py:358-432) on seeded inputs; the
    nnmnkwii.metrics functions underneath are the restatements of oracle/gantts_oracle.py."""
    train, hparams, gantts = ref_loader.load_reference()
    out = {}
    for name, case in C.DISTORTION_CASES.items():
        hp = getattr(hparams, case["hp"])
        saved = dict(hp.__dict__)
        try:
            hp.__dict__.update(stream_sizes=case["stream_sizes"], has_dynamic_features=case["has_dynamic_features"],
                               windows=C.WINDOWS[:case["windows"]], order=sum(C.static_sizes(case)))
            train.hp = hp
            y, yh, mean, std, lengths = C.make_distortion_inputs(case)
            ty, tyh, tm, ts = (torch.from_numpy(a) for a in (y, yh, mean, std))
            d = train.compute_distortions(ty, tyh, tm, ts, torch.from_numpy(lengths))
            for k, v in d.items():
                out["%s.%s" % (name, k)] = np.float64(v)
            if case["name"] == "acoustic":
                for tag, t in (("y", ty), ("yh", tyh)):
                    mgc, lf0, vuv, bap = train.split_streams(t, tm, ts)
                    out["%s.split.%s.vuv" % (name, tag)] = vuv.numpy().copy()
                    out["%s.split.%s.lf0" % (name, tag)] = lf0.numpy().copy()
        finally:
            hp.__dict__.clear()
            hp.__dict__.update(saved)
    path = os.path.join(HERE, "distortions.npz")
    np.savez_compressed(path, **out)
    print("distortions -> %s" % os.path.relpath(path, ROOT))
    for k in sorted(out):
        if ".split." not in k:
            print("    %-36s %r" % (k, float(out[k])))


class _Loader(list):
    """What train_loop needs from a DataLoader: iteration, len() and .dataset statistics."""


def run_train_loop(name, case):
    import types
    train, hparams, gantts = ref_loader.load_reference()
    hp = getattr(hparams, case["hp"])
    saved = dict(hp.__dict__)
    try:
        hp.__dict__.update(
            stream_sizes=case["stream_sizes"], has_dynamic_features=case["has_dynamic_features"],
            windows=C.WINDOWS[:case["windows"]], adversarial_streams=case["adversarial_streams"],
            mask_nth_mgc_for_adv_loss=case["mask_nth_mgc"], discriminator_linguistic_condition=case["cond"],
            nepoch=case["nepoch"], lr_decay_schedule=case["lr_decay_schedule"], lr_decay_epoch=case["lr_decay_epoch"],
            generator_add_noise=False, optimizer_g_params=dict(case["opt_g"][1]), optimizer_d_params=dict(case["opt_d"][1]))
        if "order" in case:
            hp.__dict__["order"] = case["order"]
        train.hp = hp
        train.global_epoch = 0

        def build(spec, seed):
            kw = {k: v for k, v in spec.items() if k != "kind"}
            m = getattr(gantts.models, spec["kind"])(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in C.make_weights(spec, seed).items()})
            return m

        model_g, model_d = build(case["g"], 11), build(case["d"], 22)
        ref_d = build(case["d"], 33) if case["reference_d"] else None
        og = getattr(torch.optim, case["opt_g"][0])(model_g.parameters(), **case["opt_g"][1])
        od = getattr(torch.optim, case["opt_d"][0])(model_d.parameters(), **case["opt_d"][1])
        data, mean, std = C.make_train_loop_data(case)
        loaders = {}
        for phase in ("train", "test"):
            ld = _Loader((torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(l)) for x, y, l in data[phase])
            if case["hp"] == "vc":
                ld.dataset = types.SimpleNamespace(data_mean=mean, data_std=std)
            else:
                ld.dataset = types.SimpleNamespace(Y_data_mean=mean, Y_data_std=std)
            loaders[phase] = ld
        logs = []
        train.log_value = lambda n, v, e: logs.append((n, float(v), int(e)))
        train.tqdm = lambda it: it
        rc = train.train_loop((model_g, model_d), (og, od), loaders, w_d=case["w_d"], mse_w=case["mse_w"],
                              mge_w=case["mge_w"], update_d=case["update_d"], update_g=case["update_g"],
                              reference_discriminator=ref_d)
        assert rc == 0
        out = {"log.names": np.array([n for n, _, _ in logs]), "log.values": np.array([v for _, v, _ in logs]),
               "log.epochs": np.array([e for _, _, e in logs]),
               "lr_g": np.float64(og.param_groups[0]["lr"]), "lr_d": np.float64(od.param_groups[0]["lr"])}
        for k, v in model_g.state_dict().items():
            out["G." + k] = v.numpy().copy()
        for k, v in model_d.state_dict().items():
            out["D." + k] = v.numpy().copy()
        return out
    finally:
        hp.__dict__.clear()
        hp.__dict__.update(saved)


def run_inference():
    """Eval-mode generator forwards with the real gantts.models classes, the real
    evaluation_tts.gen_parameters (evaluation_tts.py:47-100) and the lines of gen_duration /
    test_vc_from_path that sit between the front end / vocoder and the model."""
    train, hparams, gantts = ref_loader.load_reference()
    ev = ref_loader.load_reference_evaluation_tts()
    import gantts_oracle as O
    from nnmnkwii.paramgen import unit_variance_mlpg_matrix
    I, inp = C.INFERENCE, C.make_inference_inputs()
    out = {}

    def build(spec, seed):
        kw = {k: v for k, v in spec.items() if k != "kind"}
        m = getattr(gantts.models, spec["kind"])(**kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in C.make_weights(spec, seed).items()})
        return m.eval()

    def mm(x, lo, hi):       # nnmnkwii.preprocessing.minmax_scale(x, min, max, feature_range=(0.01, 0.99))
        return (x - lo) / (hi - lo) * (0.99 - 0.01) + 0.01

    with torch.no_grad():
        for tag, spec in (("lstm", I["acoustic"]), ("mlp", I["acoustic_mlp"])):
            x = torch.from_numpy(mm(inp["feats_acoustic"], inp["X_min_acoustic"], inp["X_max_acoustic"])).float()
            xl = len(x)
            pred = build(spec, 41)(x.view(1, -1, x.size(-1)), [xl]).numpy().reshape(-1, 187)
            out["acoustic_predicted." + tag] = pred
            mgc, lf0, vuv, bap = ev.gen_parameters(pred, {"acoustic": inp["Y_mean_acoustic"]}, {"acoustic": inp["Y_std_acoustic"]})
            for n, v in (("mgc", mgc), ("lf0", lf0), ("vuv", vuv), ("bap", bap)):
                out["%s.%s" % (n, tag)] = np.asarray(v)
        x = torch.from_numpy(mm(inp["feats_duration"], inp["X_min_duration"], inp["X_max_duration"])).float()
        pred = build(I["duration"], 42)(x.view(1, -1, x.size(-1)), [len(x)]).numpy().reshape(-1, 5)
        d = np.round(pred * inp["Y_std_duration"] + inp["Y_mean_duration"])      # evaluation_tts.py:171-172
        d[d <= 0] = 1
        out["durations"] = d
        # evaluation_vc.py:56-92
        hp = hparams.vc
        mc = O.unit_variance_mlpg_matrix  # (silence linters)
        W = np.vstack([O._window_matrix(l, u, c, I["T_vc"]) for (l, u, c) in C.WINDOWS])
        st = inp["vc_static"].astype(np.float64)
        mc = (W @ st).reshape(3, I["T_vc"], 25).transpose(1, 0, 2).reshape(I["T_vc"], 75).astype(np.float32)   # P.delta_features
        mc_scaled = torch.from_numpy(((mc - inp["vc_mean"]) / inp["vc_std"]).astype(np.float32)).view(1, I["T_vc"], 75)
        R = torch.from_numpy(unit_variance_mlpg_matrix(C.WINDOWS, I["T_vc"]))
        y_hat, y_hat_static = build(I["vc"], 43)(mc_scaled, R, lengths=[I["T_vc"]])
        pred = y_hat_static.numpy().reshape(-1, 25)
        pred = pred * inp["vc_std"][:25] + inp["vc_mean"][:25]
        out["vc_mc"] = mc
        out["vc_outputs"] = pred
        out["vc_diff"] = pred - mc[:, :25]
    path = os.path.join(HERE, "inference.npz")
    np.savez_compressed(path, **out)
    print("inference -> %s: %s" % (os.path.relpath(path, ROOT), {k: v.shape for k, v in out.items()}))


def main():
    only = sys.argv[1:]
    if not only or "inference" in only:
        run_inference()
    for name, case in C.TRAIN_LOOP_CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        out = run_train_loop(name, case)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s -> %s (%d log values)" % (name, os.path.relpath(path, ROOT), len(out["log.values"])))
        for n, v, e in list(zip(out["log.names"], out["log.values"], out["log.epochs"]))[-12:]:
            print("    [%d] %-28s %.6f" % (e, n, v))
    if not only or "distortions" in only:
        run_distortions()
    for name, case in C.CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        out = run_case(name, case)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s -> %s (%.1f KB)" % (name, os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))
        for k in sorted(out):
            if "scalars" in k or "leak" in k:
                print("    %-18s %s" % (k, np.array2string(np.atleast_1d(out[k]), precision=6)))


if __name__ == "__main__":
    main()
