"""Generate the at-size fixtures tests/golden/at_size_<case>.npz (build container only; minutes of CPU per case).

    python tests/golden/make_at_size.py [case ...]

Every case of ``at_size.AT_SIZE_CASES`` is run twice -- float32 (the reference's arithmetic) and float64 (the arbiter) --
and digested by ``at_size.digest``.  ``source="reference"`` cases execute the REAL reference (``train.apply_generator /
update_discriminator / update_generator`` with ``gantts.models`` modules and ``torch.optim``, imported from
/root/reference through oracle/ref_loader.py, in the order train.py:528-585 uses them; generator noise z is concatenated
as train.py:542 does); ``source="oracle"`` cases (SRU: un-vendored third-party cell, parity unpinned) run
oracle/gantts_oracle.py.  Dropout masks are injected exactly as in make_golden.py.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import at_size as A  # noqa: E402
import cases as C  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy().copy()


class SlopeCensus(object):
    """Records, for every LeakyReLU evaluation of a run (gantts/models.py:132: nn.LeakyReLU(inplace=True) -> F.leaky_relu; the
    oracle calls F.leaky_relu directly), which slope each activation took (output > 0).  Two runs of one case (float32, float64)
    make the same calls in the same order, so the number of activations whose slope DIFFERS between them is a direct count of
    the LeakyReLU flips a correct float32 evaluation has on this case -- what at_size.kink_allowance() is derived from."""

    def __init__(self):
        self.signs = []
        self._orig = None

    def __enter__(self):
        import torch.nn.functional as F
        self._orig = F.leaky_relu
        census = self

        def patched(input, negative_slope=0.01, inplace=False):
            out = census._orig(input, negative_slope, inplace)
            census.signs.append((out > 0).detach().cpu().numpy())
            return out

        F.leaky_relu = patched
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.leaky_relu = self._orig

    @staticmethod
    def compare(a, b, n_g_calls_per_step, calls_per_step, first_step_only=False):
        """-> {"G": (flips, activations), "D": (flips, activations)}: the first n_g_calls_per_step calls of a step are G's."""
        assert len(a.signs) == len(b.signs), (len(a.signs), len(b.signs))
        out = {"G": [0, 0], "D": [0, 0]}
        for i, (x, y) in enumerate(zip(a.signs, b.signs)):
            if first_step_only and i >= calls_per_step:
                break
            assert x.shape == y.shape
            net = "G" if (i % calls_per_step) < n_g_calls_per_step else "D"
            out[net][0] += int((x != y).sum())
            out[net][1] += int(x.size)
        return out


def _mlp_hidden_perms(spec, seed):
    """One permutation per hidden activation of an MLP discriminator (make_drift_spread.py)."""
    rs = np.random.RandomState(seed)
    return [rs.permutation(spec["hidden_dim"]) for _ in range(spec["num_hidden"])]


def _permute_mlp(sd, perms, inverse=False):
    """Relabel the hidden units of a gantts.models.MLP state dict (layers.<l>.weight / .bias, last_linear.weight): unit i of the
    relabelled network's layer l is unit perms[l][i] of the original.  The function the network computes is unchanged; the ORDER in
    which every product over a hidden layer sums its terms is not -- another correct float32 evaluation of the same step.
    inverse: map tensors of the relabelled network (gradients, updates) back to the original labelling."""
    out = dict(sd)
    L = len(perms)
    for l in range(L):
        w, b = "layers.%d.weight" % l, "layers.%d.bias" % l
        W, B = np.array(out[w]), np.array(out[b])
        if not inverse:
            W, B = W[perms[l]], B[perms[l]]
            if l > 0:
                W = W[:, perms[l - 1]]
        else:
            Wo, Bo = np.empty_like(W), np.empty_like(B)
            Wo[perms[l]], Bo[perms[l]] = W, B
            W, B = Wo, Bo
            if l > 0:
                Wo = np.empty_like(W)
                Wo[:, perms[l - 1]] = W
                W = Wo
        out[w], out[b] = W, B
    W = np.array(out["last_linear.weight"])
    if not inverse:
        W = W[:, perms[L - 1]]
    else:
        Wo = np.empty_like(W)
        Wo[:, perms[L - 1]] = W
        W = Wo
    out["last_linear.weight"] = W
    return out


def run_reference(case, dtype, d_perm_seed=None):
    """The real reference on one at-size case; returns {name: full tensor}.
    d_perm_seed (make_drift_spread.py only; MLP discriminators): run the step on a discriminator whose hidden units are relabelled
    (_permute_mlp; its dropout masks relabelled alike) and report its gradients / updates in the original labelling."""
    import ref_loader
    train, hparams, gantts = ref_loader.load_reference()
    from gantts.multistream import get_static_features
    from gantts.seqloss import sequence_mask
    from nnmnkwii.paramgen import unit_variance_mlpg_matrix
    hp = getattr(hparams, case["hp"])
    saved = dict(hp.__dict__)
    orig_dropout = torch.nn.Dropout.forward
    try:
        windows = C.WINDOWS[:case["windows"]]
        hp.__dict__.update(
            stream_sizes=case["stream_sizes"], has_dynamic_features=case["has_dynamic_features"], windows=windows,
            adversarial_streams=case["adversarial_streams"], mask_nth_mgc_for_adv_loss=case["mask_nth_mgc"],
            discriminator_linguistic_condition=case["cond"])
        train.hp = hp

        def build(spec, seed):
            kw = {k: v for k, v in spec.items() if k != "kind"}
            m = getattr(gantts.models, spec["kind"])(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in C.make_weights(spec, seed).items()})
            return m.to(dtype)

        model_g, model_d = build(case["g"], 11), build(case["d"], 22)
        d_perms = None
        if d_perm_seed is not None:
            assert case["d"]["kind"] == "MLP"
            d_perms = _mlp_hidden_perms(case["d"], d_perm_seed)
            sd0 = {k: _np(v) for k, v in model_d.state_dict().items()}
            model_d.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in _permute_mlp(sd0, d_perms).items()})
            model_d.to(dtype)
        w0 = {"G." + k: _np(v) for k, v in model_g.state_dict().items()}
        w0.update({"D." + k: _np(v) for k, v in model_d.state_dict().items()})
        og = getattr(torch.optim, case["opt_g"][0])(model_g.parameters(), **case["opt_g"][1])
        od = getattr(torch.optim, case["opt_d"][0])(model_d.parameters(), **case["opt_d"][1])
        queue = []

        def patched(self, inp):
            if not self.training or self.p == 0:
                return inp
            m = queue.pop(0)
            assert m.shape == inp.shape, (m.shape, inp.shape)
            return inp * m.to(inp.dtype) / (1.0 - self.p)

        torch.nn.Dropout.forward = patched
        model_g.train(), model_d.train()
        x_np, y_np, lengths, z_np = A.make_inputs(case)
        x, y = torch.from_numpy(x_np).to(dtype), torch.from_numpy(y_np).to(dtype)
        gin = torch.cat((x, torch.from_numpy(z_np).to(dtype)), -1) if z_np is not None else x
        T = case["T"]
        has_dyn = bool(np.any(case["has_dynamic_features"]))
        R = torch.from_numpy(unit_variance_mlpg_matrix(windows, T)).to(dtype) if has_dyn else None
        sl = torch.from_numpy(lengths)
        cpu_lengths = list(sl)
        out = {}
        for step in range(case["steps"]):
            gm, dm = C.make_dropout_masks(case, step)
            if d_perms is not None:      # the masks of D's hidden layer l follow its relabelling (three passes of D per step)
                dm = [np.ascontiguousarray(m[..., d_perms[i % len(d_perms)]]) for i, m in enumerate(dm)]
            queue[:] = [torch.from_numpy(m) for m in gm + dm]
            y_static = get_static_features(y, len(windows), hp.stream_sizes, hp.has_dynamic_features)
            mask = sequence_mask(sl).unsqueeze(-1).to(dtype)
            og.zero_grad(), od.zero_grad()
            y_hat, y_hat_static = train.apply_generator(model_g, gin, R, cpu_lengths)
            if step == 0:
                out["y_hat"], out["y_hat_static"] = _np(y_hat), _np(y_hat_static)
            res = train.update_discriminator(model_d, od, x, y_static, y_hat_static, cpu_lengths, mask, "train")
            out["d_scalars_%d" % step] = np.array(res, dtype=np.float64)
            if step == 0:
                for (k, _), p in zip(model_d.state_dict().items(), model_d.parameters()):
                    out["Dgrad." + k] = _np(p.grad)
            res = train.update_generator(model_g, model_d, og, x, y, y_hat, y_static, y_hat_static, case["adv_w"],
                                         cpu_lengths, mask, "train", mse_w=case["mse_w"], mge_w=case["mge_w"])
            out["g_scalars_%d" % step] = np.array(res, dtype=np.float64)
            if step == 0:
                for (k, _), p in zip(model_g.state_dict().items(), model_g.parameters()):
                    out["Ggrad." + k] = _np(p.grad)
            assert not queue, "unused dropout masks: %d" % len(queue)
        for k, v in model_g.state_dict().items():
            out["Gupd." + k] = _np(v) - w0["G." + k]
        for k, v in model_d.state_dict().items():
            out["Dupd." + k] = _np(v) - w0["D." + k]
        if d_perms is not None:
            for pre in ("Dupd.", "Dgrad."):
                back = _permute_mlp({k[len(pre):]: v for k, v in out.items() if k.startswith(pre)}, d_perms, inverse=True)
                out.update({pre + k: v for k, v in back.items()})
        return out
    finally:
        torch.nn.Dropout.forward = orig_dropout
        hp.__dict__.clear()
        hp.__dict__.update(saved)


def run_oracle(case, dtype):
    """oracle/gantts_oracle.py on one at-size case (same record layout as run_reference)."""
    import gantts_oracle as O
    from oracle_runner import build_oracle_model, stream_config
    cfg = stream_config(case)
    mg, md = build_oracle_model(case["g"], 11), build_oracle_model(case["d"], 22)
    O.cast_model(mg, dtype), O.cast_model(md, dtype)
    mg.training = md.training = True
    w0 = {"G." + n: _np(p) for n, p in zip(mg.names, mg.params)}
    w0.update({"D." + n: _np(p) for n, p in zip(md.names, md.params)})
    og = O.make_optimizer(case["opt_g"][0], mg.params, **case["opt_g"][1])
    od = O.make_optimizer(case["opt_d"][0], md.params, **case["opt_d"][1])
    x_np, y_np, lengths, z_np = A.make_inputs(case)
    x, y = torch.from_numpy(x_np).to(dtype), torch.from_numpy(y_np).to(dtype)
    gin = torch.cat((x, torch.from_numpy(z_np).to(dtype)), -1) if z_np is not None else x
    T = case["T"]
    has_dyn = bool(np.any(case["has_dynamic_features"]))
    R = torch.from_numpy(O.unit_variance_mlpg_matrix(C.WINDOWS[:case["windows"]], T)).to(dtype) if has_dyn else None
    mask = O.sequence_mask(lengths).unsqueeze(-1).to(dtype)
    out = {}
    for step in range(case["steps"]):
        gm, dm = C.make_dropout_masks(case, step)
        dg = O._DropoutSource([torch.from_numpy(m).to(dtype) for m in gm])
        dd = O._DropoutSource([torch.from_numpy(m).to(dtype) for m in dm])
        y_static = O.get_static_features(y, cfg.num_windows, cfg.stream_sizes, cfg.has_dynamic_features)
        og.zero_grad(), od.zero_grad()
        y_hat, y_hat_static = O.apply_generator(cfg, mg, gin, R, list(lengths), drop=dg)
        if step == 0:
            out["y_hat"], out["y_hat_static"] = _np(y_hat), _np(y_hat_static)
        res = O.update_discriminator(cfg, md, od, x, y_static, y_hat_static, list(lengths), mask, "train", drop=dd)
        out["d_scalars_%d" % step] = np.array(res, dtype=np.float64)
        if step == 0:
            for n, p in zip(md.names, md.params):
                out["Dgrad." + n] = _np(p.grad)
        res = O.update_generator(cfg, mg, md, og, x, y, y_hat, y_static, y_hat_static, case["adv_w"], list(lengths), mask,
                                 "train", mse_w=case["mse_w"], mge_w=case["mge_w"], drop=dd)
        out["g_scalars_%d" % step] = np.array(res, dtype=np.float64)
        if step == 0:
            for n, p in zip(mg.names, mg.params):
                out["Ggrad." + n] = _np(p.grad)
    for n, p in zip(mg.names, mg.params):
        out["Gupd." + n] = _np(p) - w0["G." + n]
    for n, p in zip(md.names, md.params):
        out["Dupd." + n] = _np(p) - w0["D." + n]
    return out


def add_projections(names):
    """python tests/golden/make_at_size.py --add-proj [case ...]: adds the full-tensor random projections (at_size.proj_of) to EXISTING
    fixtures from a fresh float64 run of the case -- every key the fixture already holds must come out bit-identical (norm and sample
    are re-derived and compared), only `<tensor>.proj` keys are new."""
    for name, case in A.AT_SIZE_CASES.items():
        if names and name not in names:
            continue
        path = os.path.join(HERE, "at_size_%s.npz" % name)
        fx = dict(np.load(path))
        if all((k[:-7] + ".proj") in fx for k in fx if k.endswith(".sample")):
            print("%-14s already has projections" % name, flush=True)
            continue
        runner = run_reference if case["source"] == "reference" else run_oracle
        torch.manual_seed(0)
        t0 = time.time()
        run = runner(case, torch.float64)
        n = 0
        for k, v in run.items():
            if "scalars" in k:
                assert np.array_equal(fx[k + ".f64"], np.asarray(v, dtype=np.float64)), k
                continue
            v64 = np.asarray(v, dtype=np.float64)
            assert np.float64(np.sqrt((v64 * v64).sum())) == fx[k + ".norm"], (name, k, "norm differs from the committed fixture")
            assert np.array_equal(A.sample_of(k, v64).astype(np.float32), fx[k + ".sample"]), (name, k)
            fx[k + ".proj"] = A.proj_of(k, v64)
            n += 1
        np.savez_compressed(path, **fx)
        print("%-14s %d tensors projected in %.0f s (float64 run reproduced the committed digest bit for bit)" % (name, n, time.time() - t0), flush=True)


def main():
    only = sys.argv[1:]
    torch.set_num_threads(int(os.environ.get("AT_SIZE_THREADS", os.cpu_count())))
    if only and only[0] == "--add-proj":
        return add_projections(only[1:])
    for name, case in A.AT_SIZE_CASES.items():
        if only and name not in only:
            continue
        runner = run_reference if case["source"] == "reference" else run_oracle
        runs, secs, census = {}, {}, {}
        for dtype in (torch.float32, torch.float64):
            torch.manual_seed(0)
            t0 = time.time()
            with SlopeCensus() as census[dtype]:
                runs[dtype] = runner(case, dtype)
            secs[dtype] = time.time() - t0
            print("%-14s %-8s %s: %.1f s" % (name, case["source"], str(dtype).split(".")[1], secs[dtype]), flush=True)
        fx = A.digest(runs[torch.float32], runs[torch.float64], cold=bool(case.get("cold")))
        n_g = case["g"]["num_hidden"] if case["g"]["kind"] in ("MLP", "In2OutHighwayNet") else 0
        calls = len(census[torch.float32].signs) // case["steps"]
        for net, (flips, acts) in SlopeCensus.compare(census[torch.float32], census[torch.float64], n_g, calls).items():
            fx["kink.%s.flips" % net], fx["kink.%s.activations" % net] = np.int64(flips), np.int64(acts)
            print("    LeakyReLU slope census %s: %d of %d activations differ between the float32 and the float64 run" % (net, flips, acts), flush=True)
        # the first step alone (its gradients see only its own flips; after the first update the two precisions' parameters differ
        # and every pre-activation within that difference of 0 flips)
        for net, (flips, acts) in SlopeCensus.compare(census[torch.float32], census[torch.float64], n_g, calls, first_step_only=True).items():
            fx["kink.%s.flips_step0" % net], fx["kink.%s.activations_step0" % net] = np.int64(flips), np.int64(acts)
        fx["meta.source"] = np.array(case["source"])
        fx["meta.seconds_f32"], fx["meta.seconds_f64"] = np.float64(secs[torch.float32]), np.float64(secs[torch.float64])
        path = os.path.join(os.environ.get("AT_SIZE_OUT", HERE), "at_size_%s.npz" % name)
        np.savez_compressed(path, **fx)
        print("%-14s -> %s (%.0f KB)" % (name, os.path.relpath(path, ROOT), os.path.getsize(path) / 1024), flush=True)
        for k in sorted(fx):
            if "scalars" in k:
                print("    %-22s %s" % (k, np.array2string(fx[k], precision=7)))
        worst = sorted(((float(fx[k]), k) for k in fx if k.endswith(".err32")), reverse=True)[:6]
        print("    largest float32-vs-float64 distances: " + ", ".join("%s %.2e" % (k[:-6], v) for v, k in worst), flush=True)


if __name__ == "__main__":
    main()
