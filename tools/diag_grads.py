#!/usr/bin/env python
"""Diagnostic (GPU box): per-tensor gradient error of one full-size cfg2 step, HIP vs CPU oracle, with dropout
off / Philox / the same masks injected as buffers.  Prints one line per parameter tensor."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases as C  # noqa: E402
import gantts_oracle as O  # noqa: E402
import gantts_amd.train as T  # noqa: E402
from gantts_amd import _lib as L, hparams, optim, paramgen  # noqa: E402
from gantts_amd.engine import engine_for  # noqa: E402
from gantts_amd.multistream import get_static_features  # noqa: E402
from gantts_amd.seqloss import sequence_mask  # noqa: E402
from hip_runner import build_model  # noqa: E402

B, Tn, gh, dh, p = int(os.environ.get("DB", 32)), int(os.environ.get("DT", 512)), int(os.environ.get("DGH", 512)), int(os.environ.get("DDH", 256)), 0.5
N = B * Tn
gs = dict(kind="MLP", in_dim=425, out_dim=187, num_hidden=3, hidden_dim=gh, dropout=p, last_sigmoid=False)
ds = dict(kind="MLP", in_dim=483, out_dim=1, num_hidden=3, hidden_dim=dh, dropout=p, last_sigmoid=True)
case = dict(B=B, T=Tn, din=425, dout=187, stream_sizes=[180, 3, 1, 3])
x_np, y_np, lengths = C.make_batch(case, seed=11)
hp = types.SimpleNamespace(**hparams.tts_acoustic.values())
T.hp = hp
R_np = np.array(paramgen.unit_variance_mlpg_matrix(hp.windows, Tn))
okw = dict(lr=0.01, weight_decay=1e-7)
cfg = O.StreamConfig([180, 3, 1, 3], [True, True, False, True], 3, [True, False, False, False], 2, True)


def run(mode):
    mg, md = build_model(gs, 1), build_model(ds, 2)
    (mg.train(), md.train()) if mode != "off" else (mg.eval(), md.eval())
    og, od = optim.Adagrad(mg.parameters(), initial_accumulator_value=1e-4, **okw), optim.Adagrad(md.parameters(), initial_accumulator_value=1e-4, **okw)
    eng = engine_for(hp, mg)
    eng.set_seed(1234)
    x, y, R = torch.from_numpy(x_np).cuda(), torch.from_numpy(y_np).cuda(), torch.from_numpy(R_np).cuda()
    ys = get_static_features(y, 3, hp.stream_sizes, hp.has_dynamic_features)
    mask = sequence_mask(torch.from_numpy(lengths).cuda()).unsqueeze(-1)
    gm = dm = None
    if mode != "off":
        gmd = [eng.philox_mask(L.ROLE_G, 0, l, p, N, gh) for l in range(3)]
        d0 = [eng.philox_mask(L.ROLE_D, 0, l, p, 2 * N, dh) for l in range(3)]
        d2 = [eng.philox_mask(L.ROLE_D, 2, l, p, N, dh) for l in range(3)]
        gm = [m.cpu().view(B, Tn, gh) for m in gmd]
        dm = [m[:N].cpu().view(B, Tn, dh) for m in d0] + [m[N:].cpu().view(B, Tn, dh) for m in d0] + [m.cpu().view(B, Tn, dh) for m in d2]
        if mode == "buffer":
            mg.set_dropout_masks(0, gm)
            md.set_dropout_masks(0, dm[0:3]); md.set_dropout_masks(1, dm[3:6]); md.set_dropout_masks(2, dm[6:9])
    og.zero_grad(), od.zero_grad()
    yh, yhs = T.apply_generator(mg, x, R, list(lengths))
    d = T.update_discriminator(md, od, x, ys, yhs, list(lengths), mask, "train")
    dgrad = md.flat_grads().cpu().numpy().copy()
    g = T.update_generator(mg, md, og, x, y, yh, ys, yhs, 1.0, list(lengths), mask, "train", mse_w=0.0, mge_w=1.0)
    ggrad = mg.flat_grads().cpu().numpy().copy()
    # oracle
    omg = O.OracleMLP(**{k: v for k, v in gs.items() if k != "kind"}); omd = O.OracleMLP(**{k: v for k, v in ds.items() if k != "kind"})
    omg.load_state_dict(C.make_weights(gs, 1)), omd.load_state_dict(C.make_weights(ds, 2))
    omg.training = omd.training = mode != "off"
    oog, ood = O.OracleAdagrad(omg.params, **okw), O.OracleAdagrad(omd.params, **okw)
    for o in (oog, ood):
        for s_ in o.sum:
            s_.fill_(1e-4)
    xc, yc, Rc = torch.from_numpy(x_np), torch.from_numpy(y_np), torch.from_numpy(R_np)
    omask = O.sequence_mask(lengths, Tn).unsqueeze(-1)
    oys = O.get_static_features(yc, 3, cfg.stream_sizes, cfg.has_dynamic_features)
    dd = O._DropoutSource(dm) if dm else None
    oyh, oyhs = O.apply_generator(cfg, omg, xc, Rc, list(lengths), drop=O._DropoutSource(gm) if gm else None)
    od_ = O.update_discriminator(cfg, omd, ood, xc, oys, oyhs, list(lengths), omask, "train", drop=dd)
    rd = [q.grad.numpy().copy() for q in omd.params]
    leak = [q.grad.numpy().copy() for q in omg.params]
    og_ = O.update_generator(cfg, omg, omd, oog, xc, yc, oyh, oys, oyhs, 1.0, list(lengths), omask, "train", mse_w=0.0, mge_w=1.0, drop=dd)
    rg = [q.grad.numpy().copy() for q in omg.params]
    print("== mode %s: d %s | ref %s ; g %s | ref %s" % (mode, ["%.5f" % v for v in d], ["%.5f" % v for v in od_], ["%.5f" % v for v in g], ["%.5f" % v for v in og_]))
    for tag, flat, refs, names, model in (("G", ggrad, rg, omg.names, mg), ("D", dgrad, rd, omd.names, md)):
        off = 0
        for nm, r in zip(names, refs):
            n = r.size
            a = flat[off:off + n].reshape(r.shape).astype(np.float64)
            off += n
            err = np.abs(a - r)
            print("   %s.%-22s |ref|max %.3e  rms %.3e   max err %.3e  rel(max) %.2e  rms err %.3e  rel(rms) %.2e" % (
                tag, nm, np.abs(r).max(), np.sqrt((r.astype(np.float64) ** 2).mean()), err.max(), err.max() / np.abs(r).max(),
                np.sqrt((err ** 2).mean()), np.sqrt((err ** 2).mean()) / np.sqrt((r.astype(np.float64) ** 2).mean())))
            if tag == "G" and nm == "layers.2.bias":
                w = np.argsort(-err)[:6]
                print("      worst bias idx %s got %s ref %s" % (w, a[w], r[w]))


for mode in os.environ.get("DMODES", "off,philox,buffer").split(","):
    run(mode)
