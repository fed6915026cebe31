#!/usr/bin/env python
"""Build-container only: times the REAL reference step (train.apply_generator / update_discriminator / update_generator of
/root/reference, imported through oracle/ref_loader.py) and the oracle port (oracle/gantts_oracle.py -- what
bench.py's cpu_baseline leg times on the GPU box, where the reference tree does not exist) side by side on THIS host, same
cfg2 workload (B=32, T=512, MLP G 425-512x3-187 + conditioned MLP D 483-256x3-1, Adagrad, dropout 0.5 drawn by torch's own
CPU RNG in both), interleaved steps, median of --steps after one warm-up each.  Prints one JSON line; BASELINE.md quotes it
so that `cpu_baseline.kind = "port"` is backed by a measured equivalence."""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import gantts_oracle as O  # noqa: E402
import ref_loader  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
B, T = args.batch, args.frames
G_SPEC = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=512, dropout=0.5, last_sigmoid=False)
D_SPEC = dict(in_dim=483, out_dim=1, num_hidden=3, hidden_dim=256, dropout=0.5, last_sigmoid=True)
OPT = dict(lr=0.01, weight_decay=1e-7)
torch.set_num_threads(os.cpu_count())
g = torch.Generator().manual_seed(0)
x = torch.rand(B, T, 425, generator=g)
y = torch.randn(B, T, 187, generator=g)
lengths = [T] * B

train, hparams, gantts = ref_loader.load_reference()
from gantts.multistream import get_static_features  # noqa: E402
from gantts.seqloss import sequence_mask  # noqa: E402
from nnmnkwii.paramgen import unit_variance_mlpg_matrix  # noqa: E402
hp = hparams.tts_acoustic
train.hp = hp
R = torch.from_numpy(unit_variance_mlpg_matrix(hp.windows, T))
torch.manual_seed(0)
rg, rd = gantts.models.MLP(**G_SPEC).train(), gantts.models.MLP(**D_SPEC).train()
rog, rod = torch.optim.Adagrad(rg.parameters(), **OPT), torch.optim.Adagrad(rd.parameters(), **OPT)
sl = torch.tensor(lengths)
mask = sequence_mask(sl).unsqueeze(-1)
y_static = get_static_features(y, len(hp.windows), hp.stream_sizes, hp.has_dynamic_features)


def ref_step():
    rog.zero_grad(), rod.zero_grad()
    yh, yhs = train.apply_generator(rg, x, R, lengths)
    train.update_discriminator(rd, rod, x, y_static, yhs, lengths, mask, "train")
    train.update_generator(rg, rd, rog, x, y, yh, y_static, yhs, 1.0, lengths, mask, "train", mse_w=0.0, mge_w=1.0)


og_, od_ = O.OracleMLP(seed=1, **G_SPEC), O.OracleMLP(seed=2, **D_SPEC)
oog, ood = O.OracleAdagrad(og_.params, **OPT), O.OracleAdagrad(od_.params, **OPT)
cfg = O.StreamConfig([180, 3, 1, 3], [True, True, False, True], 3, [True, False, False, False], 2, True)
omask = O.sequence_mask(lengths, T).unsqueeze(-1)


def oracle_step():
    O.train_step(cfg, og_, od_, oog, ood, x, y, R, lengths, omask, adv_w=1.0, mse_w=0.0, mge_w=1.0)


ref_step(), oracle_step()
tr, to = [], []
for _ in range(args.steps):
    t0 = time.time(); ref_step(); tr.append(time.time() - t0)
    t0 = time.time(); oracle_step(); to.append(time.time() - t0)
tr.sort(), to.sort()
mr, mo = tr[len(tr) // 2], to[len(to) // 2]
print(json.dumps({"host_threads": torch.get_num_threads(), "workload": "cfg2 B=%d T=%d fp32 dropout 0.5" % (B, T),
                  "reference_s_per_step": mr, "oracle_port_s_per_step": mo, "reference_frames_per_s": B * T / mr,
                  "oracle_port_frames_per_s": B * T / mo, "port_over_reference_time": mo / mr, "steps": args.steps}))
