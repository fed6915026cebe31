#!/usr/bin/env python
"""cfg5-shaped measurement (BASELINE.json configs[4]): TTS duration + acoustic models trained jointly -- two (G, D)
pairs, each with its own engine in one process -- with generator_add_noise (G sees cat(x, z), train.py:504-506, 542) and
discriminator_linguistic_condition (D sees cat(x, adv streams), train.py:254-256), batch 64.  One "joint step" = one G+D
step of the duration pair (phone-level: 416 -> 5, no dynamic features, Adam) followed by one of the acoustic pair
(frame-level: 425 + 200 noise -> 187, MLPG, Adagrad).  Prints ms per joint step and acoustic frames/s (GPU box only)."""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gantts_amd.train as T  # noqa: E402
from gantts_amd import hparams, models, optim, paramgen  # noqa: E402
from gantts_amd.multistream import get_static_features  # noqa: E402
from gantts_amd.seqloss import sequence_mask  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=512, help="acoustic frames per utterance")
ap.add_argument("--phones", type=int, default=40, help="phones per utterance (duration model)")
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
B = args.batch
gen = torch.Generator().manual_seed(0)


def make_pair(hp_set, in_dim, out_dim, Tn, opt, noise):
    hp = types.SimpleNamespace(**hp_set.values())
    hp.generator_add_noise = noise
    nz = hp.generator_noise_dim if noise else 0
    static_dim = sum(s // (len(hp.windows) if d else 1) for s, d in zip(hp.stream_sizes, hp.has_dynamic_features))
    T.hp = hp
    adv = get_static_features(torch.zeros(1, 1, out_dim, device="cuda"), len(hp.windows), hp.stream_sizes, hp.has_dynamic_features)
    adv_dim = T.get_selected_static_stream(adv).shape[-1] if hp.adversarial_streams is not None else static_dim
    mg = models.MLP(in_dim=in_dim + nz, out_dim=out_dim, num_hidden=3, hidden_dim=512, dropout=0.5, last_sigmoid=False).cuda().train()
    md = models.MLP(in_dim=in_dim + adv_dim, out_dim=1, num_hidden=3, hidden_dim=256, dropout=0.5, last_sigmoid=True).cuda().train()
    if opt == "Adam":
        og, od = optim.Adam(mg.parameters(), lr=0.001, weight_decay=1e-6), optim.Adam(md.parameters(), lr=0.001, weight_decay=1e-6)
    else:
        og, od = optim.Adagrad(mg.parameters(), lr=0.01, weight_decay=1e-7), optim.Adagrad(md.parameters(), lr=0.01, weight_decay=1e-7)
    x = torch.rand(B, Tn, in_dim, generator=gen).cuda()
    y = torch.randn(B, Tn, out_dim, generator=gen).cuda()
    lengths = torch.sort(torch.randint(Tn // 2, Tn + 1, (B,), generator=torch.Generator().manual_seed(5)), descending=True)[0]
    lengths[0] = Tn
    has_dyn = any(hp.has_dynamic_features)
    R = paramgen.unit_variance_mlpg_matrix_cuda(hp.windows, Tn) if has_dyn else None
    ys = get_static_features(y, len(hp.windows), hp.stream_sizes, hp.has_dynamic_features)
    mask = sequence_mask(lengths.cuda()).unsqueeze(-1)
    return types.SimpleNamespace(hp=hp, mg=mg, md=md, og=og, od=od, x=x, y=y, ys=ys, R=R, mask=mask, cl=[int(v) for v in lengths],
                                 nz=nz, Tn=Tn, noise_gen=torch.Generator(device="cuda").manual_seed(11))


def pair_step(p):
    T.hp = p.hp
    p.og.zero_grad(), p.od.zero_grad()
    gin = p.x
    if p.nz:
        z = torch.rand(B, p.Tn, p.nz, device="cuda", generator=p.noise_gen)
        gin = torch.cat((p.x, z), -1)
    yh, yhs = T.apply_generator(p.mg, gin, p.R, p.cl)
    d = T.update_discriminator(p.md, p.od, p.x, p.ys, yhs, p.cl, p.mask, "train")
    g = T.update_generator(p.mg, p.md, p.og, p.x, p.y, yh, p.ys, yhs, 1.0, p.cl, p.mask, "train", mse_w=0.0, mge_w=1.0)
    return d, g


duration = make_pair(hparams.tts_duration, 416, 5, args.phones, "Adam", True)
acoustic = make_pair(hparams.tts_acoustic, 425, 187, args.frames, "Adagrad", True)


def joint_step():
    return pair_step(duration), pair_step(acoustic)


joint_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    out = joint_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
t1 = time.perf_counter()
for _ in range(args.steps):
    pair_step(acoustic)
torch.cuda.synchronize()
da = (time.perf_counter() - t1) / args.steps
print("cfg5 duration (B=%d, %d phones, 416+%d -> 5, Adam) + acoustic (B=%d, T=%d, 425+%d -> 187, Adagrad), noise + conditioned D: "
      "%.2f ms per joint step (acoustic pair alone %.2f ms), %.0f acoustic frames/s, scalars %s"
      % (B, args.phones, duration.nz, B, args.frames, acoustic.nz, dt * 1e3, da * 1e3, B * args.frames / dt, out))
