#!/usr/bin/env python
"""cfg3-shaped measurement (BASELINE.json configs[2], fp32 here): BiLSTM generator 3x256 + conditioned MLP D,
B sequences x T frames, variable lengths.  Prints ms/step and frames/s (GPU box only)."""
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
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--gen", default="lstm", choices=["lstm", "sru", "mlp"])
ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                help="bf16: GEMM operands rounded to bfloat16, f32 accumulation, f32 master weights / state (GT_OPT_MATMUL_BF16)")
args = ap.parse_args()
B, Tn = args.batch, args.frames
hp = types.SimpleNamespace(**hparams.tts_acoustic.values())
T.hp = hp
if args.gen == "mlp":
    mg = models.MLP(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=512, dropout=0.5, last_sigmoid=False).cuda().train()
elif args.gen == "lstm":
    mg = models.LSTMRNN(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=256, bidirectional=True, dropout=0.0).cuda().train()
else:   # hparams.tts_acoustic default generator (hparams.py:211-222)
    mg = models.SRURNN(in_dim=425, out_dim=187, num_hidden=6, hidden_dim=512, bidirectional=True, dropout=0.2,
                       use_relu=1, rnn_dropout=0.2).cuda().train()
md = models.MLP(in_dim=483, out_dim=1, num_hidden=3, hidden_dim=256, dropout=0.5, last_sigmoid=True).cuda().train()
og, od = optim.Adagrad(mg.parameters(), lr=0.01, weight_decay=1e-7), optim.Adagrad(md.parameters(), lr=0.01, weight_decay=1e-7)
g = torch.Generator().manual_seed(0)
x = torch.rand(B, Tn, 425, generator=g).cuda()
y = torch.randn(B, Tn, 187, generator=g).cuda()
lengths = torch.sort(torch.randint(Tn // 2, Tn + 1, (B,), generator=torch.Generator().manual_seed(1234)), descending=True)[0]
lengths[0] = Tn
R = paramgen.unit_variance_mlpg_matrix_cuda(hp.windows, Tn)
ys = get_static_features(y, 3, hp.stream_sizes, hp.has_dynamic_features)
mask = sequence_mask(lengths.cuda()).unsqueeze(-1)
cl = [int(v) for v in lengths]
if args.dtype == "bf16":
    from gantts_amd.engine import engine_for  # noqa: E402
    engine_for(hp, mg).set_option("matmul_bf16", 1)


def step():
    og.zero_grad(), od.zero_grad()
    yh, yhs = T.apply_generator(mg, x, R, cl)
    d = T.update_discriminator(md, od, x, ys, yhs, cl, mask, "train")
    gg = T.update_generator(mg, md, og, x, y, yh, ys, yhs, 1.0, cl, mask, "train", mse_w=0.0, mge_w=1.0)
    return d, gg


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
name = {"lstm": "cfg3 BiLSTM 3x256", "sru": "SRU 6x512 bi (hparams default G)", "mlp": "cfg2 MLP 3x512"}[args.gen]
print("%s %s B=%d T=%d: %.2f ms/step, %.0f padded frames/s, scalars %s" % (name, args.dtype, B, Tn, dt * 1e3, B * Tn / dt, out))
