#!/usr/bin/env python
"""Headline benchmark: acoustic frames/sec for one G+D GAN step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "cfg2"): TTS acoustic MLP G (425->512x3->187, LeakyReLU 0.01,
dropout 0.5) + conditioned MLP D (483->256x3->1, dropout 0.5, sigmoid), Adagrad(lr 0.01, wd 1e-7),
w_d=1, mse_w=0, mge_w=1, adv_w=1, B=32 sequences x T=512 frames per GPU, float32, all lengths = T.
One step = zero_grad x2 -> apply_generator -> update_discriminator("train") -> update_generator("train")
including both optimizer steps and the D2H of the 9 scalars (train.py:538-585); inputs are resident
in HBM before the timed region.  Scaling (`--scaling`, printed in the JSON line): **strong** by default -- the metric
is quoted on a GLOBAL minibatch of B=32 sequences ("frames = B*T padded frames per global step"), which N
ranks split into B/N whole sequences each, dealt round-robin; `--scaling weak` gives every rank its own
32x512 shard instead.  Gradients, loss sums and the valid-frame count are all-reduced over RCCL by the engine's
communicator (gt_comm_*), overlapped with the backward pass.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel = the f32
MFMA GEMM family, per-launch HIP-event timing on the launch stream) and `cpu_baseline` (the CPU
oracle -- a torch-CPU restatement of the reference step -- timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
# one entry per kernel (template instantiation family): include/gantts_hip.h, GT_PROFILE_SLOTS
VARIANTS = ["fwd X.W^T BN64 (run-time epilogue)", "fwd X.W^T BN128", "bwd-data dZ.W BN64 (run-time epilogue)", "bwd-data dZ.W BN128",
            "bwd-weight dZ^T.X BN64", "bwd-weight dZ^T.X BN128", "fwd X.W^T 64x64 <no activation>", "fwd X.W^T 64x64 <LeakyReLU + Philox>",
            "pair bwd-data dZ.W + bwd-weight dZ^T.X BN64", "fwd X.W^T 64x64 <LeakyReLU + Philox, added matrix>",
            "bwd-data dZ.W 64x64 <no activation>", "bwd-data dZ.W 64x64 <LeakyReLU + Philox>",
            "bwd-weight pair of the split first layer (x block + adversarial block)",
            "fwd split first layer 64x64 <two K segments, LeakyReLU + Philox>",
            "fused discriminator stack: layers 1..L-1 + head (+ the generator step's backward-data chain), one launch per pass", "-"]

G_SPEC = dict(in_dim=425, out_dim=187, num_hidden=3, hidden_dim=512, dropout=0.5, last_sigmoid=False)
D_SPEC = dict(in_dim=483, out_dim=1, num_hidden=3, hidden_dim=256, dropout=0.5, last_sigmoid=True)
OPT = dict(lr=0.01, weight_decay=1e-7)


def pmc_traffic(variant):
    """HBM bytes per launch of a GEMM variant from the committed PMC passes (profiles/<round>_pmc.json, written by
    tools/summarize_profile.py from `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` runs of this same command).  Counters
    cannot be collected inside this process; null when no profile of the build is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        key = "pair" if variant == 8 else ("slot%d" % variant if variant >= 6 else "%d,%d" % (variant // 2, 64 if variant % 2 == 0 else 128))
        return d["gemm_variants"][key]["hbm_bytes_per_launch"], os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__))) + ": " + d["source"]
    except Exception:
        return None, None


def algorithmic_flops_per_frame():
    """3g - g1 + 8d - d1 MACs per frame (minimal equivalent step)."""
    def macs(spec):
        ins = [spec["in_dim"]] + [spec["hidden_dim"]] * (spec["num_hidden"] - 1)
        per = [i * spec["hidden_dim"] for i in ins] + [spec["hidden_dim"] * spec["out_dim"]]
        return sum(per), per[0]
    g, g1 = macs(G_SPEC)
    d, d1 = macs(D_SPEC)
    return 2.0 * (3 * g - g1 + 8 * d - d1)


def make_hp():
    from gantts_amd import hparams
    hp = types.SimpleNamespace(**hparams.tts_acoustic.values())
    hp.generator, hp.generator_params = "MLP", dict(G_SPEC)
    hp.discriminator_params = dict(D_SPEC)
    return hp


def synthetic_batch(B, T, seed, device):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, T, 425, generator=g)
    y = torch.randn(B, T, 187, generator=g)
    vuv = (torch.rand(B, T, generator=g) > 0.5).float()
    y[:, :, 183] = (vuv - 0.5) / 0.5
    return x.to(device), y.to(device)


def cpu_baseline(B, T, budget_s=20.0):
    """Times the CPU oracle (torch-CPU restatement of train.py:245-320) on the same workload."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gantts_oracle as O
    torch.manual_seed(0)
    mg, md = O.OracleMLP(seed=1, **G_SPEC), O.OracleMLP(seed=2, **D_SPEC)
    og, od = O.OracleAdagrad(mg.params, **OPT), O.OracleAdagrad(md.params, **OPT)
    cfg = O.StreamConfig([180, 3, 1, 3], [True, True, False, True], 3, [True, False, False, False], 2, True)
    x, y = synthetic_batch(B, T, 0, "cpu")
    from gantts_amd import hparams, paramgen
    R = torch.from_numpy(__import__("numpy").array(paramgen.unit_variance_mlpg_matrix(hparams.tts_acoustic.windows, T)))
    lengths = [T] * B
    mask = O.sequence_mask(lengths, T).unsqueeze(-1)
    # Thread sweep (128 threads on a shared host were SLOWER than 8 in the build container): one warm-up + one timed step per
    # thread count, then the best count is timed for the rest of the budget.  `cores` = the threads of the reported figure.
    ncpu = os.cpu_count() or 1
    counts = sorted({n for n in (4, 8, 16, 32, 64, 128, ncpu) if n <= ncpu})

    def one_step():
        t0 = time.time()
        O.train_step(cfg, mg, md, og, od, x, y, R, lengths, mask, adv_w=1.0, mse_w=0.0, mge_w=1.0)
        return time.time() - t0
    sweep = {}
    t_start = time.time()
    for n in counts:      # ascending; stops once more threads have stopped paying (256 threads on a shared 256-CPU host: 45 s per step)
        torch.set_num_threads(n)
        one_step()
        sweep[n] = one_step()
        if sweep[n] > 1.3 * min(sweep.values()) or time.time() - t_start > 2.0 * budget_s:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]]
    t_start = time.time()
    while len(times) < 64 and (len(times) < 4 or time.time() - t_start < budget_s):
        times.append(one_step())
    times.sort()
    med = times[len(times) // 2]
    return {"value": B * T / med, "unit": "frames/s", "cores": best, "kind": "port", "host_cpus": ncpu,
            "thread_sweep_s_per_step": {str(n): sweep[n] for n in sorted(sweep)},
            "sample": "%d timed steps at the best thread count of the sweep (%d of %d host CPUs; one warm-up + one timed step per count) of the "
                      "same cfg2 workload B=%d T=%d fp32 dropout 0.5 through oracle/gantts_oracle.py (torch-CPU autograd restatement of "
                      "train.py:245-320), median %.3f s/step" % (len(times), best, ncpu, B, T, med)}


XGMI_LINK_GBS = 153.0       # per link and direction (MI355X_MICROARCH.md)


def traced_schedule():
    """The data-parallel step's messages as the engine's own trace recorded them with one rank over RCCL: the NEWEST committed
    `profiles/r??_comm_schedule_dp1_rccl.json` (`bench.py --force-dp --comm-trace 20`, collected by tools/profile_round.sh every round).
    ADVICE r5: no constants copied from a past profile -- the file is read, named in the line, and without one the with-communication
    figures are omitted."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_comm_schedule_dp1_rccl.json")))
    if not files:
        return None
    try:
        cs = json.load(open(files[-1])).get("comm_schedule", {})
        msgs = cs["messages"]
        return {"messages_per_step": len(msgs), "bytes_per_step": float(cs["bytes_per_step"]),
                "exposed_message_bytes": [float(m["bytes"]) for m in msgs if m["on_step_stream"]],
                "one_rank_exposed_us": float(cs["exposed_us_per_step"]), "one_rank_hidden_us": float(cs["hidden_us_per_step"]),
                "source": os.path.relpath(files[-1], ROOT) + " (a committed trace of an earlier run of this command with --force-dp --comm-trace, not of this run)"}
    except Exception:      # noqa: BLE001
        return None


def scaling_model(proxy_ms, one_gpu_ms):
    """`scaling_model` of the JSON line: the measured single-GPU proxy of a rank's step at B/N sequences, plus -- when a traced schedule is
    on file -- what it says must be added: the exposed messages' launch cost (measured with one rank) and the BANDWIDTH LOWER BOUND of
    their ring all-reduce over xGMI (2 (N-1)/N x bytes at one link's rate; per-hop latency not modelled).  Everything here is labelled unmeasured."""
    out = {
        "status": "UNMEASURED on more than one GPU: a single-GPU proxy (+ a bandwidth lower bound for the exposed messages of a traced schedule)",
        "ms_per_rank_step": {str(n): proxy_ms[n] for n in proxy_ms},
        "speedup_over_one_gpu_without_communication": {str(n): one_gpu_ms / proxy_ms[n] for n in proxy_ms},
    }
    sched = traced_schedule()
    if sched is None:
        out["method"] = "per-rank step of N ranks = this step on B/N whole sequences (measured below, one GPU, no communication); no traced schedule on file: nothing added for the messages"
        return out
    exposed_bytes = float(sum(sched["exposed_message_bytes"]))
    wire = {n: 1e6 * 2.0 * (n - 1) / n * exposed_bytes / (XGMI_LINK_GBS * 1e9) for n in proxy_ms}
    with_comm = {n: proxy_ms[n] + 1e-3 * (sched["one_rank_exposed_us"] + wire[n]) for n in proxy_ms}
    out.update({
        "method": "per-rank step of N ranks = this step on B/N whole sequences (measured below, one GPU, no communication); + the closing "
                  "messages of the traced schedule (%s bytes per step, on the step stream: nothing overlaps them): their one-rank cost as "
                  "traced (%.1f us) + the ring all-reduce's wire time at one xGMI link's 153 GB/s, latency not modelled -- so the speed-up WITH "
                  "communication is an UPPER bound" % (" + ".join("%d" % b for b in sched["exposed_message_bytes"]), sched["one_rank_exposed_us"]),
        "collectives": sched,
        "exposed_wire_time_lower_bound_us": {str(n): wire[n] for n in proxy_ms},
        "speedup_over_one_gpu_upper_bound_with_exposed_messages": {str(n): one_gpu_ms / with_comm[n] for n in proxy_ms},
    })
    # the same with a per-message latency on top of the wire time: a ring all-reduce of N ranks is 2 (N - 1) hops, and the builder has no
    # device pair to measure a hop on -- two assumed values per EXPOSED message bracket what RCCL over xGMI is usually quoted at for ~1 MB
    n_exposed = len(sched["exposed_message_bytes"])
    out["speedup_over_one_gpu_with_exposed_messages_and_assumed_latency"] = {
        "assumed_us_per_exposed_message": [30.0, 60.0],
        "speedup": {str(n): [one_gpu_ms / (with_comm[n] + 1e-3 * lat * n_exposed) for lat in (30.0, 60.0)] for n in proxy_ms}}
    return out


def comm_schedule(rec, steps, traced_ms, args):
    """Condenses the engine's schedule trace (gt_comm_trace_read: kind, bytes, on the step stream, start us, end us) of `steps` steps."""
    import numpy as np
    msgs, waits = rec[rec[:, 0] == 0], rec[rec[:, 0] == 1]
    n_msg = len(msgs) // steps
    out = {"source": "gt_comm_trace: timed HIP events around every message (on the stream that carries it) and around every wait of the "
                     "step stream for the communicator's stream; rank 0, %d steps behind the timed region" % steps,
           "transport": ("tests/fake_rccl.cpp (host-staged test double), every rank on ONE device: message DURATIONS are the double's, "
                         "only the schedule (count, sizes, what overlaps what) carries over to RCCL over xGMI") if args.one_device
                        else ("two-shot all-reduce over hipIpc arenas" if args.dp_ipc else "RCCL"),
           "ms_per_step_traced": traced_ms, "messages_per_step": len(msgs) / steps, "waits_per_step": len(waits) / steps,
           "bytes_per_step": float(msgs[:, 1].sum()) / steps,
           "step_stream_wait_us_per_step": float((waits[:, 4] - waits[:, 3]).sum()) / steps}
    # per position in the step (the schedule is the same every step): size, duration, exposed share
    per = []
    if n_msg and len(msgs) == n_msg * steps:
        m = msgs.reshape(steps, n_msg, 5)
        for j in range(n_msg):
            dur = m[:, j, 4] - m[:, j, 3]
            exposed = np.zeros(steps)
            for i in range(steps):
                if m[i, j, 2]:            # a closing message on the step stream itself: nothing of the step runs beside it
                    exposed[i] = dur[i]
                else:                     # the part of the message during which the step stream stood waiting
                    lo = np.maximum(waits[:, 3], m[i, j, 3])
                    hi = np.minimum(waits[:, 4], m[i, j, 4])
                    exposed[i] = np.clip(hi - lo, 0, None).sum()
            per.append({"bytes": float(m[0, j, 1]), "on_step_stream": bool(m[0, j, 2]), "duration_us": float(np.median(dur)),
                        "exposed_us": float(np.median(exposed)), "hidden_us": float(np.median(dur - exposed))})
        out["messages"] = per
        out["exposed_us_per_step"] = sum(p["exposed_us"] for p in per)
        out["hidden_us_per_step"] = sum(p["hidden_us"] for p in per)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): --batch is the GLOBAL minibatch, every rank gets batch/N sequences; "
                         "weak: every rank gets its own --batch sequences")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-companion", action="store_true", help="skip the companion measurements behind the timed region (the other x layout, "
                    "the single-GPU scaling proxy): profile runs, whose kernel trace must hold the headline's steps only")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE.json configs[2..4] behind the timed region")
    ap.add_argument("--spinup-ms", type=float, default=400.0,
                    help="GPU clock spin-up before the warm-up steps: a generic matrix product on scratch operands (NOT steps of the "
                         "workload) keeps the device busy for this long, so that the W warm-up + K timed steps run at the shader clock a "
                         "training run holds, not on the first milliseconds' ramp of an idle device (0 = off; DESIGN.md 4)")
    ap.add_argument("--spinup-order", choices=["before", "after-first"], default="after-first",
                    help="where the clock spin-up runs: before the warm-up steps, or behind the first of them (which loads the kernels)")
    ap.add_argument("--trace-steps", action="store_true", help="diagnostic: host time of every timed step to stderr")
    ap.add_argument("--dense-x", action="store_true", help="x resident with dense 425-float rows (the engine then makes the 16-byte-pitch "
                    "copy its weight-gradient products read, once per step) instead of the pitched rows the batch pipeline stages")
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel code path (RCCL all-reduce) even with one rank")
    ap.add_argument("--dp-ipc", action="store_true", help="data parallel: the step's messages over the engine's two-shot all-reduce on hipIpc "
                    "peer arenas (gt_comm_ipc_*) instead of RCCL; falls back to RCCL with a warning if the arenas cannot be attached")
    ap.add_argument("--comm-trace", type=int, default=0, metavar="STEPS",
                    help="data parallel: BEHIND the timed region, run STEPS more steps with the engine's schedule trace on (gt_comm_trace) and "
                         "add `comm_schedule` to the line: messages per step, how long each holds the communicator's stream, how long the "
                         "step stream waits for them (exposed) and how much ran under compute (hidden)")
    ap.add_argument("--engine-option", action="append", default=[], metavar="NAME=VALUE",
                    help="measurement: set an engine switch before the run (Engine.set_option), repeatable; recorded in config.engine_options")
    ap.add_argument("--one-device", action="store_true",
                    help="SCHEDULE measurement without a multi-GPU node, never a throughput claim: all ranks share device 0 and the engine's "
                         "communicator binds the tests' RCCL double (tests/fake_rccl.cpp: all-reduce staged through host shared memory), "
                         "because RCCL refuses two ranks on one device; the bootstrap group is gloo")
    ap.add_argument("--dp-python", action="store_true", help="data parallelism orchestrated from Python (torch.distributed "
                    "all-reduce between the split-phase calls) instead of the engine's own RCCL communicator")
    args = ap.parse_args()

    if args.force_dp:      # a one-rank communicator would otherwise take the plain path (a sum over one rank is the identity)
        os.environ["GT_COMM_FORCE_COLLECTIVES"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU over RCCL, rendezvous on 127.0.0.1
        # (rank 0 prints the one JSON line; this process is replaced, so its exit status is the job's)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU path")
    local_rank %= max(1, torch.cuda.device_count())
    if args.one_device:
        if args.dp_python:
            raise SystemExit("bench.py --one-device: the engine's own communicator only")
        local_rank = 0
        fake, fake_src = os.path.join(ROOT, "tests", "libfake_rccl.so"), os.path.join(ROOT, "tests", "fake_rccl.cpp")
        if rank == 0 and (not os.path.isfile(fake) or os.path.getmtime(fake) < os.path.getmtime(fake_src)):
            import subprocess
            hipcc = os.environ.get("HIPCC") or ("/opt/rocm/bin/hipcc" if os.path.isfile("/opt/rocm/bin/hipcc") else "hipcc")
            subprocess.check_call([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=" + os.environ.get("ARCH", "gfx950"),
                                   fake_src, "-o", fake, "-lrt"])
        os.environ["GT_RCCL_LIB"] = fake
        os.environ["GT_IPC_ALLOW_COARSE"] = "1"       # one device, one L2: a coarse-grained arena is coherent here (eng_ipc.hip)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    boot_dev = torch.device("cpu") if args.one_device else dev       # where the bootstrap group's tensors live
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()       # (rank 0 has built the double)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if torch.cuda.device_count() < world and rank == 0:
        print("warning: %d ranks on %d visible GPU(s)" % (world, torch.cuda.device_count()), file=sys.stderr)

    import gantts_amd.train as T
    from gantts_amd import _lib as L
    from gantts_amd import models, optim, paramgen
    from gantts_amd.engine import HipStepBackend
    from gantts_amd.multistream import get_static_features
    from gantts_amd.parallel import DataParallelStep
    from gantts_amd.seqloss import sequence_mask

    Bglobal, Tn = args.batch, args.frames
    if args.scaling == "strong":
        if Bglobal % world:
            raise SystemExit("bench.py --scaling strong: the global batch of %d sequences does not split over %d ranks" % (Bglobal, world))
        B = Bglobal // world
    else:
        B, Bglobal = args.batch, args.batch * world
    hp = make_hp()
    T.hp = hp
    torch.manual_seed(0)
    mg = models.MLP(**G_SPEC).cuda().train()
    md = models.MLP(**D_SPEC).cuda().train()
    og, od = optim.Adagrad(mg.parameters(), **OPT), optim.Adagrad(md.parameters(), **OPT)
    if args.engine_option:             # measurement: engine switches by name (gantts_amd.engine.Engine.set_option), e.g. fused_dstack=2
        from gantts_amd.engine import engine_for
        for kv in args.engine_option:
            name, _, val = kv.partition("=")
            engine_for(hp, mg).set_option(name, int(val))
    if args.scaling == "strong":       # one global batch, sequences dealt round-robin over the ranks
        xg, yg = synthetic_batch(Bglobal, Tn, 1000, "cpu")
        x, y = xg[rank::world].contiguous().to(dev), yg[rank::world].contiguous().to(dev)
        del xg, yg
    else:                              # every rank its own shard
        x, y = synthetic_batch(B, Tn, 1000 + rank, dev)
    if not args.dense_x:
        # x resident as gantts_amd.data.DevicePrefetcher(pitch_x=True) stages it: rows of 425 floats on a 428-float pitch
        # (include/gantts_hip.h: gt_set_x_pitch -- the lda of the step functions' x)
        from gantts_amd.engine import pitched_empty
        xp = pitched_empty(B, Tn, 425, dev)
        xp.copy_(x)
        x = xp
    R = paramgen.unit_variance_mlpg_matrix_cuda(hp.windows, Tn, dev)
    lengths = torch.full((B,), Tn, dtype=torch.long, device=dev)
    cpu_lengths = [Tn] * B
    y_static = get_static_features(y, len(hp.windows), hp.stream_sizes, hp.has_dynamic_features)
    mask = sequence_mask(lengths).unsqueeze(-1)

    if (world > 1 or args.force_dp) and args.dp_python:
        backend = HipStepBackend(hp, mg, md, og, od)
        dp = DataParallelStep(backend, always_reduce=args.force_dp)
        dp.broadcast_parameters(mg.flat_params(), md.flat_params())
        batch = dict(x=x, y=y, y_static=y_static, mask=mask, R=R)

        def step():
            # the global valid-frame count is all-reduced every step (device-resident, overlapped with the generator
            # forward) -- it is part of a data-parallel step for real, ragged batches; no constant is passed in
            return dp.step(batch, adv_w=1.0, mse_w=0.0, mge_w=1.0, lazy_g=True)
    else:
        if world > 1 or args.force_dp:
            # the engine's own communicator (include/gantts_hip.h gt_comm_*): rank 0's id travels over the bootstrap
            # process group once; from then on every collective of a step is issued by the engine itself -- gradient
            # buckets per layer on its communicator stream, overlapped with the backward pass
            from gantts_amd.engine import engine_for
            eng = engine_for(hp, mg)
            idt = torch.zeros(L.COMM_ID_BYTES, dtype=torch.uint8, device=boot_dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(eng.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            for m in (mg, md):
                fp = m.flat_params()
                staged = fp.to(boot_dev)
                dist.broadcast(staged, src=0)
                if staged is not fp:
                    fp.copy_(staged)
            eng.comm_init(rank, world, bytes(idt.cpu().numpy().tobytes()))
            if args.dp_ipc:
                try:
                    mine = torch.frombuffer(bytearray(eng.comm_ipc_export()), dtype=torch.uint8).to(boot_dev)
                    allh = [torch.zeros_like(mine) for _ in range(world)]
                    if world > 1:
                        dist.all_gather(allh, mine)
                    else:
                        allh = [mine]
                    eng.comm_ipc_attach(rank, world, b"".join(bytes(h.cpu().numpy().tobytes()) for h in allh))
                except Exception as e:      # noqa: BLE001 -- the collective of record is RCCL; the arenas are an option
                    print("warning: rank %d: interprocess arenas not attached (%s); RCCL carries every message" % (rank, e), file=sys.stderr)


        def step():
            og.zero_grad()
            od.zero_grad()
            y_hat, y_hat_static = T.apply_generator(mg, x, R, cpu_lengths)
            d = T.update_discriminator(md, od, x, y_static, y_hat_static, cpu_lengths, mask, "train")
            g = T.update_generator(mg, md, og, x, y, y_hat, y_static, y_hat_static, 1.0, cpu_lengths, mask,
                                   "train", mse_w=0.0, mge_w=1.0)
            return d, g

    def barrier():
        if world > 1 or args.force_dp:
            dist.barrier()

    def spin_up():
        # an idle MI355X sits at its lowest DPM state; the driver's 25 steps are 40 ms of work in total, all of it inside the
        # ramp.  A training run is never in that state, so the device is brought to its working clocks first -- by a
        # generic matrix product on scratch operands (the library's stand-alone gt_op_linear_forward: matrix-pipe load, like the
        # step's), nothing of the workload: no step function, no model parameter, no tensor the step would find warm.
        # It runs BEHIND the first warm-up step (--spinup-order): that step loads every kernel's code object and sizes every
        # scratch buffer with the host in the way, i.e. it leaves the device half idle again.
        import ctypes as C1
        ba = torch.rand(8192, 512, device=dev)
        bw = torch.rand(512, 512, device=dev)
        bo = torch.empty(8192, 512, device=dev)
        torch.cuda.synchronize()
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            for _ in range(32):
                L.check(L.lib.gt_op_linear_forward(L.ptr(ba), 512, L.ptr(bw), None, L.ptr(bo), 512, 8192, 512, 512, 0, None, C1.c_float(0.0),
                                                   L.current_stream()))
            torch.cuda.synchronize()
        del ba, bw, bo

    spin_after_first = args.spinup_order == "after-first" and args.warmup >= 1
    if args.spinup_ms > 0 and not spin_after_first:
        spin_up()
    profile = not args.no_roofline
    pre_stats = None
    for w in range(args.warmup):
        # the last warm-up step runs with EVERY product launch instrumented: the per-kernel table of the line (`variants`, the
        # family figure) comes from it, and it tells which kernel dominates -- inside the timed region only that kernel's
        # launches carry events (below)
        pre = profile and w == args.warmup - 1
        if pre:
            L.lib.gt_profile_enable(1)
        last = step()
        if w == 0 and args.spinup_ms > 0 and spin_after_first:
            spin_up()
        if pre:
            L.lib.gt_profile_enable(0)
            import ctypes as C0
            _a, _b, _c = (C0.c_double * L.PROFILE_SLOTS)(), (C0.c_double * L.PROFILE_SLOTS)(), (C0.c_int64 * L.PROFILE_SLOTS)()
            L.check(L.lib.gt_profile_read(_a, _b, _c))
            _d = (C0.c_double * L.PROFILE_SLOTS)()
            L.check(L.lib.gt_profile_bytes(_d))
            pre_stats = (list(_a), list(_b), list(_c), list(_d))
    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides --------------------
    # The dominant kernel's HIP-event timing is taken LIVE inside the timed region, on a sample of its steps: two
    # hipEventRecord per product launch are not free (~3 us each, ~25 launches), so instrumenting every launch of every step
    # would itself cost several % of `value` (measured: 1.515 ms/step un-instrumented, 1.535-1.547 with every 5th step
    # instrumented).  When the instrumented warm-up step says the pair launches dominate (slot 8: they do at cfg2), only THOSE five
    # launches of every 10th step carry events, starting with the 6th step (two steps of the driver's 20, five of the default
    # 50; ~30 us per sampled step); otherwise every launch of every 20th step, starting with the 11th.
    PAIR_SLOT, PAIR_KIND = 8, 5
    pair_only = bool(pre_stats) and max(range(L.PROFILE_SLOTS), key=lambda v: pre_stats[0][v]) == PAIR_SLOT
    PROFILE_EVERY, PROFILE_PHASE = (10, 5) if pair_only else (20, 10)
    profiled_steps = 0
    step_marks = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = profile and (i % PROFILE_EVERY == PROFILE_PHASE or args.steps <= PROFILE_PHASE and i == args.steps - 1)
        if on:
            L.lib.gt_profile_enable(2 + PAIR_KIND if pair_only else 1)
            profiled_steps += 1
        last = step()
        if on:
            L.lib.gt_profile_enable(0)
        if args.trace_steps:
            step_marks.append(time.perf_counter())
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if args.trace_steps and rank == 0:
        marks = [t0] + step_marks
        print("host ms per step: " + " ".join("%.3f" % (1e3 * (b - a)) for a, b in zip(marks, marks[1:])) +
              " | drain %.3f" % (1e3 * (t0 + elapsed - marks[-1])), file=sys.stderr)
    L.check(L.lib.gt_profile_enable(0))
    if world > 1:
        t = torch.tensor([elapsed], device=boot_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    roofline = None
    if profile:
        import ctypes as C
        NS = L.PROFILE_SLOTS
        ms, fl, cnt = (C.c_double * NS)(), (C.c_double * NS)(), (C.c_int64 * NS)()
        L.check(L.lib.gt_profile_read(ms, fl, cnt))
        alg = (C.c_double * NS)()
        L.check(L.lib.gt_profile_bytes(alg))
        live = {v: (ms[v], fl[v], cnt[v], alg[v]) for v in range(NS) if cnt[v]}
        table_steps, table_src = profiled_steps, "timed region (sampled steps)"
        if pair_only:      # the table of all kernels: the instrumented warm-up step; the dominant kernel's row: live, from the timed region
            ms, fl, cnt, alg = [list(a) for a in pre_stats]
            table_steps, table_src = 1, "the instrumented last warm-up step (untimed); the dominant kernel's row is live"
            if PAIR_SLOT in live:
                ms[PAIR_SLOT], fl[PAIR_SLOT], alg[PAIR_SLOT] = [live[PAIR_SLOT][i] / profiled_steps for i in (0, 1, 3)]
                cnt[PAIR_SLOT] = live[PAIR_SLOT][2] // profiled_steps
        per = []
        for v in range(NS):
            if cnt[v]:
                per.append({"kernel": ("dstack_kernel<%s>" if v == 14 else "gemm_f32_kernel<%s>") % VARIANTS[v], "launches_per_step": cnt[v] / table_steps,
                            "avg_us": 1e3 * ms[v] / cnt[v], "tflops": fl[v] / (ms[v] * 1e-3) / 1e12,
                            "share_of_step_ms": ms[v] / table_steps})
        tot_ms, tot_fl = sum(ms), sum(fl)
        profiled_steps_live, profiled_steps = profiled_steps, table_steps
        dom = max(per, key=lambda p: p["share_of_step_ms"])
        dom_v = [v for v in range(NS) if cnt[v] and dom["kernel"].endswith("<%s>" % VARIANTS[v])][0]
        traffic, traffic_src = pmc_traffic(dom_v)
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": dom["tflops"] / F32_MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                    "traffic_algorithmic": alg[dom_v] / cnt[dom_v],
                    "avg_launch_us": dom["avg_us"],
                    "gemm_family": {"achieved": tot_fl / (tot_ms * 1e-3) / 1e12,
                                    "frac": tot_fl / (tot_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                    "ms_per_step": tot_ms / profiled_steps, "gflop_per_step": tot_fl / profiled_steps / 1e9},
                    "sampled_steps": profiled_steps_live, "sampled_launches": "the dominant kernel's" if pair_only else "all product launches",
                    "variants_source": table_src,
                    "variants": per}

    # ---- companion measurements BEHIND the timed region (never part of `value`) -----------------------------------------------------
    def timed_steps(step_fn, n, w=3):
        for _ in range(w):
            step_fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            step_fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t1) / n

    companion = {}
    if world == 1 and not args.force_dp and not args.no_roofline and not args.no_companion:
        # (a) the other x layout: dense 425-float rows when the headline ran on the pitched rows train_loop stages (or vice versa)
        x_other = x.contiguous() if not args.dense_x else None
        if x_other is None:
            from gantts_amd.engine import pitched_empty
            x_other = pitched_empty(B, Tn, 425, dev)
            x_other.copy_(x)

        def step_other():
            og.zero_grad()
            od.zero_grad()
            y_hat, y_hat_static = T.apply_generator(mg, x_other, R, cpu_lengths)
            T.update_discriminator(md, od, x_other, y_static, y_hat_static, cpu_lengths, mask, "train")
            T.update_generator(mg, md, og, x_other, y, y_hat, y_static, y_hat_static, 1.0, cpu_lengths, mask, "train", mse_w=0.0, mge_w=1.0)
        companion["ms_per_step_dense_x" if not args.dense_x else "ms_per_step_pitched_x"] = timed_steps(step_other, args.steps)
        del x_other
        # (b) single-GPU PROXY of strong scaling (no multi-GPU node is available to the builder): the per-rank step of N ranks is the step
        # on B / N whole sequences; measured here for N = 2, 4, 8, communication NOT included
        if args.scaling == "strong" and B % 8 == 0:
            proxy = {}
            for n in (2, 4, 8):
                b = B // n
                xs, ys_, yss, ms_, cl = x[:b], y[:b].contiguous(), y_static[:b].contiguous(), mask[:b].contiguous(), cpu_lengths[:b]
                if not args.dense_x:
                    from gantts_amd.engine import pitched_empty
                    xq = pitched_empty(b, Tn, 425, dev)
                    xq.copy_(xs)
                    xs = xq
                else:
                    xs = xs.contiguous()

                def step_b(xs=xs, ys_=ys_, yss=yss, ms_=ms_, cl=cl):
                    og.zero_grad()
                    od.zero_grad()
                    y_hat, y_hat_static = T.apply_generator(mg, xs, R, cl)
                    T.update_discriminator(md, od, xs, yss, y_hat_static, cl, ms_, "train")
                    T.update_generator(mg, md, og, xs, ys_, y_hat, yss, y_hat_static, 1.0, cl, ms_, "train", mse_w=0.0, mge_w=1.0)
                proxy[n] = timed_steps(step_b, max(10, args.steps))
            companion["scaling_model"] = scaling_model(proxy, 1e3 * elapsed / args.steps)

    # (c) data parallel: the step's communication schedule from the engine's own trace (timed events around every message and every wait)
    if args.comm_trace > 0 and (world > 1 or args.force_dp) and not args.dp_python:
        for _ in range(3):
            step()
        eng.comm_trace(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.comm_trace):
            step()
        torch.cuda.synchronize()
        traced_ms = 1e3 * (time.perf_counter() - t1) / args.comm_trace
        eng.comm_trace(False)
        companion["comm_schedule"] = comm_schedule(eng.comm_trace_read(), args.comm_trace, traced_ms, args)

    # every rank flushes its C stdio (RCCL's NCCL_DEBUG=VERSION banner is block-buffered when piped)
    # before rank 0 prints, so the JSON line is the last line of the job's stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    barrier()
    if rank == 0:
        frames = Bglobal * Tn * args.steps
        value = frames / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        step_flops = algorithmic_flops_per_frame() * B * Tn          # per GPU
        out = {"metric": "acoustic frames/sec per G+D GAN step (B=32,T=512)", "value": value, "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"x_layout": "dense rows (425 floats)" if args.dense_x else "rows of 425 floats on a 428-float pitch (gt_set_x_pitch; DevicePrefetcher(pitch_x=True))",
                          "workload": "cfg2: TTS acoustic MLP G 425-512x3-187 + conditioned MLP D 483-256x3-1, "
                                      "Adagrad, MGE+ADV loss, global B=%d T=%d (%d sequences per GPU, %s scaling), fp32, "
                                      "dropout 0.5 (Philox)" % (Bglobal, Tn, B, args.scaling),
                          "global_batch": Bglobal, "per_gpu_batch": B, "frames_per_step": Bglobal * Tn,
                          "parallelism": "dp%d" % world, "collective": None if (world == 1 and not args.force_dp) else ("two-shot all-reduce over hipIpc arenas" if args.dp_ipc else
                                                                                          ("tests/fake_rccl.cpp, ALL RANKS ON ONE DEVICE: a schedule measurement, not a throughput" if args.one_device else "rccl"))},
               "step_algorithmic_gflop": step_flops / 1e9,
               "step_mfma_frac": step_flops / (ms_per_step * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
               "last_step_scalars": {"d": [float(v) for v in last[0]], "g": [float(v) for v in last[1]]},
               "roofline": roofline}
        out["config"]["clock_spinup_ms"] = args.spinup_ms
        if args.engine_option:
            out["config"]["engine_options"] = list(args.engine_option)
        if world > 1 or args.force_dp:      # what the communicator itself says (gt_comm_info): a SCALE run proves its N ranks with this
            try:
                cr, cw = eng.comm_info()
                out["config"]["rccl_ranks"] = int(cw)
                out["config"]["rccl_rank_of_reporter"] = int(cr)
            except Exception as e:      # noqa: BLE001
                out["config"]["rccl_ranks"] = "unavailable: %s" % e
        out.update(companion)
        if world == 1 and not args.no_other_configs and not args.force_dp:
            # the other BASELINE.json configurations, a few steps each, BEHIND the headline's timed region
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs
            del x, y, y_static, mask, R
            torch.cuda.empty_cache()
            out["other_configs"] = {}
            for name in bench_configs.ALL:
                try:
                    out["other_configs"][name] = bench_configs.run_config(name)
                except Exception as e:      # noqa: BLE001 -- a failing side measurement must not lose the headline line
                    out["other_configs"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Bglobal, Tn)
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
