"""Runs a golden case through the HIP engine via the reference-shaped Python API (gantts_amd)."""
import types

import numpy as np
import torch

import cases as C


def make_hp(case):
    from gantts_amd import hparams
    base = getattr(hparams, case["hp"])
    hp = types.SimpleNamespace(**base.values())
    hp.stream_sizes = case["stream_sizes"]
    hp.has_dynamic_features = case["has_dynamic_features"]
    hp.windows = C.WINDOWS[:case["windows"]]
    hp.adversarial_streams = case["adversarial_streams"]
    hp.mask_nth_mgc_for_adv_loss = case["mask_nth_mgc"]
    hp.discriminator_linguistic_condition = case["cond"]
    return hp


def build_model(spec, seed, saturate_gates=False):
    from gantts_amd import models
    kw = {k: v for k, v in spec.items() if k != "kind"}
    m = getattr(models, spec["kind"])(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in C.make_weights(spec, seed, saturate_gates).items()})
    return m.cuda()


def run_hip_case(case, return_objects=False, engine_options=None, comm_world_1=False, shard=None, comm_id=None, extra=None,
                 philox=False, pitch_x=False):
    """shard = (rank, world): this process holds sequences rank, rank + world, ... of the case's batch (whole
    sequences dealt round-robin) and, with `comm_id`, attaches the engine's communicator (gt_comm_init) first -- the
    fused step functions are then data-parallel by themselves and every rank must reproduce the WHOLE batch's result.
    philox=True: a dropout case runs on the engine's own Philox streams (the production path) instead of injected masks;
    `extra` then receives the keep masks of the NEXT step's first G site and first D-step site (gt_op_philox_mask)."""
    import gantts_amd.train as T
    from gantts_amd import optim, paramgen
    from gantts_amd.multistream import get_static_features
    from gantts_amd.seqloss import sequence_mask
    hp = make_hp(case)
    T.hp = hp
    mg, md = build_model(case["g"], 11, case.get("saturate_gates", False)), build_model(case["d"], 22)
    if case["dropout_on"]:
        mg.train(), md.train()
    else:
        mg.eval(), md.eval()
    og = getattr(optim, case["opt_g"][0])(mg.parameters(), **case["opt_g"][1])
    od = getattr(optim, case["opt_d"][0])(md.parameters(), **case["opt_d"][1])
    if engine_options:
        from gantts_amd.engine import engine_for
        for k, v in engine_options.items():
            engine_for(hp, mg).set_option(k, v)
    if comm_world_1:      # the engine's own data-parallel path (RCCL communicator, bucketed all-reduce) with a single rank
        from gantts_amd.engine import engine_for
        eng = engine_for(hp, mg)
        eng.comm_init(0, 1, eng.comm_unique_id())
        if extra is not None and callable(extra.get("after_comm")):
            extra["after_comm"](eng)
    x_np, y_np, lengths = C.make_batch(case)
    rows = slice(None)
    if shard is not None:
        rank, world = shard
        rows = np.arange(case["B"])[rank::world]
        x_np, y_np, lengths = x_np[rows], y_np[rows], lengths[rows]
        if comm_id is not None:
            from gantts_amd.engine import engine_for
            eng = engine_for(hp, mg)
            eng.comm_init(rank, world, comm_id)
            if extra is not None and callable(extra.get("after_comm")):      # (test hook: e.g. attach the interprocess arenas)
                extra["after_comm"](eng)
    x, y = torch.from_numpy(x_np).cuda(), torch.from_numpy(y_np).cuda()
    if pitch_x:       # rows on a 16-byte pitch, as DevicePrefetcher(pitch_x=True) stages them (gt_set_x_pitch)
        from gantts_amd.engine import pitched_empty
        xp = pitched_empty(x.size(0), x.size(1), x.size(2))
        xp.copy_(x)
        x = xp
    Tn = case["T"]
    has_dyn = bool(np.any(case["has_dynamic_features"]))
    R = paramgen.unit_variance_mlpg_matrix_cuda(hp.windows, Tn) if has_dyn else None
    sl = torch.from_numpy(np.ascontiguousarray(lengths)).cuda()
    cpu_lengths = list(lengths)
    out = {}
    for step in range(case["steps"]):
        if case["dropout_on"] and not philox:
            gm, dm = C.make_dropout_masks(case, step)
            nh = len(C.hidden_sites(case["d"]))      # dropout sites of one D pass (an MLP: every hidden layer; nn.LSTM: between layers)
            mg.set_dropout_masks(0, [torch.from_numpy(m[rows]) for m in gm])
            for p in range(3):
                md.set_dropout_masks(p, [torch.from_numpy(m[rows]) for m in dm[p * nh:(p + 1) * nh]])
        y_static = get_static_features(y, len(hp.windows), hp.stream_sizes, hp.has_dynamic_features)
        mask = sequence_mask(sl, max_len=Tn).unsqueeze(-1)
        og.zero_grad()
        od.zero_grad()
        y_hat, y_hat_static = T.apply_generator(mg, x, R, cpu_lengths)
        if step == 0:
            out["y_hat"] = y_hat.cpu().numpy()
            out["y_hat_static"] = y_hat_static.cpu().numpy()
        if case["update_d"]:
            res = T.update_discriminator(md, od, x, y_static, y_hat_static, cpu_lengths, mask, "train")
            out["d_scalars_%d" % step] = np.array(res, dtype=np.float64)
            if not case["update_g"] or step == case["steps"] - 1 and False:
                pass
        if case["update_g"]:
            res = T.update_generator(mg, md, og, x, y, y_hat, y_static, y_hat_static, case["adv_w"],
                                     cpu_lengths, mask, "train", mse_w=case["mse_w"], mge_w=case["mge_w"])
            out["g_scalars_%d" % step] = np.array(res, dtype=np.float64)
    torch.cuda.synchronize()
    if extra is not None:      # the keep masks the engine would draw for this shard's rows in the next step
        from gantts_amd.engine import engine_for
        eng = engine_for(hp, mg)
        rows = x.shape[0] * Tn
        extra["philox_g"] = eng.philox_mask(0, 0, 0, 0.5, rows, 48).cpu().numpy()
        extra["philox_d"] = eng.philox_mask(1, 0, 0, 0.5, 2 * rows, 40).cpu().numpy()
    for k, v in mg.state_dict().items():
        out["G." + k] = v.cpu().numpy()
    for k, v in md.state_dict().items():
        out["D." + k] = v.cpu().numpy()
    for tag, opt, model in (("G", og, mg), ("D", od, md)):
        names = list(model.state_dict().keys())
        for i, st in opt.state_dict()["state"].items():
            for key in ("sum", "exp_avg", "exp_avg_sq"):
                if key in st:
                    out["%s.opt.%s.%s" % (tag, key, names[i])] = st[key].cpu().numpy()
    if return_objects:
        return out, (mg, md, og, od)
    return out
