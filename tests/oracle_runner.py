"""Runs a golden case through the CPU oracle (oracle/gantts_oracle.py)."""
import numpy as np
import torch

import cases as C
import gantts_oracle as O


def build_oracle_model(spec, seed, saturate_gates=False):
    kw = {k: v for k, v in spec.items() if k != "kind"}
    cls = {"MLP": O.OracleMLP, "In2OutHighwayNet": O.OracleIn2OutHighwayNet,
           "LSTMRNN": O.OracleLSTMRNN, "GRURNN": O.OracleLSTMRNN, "SRURNN": O.OracleSRURNN,
           "In2OutRNNHighwayNet": O.OracleIn2OutRNNHighwayNet}[spec["kind"]]
    if spec["kind"] == "GRURNN":
        kw["prefix"] = "gru"
    m = cls(**kw)
    m.load_state_dict(C.make_weights(spec, seed, saturate_gates))
    return m


def stream_config(case):
    return O.StreamConfig(case["stream_sizes"], case["has_dynamic_features"], case["windows"],
                          case["adversarial_streams"], case["mask_nth_mgc"], case["cond"])


def run_oracle_case(case):
    cfg = stream_config(case)
    mg, md = build_oracle_model(case["g"], 11, case.get("saturate_gates", False)), build_oracle_model(case["d"], 22)
    mg.training = md.training = bool(case["dropout_on"])
    og = O.make_optimizer(case["opt_g"][0], mg.params, **case["opt_g"][1])
    od = O.make_optimizer(case["opt_d"][0], md.params, **case["opt_d"][1])
    x_np, y_np, lengths = C.make_batch(case)
    x, y = torch.from_numpy(x_np), torch.from_numpy(y_np)
    T = case["T"]
    has_dyn = bool(np.any(case["has_dynamic_features"]))
    R = torch.from_numpy(O.unit_variance_mlpg_matrix(C.WINDOWS[:case["windows"]], T)) if has_dyn else None
    mask = O.sequence_mask(lengths).unsqueeze(-1)
    out = {}
    for step in range(case["steps"]):
        dg = dd = None
        if case["dropout_on"]:
            gm, dm = C.make_dropout_masks(case, step)
            dg = O._DropoutSource([torch.from_numpy(m) for m in gm])
            dd = O._DropoutSource([torch.from_numpy(m) for m in dm])
        y_static = O.get_static_features(y, cfg.num_windows, cfg.stream_sizes, cfg.has_dynamic_features)
        og.zero_grad(), od.zero_grad()
        y_hat, y_hat_static = O.apply_generator(cfg, mg, x, R, list(lengths), drop=dg)
        if step == 0:
            out["y_hat"] = y_hat.detach().numpy().copy()
            out["y_hat_static"] = y_hat_static.detach().numpy().copy()
        if case["update_d"]:
            res = O.update_discriminator(cfg, md, od, x, y_static, y_hat_static, list(lengths), mask,
                                         "train", drop=dd)
            out["d_scalars_%d" % step] = np.array(res, dtype=np.float64)
            gn = [p.grad for p in mg.params if p.grad is not None]
            out["g_leak_norm_%d" % step] = np.array(
                float(torch.sqrt(sum((g ** 2).sum() for g in gn))) if gn else 0.0)
        if case["update_g"]:
            res = O.update_generator(cfg, mg, md, og, x, y, y_hat, y_static, y_hat_static,
                                     case["adv_w"], list(lengths), mask, "train",
                                     mse_w=case["mse_w"], mge_w=case["mge_w"], drop=dd)
            out["g_scalars_%d" % step] = np.array(res, dtype=np.float64)
    for k, v in mg.state_dict().items():
        out["G." + k] = v.numpy()
    for k, v in md.state_dict().items():
        out["D." + k] = v.numpy()
    for tag, opt, model in (("G", og, mg), ("D", od, md)):
        for i, name in enumerate(model.names):
            if isinstance(opt, O.OracleAdagrad):  # torch creates "sum" at construction
                out["%s.opt.sum.%s" % (tag, name)] = opt.sum[i].numpy()
            elif opt.step_count:
                out["%s.opt.exp_avg.%s" % (tag, name)] = opt.m[i].numpy()
                out["%s.opt.exp_avg_sq.%s" % (tag, name)] = opt.v[i].numpy()
    return out
