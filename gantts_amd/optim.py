"""``Adagrad`` / ``Adam`` with torch.optim's constructor signature, ``param_groups`` and
``state_dict`` layout (reference train.py:796-799 builds them with
``getattr(optim, hp.optimizer_g)(model.parameters(), **hp.optimizer_g_params)``; checkpoints
store ``optimizer.state_dict()``, train.py:162-171).

The update itself is the fused clip-norm + optimizer HIP kernel over the network's flat
parameter buffer (one launch per network, executed inside ``update_discriminator`` /
``update_generator`` exactly where the reference calls ``clip_grad_norm_`` + ``step()``,
train.py:275-276, 317-318).  State lives in flat float32 device buffers owned here.
"""
import torch

from . import _lib as L


def _owner_of(params):
    params = list(params)
    if not params:
        raise ValueError("optimizer got an empty parameter list")
    if isinstance(params[0], dict):
        if len(params) != 1:
            raise ValueError("gantts_amd optimizers support a single param group")
        params = list(params[0]["params"])
    owners = {id(getattr(p, "_gt_owner", lambda: None)()) for p in params}
    owner = getattr(params[0], "_gt_owner", lambda: None)()
    if owner is None or len(owners) != 1:
        raise TypeError("gantts_amd.optim optimizers take model.parameters() of ONE gantts_amd network")
    if len(params) != len(list(owner.parameters())):
        raise ValueError("pass all parameters of the network (the update runs over its flat buffer)")
    return owner, params


class _FlatOptimizer(object):
    KIND = None
    STATE_KEYS = ()

    def __init__(self, params, defaults):
        self.model, self._params = _owner_of(params)
        self.defaults = dict(defaults)
        group = dict(defaults)
        group["params"] = list(range(len(self._params)))
        self.param_groups = [group]
        self._state = [None, None]
        self._step = 0
        self._version = 0
        self._engines = {}
        self.max_grad_norm = 1.0      # clip_grad_norm_(params, 1.0), train.py:275,317

    # -- flat state -------------------------------------------------------------------------
    def _ensure_state(self):
        flat = self.model.flat_params()
        for i in range(len(self.STATE_KEYS)):
            st = self._state[i]
            if st is None:
                self._state[i] = torch.zeros_like(flat)
                self._version += 1
            elif st.device != flat.device:
                self._state[i] = st.to(flat.device)
                self._version += 1

    def _desc(self):
        self._ensure_state()
        g = self.param_groups[0]
        d = L.OptimDesc()
        d.kind = self.KIND
        d.lr = float(g["lr"])
        d.weight_decay = float(g.get("weight_decay", 0.0))
        d.eps = float(g["eps"])
        d.lr_decay = float(g.get("lr_decay", 0.0))
        b1, b2 = g.get("betas", (0.0, 0.0))
        d.beta1, d.beta2 = float(b1), float(b2)
        d.max_grad_norm = float(self.max_grad_norm)
        d.step = int(self._step)
        d.state0 = self._state[0].data_ptr()
        d.state1 = self._state[1].data_ptr() if len(self.STATE_KEYS) > 1 else None
        return d

    def _hyper(self):
        """(lr, everything else the fused update kernel reads) -- compared by StepEngine.bind_optimizer on every step."""
        g = self.param_groups[0]
        b1, b2 = g.get("betas", (0.0, 0.0))
        return (float(g["lr"]), float(g.get("weight_decay", 0.0)), float(g["eps"]), float(g.get("lr_decay", 0.0)),
                float(b1), float(b2), float(self.max_grad_norm))

    def _note_step(self, engine, role):
        self._step = engine.optimizer_step_count(role)

    # -- torch.optim API --------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        """Marks the gradients as cleared (the next backward overwrites them) and drops the
        pending D-loss -> G gradient, like ``optimizer_g.zero_grad()`` at train.py:538."""
        for key, (ref, role) in list(self.model._bound_engines.items()):
            eng = ref()
            if eng is None:
                del self.model._bound_engines[key]
            else:
                eng.zero_grad(role)

    def step(self, closure=None):
        raise RuntimeError("gantts_amd optimizers step inside update_discriminator/update_generator "
                           "(fused clip-norm + update kernel); a separate step() would apply the update twice")

    def _views(self, flat):
        out, off = [], 0
        for p in self._params:
            n = p.numel()
            out.append(flat[off:off + n].view(p.shape))
            off += n
        return out

    def state_dict(self):
        state = {}
        if self.KIND == L.OPT_ADAGRAD:
            self._ensure_state()      # torch.optim.Adagrad creates "sum" at construction
        if self._state[0] is not None and (self._step > 0 or self.KIND == L.OPT_ADAGRAD):
            views = [self._views(s) for s in self._state[:len(self.STATE_KEYS)]]
            for i in range(len(self._params)):
                st = {"step": torch.tensor(float(self._step))}
                for k, key in enumerate(self.STATE_KEYS):
                    st[key] = views[k][i].clone()
                state[i] = st
        return {"state": state, "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        g = dict(groups[0])
        g["params"] = list(range(len(self._params)))
        self.param_groups = [g]
        state = sd.get("state", {})
        if state:
            self._ensure_state()
            views = [self._views(s) for s in self._state[:len(self.STATE_KEYS)]]
            steps = set()
            for i in range(len(self._params)):
                st = state[i]
                steps.add(int(float(st["step"])))
                for k, key in enumerate(self.STATE_KEYS):
                    views[k][i].copy_(st[key])
            if len(steps) != 1:
                raise ValueError("per-parameter step counts differ; not a checkpoint of this optimizer")
            self._step = steps.pop()
        self._version += 1


class Adagrad(_FlatOptimizer):
    """p -= lr/(1+(t-1)*lr_decay) * g / (sqrt(sum g^2) + eps)   (torch.optim.Adagrad semantics)."""
    KIND = L.OPT_ADAGRAD
    STATE_KEYS = ("sum",)

    def __init__(self, params, lr=1e-2, lr_decay=0, weight_decay=0, initial_accumulator_value=0, eps=1e-10):
        if lr < 0 or lr_decay < 0 or weight_decay < 0 or eps < 0:
            raise ValueError("invalid Adagrad hyper-parameter")
        super(Adagrad, self).__init__(params, dict(lr=lr, lr_decay=lr_decay, eps=eps, weight_decay=weight_decay,
                                                   initial_accumulator_value=initial_accumulator_value))
        self._init_acc = float(initial_accumulator_value)

    def _ensure_state(self):
        fresh = self._state[0] is None
        super(Adagrad, self)._ensure_state()
        if fresh and self._init_acc != 0.0:
            self._state[0].fill_(self._init_acc)


class Adam(_FlatOptimizer):
    """torch.optim.Adam semantics (bias-corrected, no amsgrad)."""
    KIND = L.OPT_ADAM
    STATE_KEYS = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise ValueError("amsgrad is not supported")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super(Adam, self).__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                                amsgrad=False))
