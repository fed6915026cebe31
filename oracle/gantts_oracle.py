"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the r9y9/gantts G+D training step.

This module is the *checker* for the HIP engine in ``gantts_amd/``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path never does (and fails loudly when the HIP library is missing).

What it restates (reference file:line, relative to the upstream tree):

* ``MLP``                       gantts/models.py:121-141
* ``In2OutHighwayNet``          gantts/models.py:21-69
* ``LSTMRNN`` / ``GRURNN``      gantts/models.py:170-213 (nn.LSTM over packed sequences, restated
                                as an explicit masked time loop; pinned by the golden case
                                ``acoustic_lstm`` generated from the real reference)
* ``sequence_mask`` / masked MSE gantts/seqloss.py:9-43
* stream index arithmetic       gantts/multistream.py:33-123
* ``apply_generator``           train.py:336-355
* ``update_discriminator``      train.py:245-279
* ``update_generator``          train.py:282-320
* ``clip_grad_norm_`` + ``torch.optim.Adagrad/Adam`` as used at train.py:275-276,
  317-318, 796-799 (restated from the published update rules, checked against
  ``torch.optim`` in tests/test_oracle.py)
* ``SRURNN``                    gantts/models.py:144-167 over the un-vendored third-party SRU cell:
                                **parity unpinned** (see OracleSRURNN)
* ``unit_variance_mlpg_matrix`` / ``unit_variance_mlpg`` -- third-party nnmnkwii
  (>= 0.0.14, reference setup.py:58-68, NOT vendored under /root/reference).
  Restated from its published definition: W = vstack of per-window band matrices with
  zero edges, R = (W^T W)^-1 W^T in float64, cast to float32; apply = window-major
  reshape then R @ means.  The reference tests pin only shapes and stream slicing at
  this boundary (tests/test_gantts.py:132-163) => numeric values of R are
  **parity unpinned**; self-consistency (R W = I) is checked in tests/test_oracle.py.

Pinning: ``tests/golden/make_golden.py`` (run in the build container, where the
reference tree can be imported through ``oracle/ref_loader.py``) executes the REAL
reference ``train.update_*`` on seeded inputs and stores the results under
``tests/golden/``; ``tests/test_oracle.py`` checks this restatement against those
fixtures, and (container only) live against the reference.

The numeric backend is torch-CPU float32 because that *is* the reference's backend
(ATen/MKL); gradients come from autograd exactly as in the reference, including the
un-detached D-loss -> G gradient leak (train.py:265,274,316-318).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.01  # nn.LeakyReLU() default, gantts/models.py:132


# ----------------------------------------------------------------------------
# MLPG (nnmnkwii restatement)
# ----------------------------------------------------------------------------
def _window_matrix(left, right, coef, T):
    """(W c)[t] = sum_k coef[k + left] * c[t + k], zero outside [0, T)."""
    W = np.zeros((T, T), dtype=np.float64)
    for k in range(-left, right + 1):
        c = float(coef[k + left])
        if c == 0.0:
            continue
        idx = np.arange(max(0, -k), min(T, T - k))
        W[idx, idx + k] = c
    return W


def unit_variance_mlpg_matrix(windows, T):
    """R = (W^T W)^-1 W^T, shape (T, num_windows*T), float32 (window-major columns)."""
    T = int(T)
    W = np.vstack([_window_matrix(l, u, c, T) for (l, u, c) in windows])
    R = np.linalg.solve(W.T @ W, W.T)
    return R.astype(np.float32)


def unit_variance_mlpg(R, means):
    """means (B,T,nW*d) with per-frame layout [static(d)|delta(d)|delta2(d)] -> (B,T,d)."""
    squeeze = means.dim() == 2
    if squeeze:
        means = means.unsqueeze(0)
    B, T, D = means.shape
    nW = R.shape[1] // R.shape[0]
    d = D // nW
    m = means.contiguous().view(B, T, nW, d).transpose(1, 2).contiguous().view(B, nW * T, d)
    out = torch.matmul(R, m)
    return out.squeeze(0) if squeeze else out


# ----------------------------------------------------------------------------
# stream index arithmetic (gantts/multistream.py)
# ----------------------------------------------------------------------------
def get_static_stream_sizes(stream_sizes, has_dynamic_features, num_windows):
    # multistream.py:46-53 -- integer array; dynamic streams divided by num_windows
    out = []
    for s, dyn in zip(stream_sizes, has_dynamic_features):
        out.append(int(s // num_windows) if dyn else int(s))
    return np.array(out, dtype=np.int64)


def select_streams(inputs, stream_sizes, streams):
    # multistream.py:33-43
    cols, start = [], 0
    for size, enabled in zip(stream_sizes, streams):
        if enabled:
            cols.append(inputs[:, :, start:start + int(size)])
        start += int(size)
    return torch.cat(cols, dim=-1)


def get_static_features(inputs, num_windows, stream_sizes, has_dynamic_features, streams=None):
    # multistream.py:56-79
    D = inputs.shape[-1]
    if stream_sizes is None or (len(stream_sizes) == 1 and has_dynamic_features[0]):
        return inputs[:, :, :D // num_windows]
    if len(stream_sizes) == 1 and not has_dynamic_features[0]:
        return inputs
    if streams is None:
        streams = [True] * len(stream_sizes)
    cols, start = [], 0
    for size, dyn, enabled in zip(stream_sizes, has_dynamic_features, streams):
        if enabled:
            w = size // num_windows if dyn else size
            cols.append(inputs[:, :, start:start + w])
        start += size
    return torch.cat(cols, dim=-1)


def multi_stream_mlpg(inputs, R, stream_sizes, has_dynamic_features, streams=None):
    # multistream.py:82-123
    if inputs.shape[-1] != sum(stream_sizes):
        raise RuntimeError("You probably have specified wrong dimention params.")
    if streams is None:
        streams = [True] * len(stream_sizes)
    cols, start = [], 0
    for size, dyn, enabled in zip(stream_sizes, has_dynamic_features, streams):
        if enabled:
            x = inputs[:, :, start:start + size]
            cols.append(unit_variance_mlpg(R, x) if dyn else x)
        start += size
    return torch.cat(cols, dim=-1)


def adversarial_columns(stream_sizes, has_dynamic_features, num_windows,
                        adversarial_streams, mask_nth_mgc):
    """Column indices (into the static layout) fed to D: train.py:232-242."""
    ss = get_static_stream_sizes(stream_sizes, has_dynamic_features, num_windows)
    if adversarial_streams is None:
        return list(range(int(ss.sum())))
    cols, start = [], 0
    for size, enabled in zip(ss, adversarial_streams):
        if enabled:
            cols.extend(range(start, start + int(size)))
        start += int(size)
    return cols[int(mask_nth_mgc):] if mask_nth_mgc > 0 else cols


# ----------------------------------------------------------------------------
# loss pieces (gantts/seqloss.py)
# ----------------------------------------------------------------------------
def sequence_mask(lengths, max_len=None):
    lengths = torch.as_tensor(lengths).long()
    if max_len is None:
        max_len = int(lengths.max())
    ar = torch.arange(0, max_len).long().unsqueeze(0)
    return (ar < lengths.unsqueeze(1)).float()


def masked_mse(inp, target, mask):
    """sum((inp*m - target*m)^2) / sum(m)  -- divides by valid FRAMES (seqloss.py:42-43)."""
    m = mask.expand_as(inp)
    return F.mse_loss(inp * m, target * m, reduction="sum") / mask.sum()


# ----------------------------------------------------------------------------
# models
# ----------------------------------------------------------------------------
def linear_init(out_dim, in_dim, gen):
    """nn.Linear default init: U(+-1/sqrt(in)) for weight and bias."""
    k = 1.0 / math.sqrt(in_dim)
    W = (torch.rand(out_dim, in_dim, generator=gen) * 2 - 1) * k
    b = (torch.rand(out_dim, generator=gen) * 2 - 1) * k
    return W, b


class _DropoutSource(object):
    """Hands out dropout masks: injected (list), or fresh Bernoulli(1-p) draws."""

    def __init__(self, masks=None, generator=None):
        self.masks = list(masks) if masks is not None else None
        self.generator = generator
        self.drawn = []

    def next(self, like, p):
        if self.masks is not None:
            m = self.masks.pop(0)
            assert m.shape == like.shape, (m.shape, like.shape)
        else:
            m = torch.bernoulli(torch.full_like(like, 1.0 - p), generator=self.generator)
        self.drawn.append(m)
        return m


def cast_model(model, dtype):
    """Re-creates the model's parameters in `dtype` (float64 = the arbiter run of the at-size fixtures and of the
    tolerance tests: same inputs and masks, arithmetic exact to 1e-16).  Call before building its optimizer."""
    model.params = [p.detach().to(dtype).clone().requires_grad_(True) for p in model.params]
    return model


class OracleMLP(object):
    """gantts/models.py:121-141.  state-dict keys: layers.{i}.weight/bias, last_linear.*"""

    def __init__(self, in_dim=118, out_dim=1, num_hidden=2, hidden_dim=256,
                 dropout=0.5, last_sigmoid=True, bidirectional=None, seed=0):
        gen = torch.Generator().manual_seed(seed)
        ins = [in_dim] + [hidden_dim] * (num_hidden - 1)
        self.names, self.params = [], []
        for i, n_in in enumerate(ins):
            W, b = linear_init(hidden_dim, n_in, gen)
            self.names += ["layers.%d.weight" % i, "layers.%d.bias" % i]
            self.params += [W, b]
        W, b = linear_init(out_dim, hidden_dim, gen)
        self.names += ["last_linear.weight", "last_linear.bias"]
        self.params += [W, b]
        for p in self.params:
            p.requires_grad_(True)
        self.num_hidden, self.p, self.last_sigmoid = num_hidden, float(dropout), last_sigmoid
        self.training = True

    def include_parameter_generation(self):
        return False

    def state_dict(self):
        return {n: p.detach().clone() for n, p in zip(self.names, self.params)}

    def load_state_dict(self, sd):
        with torch.no_grad():
            for n, p in zip(self.names, self.params):
                p.copy_(torch.as_tensor(sd[n]))

    def forward(self, x, lengths=None, drop=None):
        drop = drop or _DropoutSource()
        for i in range(self.num_hidden):
            W, b = self.params[2 * i], self.params[2 * i + 1]
            x = F.leaky_relu(F.linear(x, W, b), LEAKY_SLOPE)
            if self.training and self.p > 0:
                x = x * drop.next(x, self.p) / (1.0 - self.p)
        x = F.linear(x, self.params[-2], self.params[-1])
        return torch.sigmoid(x) if self.last_sigmoid else x

    __call__ = forward


class OracleIn2OutHighwayNet(object):
    """gantts/models.py:21-69.  keys: T.*, H.{i}.*, last_linear.*"""

    def __init__(self, in_dim=118, out_dim=118, static_dim=59, num_hidden=3,
                 hidden_dim=512, dropout=0.5, seed=0):
        gen = torch.Generator().manual_seed(seed)
        self.static_dim = static_dim
        self.names, self.params = ["T.weight", "T.bias"], list(linear_init(static_dim, static_dim, gen))
        ins = [in_dim] + [hidden_dim] * (num_hidden - 1)
        for i, n_in in enumerate(ins):
            W, b = linear_init(hidden_dim, n_in, gen)
            self.names += ["H.%d.weight" % i, "H.%d.bias" % i]
            self.params += [W, b]
        W, b = linear_init(out_dim, hidden_dim, gen)
        self.names += ["last_linear.weight", "last_linear.bias"]
        self.params += [W, b]
        for p in self.params:
            p.requires_grad_(True)
        self.num_hidden, self.p = num_hidden, float(dropout)
        self.training = True

    def include_parameter_generation(self):
        return True

    state_dict = OracleMLP.state_dict
    load_state_dict = OracleMLP.load_state_dict

    def forward(self, x, R, lengths=None, drop=None):
        drop = drop or _DropoutSource()
        x_static = x[:, :, :self.static_dim]
        Tx = torch.sigmoid(F.linear(x_static, self.params[0], self.params[1]))
        h = x
        for i in range(self.num_hidden):
            W, b = self.params[2 + 2 * i], self.params[3 + 2 * i]
            h = F.leaky_relu(F.linear(h, W, b), LEAKY_SLOPE)
            if self.training and self.p > 0:
                h = h * drop.next(h, self.p) / (1.0 - self.p)
        h = F.linear(h, self.params[-2], self.params[-1])
        Gx = unit_variance_mlpg(R, h)
        return h, x_static + Tx * Gx

    __call__ = forward


class OracleLSTMRNN(object):
    """gantts/models.py:193-213 (LSTMRNN; GRURNN :170-190 is the same network under the attribute
    name ``gru``).  pack_padded_sequence -> nn.LSTM(batch_first, bidirectional, dropout) ->
    pad_packed_sequence -> hidden2out -> optional sigmoid, restated with an explicit time loop:
    a sequence is active at frame t iff t < length; the reverse direction therefore starts at each
    sequence's own last valid frame with zero state; outputs beyond the length are zero (so
    hidden2out yields its bias there).  Gate order i, f, g, o; keys lstm.weight_ih_l{k}[_reverse] ..."""

    def __init__(self, in_dim=118, out_dim=118, num_hidden=2, hidden_dim=256, bidirectional=False,
                 dropout=0, last_sigmoid=False, seed=0, prefix="lstm"):
        gen = torch.Generator().manual_seed(seed)
        self.H, self.L, self.dirs = hidden_dim, num_hidden, 2 if bidirectional else 1
        self.p, self.last_sigmoid = float(dropout), last_sigmoid
        k = 1.0 / math.sqrt(hidden_dim)
        self.names, self.params = [], []
        for l in range(num_hidden):
            n_in = in_dim if l == 0 else hidden_dim * self.dirs
            for d in range(self.dirs):
                sfx = "_l%d%s" % (l, "_reverse" if d else "")
                for nm, shape in (("weight_ih", (4 * hidden_dim, n_in)), ("weight_hh", (4 * hidden_dim, hidden_dim)),
                                  ("bias_ih", (4 * hidden_dim,)), ("bias_hh", (4 * hidden_dim,))):
                    self.names.append("%s.%s%s" % (prefix, nm, sfx))
                    self.params.append((torch.rand(*shape, generator=gen) * 2 - 1) * k)
        W, b = linear_init(out_dim, hidden_dim * self.dirs, gen)
        self.names += ["hidden2out.weight", "hidden2out.bias"]
        self.params += [W, b]
        for p in self.params:
            p.requires_grad_(True)
        self.training = True

    def include_parameter_generation(self):
        return False

    state_dict = OracleMLP.state_dict
    load_state_dict = OracleMLP.load_state_dict

    def _recurrent(self, x, lengths, drop, idx=0):
        """nn.LSTM over the packed batch; returns the padded (B,T,H*dirs) outputs."""
        B, T, _ = x.shape
        lens = torch.as_tensor([int(v) for v in lengths]) if lengths is not None else torch.full((B,), T)
        H = self.H
        inp = x
        for l in range(self.L):
            outs = []
            for d in range(self.dirs):
                Wih, Whh, bih, bhh = self.params[idx:idx + 4]
                idx += 4
                xp = F.linear(inp, Wih, bih + bhh)
                h, c = x.new_zeros(B, H), x.new_zeros(B, H)
                seq = [None] * T
                for t in (range(T - 1, -1, -1) if d else range(T)):
                    act = (t < lens).to(x.dtype).unsqueeze(1)
                    g = xp[:, t] + F.linear(h, Whh)
                    i, f, gg, o = g[:, :H].sigmoid(), g[:, H:2 * H].sigmoid(), g[:, 2 * H:3 * H].tanh(), g[:, 3 * H:].sigmoid()
                    c = act * (f * c + i * gg)
                    h = act * (o * c.tanh())
                    seq[t] = h
                outs.append(torch.stack(seq, 1))
            inp = torch.cat(outs, -1)
            if self.training and self.p > 0 and l + 1 < self.L:
                drop = drop or _DropoutSource()
                inp = inp * drop.next(inp, self.p) / (1.0 - self.p)
        return inp

    def forward(self, x, lengths=None, drop=None):
        out = F.linear(self._recurrent(x, lengths, drop), self.params[-2], self.params[-1])
        return torch.sigmoid(out) if self.last_sigmoid else out

    __call__ = forward


class OracleIn2OutRNNHighwayNet(OracleLSTMRNN):
    """gantts/models.py:72-118.  keys: T.*, lstm.*, hidden2out.*.  Returns (x, x_static + T(x) *
    MLPG(hidden2out(LSTM(x)))) -- the first output is the input itself (:118)."""

    def __init__(self, in_dim=118, out_dim=118, static_dim=59, num_hidden=3, hidden_dim=512,
                 bidirectional=False, dropout=0.5, seed=0):
        OracleLSTMRNN.__init__(self, in_dim, out_dim, num_hidden, hidden_dim, bidirectional, dropout, False, seed)
        gen = torch.Generator().manual_seed(seed + 7919)
        self.static_dim = static_dim
        W, b = linear_init(static_dim, static_dim, gen)
        self.names = ["T.weight", "T.bias"] + self.names
        self.params = [W.requires_grad_(True), b.requires_grad_(True)] + self.params

    def include_parameter_generation(self):
        return True

    def forward(self, x, R, lengths=None, drop=None):
        x_static = x[:, :, :self.static_dim]
        Tx = torch.sigmoid(F.linear(x_static, self.params[0], self.params[1]))
        out = F.linear(self._recurrent(x, lengths, drop, idx=2), self.params[-2], self.params[-1])
        return x, x_static + Tx * unit_variance_mlpg(R, out)

    __call__ = forward


class OracleSRURNN(object):
    """gantts/models.py:144-167 (SRURNN) on top of the THIRD-PARTY, un-vendored SRU cell
    (`cuda_functional.SRU` of github.com/taolei87/sru, 2017 layout; no version pinned anywhere in
    the reference, CUDA-only, no reference test touches it) => **parity unpinned**: restated from
    the published recurrence (Lei et al. 2017, arXiv:1709.02755, and SURVEY Appendix B) and checked
    for self-consistency only (finite-difference gradients, tests/test_oracle.py).

    Per layer (n_in -> H per direction, ncols = H*dirs, k = 3 if n_in == ncols else 4):
        U = x @ weight                       weight (n_in, ncols*k), column j owns U[..., j*k : (j+1)*k]
        f = sigmoid(u1 + b_f[j]); r = sigmoid(u2 + b_r[j])          bias = [b_f (ncols) | b_r (ncols)]
        c_t = (c_{t-1} - u0) * f + u0
        h_t = (g(c_t) * mask_h - x') * r + x'      x' = x_t[j] if k == 3 else u3; g = relu / tanh / identity
    columns j >= H of a bidirectional layer run backwards in time.  `lengths` are ignored (padded
    frames run through the recurrence).  Training only: `rnn_dropout` = one Bernoulli mask (B, n_in)
    shared over time on x before the GEMM; `dropout` = mask (B, ncols) on g(c) (not on the last layer)."""

    def __init__(self, in_dim=118, out_dim=118, num_hidden=2, hidden_dim=256, bidirectional=False,
                 dropout=0, last_sigmoid=False, use_relu=0, rnn_dropout=0.0, seed=0):
        gen = torch.Generator().manual_seed(seed)
        self.H, self.L, self.dirs = hidden_dim, num_hidden, 2 if bidirectional else 1
        self.p, self.rnn_p, self.act = float(dropout), float(rnn_dropout), ("relu" if use_relu else "tanh")
        self.last_sigmoid = last_sigmoid
        ncols = hidden_dim * self.dirs
        self.names, self.params, self.ks = [], [], []
        for l in range(num_hidden):
            n_in = in_dim if l == 0 else ncols
            k = 3 if n_in == ncols else 4
            self.ks.append(k)
            r = math.sqrt(3.0 / n_in)
            self.names += ["gru.rnn_lst.%d.weight" % l, "gru.rnn_lst.%d.bias" % l]
            self.params += [(torch.rand(n_in, ncols * k, generator=gen) * 2 - 1) * r, torch.zeros(2 * ncols)]
        W, b = linear_init(out_dim, ncols, gen)
        self.names += ["hidden2out.weight", "hidden2out.bias"]
        self.params += [W, b]
        for p in self.params:
            p.requires_grad_(True)
        self.training = True

    def include_parameter_generation(self):
        return False

    state_dict = OracleMLP.state_dict
    load_state_dict = OracleMLP.load_state_dict

    def _g(self, c):
        return torch.relu(c) if self.act == "relu" else torch.tanh(c)

    def forward(self, x, lengths=None, drop=None):
        B, T, _ = x.shape
        H, ncols = self.H, self.H * self.dirs
        inp = x
        for l in range(self.L):
            W, bias = self.params[2 * l], self.params[2 * l + 1]
            k = self.ks[l]
            xin = inp
            if self.training and self.rnn_p > 0:
                drop = drop or _DropoutSource()
                xin = inp * (drop.next(inp[:, 0], self.rnn_p) / (1.0 - self.rnn_p)).unsqueeze(1)
            U = (xin @ W).view(B, T, ncols, k)
            # per-frame views taken ONCE (unbind): slicing U[:, t] inside the time loop makes autograd allocate a zero tensor
            # of U's full size for every frame on the way back (O(T^2) traffic: hours at T = 2048); same values either way
            Ut = U.unbind(1)
            inp_t = inp.unbind(1) if k == 3 else None
            mask_h = None
            if self.training and self.p > 0 and l + 1 < self.L:
                drop = drop or _DropoutSource()
                mask_h = drop.next(inp.new_zeros(B, ncols), self.p) / (1.0 - self.p)
            bf, br = bias[:ncols], bias[ncols:]
            outs = []
            for d in range(self.dirs):
                sl = slice(d * H, (d + 1) * H)
                c = x.new_zeros(B, H)
                seq = [None] * T
                for t in (range(T - 1, -1, -1) if d else range(T)):
                    u = Ut[t][:, sl]
                    f = torch.sigmoid(u[..., 1] + bf[sl])
                    r = torch.sigmoid(u[..., 2] + br[sl])
                    c = (c - u[..., 0]) * f + u[..., 0]
                    val = self._g(c)
                    if mask_h is not None:
                        val = val * mask_h[:, sl]
                    xp = inp_t[t][:, sl] if k == 3 else u[..., 3]
                    seq[t] = (val - xp) * r + xp
                outs.append(torch.stack(seq, 1))
            inp = torch.cat(outs, -1)
        out = F.linear(inp, self.params[-2], self.params[-1])
        return torch.sigmoid(out) if self.last_sigmoid else out

    __call__ = forward


# ----------------------------------------------------------------------------
# optimizers (torch.optim.Adagrad / Adam update rules, single param group)
# ----------------------------------------------------------------------------
def clip_grad_norm_(params, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (L2): coef = clamp(max_norm/(||g||+1e-6), max=1)."""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


class OracleAdagrad(object):
    def __init__(self, params, lr=0.01, lr_decay=0.0, weight_decay=0.0, eps=1e-10, initial_accumulator_value=0.0):
        self.params = list(params)
        self.lr, self.lr_decay, self.wd, self.eps = lr, lr_decay, weight_decay, eps
        self.sum = [torch.full_like(p, float(initial_accumulator_value)) for p in self.params]     # torch.optim.Adagrad
        self.step_count = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        self.step_count += 1
        clr = self.lr / (1.0 + (self.step_count - 1) * self.lr_decay)
        with torch.no_grad():
            for p, s in zip(self.params, self.sum):
                if p.grad is None:
                    continue
                g = p.grad
                if self.wd != 0:
                    g = g + self.wd * p
                s.addcmul_(g, g, value=1.0)
                p.addcdiv_(g, s.sqrt() + self.eps, value=-clr)


class OracleAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.step_count = 0

    zero_grad = OracleAdagrad.zero_grad

    def step(self):
        self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        with torch.no_grad():
            for p, m, v in zip(self.params, self.m, self.v):
                if p.grad is None:
                    continue
                g = p.grad
                if self.wd != 0:
                    g = g + self.wd * p
                m.mul_(b1).add_(g, alpha=1.0 - b1)
                v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                denom = (v.sqrt() / math.sqrt(bc2)) + self.eps
                p.addcdiv_(m, denom, value=-self.lr / bc1)


def make_optimizer(kind, params, **kw):
    return {"Adagrad": OracleAdagrad, "Adam": OracleAdam}[kind](params, **kw)


# ----------------------------------------------------------------------------
# step functions (train.py)
# ----------------------------------------------------------------------------
class StreamConfig(object):
    """The hp.* fields the step functions read (train.py:61,233-241,248,254,299,304,352)."""

    def __init__(self, stream_sizes, has_dynamic_features, num_windows,
                 adversarial_streams=None, mask_nth_mgc_for_adv_loss=0,
                 discriminator_linguistic_condition=False):
        self.stream_sizes = list(stream_sizes)
        self.has_dynamic_features = list(has_dynamic_features)
        self.num_windows = int(num_windows)
        self.adversarial_streams = None if adversarial_streams is None else list(adversarial_streams)
        self.mask_nth_mgc_for_adv_loss = int(mask_nth_mgc_for_adv_loss)
        self.discriminator_linguistic_condition = bool(discriminator_linguistic_condition)

    def adv_cols(self):
        return adversarial_columns(self.stream_sizes, self.has_dynamic_features, self.num_windows,
                                   self.adversarial_streams, self.mask_nth_mgc_for_adv_loss)


def apply_generator(cfg, model_g, x, R, lengths, drop=None):
    """train.py:336-355"""
    if model_g.include_parameter_generation():
        return model_g(x, R, lengths=lengths, drop=drop)
    y_hat = model_g(x, lengths=lengths, drop=drop)
    y_hat_static = multi_stream_mlpg(y_hat, R, cfg.stream_sizes, cfg.has_dynamic_features) \
        if R is not None else y_hat
    return y_hat, y_hat_static


def _adv_input(cfg, x, feats):
    sel = feats[:, :, cfg.adv_cols()] if cfg.adversarial_streams is not None else feats
    return torch.cat((x, sel), -1) if cfg.discriminator_linguistic_condition else sel


def update_discriminator(cfg, model_d, optimizer_d, x, y_static, y_hat_static, lengths,
                         mask, phase, eps=1e-20, drop=None, Tv=None, step=True):
    """train.py:245-279.  y_hat_static is NOT detached: loss_d.backward also fills G grads.
    Tv / step: hooks for the data-parallel test backend (global normaliser; stop before clip+step)."""
    real_in = _adv_input(cfg, x, y_static)
    fake_in = _adv_input(cfg, x, y_hat_static)
    Tv = mask.sum().item() if Tv is None else Tv
    D_real = model_d(real_in, lengths=lengths, drop=drop)
    real_correct = ((D_real > 0.5).float() * mask).sum().item()
    D_fake = model_d(fake_in, lengths=lengths, drop=drop)
    fake_correct = ((D_fake < 0.5).float() * mask).sum().item()
    loss_real = -(torch.log(D_real + eps) * mask).sum() / Tv
    loss_fake = -(torch.log(1 - D_fake + eps) * mask).sum() / Tv
    loss_d = loss_real + loss_fake
    if phase == "train":
        loss_d.backward(retain_graph=True)
        if step:
            clip_grad_norm_(model_d.params, 1.0)
            optimizer_d.step()
    return loss_d.item(), loss_fake.item(), loss_real.item(), real_correct, fake_correct


def update_generator(cfg, model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static,
                     adv_w, lengths, mask, phase, mse_w=None, mge_w=None, eps=1e-20, drop=None,
                     Tv=None, step=True):
    """train.py:282-320.  D forward here uses the ALREADY UPDATED D weights."""
    local_tv = mask.sum()
    Tv = local_tv.item() if Tv is None else Tv
    loss_mge = masked_mse(y_hat_static, y_static, mask) * (local_tv / Tv)
    loss_mse = masked_mse(y_hat, y, mask) * (local_tv / Tv)
    if adv_w > 0:
        fake_in = _adv_input(cfg, x, y_hat_static)
        loss_adv = -(torch.log(model_d(fake_in, lengths=lengths, drop=drop) + eps) * mask).sum() / Tv
    else:
        loss_adv = torch.zeros(1)
    loss_g = (mse_w * loss_mse + mge_w * loss_mge) + adv_w * loss_adv
    if phase == "train":
        loss_g.backward()
        if step:
            clip_grad_norm_(model_g.params, 1.0)
            optimizer_g.step()
    return loss_mse.item(), loss_mge.item(), loss_adv.item(), loss_g.item()


def train_step(cfg, model_g, model_d, opt_g, opt_d, x, y, R, lengths, mask,
               adv_w=1.0, mse_w=0.0, mge_w=1.0, update_d=True, update_g=True,
               phase="train", drop_g=None, drop_d=None):
    """One batch of train.py:train_loop (:528-585) without data loading / metrics."""
    y_static = get_static_features(y, cfg.num_windows, cfg.stream_sizes, cfg.has_dynamic_features)
    opt_g.zero_grad()
    opt_d.zero_grad()
    y_hat, y_hat_static = apply_generator(cfg, model_g, x, R, lengths, drop=drop_g)
    out = {"y_hat": y_hat.detach(), "y_hat_static": y_hat_static.detach()}
    if update_d:
        out["d"] = update_discriminator(cfg, model_d, opt_d, x, y_static, y_hat_static,
                                        lengths, mask, phase, drop=drop_d)
    if update_g:
        out["g"] = update_generator(cfg, model_g, model_d, opt_g, x, y, y_hat, y_static,
                                    y_hat_static, adv_w, lengths, mask, phase,
                                    mse_w=mse_w, mge_w=mge_w, drop=drop_d)
    return out


# ---------------------------------------------------------------------------------------------
# Distortion metrics of the training loop (reference train.py:358-432).
#
# The four metric functions live in the THIRD-PARTY, un-vendored nnmnkwii (>= 0.0.14 per reference
# setup.py:58-68; `from nnmnkwii import metrics`, train.py:51): restated here from that package's
# published definitions -> **parity unpinned** for their numeric conventions (no reference test or
# fixture touches them).  The stream split / inverse scaling / vuv binarisation around them IS
# reference code (train.py:358-396) and is pinned through tests/golden/distortions.npz, generated by
# the real train.compute_distortions with these metric restatements plugged in as the nnmnkwii stub.
# ---------------------------------------------------------------------------------------------
_LOGDB_CONST = 10.0 / math.log(10.0) * math.sqrt(2.0)


def _len_list(lengths, B, T):
    if lengths is None:
        return [T] * B
    if isinstance(lengths, torch.Tensor):
        return [int(v) for v in lengths.view(-1).tolist()]
    return [int(v) for v in lengths]


def melcd(X, Y, lengths=None):
    """nnmnkwii.metrics.melcd: (10/ln10)*sqrt(2) * mean over valid frames of ||x_t - y_t||_2."""
    if X.dim() == 2:
        X, Y = X.unsqueeze(-1), Y.unsqueeze(-1)
    s, Tn = 0.0, 0
    for x, y, n in zip(X, Y, _len_list(lengths, X.size(0), X.size(1))):
        z = x[:n] - y[:n]
        s += float(torch.sqrt((z * z).sum(-1)).sum())
        Tn += n
    return _LOGDB_CONST * s / float(Tn)


def mean_squared_error(X, Y, lengths=None):
    """nnmnkwii.metrics.mean_squared_error: sum of squared errors over valid frames / (frames * D).
    (The reference takes sqrt() of it for "dur_rmse", train.py:420.)"""
    s, Tn = 0.0, 0
    for x, y, n in zip(X, Y, _len_list(lengths, X.size(0), X.size(1))):
        z = x[:n] - y[:n]
        s += float((z * z).sum())
        Tn += n
    return s / float(Tn * X.shape[-1])


def lf0_mean_squared_error(src_f0, src_vuv, tgt_f0, tgt_vuv, lengths=None, linear_domain=False):
    """nnmnkwii.metrics.lf0_mean_squared_error: MSE over frames voiced in BOTH; ZeroDivisionError if none."""
    s, Tn = 0.0, 0
    for x, xv, y, yv, n in zip(src_f0, src_vuv, tgt_f0, tgt_vuv, _len_list(lengths, src_f0.size(0), src_f0.size(1))):
        voiced = (xv[:n] + yv[:n]) >= 2
        Tn += int(voiced.sum())
        x, y = x[:n][voiced], y[:n][voiced]
        if linear_domain:
            x, y = torch.exp(x), torch.exp(y)
        z = x - y
        s += float((z * z).sum())
    return float(s) / float(Tn)      # python float division: ZeroDivisionError when nothing is voiced


def vuv_error(src_vuv, tgt_vuv, lengths=None):
    """nnmnkwii.metrics.vuv_error: fraction of valid frames whose binary decisions differ."""
    s, Tn = 0, 0
    for x, y, n in zip(src_vuv, tgt_vuv, _len_list(lengths, src_vuv.size(0), src_vuv.size(1))):
        s += int((x[:n] != y[:n]).sum())
        Tn += n
    return float(s) / float(Tn)


def _inv_scale(x, m, s):
    return x * s + m      # nnmnkwii.preprocessing.inv_scale


def split_streams(cfg, y_static, Y_mean, Y_std):
    """train.py:358-396 (inv_scale + split_streams).  Statistics are indexed in the static+dynamic
    domain (lf0 at mgc_dim, vuv after lf0, bap after vuv), the features in the static domain;
    vuv is binarised (> 0.5 -> 1, else 0) to int64."""
    mgc_dim, lf0_dim, vuv_dim, bap_dim = cfg.stream_sizes
    nw = cfg.num_windows
    lf0_0 = mgc_dim
    vuv_0 = lf0_0 + lf0_dim
    bap_0 = vuv_0 + vuv_dim
    smgc, slf0, svuv, sbap = get_static_stream_sizes(cfg.stream_sizes, cfg.has_dynamic_features, nw)
    mgc = y_static[:, :, :smgc]
    lf0 = y_static[:, :, smgc:smgc + slf0]
    vuv = y_static[:, :, smgc + slf0]
    bap = y_static[:, :, smgc + slf0 + svuv:]
    mgc = _inv_scale(mgc, Y_mean[:mgc_dim // nw], Y_std[:mgc_dim // nw])
    lf0 = _inv_scale(lf0, Y_mean[lf0_0:lf0_0 + lf0_dim // nw], Y_std[lf0_0:lf0_0 + lf0_dim // nw])
    bap = _inv_scale(bap, Y_mean[bap_0:bap_0 + bap_dim // nw], Y_std[bap_0:bap_0 + bap_dim // nw])
    vuv = _inv_scale(vuv, Y_mean[vuv_0], Y_std[vuv_0])
    vuv = (vuv > 0.5).long()
    return mgc, lf0, vuv, bap


def compute_distortions(cfg, name, y_static, y_hat_static, Y_mean, Y_std, lengths=None):
    """train.py:399-432.  name = hp.name ("acoustic" | "duration" | "vc")."""
    if name == "acoustic":
        mgc, lf0, vuv, bap = split_streams(cfg, y_static, Y_mean, Y_std)
        mgc_h, lf0_h, vuv_h, bap_h = split_streams(cfg, y_hat_static, Y_mean, Y_std)
        try:
            f0_mse = lf0_mean_squared_error(lf0, vuv, lf0_h, vuv_h, lengths=lengths, linear_domain=True)
        except ZeroDivisionError:
            f0_mse = float("nan")
        return {"mcd": melcd(mgc[:, :, 1:], mgc_h[:, :, 1:], lengths=lengths),
                "bap_mcd": melcd(bap, bap_h, lengths=lengths) / 10.0,
                "f0_rmse": float(np.sqrt(f0_mse)),
                "vuv_err": vuv_error(vuv, vuv_h, lengths=lengths)}
    if name == "duration":
        a, b = _inv_scale(y_static, Y_mean, Y_std), _inv_scale(y_hat_static, Y_mean, Y_std)
        return {"dur_rmse": math.sqrt(mean_squared_error(a, b, lengths=lengths))}
    if name == "vc":
        sd = y_static.size(-1)      # == hp.order (train.py:423)
        a, b = _inv_scale(y_static, Y_mean[:sd], Y_std[:sd]), _inv_scale(y_hat_static, Y_mean[:sd], Y_std[:sd])
        return {"mcd": melcd(a, b, lengths=lengths)}
    raise AssertionError(name)


# ---------------------------------------------------------------------------------------------
# Inference path (reference evaluation_tts.py:47-100, 133-221; evaluation_vc.py:40-92)
# ---------------------------------------------------------------------------------------------
def mlpg(mean_frames, variance_frames, windows):
    """nnmnkwii.paramgen.mlpg (un-vendored; restated from its published definition, parity unpinned):
    maximum-likelihood static trajectory c = (W' P W)^-1 W' P mu per static dimension, P = diagonal
    precisions.  mean_frames (T, D) in [static | delta | delta-delta] layout, variance_frames (T, D) or
    (D,), result (T, D // len(windows)) in float64.  Dense float64 solve (test sizes only)."""
    mu = np.asarray(mean_frames, dtype=np.float64)
    T, D = mu.shape
    nw = len(windows)
    sd = D // nw
    var = np.asarray(variance_frames, dtype=np.float64)
    if var.ndim == 1:
        var = np.tile(var, (T, 1))
    Ws = [_window_matrix(l, u, c, T) for (l, u, c) in windows]
    out = np.empty((T, sd))
    for d in range(sd):
        A, b = np.zeros((T, T)), np.zeros(T)
        for w, Ww in enumerate(Ws):
            p = 1.0 / var[:, w * sd + d]
            A += Ww.T @ (p[:, None] * Ww)
            b += Ww.T @ (p * mu[:, w * sd + d])
        out[:, d] = np.linalg.solve(A, b)
    return out


def gen_parameters(stream_sizes, windows, y_predicted, Y_mean, Y_std):
    """evaluation_tts.py:47-83, the ``mge_training=True`` branch (the other branch multiplies a dict,
    :86, and cannot run): split -> per-stream MLPG on the NORMALISED features with unit variance ->
    inverse scaling with statistics indexed in the static+dynamic domain; vuv is only inverse-scaled."""
    mgc_dim, lf0_dim, vuv_dim, bap_dim = stream_sizes
    nw = len(windows)
    lf0_0, vuv_0, bap_0 = mgc_dim, mgc_dim + lf0_dim, mgc_dim + lf0_dim + vuv_dim
    mgc, lf0 = y_predicted[:, :lf0_0], y_predicted[:, lf0_0:vuv_0]
    vuv, bap = y_predicted[:, vuv_0], y_predicted[:, bap_0:]
    mgc = mlpg(mgc, np.ones(mgc.shape[-1]), windows)
    lf0 = mlpg(lf0, np.ones(lf0.shape[-1]), windows)
    bap = mlpg(bap, np.ones(bap.shape[-1]), windows)
    mgc = _inv_scale(mgc, Y_mean[:mgc_dim // nw], Y_std[:mgc_dim // nw])
    lf0 = _inv_scale(lf0, Y_mean[lf0_0:lf0_0 + lf0_dim // nw], Y_std[lf0_0:lf0_0 + lf0_dim // nw])
    bap = _inv_scale(bap, Y_mean[bap_0:bap_0 + bap_dim // nw], Y_std[bap_0:bap_0 + bap_dim // nw])
    vuv = _inv_scale(vuv, Y_mean[vuv_0], Y_std[vuv_0])
    return mgc, lf0, vuv, bap


def predict_durations(duration_pred, Y_mean, Y_std):
    """evaluation_tts.py:170-176: denormalise, round to frames, at least one frame per state."""
    d = np.round(_inv_scale(np.asarray(duration_pred), Y_mean, Y_std))
    d[d <= 0] = 1
    return d
