"""Generator / discriminator networks with the reference's constructor and module API
(``gantts/models.py``), computed by the HIP engine.

Each class is an ``nn.Module`` only for *plumbing*: parameters are ``nn.Parameter`` views into
ONE flat float32 device buffer per network (state_dict order), so that
  * ``state_dict()/load_state_dict()/parameters()/cuda()/train()/eval()`` and
    ``torch.save`` checkpoints keep the reference's keys and (out, in) weight layout
    (reference train.py:162-171, 651-658; SURVEY 5 "Checkpoint / resume"),
  * the engine sees a single contiguous parameter / gradient / optimizer-state range
    (fused clip-norm + optimizer kernel, one all-reduce bucket per network).
No torch op touches the numbers: forward/backward/step run in libgantts_hip.so.
"""
import math
import weakref

import torch
from torch import nn

from . import _lib as L


class AbstractModel(object):
    """Interface for VC and TTS models (reference gantts/models.py:11-18)."""

    def include_parameter_generation(self):
        return False


class _LinearParams(nn.Module):
    """Holds ``weight`` (out, in) and ``bias`` (out) of one nn.Linear; never called."""

    def __init__(self, out_dim, in_dim):
        super(_LinearParams, self).__init__()
        self.in_features, self.out_features = in_dim, out_dim
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim))
        self.bias = nn.Parameter(torch.empty(out_dim))

    def init_bound(self, name=None):
        return 1.0 / math.sqrt(self.in_features)     # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(in))

    def extra_repr(self):
        return "in_features=%d, out_features=%d" % (self.in_features, self.out_features)


class _LSTMParams(nn.Module):
    """Parameters of an ``nn.LSTM(batch_first=True)`` in torch's registration order and naming
    (``weight_ih_l{k}[_reverse]``, ``weight_hh_l{k}..``, ``bias_ih_l{k}..``, ``bias_hh_l{k}..``;
    gate order i, f, g, o); never called."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional):
        super(_LSTMParams, self).__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.bidirectional = bool(bidirectional)
        dirs = 2 if bidirectional else 1
        for layer in range(num_layers):
            n_in = input_size if layer == 0 else hidden_size * dirs
            for d in range(dirs):
                sfx = "_l%d%s" % (layer, "_reverse" if d else "")
                self.register_parameter("weight_ih" + sfx, nn.Parameter(torch.empty(4 * hidden_size, n_in)))
                self.register_parameter("weight_hh" + sfx, nn.Parameter(torch.empty(4 * hidden_size, hidden_size)))
                self.register_parameter("bias_ih" + sfx, nn.Parameter(torch.empty(4 * hidden_size)))
                self.register_parameter("bias_hh" + sfx, nn.Parameter(torch.empty(4 * hidden_size)))

    def init_bound(self, name=None):
        return 1.0 / math.sqrt(self.hidden_size)      # nn.LSTM.reset_parameters

    def extra_repr(self):
        return "%d, %d, num_layers=%d, batch_first=True, bidirectional=%s" % (
            self.input_size, self.hidden_size, self.num_layers, self.bidirectional)


class _SRUCellParams(nn.Module):
    """``weight`` (n_in, ncols*k) and ``bias`` (2*ncols) of one SRU cell in the layout of the 2017
    ``cuda_functional.SRUCell`` the reference imports (gantts/models.py:150): k = 3 when
    n_in == ncols else 4; init U(+-sqrt(3/n_in)), zero bias; never called."""

    def __init__(self, n_in, n_out, bidirectional):
        super(_SRUCellParams, self).__init__()
        self.n_in, self.n_out, self.bidirectional = n_in, n_out, bool(bidirectional)
        ncols = n_out * (2 if bidirectional else 1)
        self.k = 3 if n_in == ncols else 4
        self.weight = nn.Parameter(torch.empty(n_in, ncols * self.k))
        self.bias = nn.Parameter(torch.empty(2 * ncols))

    def init_bound(self, name=None):
        return math.sqrt(3.0 / self.n_in) if name == "weight" else 0.0

    def extra_repr(self):
        return "n_in=%d, n_out=%d, k=%d, bidirectional=%s" % (self.n_in, self.n_out, self.k, self.bidirectional)


class _SRUStack(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, bidirectional):
        super(_SRUStack, self).__init__()
        ncols = hidden_size * (2 if bidirectional else 1)
        self.rnn_lst = nn.ModuleList([_SRUCellParams(input_size if i == 0 else ncols, hidden_size, bidirectional)
                                      for i in range(num_layers)])


class _FlatNetwork(AbstractModel, nn.Module):
    ARCH = None

    def _finish_init(self):
        """Re-homes all parameters into one flat buffer with nn.Linear's default init."""
        params = list(self.parameters())
        total = sum(p.numel() for p in params)
        self._flat = torch.empty(total, dtype=torch.float32)
        self._flat_grad = None
        self._repoint(init=True)
        self._engine = None          # private engine for plain forward calls
        self._masks = {}             # (pass, layer) -> injected dropout mask tensor (tests)
        self._bound_engines = {}     # id(engine) -> (weakref(engine), role): who must hear zero_grad()
        self._version = 0            # bumped whenever buffers are re-homed (engines re-bind)

    def _holders(self):
        return [m for m in self.modules() if isinstance(m, (_LinearParams, _LSTMParams, _SRUCellParams))]

    def _repoint(self, init=False):
        off = 0
        for holder in self._holders():
            for name, p in holder._parameters.items():     # registration order == state_dict order
                n = p.numel()
                view = self._flat[off:off + n].view(p.shape)
                if init:
                    k = holder.init_bound(name)
                    view.uniform_(-k, k) if k > 0 else view.zero_()
                p.data = view
                p._gt_owner = weakref.ref(self)
                p.grad = None if self._flat_grad is None else self._flat_grad[off:off + n].view(p.shape)
                off += n
        self._version = getattr(self, "_version", 0) + 1

    def _apply(self, fn, recurse=True):
        # .cuda()/.cpu()/.to(): move the flat buffers, then re-create the parameter views
        self._flat = fn(self._flat)
        if self._flat.dtype != torch.float32:
            raise TypeError("gantts_amd networks are float32 only")
        if self._flat_grad is not None:
            self._flat_grad = fn(self._flat_grad)
        self._repoint()
        return self

    def flat_params(self):
        return self._flat

    def flat_grads(self):
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
            self._repoint()
        return self._flat_grad

    def num_flat_params(self):
        return self._flat.numel()

    def set_dropout_masks(self, pass_index, masks):
        """Parity hook: inject 0/1 keep-masks (one per hidden layer, shape (B,T,hidden)) for forward
        pass ``pass_index`` (G: 0; D: 0 real / 1 fake of the D step, 2 fake of the G step)."""
        for layer in range(self.num_hidden):
            m = None if masks is None or layer >= len(masks) else masks[layer]
            if m is not None:
                m = m.to(self._flat.device, torch.float32).contiguous()
            self._masks[(pass_index, layer)] = m
        self._version += 1

    def _check_masks(self, B, T):
        """Injected dropout masks are read by the kernels for every row of the batch: a mask with fewer rows than the
        batch would be read out of bounds on the device (the C ABI carries no shape for them)."""
        for (pass_idx, layer), m in self._masks.items():
            if m is not None and m.numel() // m.size(-1) != B * T:
                raise RuntimeError("injected dropout mask (pass %d, layer %d) has %d rows, the batch has B*T = %d frames"
                                   % (pass_idx, layer, m.numel() // m.size(-1), B * T))

    def _desc(self, with_grads):
        d = L.ModelDesc()
        d.arch = self.ARCH
        d.in_dim, d.out_dim = self.in_dim, self.out_dim
        d.num_hidden, d.hidden_dim = self.num_hidden, self.hidden_dim
        d.static_dim = getattr(self, "static_dim", 0)
        d.dropout = float(self.dropout_p)
        d.last_sigmoid = int(bool(getattr(self, "last_sigmoid", False)))
        d.bidirectional = int(getattr(self, "num_direction", 1) == 2)
        d.use_relu = int(bool(getattr(self, "use_relu", 0)))
        d.rnn_dropout = float(getattr(self, "rnn_dropout", 0.0))
        if not self._flat.is_cuda:
            raise RuntimeError("gantts_amd: model is on %s -- call .cuda() first (the HIP engine has no CPU path)"
                               % self._flat.device)
        d.params = self._flat.data_ptr()
        d.grads = self.flat_grads().data_ptr() if with_grads else None
        d.n_params = self._flat.numel()
        return d

    def _own_engine(self):
        from .engine import StepEngine
        if self._engine is None:
            self._engine = StepEngine.for_forward_only(self)
        return self._engine


class MLP(_FlatNetwork):
    """Stack of Linear -> LeakyReLU(0.01) -> Dropout, final Linear, optional sigmoid
    (reference gantts/models.py:121-141).  ``bidirectional`` is a dummy, as in the reference."""
    ARCH = L.ARCH_MLP

    def __init__(self, in_dim=118, out_dim=1, num_hidden=2, hidden_dim=256,
                 dropout=0.5, last_sigmoid=True, bidirectional=None):
        super(MLP, self).__init__()
        self.in_dim, self.out_dim, self.num_hidden, self.hidden_dim = in_dim, out_dim, num_hidden, hidden_dim
        self.dropout_p, self.last_sigmoid = dropout, last_sigmoid
        ins = [in_dim] + [hidden_dim] * (num_hidden - 1)
        self.layers = nn.ModuleList([_LinearParams(hidden_dim, i) for i in ins])
        self.last_linear = _LinearParams(out_dim, hidden_dim)
        self._finish_init()

    def forward(self, x, lengths=None):
        return self._own_engine().model_forward(self, x)


class In2OutHighwayNet(_FlatNetwork):
    """Input-to-output highway network for VC with MLPG inside the model
    (reference gantts/models.py:21-69): returns ``(G(x), x_static + T(x) * MLPG(G(x)))``."""
    ARCH = L.ARCH_IN2OUT

    def __init__(self, in_dim=118, out_dim=118, static_dim=118 // 2,
                 num_hidden=3, hidden_dim=512, dropout=0.5):
        super(In2OutHighwayNet, self).__init__()
        self.in_dim, self.out_dim, self.num_hidden, self.hidden_dim = in_dim, out_dim, num_hidden, hidden_dim
        self.static_dim, self.dropout_p = static_dim, dropout
        self.T = _LinearParams(static_dim, static_dim)
        ins = [in_dim] + [hidden_dim] * (num_hidden - 1)
        self.H = nn.ModuleList([_LinearParams(hidden_dim, i) for i in ins])
        self.last_linear = _LinearParams(out_dim, hidden_dim)
        self._finish_init()

    def include_parameter_generation(self):
        return True

    def forward(self, x, R, lengths=None):
        return self._own_engine().model_forward(self, x, R)


class LSTMRNN(_FlatNetwork):
    """``pack_padded_sequence -> nn.LSTM(in, H, num_hidden, batch_first, bidirectional, dropout) ->
    pad_packed_sequence -> hidden2out -> optional sigmoid`` (reference gantts/models.py:193-213).
    ``forward(sequence, lengths)``: a sequence is active at frame t iff t < length; outputs beyond a
    length are zero before hidden2out, the reverse direction starts at each sequence's last valid
    frame.  Unlike pack_padded_sequence the batch does not have to be sorted by length."""
    ARCH = L.ARCH_LSTM
    RNN_ATTR = "lstm"
    needs_lengths = True

    def __init__(self, in_dim=118, out_dim=118, num_hidden=2, hidden_dim=256,
                 bidirectional=False, dropout=0, last_sigmoid=False):
        super(LSTMRNN, self).__init__()
        self.in_dim, self.out_dim, self.num_hidden, self.hidden_dim = in_dim, out_dim, num_hidden, hidden_dim
        self.num_direction = 2 if bidirectional else 1
        self.dropout_p, self.last_sigmoid = dropout, last_sigmoid
        setattr(self, self.RNN_ATTR, _LSTMParams(in_dim, hidden_dim, num_hidden, bidirectional))
        self.hidden2out = _LinearParams(out_dim, hidden_dim * self.num_direction)
        self._finish_init()

    def set_dropout_masks(self, pass_index, masks):
        """Parity hook for nn.LSTM's inter-layer dropout: ``masks[l]`` (B,T,H*dirs) 0/1 keep-mask on
        the outputs of layer ``l`` (l < num_hidden-1); entries for the last layer are ignored."""
        if pass_index not in (0, 1, 2):      # (0: a generator's pass / D(real); 1, 2: D(fake) of the D and of the G step -- a recurrent discriminator)
            raise ValueError("pass_index must be 0, 1 or 2")
        for layer in range(self.num_hidden):
            m = None if masks is None or layer >= len(masks) else masks[layer]
            if m is not None:
                m = m.to(self._flat.device, torch.float32).contiguous()
            self._masks[(pass_index, layer)] = m
        self._version += 1

    def forward(self, sequence, lengths):
        return self._own_engine().model_forward(self, sequence, lengths=lengths)


class In2OutRNNHighwayNet(LSTMRNN):
    """Recurrent input-to-output highway network (reference gantts/models.py:72-118):
    ``Tx = sigmoid(T(x_static))``, ``Gx = MLPG(R, hidden2out(LSTM(x, lengths)))`` and the model
    returns ``(x, x_static + Tx * Gx)`` -- the first output is the INPUT itself (models.py:118), so
    the MSE term of update_generator has no path into the weights.  state_dict keys ``T.*``,
    ``lstm.*``, ``hidden2out.*``."""
    ARCH = L.ARCH_IN2OUT_RNN

    def __init__(self, in_dim=118, out_dim=118, static_dim=118 // 2,
                 num_hidden=3, hidden_dim=512, bidirectional=False, dropout=0.5):
        _FlatNetwork.__init__(self)
        if in_dim != out_dim:
            raise ValueError("In2OutRNNHighwayNet returns its input as y_hat: in_dim must equal out_dim")
        self.in_dim, self.out_dim, self.num_hidden, self.hidden_dim = in_dim, out_dim, num_hidden, hidden_dim
        self.static_dim = static_dim
        self.num_direction = 2 if bidirectional else 1
        self.dropout_p, self.last_sigmoid = dropout, False
        self.T = _LinearParams(static_dim, static_dim)
        self.lstm = _LSTMParams(in_dim, hidden_dim, num_hidden, bidirectional)
        self.hidden2out = _LinearParams(out_dim, hidden_dim * self.num_direction)
        self._finish_init()

    def include_parameter_generation(self):
        return True

    def forward(self, x, R, lengths=None):
        return self._own_engine().model_forward(self, x, R, lengths=lengths)


class GRURNN(LSTMRNN):
    """Named GRU in the reference but built on ``nn.LSTM`` under the attribute name ``gru``
    (gantts/models.py:170-190); state_dict keys are ``gru.weight_ih_l0`` ..."""
    RNN_ATTR = "gru"


class SRURNN(_FlatNetwork):
    """``SRU(in_dim, hidden_dim, num_hidden, bidirectional, dropout, use_relu, rnn_dropout)`` on the
    time-major view of the batch, then ``hidden2out`` and an optional sigmoid (reference
    gantts/models.py:144-167; the default generator of both TTS hparams sets, hparams.py:111-124,
    211-222).  The SRU cell itself is third-party code the reference does not vendor
    (github.com/taolei87/sru): this class follows its 2017 ``cuda_functional`` layout (state_dict keys
    ``gru.rnn_lst.{i}.weight/bias``) and recurrence; ``lengths`` are ignored like in the reference."""
    ARCH = L.ARCH_SRU

    def __init__(self, in_dim=118, out_dim=118, num_hidden=2, hidden_dim=256,
                 bidirectional=False, dropout=0, last_sigmoid=False,
                 use_relu=0, rnn_dropout=0.0):
        super(SRURNN, self).__init__()
        self.in_dim, self.out_dim, self.num_hidden, self.hidden_dim = in_dim, out_dim, num_hidden, hidden_dim
        self.num_direction = 2 if bidirectional else 1
        self.dropout_p, self.last_sigmoid = dropout, last_sigmoid
        self.use_relu, self.rnn_dropout = use_relu, rnn_dropout
        self.gru = _SRUStack(in_dim, hidden_dim, num_hidden, bidirectional)
        self.hidden2out = _LinearParams(out_dim, hidden_dim * self.num_direction)
        self._finish_init()

    def set_dropout_masks(self, pass_index, masks):
        """Parity hook for the SRU cell's variational dropout: ``masks`` in the order a forward pass draws them --
        per layer the input mask (B, n_in) if ``rnn_dropout > 0``, then the output mask (B, H*dirs) if ``dropout > 0``
        and the layer is not the last -- as 0/1 keep tensors; ``None`` restores the Philox stream."""
        if pass_index != 0:
            raise ValueError("recurrent generators have a single forward pass (index 0)")
        masks = list(masks) if masks is not None else None
        ncols = self.hidden_dim * self.num_direction
        for layer in range(self.num_hidden):
            n_in = self.in_dim if layer == 0 else ncols
            for which, on, width in ((0, self.rnn_dropout > 0, n_in), (1, self.dropout_p > 0 and layer + 1 < self.num_hidden, ncols)):
                m = None
                if masks is not None and on:
                    if not masks:
                        raise ValueError("too few dropout masks for this SRURNN")
                    m = masks.pop(0).to(self._flat.device, torch.float32).contiguous()
                    if m.dim() != 2 or m.size(1) != width:
                        raise ValueError("SRU layer %d %s mask must be (B, %d), got %s" % (layer, "input" if which == 0 else "output", width, tuple(m.shape)))
                self._masks[(0, 2 * layer + which)] = m
        if masks:
            raise ValueError("too many dropout masks for this SRURNN")
        self._version += 1

    def _check_masks(self, B, T):
        for (_, site), m in self._masks.items():
            if m is not None and m.size(0) != B:
                raise RuntimeError("injected SRU dropout mask of layer %d has %d rows, the batch has %d sequences "
                                   "(variational masks are (B, width))" % (site // 2, m.size(0), B))

    def forward(self, sequence, lengths=None):
        return self._own_engine().model_forward(self, sequence)
