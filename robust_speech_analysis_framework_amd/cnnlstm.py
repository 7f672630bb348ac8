"""CNN-LSTM-with-attention classifier on the HIP path (drop-in for ``src/models.py``).

``CNNLSTM`` keeps the reference's constructor signature, attribute tree and ``state_dict`` keys
(``src/models.py:129-159``) so the shipped checkpoints load unchanged and
``model.res_block1.conv1.weight`` style access (``src/dl_cv_strategies.py:336,426``) works.  The
parameters live in ordinary ``torch.nn`` containers; ``forward`` does not call them: in eval mode
it folds BatchNorm into the convolutions, packs everything into one device blob and runs
``rsaf_cnnlstm_forward`` (fp32 MFMA GEMMs + persistent LSTM kernel).

Training: in ``model.train()`` mode ``forward`` runs ``rsaf_cnnlstm_train_forward``
(BatchNorm on batch statistics, dropout masks drawn from torch's device RNG, running statistics updated as
``nn.BatchNorm1d`` does) inside a ``torch.autograd.Function`` whose backward is ``rsaf_cnnlstm_train_backward``:
``loss.backward()`` fills ``.grad`` of the ordinary parameters, so the reference's loops
(``src/dl_cv_strategies.py:118-125,241-243``) and ``torch.optim.Adam(model.parameters())`` work unchanged.
``forward`` raises for CPU tensors instead of silently using a PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

_ACT_CODE = {"gelu": 1, "silu": 2}
BN_EPS = 1e-5


def get_activation_fn(name):
    """Same contract as ``src/models.py:7-25``: 'silu' / 'gelu', else ValueError."""
    table = {"silu": F.silu, "gelu": F.gelu}
    if name not in table:
        raise ValueError(f"Unsupported activation function: {name}")
    return table[name]


class ResidualBlock(nn.Module):
    """Parameter container with the reference's layout (``src/models.py:43-62``)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dropout=0.2, activation_fn="silu"):
        super().__init__()
        self.activation = get_activation_fn(activation_fn)
        self.activation_name = activation_fn
        pad = (kernel_size - 1) // 2
        self.conv1 = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding=pad)
        self.bn1 = nn.BatchNorm1d(out_channels)
        self.conv2 = nn.Conv1d(out_channels, out_channels, kernel_size, stride, padding=pad)
        self.bn2 = nn.BatchNorm1d(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.shortcut = nn.Sequential()
        if in_channels != out_channels:
            self.shortcut = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=stride),
                                          nn.BatchNorm1d(out_channels))

        self._packed = None
        self._packed_key = None
        self._workspace = None

    def _folded(self, device):
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._packed is None or self._packed_key != key:
            parts = list(_fold_conv_bn(self.conv1, self.bn1))
            parts += list(_fold_conv_bn(self.shortcut[0], self.shortcut[1])) if len(self.shortcut) > 0 else [None, None]
            parts += list(_fold_conv_bn(self.conv2, self.bn2))
            self._packed = [None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
                            for a in parts]
            self._packed_key = key
        return self._packed

    def forward(self, x):
        """Standalone block (``src/models.py:64-76``): x [B, Cin, T] -> [B, Cout, T], eval mode, on the HIP path
        (``rsaf_cnn_resblock_forward``; the kernels are channels-last, so the two permutes are real copies here while
        ``CNNLSTM.forward`` reads its [B, T, D] input in place)."""
        if not x.is_cuda:
            raise _lib.RsafError("ResidualBlock.forward needs a HIP (cuda) tensor: there is no CPU fallback")
        if self.training:
            raise NotImplementedError("a standalone ResidualBlock runs in eval mode on the HIP path; the training step "
                                      "(batch statistics, dropout, backward) runs through CNNLSTM")
        if any(c.stride[0] != 1 or c.kernel_size[0] != k for c, k in ((self.conv1, 3), (self.conv2, 3))):
            raise NotImplementedError("the HIP block implements kernel_size 3 / stride 1 (all the reference uses)")
        cin, cout = self.conv1.in_channels, self.conv1.out_channels
        if x.dim() != 3 or x.shape[1] != cin:
            raise ValueError(f"expected input [B, {cin}, T], got {tuple(x.shape)}")
        lib = _lib.load()
        w1, b1, wsc, bsc, w2, b2 = self._folded(x.device)
        xt = x.to(torch.float32).permute(0, 2, 1).contiguous()
        B, T = xt.shape[0], xt.shape[1]
        need = max(int(lib.rsaf_cnn_resblock_workspace_bytes(B, T, cout)), 16)
        if self._workspace is None or self._workspace.numel() * 4 < need or self._workspace.device != x.device:
            self._workspace = torch.empty(need // 4, dtype=torch.float32, device=x.device)
        y = torch.empty((B, T, cout), dtype=torch.float32, device=x.device)
        optr = lambda t: _lib.ptr(t) if t is not None else None                      # noqa: E731
        with torch.no_grad():
            _lib.check(lib.rsaf_cnn_resblock_forward(
                _lib.ptr(xt), B, T, cin, cout, _ACT_CODE[self.activation_name], _lib.ptr(w1), _lib.ptr(b1), optr(wsc),
                optr(bsc), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(self._workspace), self._workspace.numel() * 4,
                _lib.ptr(y), _lib.stream_ptr(None)), "rsaf_cnn_resblock_forward")
        return y.permute(0, 2, 1)


class AttentionPooling(nn.Module):
    """Parameter container (``src/models.py:88-92``)."""

    def __init__(self, input_dim):
        super().__init__()
        self.attention_weights = nn.Linear(input_dim, 1)

    def forward(self, lstm_out):
        """Standalone pooling (``src/models.py:94-107``): [B, T, F] -> [B, F] through ``rsaf_attnpool_forward``."""
        if not lstm_out.is_cuda:
            raise _lib.RsafError("AttentionPooling.forward needs a HIP (cuda) tensor: there is no CPU fallback")
        F_ = self.attention_weights.in_features
        if lstm_out.dim() != 3 or lstm_out.shape[2] != F_:
            raise ValueError(f"expected input [B, T, {F_}], got {tuple(lstm_out.shape)}")
        if F_ not in (128, 256):
            raise NotImplementedError("the HIP pooling kernel covers 2 * lstm_hidden_dim = 128 or 256")
        lib = _lib.load()
        x = lstm_out.detach().to(torch.float32).contiguous()
        w = self.attention_weights.weight.detach().to(torch.float32).reshape(-1).contiguous()
        b = self.attention_weights.bias.detach().to(torch.float32).contiguous()
        out = torch.empty((x.shape[0], F_), dtype=torch.float32, device=x.device)
        _lib.check(lib.rsaf_attnpool_forward(_lib.ptr(x), x.shape[0], x.shape[1], F_, _lib.ptr(w), _lib.ptr(b),
                                             _lib.ptr(out), _lib.stream_ptr(None)), "rsaf_attnpool_forward")
        return out


def _f64(t):
    return t.detach().to("cpu", torch.float64).numpy()


def _fold_conv_bn(conv, bn):
    """BN(conv(x)) in eval mode == conv'(x): returns tap-major [Cout, k*Cin] weights and bias."""
    w, b = _f64(conv.weight), _f64(conv.bias)
    s = _f64(bn.weight) / np.sqrt(_f64(bn.running_var) + bn.eps)
    wf = w * s[:, None, None]
    bf = (b - _f64(bn.running_mean)) * s + _f64(bn.bias)
    return np.ascontiguousarray(wf.transpose(0, 2, 1)).reshape(w.shape[0], -1), bf


def weight_offsets(input_dim, channels, hidden, num_classes, layers):
    lib = _lib.load()
    buf = (C.c_int64 * 32)()
    n = C.c_int(0)
    _lib.check(lib.rsaf_cnnlstm_weight_offsets(input_dim, channels, hidden, num_classes, layers, buf, 32,
                                               C.byref(n)), "rsaf_cnnlstm_weight_offsets")
    total = lib.rsaf_cnnlstm_weight_floats(input_dim, channels, hidden, num_classes, layers)
    return [int(buf[i]) for i in range(n.value)], int(total)


def pack_weights(model: "CNNLSTM") -> np.ndarray:
    """Fold + pack the module's parameters into the blob layout of ``rsaf_cnnlstm_forward``."""
    d = model.dims
    offs, total = weight_offsets(d["input_dim"], d["channels"], d["hidden"], d["num_classes"], d["layers"])
    blob = np.zeros(total, dtype=np.float32)
    it = iter(offs)

    def put(arr):
        o = next(it)
        if o >= 0:
            a = np.asarray(arr, dtype=np.float64).reshape(-1)
            blob[o:o + a.size] = a.astype(np.float32)

    r1, r2 = model.res_block1, model.res_block2
    for part in _fold_conv_bn(r1.conv1, r1.bn1):
        put(part)
    if len(r1.shortcut) > 0:
        for part in _fold_conv_bn(r1.shortcut[0], r1.shortcut[1]):
            put(part)
    else:
        next(it), next(it)
    for conv, bn in ((r1.conv2, r1.bn2), (r2.conv1, r2.bn1), (r2.conv2, r2.bn2)):
        for part in _fold_conv_bn(conv, bn):
            put(part)
    for l in range(d["layers"]):
        g = lambda n: _f64(getattr(model.lstm, n))                                   # noqa: E731
        put(np.concatenate([g(f"weight_ih_l{l}"), g(f"weight_ih_l{l}_reverse")], axis=0))
        put(np.concatenate([g(f"bias_ih_l{l}") + g(f"bias_hh_l{l}"),
                            g(f"bias_ih_l{l}_reverse") + g(f"bias_hh_l{l}_reverse")]))
        put(np.stack([g(f"weight_hh_l{l}"), g(f"weight_hh_l{l}_reverse")]))
    put(_f64(model.attention_pooling.attention_weights.weight))
    put(_f64(model.attention_pooling.attention_weights.bias))
    put(_f64(model.fc.weight))
    put(_f64(model.fc.bias))
    return blob


def cnnlstm_forward_packed(x, blob, dims, act, workspace=None, stream=None):
    """x float32 [B,T,D] on the device, blob = packed weights on the same device -> logits [B,NC]."""
    lib = _lib.load()
    if x.dim() != 3 or x.shape[2] != dims["input_dim"]:
        raise ValueError(f"expected input [B, T, {dims['input_dim']}], got {tuple(x.shape)}")
    x = x.contiguous()
    B, T, D = x.shape
    need = lib.rsaf_cnnlstm_workspace_bytes(B, T, D, dims["channels"], dims["hidden"], dims["layers"])
    if B > 0 and need < 0:
        raise ValueError("sequence length must be >= 2")
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(int(need), 16) // 4, dtype=torch.float32, device=x.device)
    logits = torch.empty((B, dims["num_classes"]), dtype=torch.float32, device=x.device)
    _lib.check(lib.rsaf_cnnlstm_forward(
        _lib.ptr(x), B, T, D, dims["channels"], dims["hidden"], dims["num_classes"], dims["layers"],
        _ACT_CODE[act], _lib.ptr(blob), _lib.ptr(workspace), workspace.numel() * 4, _lib.ptr(logits),
        _lib.stream_ptr(stream)), "rsaf_cnnlstm_forward")
    return logits, workspace


def train_param_offsets(dims):
    lib = _lib.load()
    buf = (C.c_int64 * 48)()
    n = C.c_int(0)
    a = (dims["input_dim"], dims["channels"], dims["hidden"], dims["num_classes"], dims["layers"])
    _lib.check(lib.rsaf_cnnlstm_train_param_offsets(*a, buf, 48, C.byref(n)), "rsaf_cnnlstm_train_param_offsets")
    return [int(buf[i]) for i in range(n.value)], int(lib.rsaf_cnnlstm_train_param_floats(*a))


def _train_segments(model):
    """Blob segments in the order of ``rsaf_cnnlstm_train_param_offsets`` (include/rsaf.h): a list of
    ``(offset, n_floats, pack() -> flat tensor, [(parameter, unpack(grad segment) -> grad of that parameter)])``."""
    d = model.dims
    offs, total = train_param_offsets(d)
    it = iter(offs)
    H, L = d["hidden"], d["layers"]
    segs = []

    def plain(prm):
        segs.append((next(it), prm.numel(), (lambda q=prm: q.reshape(-1)), [(prm, lambda g, q=prm: g.view(q.shape))]))

    def conv(cv, bn):
        cout, cin, k = cv.weight.shape                          # stored tap-major [Cout][k][Cin]
        segs.append((next(it), cv.weight.numel(), (lambda w=cv.weight: w.permute(0, 2, 1).reshape(-1)),
                     [(cv.weight, lambda g, a=cout, b=k, c=cin: g.view(a, b, c).permute(0, 2, 1))]))
        for prm in (cv.bias, bn.weight, bn.bias):
            plain(prm)

    r1, r2 = model.res_block1, model.res_block2
    conv(r1.conv1, r1.bn1)
    if len(r1.shortcut) > 0:
        conv(r1.shortcut[0], r1.shortcut[1])
    else:
        for _ in range(4):
            next(it)
    conv(r1.conv2, r1.bn2)
    conv(r2.conv1, r2.bn1)
    conv(r2.conv2, r2.bn2)
    for l in range(L):
        g = lambda n: getattr(model.lstm, n)                                         # noqa: E731
        wf, wr = g(f"weight_ih_l{l}"), g(f"weight_ih_l{l}_reverse")
        segs.append((next(it), 2 * wf.numel(), (lambda a=wf, b=wr: torch.cat([a, b], 0).reshape(-1)),
                     [(wf, lambda gr: gr.view(8 * H, -1)[:4 * H]), (wr, lambda gr: gr.view(8 * H, -1)[4 * H:])]))
        bs = [g(f"bias_ih_l{l}"), g(f"bias_hh_l{l}"), g(f"bias_ih_l{l}_reverse"), g(f"bias_hh_l{l}_reverse")]
        segs.append((next(it), 8 * H, (lambda b=bs: torch.cat([b[0] + b[1], b[2] + b[3]])),
                     [(bs[0], lambda gr: gr[:4 * H]), (bs[1], lambda gr: gr[:4 * H]),
                      (bs[2], lambda gr: gr[4 * H:]), (bs[3], lambda gr: gr[4 * H:])]))
        hf, hr = g(f"weight_hh_l{l}"), g(f"weight_hh_l{l}_reverse")
        segs.append((next(it), 2 * hf.numel(), (lambda a=hf, b=hr: torch.stack([a, b]).reshape(-1)),
                     [(hf, lambda gr: gr.view(2, 4 * H, H)[0]), (hr, lambda gr: gr.view(2, 4 * H, H)[1])]))
    aw = model.attention_pooling.attention_weights
    for prm in (aw.weight, aw.bias, model.fc.weight, model.fc.bias):
        plain(prm)
    return segs, total


def _bn_modules(model):
    r1, r2 = model.res_block1, model.res_block2
    return [r1.bn1, r1.shortcut[1] if len(r1.shortcut) > 0 else None, r1.bn2, r2.bn1, r2.bn2]


def draw_masks(model, B, T, device):
    """Dropout keep masks of one training step (float32 0 or 1/(1-p); None where p == 0), from torch's device RNG."""
    d = model.dims
    Tp = T // 2

    def mk(shape, p):
        if p <= 0.0:
            return None
        if p >= 1.0:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        return (torch.rand(shape, device=device) >= p).to(torch.float32) / (1.0 - p)

    p_l = float(model.lstm.dropout)
    return {"res_block1": mk((B, T, d["channels"]), float(model.res_block1.dropout.p)),
            "res_block2": mk((B, Tp, d["channels"]), float(model.res_block2.dropout.p)),
            "lstm": [mk((B, Tp, 2 * d["hidden"]), p_l) for _ in range(d["layers"] - 1)],
            "fc": mk((B, 2 * d["hidden"]), float(model.dropout.p))}


class _TrainStep(torch.autograd.Function):
    """logits = CNNLSTM(x) in training mode; backward fills the parameter gradients (none for x)."""

    @staticmethod
    def forward(ctx, x, model, masks, *params):
        lib = _lib.load()
        d = model.dims
        B, T, D = x.shape
        segs, total = _train_segments(model)
        blob = torch.zeros(total, dtype=torch.float32, device=x.device)
        with torch.no_grad():
            for off, n, pack, _ in segs:
                blob[off:off + n] = pack()
        a = (B, T, D, d["channels"], d["hidden"], d["layers"])
        n_saved, n_scr = int(lib.rsaf_cnnlstm_train_saved_floats(*a)), int(lib.rsaf_cnnlstm_train_scratch_floats(*a))
        if n_saved < 0 or n_scr < 0:
            raise ValueError("sequence length must be >= 2")
        saved = torch.empty(n_saved, dtype=torch.float32, device=x.device)
        if model._train_scratch is None or model._train_scratch.numel() < n_scr or model._train_scratch.device != x.device:
            model._train_scratch = torch.empty(n_scr, dtype=torch.float32, device=x.device)
        scratch = model._train_scratch
        logits = torch.empty((B, d["num_classes"]), dtype=torch.float32, device=x.device)
        stats = torch.empty((5, 3, d["channels"]), dtype=torch.float32, device=x.device)
        lm = masks["lstm"]
        lstm_ptrs = (C.c_void_p * max(len(lm), 1))(*[(_lib.ptr(m) if m is not None else None) for m in lm]) if lm else None
        ctx.call = (x, B, T, D, d["channels"], d["hidden"], d["num_classes"], d["layers"], _ACT_CODE[model.activation_name])
        ctx.bufs = (blob, masks, lstm_ptrs, saved, scratch)
        ctx.segs = segs
        ctx.params = params
        ctx.model = model
        optr = lambda t: _lib.ptr(t) if t is not None else None                      # noqa: E731
        _lib.check(lib.rsaf_cnnlstm_train_forward(
            _lib.ptr(x), *ctx.call[1:], _lib.ptr(blob), optr(masks["res_block1"]), optr(masks["res_block2"]), lstm_ptrs,
            optr(masks["fc"]), _lib.ptr(saved), n_saved, _lib.ptr(scratch), scratch.numel(), _lib.ptr(logits),
            _lib.ptr(stats), _lib.stream_ptr(None)), "rsaf_cnnlstm_train_forward")
        # running statistics, as nn.BatchNorm1d in training mode (momentum, unbiased variance)
        with torch.no_grad():
            for i, (bn, n) in enumerate(zip(_bn_modules(model), (B * T, B * T, B * T, B * (T // 2), B * (T // 2)))):
                if bn is None or not bn.track_running_stats or bn.running_mean is None:
                    continue
                bn.num_batches_tracked += 1
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - m).add_(stats[i, 0], alpha=m)
                bn.running_var.mul_(1 - m).add_(stats[i, 1], alpha=m * (n / (n - 1.0) if n > 1 else 1.0))
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        if ctx.bufs is None:
            raise RuntimeError("CNNLSTM training step: backward can run once per forward (the saved activations are consumed)")
        blob, masks, lstm_ptrs, saved, scratch = ctx.bufs
        x = ctx.call[0]
        if scratch is not ctx.model._train_scratch:
            scratch = torch.empty_like(scratch)
        grads = torch.zeros_like(blob)
        dl = dlogits.to(torch.float32).contiguous()
        optr = lambda t: _lib.ptr(t) if t is not None else None                      # noqa: E731
        _lib.check(lib.rsaf_cnnlstm_train_backward(
            _lib.ptr(x), *ctx.call[1:], _lib.ptr(blob), optr(masks["res_block1"]), optr(masks["res_block2"]), lstm_ptrs,
            optr(masks["fc"]), _lib.ptr(saved), saved.numel(), _lib.ptr(scratch), scratch.numel(), _lib.ptr(dl),
            _lib.ptr(grads), _lib.stream_ptr(None)), "rsaf_cnnlstm_train_backward")
        ctx.bufs = None
        by_param = {id(prm): (off, n, unpack) for off, n, _, outs in ctx.segs for prm, unpack in outs}
        out = []
        for prm in ctx.params:
            off, n, unpack = by_param[id(prm)]
            out.append(unpack(grads[off:off + n]).reshape(prm.shape).contiguous())
        return (None, None, None, *out)


class CNNLSTM(nn.Module):
    """Drop-in for ``src/models.py:109-193`` (constructor signature and state_dict keys identical)."""

    def __init__(self, input_dim=768, num_classes=2, cnn_out_channels=128, lstm_hidden_dim=128,
                 lstm_layers=2, dropout_rate=0.5, activation_fn="silu"):
        super().__init__()
        get_activation_fn(activation_fn)                       # ValueError for unknown names
        self.activation_name = activation_fn
        self.res_block1 = ResidualBlock(input_dim, cnn_out_channels, activation_fn=activation_fn)
        self.res_block2 = ResidualBlock(cnn_out_channels, cnn_out_channels, activation_fn=activation_fn)
        self.lstm = nn.LSTM(input_size=cnn_out_channels, hidden_size=lstm_hidden_dim, num_layers=lstm_layers,
                            batch_first=True, bidirectional=True,
                            dropout=dropout_rate if lstm_layers > 1 else 0)
        self.attention_pooling = AttentionPooling(input_dim=lstm_hidden_dim * 2)
        self.dropout = nn.Dropout(dropout_rate)
        self.fc = nn.Linear(lstm_hidden_dim * 2, num_classes)
        self.dims = {"input_dim": input_dim, "channels": cnn_out_channels, "hidden": lstm_hidden_dim,
                     "num_classes": num_classes, "layers": lstm_layers}
        self._packed = None
        self._packed_key = None
        self._workspace = None
        self._train_scratch = None
        self.forced_masks = None            # tests: explicit dropout masks for the next training-mode forward

    def _weights_key(self, device):
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in
                                      list(self.parameters()) + list(self.buffers()))

    def packed_weights(self, device):
        """Folded weight blob on ``device`` (rebuilt when any parameter/buffer changed)."""
        key = self._weights_key(device)
        if self._packed is None or self._packed_key != key:
            self._packed = torch.from_numpy(pack_weights(self)).to(device)
            self._packed_key = key
        return self._packed

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.RsafError("CNNLSTM.forward needs a HIP (cuda) tensor: there is no CPU fallback")
        x = x.to(torch.float32)
        if self.training:
            if x.dim() != 3 or x.shape[2] != self.dims["input_dim"]:
                raise ValueError(f"expected input [B, T, {self.dims['input_dim']}], got {tuple(x.shape)}")
            if x.shape[0] * (x.shape[1] // 2) <= 1:
                # nn.BatchNorm1d in training mode (res_block2 sees B * (T // 2) values per channel)
                raise ValueError("Expected more than 1 value per channel when training")
            x = x.contiguous()
            masks = self.forced_masks if self.forced_masks is not None else draw_masks(self, x.shape[0], x.shape[1], x.device)
            params = [p for _, _, _, outs in _train_segments(self)[0] for p, _ in outs]
            return _TrainStep.apply(x.detach(), self, masks, *params)
        blob = self.packed_weights(x.device)
        with torch.no_grad():
            logits, self._workspace = cnnlstm_forward_packed(x, blob, self.dims, self.activation_name,
                                                             self._workspace)
        return logits


def cnnlstm_forward_stages(model: "CNNLSTM", x):
    """Eval-mode forward that also returns what the reference's sub-modules return (forward hooks on
    ``res_block1`` / ``res_block2`` / ``lstm`` / ``attention_pooling`` of ``src/models.py``), channels-last:
    dict(res1 [B,T,C], res2 [B,T/2,C], lstm [B,T/2,2H], pooled [B,2H], logits [B,NC])."""
    lib = _lib.load()
    if not x.is_cuda:
        raise _lib.RsafError("cnnlstm_forward_stages needs a HIP (cuda) tensor")
    d = model.dims
    x = x.to(torch.float32).contiguous()
    B, T, D = x.shape
    blob = model.packed_weights(x.device)
    need = lib.rsaf_cnnlstm_workspace_bytes(B, T, D, d["channels"], d["hidden"], d["layers"])
    if need < 0:
        raise ValueError("sequence length must be >= 2")
    ws = torch.empty(max(int(need), 16) // 4, dtype=torch.float32, device=x.device)
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)            # noqa: E731
    out = {"res1": e(B, T, d["channels"]), "res2": e(B, T // 2, d["channels"]), "lstm": e(B, T // 2, 2 * d["hidden"]),
           "pooled": e(B, 2 * d["hidden"]), "logits": e(B, d["num_classes"])}
    _lib.check(lib.rsaf_cnnlstm_forward_stages(
        _lib.ptr(x), B, T, D, d["channels"], d["hidden"], d["num_classes"], d["layers"], _ACT_CODE[model.activation_name],
        _lib.ptr(blob), _lib.ptr(ws), ws.numel() * 4, _lib.ptr(out["logits"]), _lib.ptr(out["res1"]), _lib.ptr(out["res2"]),
        _lib.ptr(out["lstm"]), _lib.ptr(out["pooled"]), _lib.stream_ptr(None)), "rsaf_cnnlstm_forward_stages")
    return out


def collate_zero_pad(seqs, device="cuda"):
    """Batch assembly of the reference harness (``src/dl_cv_strategies.py:81-84``): right zero-padding
    to the batch maximum, float32, no mask."""
    T = max(int(s.shape[0]) for s in seqs)
    out = torch.zeros((len(seqs), T, int(seqs[0].shape[1])), dtype=torch.float32, device=device)
    for i, s in enumerate(seqs):
        out[i, :s.shape[0]] = torch.as_tensor(s, dtype=torch.float32)
    return out


def eval_outputs(logits):
    """``_eval_model`` post-processing (``src/dl_cv_strategies.py:183-194``): P(class 1) and argmax."""
    return torch.softmax(logits, dim=1)[:, 1], torch.argmax(logits, dim=1)
