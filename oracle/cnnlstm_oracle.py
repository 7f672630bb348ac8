"""CPU restatement of the reference classifier forward, ``CNNLSTM.forward`` (``src/models.py:161-193``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pinned: ``tests/golden/cnnlstm_*.npz`` hold inputs,
state_dicts and per-stage outputs captured from the reference module itself in the build container
(``tests/golden/make_cnnlstm_golden.py``), plus logits of the two shipped checkpoints.

Two restatements of the same arithmetic:

* ``forward_numpy``  – first principles, float64 numpy (explicit conv sums, explicit LSTM cell
  loop).  Small cases only; it is what pins the semantics (eval-mode BN, gate order i,f,g,o,
  floor max-pool, no masking of padded frames, softmax over time).
* ``forward_torch``  – the same graph with torch CPU functional ops in float32 (what the reference
  executes on a CPU host: ``F.conv1d``, ``F.batch_norm``, ``nn.LSTM``).  Used for full-size parity
  and as the timed CPU baseline (``bench.py`` cpu_baseline, kind "port").

Both take a reference-format ``state_dict`` (keys of SURVEY.md App. D) as a dict of numpy arrays.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5


def _act(x, name):
    if name == "silu":
        return x / (1.0 + np.exp(-x))
    if name == "gelu":
        from scipy.special import erf
        return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))
    raise ValueError(f"Unsupported activation function: {name}")      # src/models.py:25


def _conv1d_same(x, w, b):
    """x [B, Cin, T], w [Cout, Cin, k] (k odd, padding (k-1)//2, stride 1) -> [B, Cout, T]."""
    k = w.shape[2]
    pad = (k - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    T = x.shape[2]
    out = np.zeros((x.shape[0], w.shape[0], T), dtype=np.float64)
    for j in range(k):
        out += np.einsum("oc,bct->bot", w[:, :, j], xp[:, :, j:j + T])
    return out + b[None, :, None]


def _bn_eval(x, sd, prefix):
    g, be = sd[prefix + ".weight"], sd[prefix + ".bias"]
    mu, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - mu[None, :, None]) / np.sqrt(var[None, :, None] + BN_EPS) * g[None, :, None] + be[None, :, None]


def _res_block(x, sd, p, act):
    """ResidualBlock.forward, src/models.py:64-76 (eval: dropout = identity)."""
    out = _act(_bn_eval(_conv1d_same(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]), sd, p + ".bn1"), act)
    out = _bn_eval(_conv1d_same(out, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]), sd, p + ".bn2")
    if (p + ".shortcut.0.weight") in sd:
        sc = _bn_eval(_conv1d_same(x, sd[p + ".shortcut.0.weight"], sd[p + ".shortcut.0.bias"]), sd, p + ".shortcut.1")
    else:
        sc = x
    return _act(out + sc, act)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [B, T, In] -> h [B, T, H]; zero initial state; gate order i, f, g, o (torch.nn.LSTM)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H))
    c = np.zeros((B, H))
    out = np.zeros((B, T, H))
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = x[:, t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
        i, f, gg, o = _sigmoid(g[:, :H]), _sigmoid(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), _sigmoid(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[:, t] = h
    return out


def n_lstm_layers(sd):
    n = 0
    while f"lstm.weight_ih_l{n}" in sd:
        n += 1
    return n


def forward_numpy(sd, x, activation_fn="silu", return_stages=False):
    """float64 first-principles forward.  x [B, T, D] -> logits [B, num_classes]."""
    sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    st = {}
    h = np.asarray(x, dtype=np.float64).transpose(0, 2, 1)                      # :172
    h = _res_block(h, sd, "res_block1", activation_fn)                           # :175
    st["res1"] = h.transpose(0, 2, 1)
    Tp = h.shape[2] // 2
    h = np.maximum(h[:, :, 0:2 * Tp:2], h[:, :, 1:2 * Tp:2])                    # max_pool1d(2), :177
    st["pool"] = h.transpose(0, 2, 1)
    h = _res_block(h, sd, "res_block2", activation_fn)                           # :178
    st["res2"] = h.transpose(0, 2, 1)
    seq = h.transpose(0, 2, 1)                                                   # :181
    for layer in range(n_lstm_layers(sd)):                                       # :184
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            outs.append(_lstm_dir(seq, sd[f"lstm.weight_ih_l{layer}{suffix}"], sd[f"lstm.weight_hh_l{layer}{suffix}"],
                                  sd[f"lstm.bias_ih_l{layer}{suffix}"], sd[f"lstm.bias_hh_l{layer}{suffix}"], rev))
        seq = np.concatenate(outs, axis=2)
    st["lstm"] = seq
    sc = seq @ sd["attention_pooling.attention_weights.weight"].T + sd["attention_pooling.attention_weights.bias"]  # :99
    sc = sc - sc.max(axis=1, keepdims=True)
    p = np.exp(sc)
    p = p / p.sum(axis=1, keepdims=True)                                         # softmax over time, :102
    ctx = np.sum(seq * p, axis=1)                                                # :105
    st["pooled"] = ctx
    logits = ctx @ sd["fc.weight"].T + sd["fc.bias"]                              # :191
    st["logits"] = logits
    return (logits, st) if return_stages else logits


def forward_torch(sd, x, activation_fn="silu", return_stages=False):
    """float32 torch-CPU forward with the ops the reference itself dispatches."""
    import torch
    import torch.nn.functional as F
    t = {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
    act = {"silu": F.silu, "gelu": F.gelu}.get(activation_fn)
    if act is None:
        raise ValueError(f"Unsupported activation function: {activation_fn}")

    def bn(h, p):
        return F.batch_norm(h, t[p + ".running_mean"].float(), t[p + ".running_var"].float(),
                            t[p + ".weight"].float(), t[p + ".bias"].float(), False, 0.0, BN_EPS)

    def block(h, p):
        o = act(bn(F.conv1d(h, t[p + ".conv1.weight"].float(), t[p + ".conv1.bias"].float(), padding=1), p + ".bn1"))
        o = bn(F.conv1d(o, t[p + ".conv2.weight"].float(), t[p + ".conv2.bias"].float(), padding=1), p + ".bn2")
        if (p + ".shortcut.0.weight") in t:
            s = bn(F.conv1d(h, t[p + ".shortcut.0.weight"].float(), t[p + ".shortcut.0.bias"].float()), p + ".shortcut.1")
        else:
            s = h
        return act(o + s)

    st = {}
    with torch.no_grad():
        h = torch.as_tensor(np.asarray(x), dtype=torch.float32).permute(0, 2, 1)
        h = block(h, "res_block1")
        st["res1"] = h.permute(0, 2, 1)
        h = F.max_pool1d(h, kernel_size=2)
        st["pool"] = h.permute(0, 2, 1)
        h = block(h, "res_block2")
        st["res2"] = h.permute(0, 2, 1)
        seq = h.permute(0, 2, 1)
        nl = n_lstm_layers(sd)
        Hh = t["lstm.weight_hh_l0"].shape[1]
        lstm = torch.nn.LSTM(seq.shape[2], Hh, num_layers=nl, batch_first=True, bidirectional=True)
        lstm.load_state_dict({k[5:]: v.float() for k, v in t.items() if k.startswith("lstm.")})
        lstm.eval()
        seq, _ = lstm(seq.contiguous())
        st["lstm"] = seq
        sc = F.linear(seq, t["attention_pooling.attention_weights.weight"].float(),
                      t["attention_pooling.attention_weights.bias"].float())
        p = F.softmax(sc, dim=1)
        ctx = torch.sum(seq * p, dim=1)
        st["pooled"] = ctx
        logits = F.linear(ctx, t["fc.weight"].float(), t["fc.bias"].float())
        st["logits"] = logits
    if return_stages:
        return logits.numpy(), {k: v.numpy() for k, v in st.items()}
    return logits.numpy()


def collate_zero_pad(seqs):
    """``collate_fn`` of the reference harness (``src/dl_cv_strategies.py:81-84``): right zero-pad to the
    batch maximum, float32, NO mask."""
    T = max(s.shape[0] for s in seqs)
    out = np.zeros((len(seqs), T, seqs[0].shape[1]), dtype=np.float32)
    for i, s in enumerate(seqs):
        out[i, :s.shape[0]] = s
    return out


def eval_outputs(logits):
    """``_eval_model`` (``src/dl_cv_strategies.py:183-194``): softmax[:, 1] and argmax."""
    z = logits - logits.max(axis=1, keepdims=True)
    p = np.exp(z)
    p = p / p.sum(axis=1, keepdims=True)
    return p[:, 1], np.argmax(logits, axis=1).astype(np.int64)
