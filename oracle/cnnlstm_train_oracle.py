"""CPU restatement of one CNN-LSTM training step of the reference harness.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): only tests/, smoke() and bench.py's cpu_baseline leg use it.

What it restates:

* ``CNNLSTM.forward`` in TRAINING mode (``src/models.py:64-76,161-193`` with ``model.train()``,
  ``src/dl_cv_strategies.py:118,241``): BatchNorm1d normalises with the statistics of the batch (biased
  variance over batch x time, zero-padded frames included: nothing is masked) and updates the running
  statistics with momentum 0.1 and the unbiased variance; the three kinds of dropout (after the first
  activation of each residual block, between the LSTM layers, before the classifier) multiply by a
  Bernoulli mask scaled by 1/(1-p).  The masks are explicit arguments here (float arrays holding 0 or
  1/(1-p)) so that the HIP path and this restatement can be driven with the same randomness.
* ``nn.CrossEntropyLoss()`` (mean over the batch) on the logits, ``loss.backward()``
  (``src/dl_cv_strategies.py:122-125,243``): gradients of every parameter.  The graph is written with torch
  CPU float64 functional ops and differentiated by autograd; ``lstm_layer`` is an explicit cell loop
  (gate order i, f, g, o), not ``nn.LSTM``.
* ``torch.optim.Adam(model.parameters(), lr)`` with its defaults (``src/dl_cv_strategies.py:236``):
  ``adam_step`` below is the published update (bias-corrected first/second moments, eps outside the square
  root, no weight decay, no amsgrad).

Pinned: ``tests/golden/cnnlstm_train_*.npz`` hold loss, gradients, updated BN buffers and the parameters after
one Adam step captured from the reference module itself with its dropout probabilities set to 0
(``tests/golden/make_cnnlstm_train_golden.py``); ``tests/test_cnnlstm_train_oracle.py`` checks this file against
them.  Dropout with p > 0 is pinned only in distribution (PyTorch's Philox stream is not reproduced).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _t(a):
    import torch
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def n_lstm_layers(sd):
    n = 0
    while f"lstm.weight_ih_l{n}" in sd:
        n += 1
    return n


def mask_shapes(B, T, C, H, layers=2):
    """Shapes of the dropout masks of one step, in the order the forward consumes them."""
    Tp = T // 2
    sh = {"res_block1": (B, T, C), "res_block2": (B, Tp, C)}
    for l in range(layers - 1):
        sh[f"lstm{l}"] = (B, Tp, 2 * H)
    sh["fc"] = (B, 2 * H)
    return sh


def make_masks(B, T, C, H, p_block, p_rate, seed, layers=2):
    """Bernoulli keep masks scaled by 1/(1-p) from numpy PCG64 (float32 values 0 or 1/(1-p))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, shape in mask_shapes(B, T, C, H, layers).items():
        p = p_block if k.startswith("res_block") else p_rate
        keep = rng.random(shape) >= p
        out[k] = (keep / (1.0 - p)).astype(np.float32) if p > 0 else np.ones(shape, np.float32)
    return out


def _bn_train(h, g, b, stats, name):
    """h [B, C, T]: batch statistics over (B, T); records (mean, biased var, count) under ``name``."""
    mean = h.mean(dim=(0, 2))
    var = h.var(dim=(0, 2), unbiased=False)
    stats[name] = (mean.detach().numpy().copy(), var.detach().numpy().copy(), h.shape[0] * h.shape[2])
    import torch
    return (h - mean[None, :, None]) / torch.sqrt(var[None, :, None] + BN_EPS) * g[None, :, None] + b[None, :, None]


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [B, T, In] -> h [B, T, H]; zero initial state; gate order i, f, g, o."""
    import torch
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros((B, H))
    c = x.new_zeros((B, H))
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = x[:, t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
        i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def forward_backward(sd, x, labels, activation_fn="silu", masks=None, grad_logits=None, return_stages=False):
    """One training-mode forward + backward in float64.

    sd: reference-format state_dict (numpy); x [B, T, D]; labels int [B] (ignored when ``grad_logits`` is given,
    which is then the upstream gradient of the logits).  Returns a dict with ``logits``, ``loss``, ``grads``
    (state_dict keys -> numpy float64) and ``bn_stats`` (BN module path -> (mean, biased var, count)).
    """
    import torch
    import torch.nn.functional as F
    act = {"silu": F.silu, "gelu": F.gelu}.get(activation_fn)
    if act is None:
        raise ValueError(f"Unsupported activation function: {activation_fn}")
    P = {k: _t(v).requires_grad_(True) for k, v in sd.items()
         if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))}
    masks = masks or {}
    m = lambda k: _t(masks[k]) if k in masks else None                      # noqa: E731
    stats, st = {}, {}

    def block(h, p):
        o = act(_bn_train(F.conv1d(h, P[p + ".conv1.weight"], P[p + ".conv1.bias"], padding=1),
                          P[p + ".bn1.weight"], P[p + ".bn1.bias"], stats, p + ".bn1"))
        mk = m(p)
        if mk is not None:
            o = o * mk.permute(0, 2, 1)
        o = _bn_train(F.conv1d(o, P[p + ".conv2.weight"], P[p + ".conv2.bias"], padding=1),
                      P[p + ".bn2.weight"], P[p + ".bn2.bias"], stats, p + ".bn2")
        if (p + ".shortcut.0.weight") in P:
            s = _bn_train(F.conv1d(h, P[p + ".shortcut.0.weight"], P[p + ".shortcut.0.bias"]),
                          P[p + ".shortcut.1.weight"], P[p + ".shortcut.1.bias"], stats, p + ".shortcut.1")
        else:
            s = h
        return act(o + s)

    h = _t(x).permute(0, 2, 1)
    h = block(h, "res_block1")
    st["res1"] = h.permute(0, 2, 1)
    h = F.max_pool1d(h, kernel_size=2)
    h = block(h, "res_block2")
    st["res2"] = h.permute(0, 2, 1)
    seq = h.permute(0, 2, 1)
    nl = n_lstm_layers(sd)
    for l in range(nl):
        outs = [lstm_layer(seq, P[f"lstm.weight_ih_l{l}{s}"], P[f"lstm.weight_hh_l{l}{s}"],
                           P[f"lstm.bias_ih_l{l}{s}"], P[f"lstm.bias_hh_l{l}{s}"], rev)
                for s, rev in (("", False), ("_reverse", True))]
        seq = torch.cat(outs, dim=2)
        if l < nl - 1 and m(f"lstm{l}") is not None:
            seq = seq * m(f"lstm{l}")
    st["lstm"] = seq
    sc = F.linear(seq, P["attention_pooling.attention_weights.weight"], P["attention_pooling.attention_weights.bias"])
    ctx = torch.sum(seq * F.softmax(sc, dim=1), dim=1)
    st["pooled"] = ctx
    if m("fc") is not None:
        ctx = ctx * m("fc")
    logits = F.linear(ctx, P["fc.weight"], P["fc.bias"])
    if grad_logits is not None:
        loss = None
        logits.backward(_t(grad_logits))
    else:
        loss = F.cross_entropy(logits, torch.as_tensor(np.asarray(labels), dtype=torch.long))
        loss.backward()
    out = {"logits": logits.detach().numpy(), "loss": None if loss is None else loss.item(),
           "grads": {k: (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in P.items()},
           "bn_stats": stats}
    if return_stages:
        out["stages"] = {k: v.detach().numpy() for k, v in st.items()}
    return out


def updated_bn_buffers(sd, bn_stats):
    """Running statistics after the step: (1-m)*old + m*batch, with the UNBIASED batch variance."""
    new = {}
    for name, (mean, var, n) in bn_stats.items():
        new[name + ".running_mean"] = (1 - BN_MOMENTUM) * np.asarray(sd[name + ".running_mean"], np.float64) + BN_MOMENTUM * mean
        unb = var * (n / (n - 1.0)) if n > 1 else var
        new[name + ".running_var"] = (1 - BN_MOMENTUM) * np.asarray(sd[name + ".running_var"], np.float64) + BN_MOMENTUM * unb
    return new


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults, in place on float64 numpy dicts; ``state`` holds step / exp_avg / exp_avg_sq."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, g in grads.items():
        m = state.setdefault("m", {}).setdefault(k, np.zeros_like(g))
        v = state.setdefault("v", {}).setdefault(k, np.zeros_like(g))
        m *= b1
        m += (1 - b1) * g
        v *= b2
        v += (1 - b2) * g * g
        denom = np.sqrt(v) / np.sqrt(1 - b2 ** t) + eps
        params[k] = params[k] - (lr / (1 - b1 ** t)) * m / denom
    return params
