"""Randomized parity sweep of the CNN-LSTM training step (HIP path vs oracle/cnnlstm_train_oracle.py).

    python tests/sweeps/train_fuzz.py <first_seed> <count>

Each case draws the architecture from the reference's Optuna search space (src/dl_cv_strategies.py:213-219: C in
{32, 64, 128}, H in {64, 128}, silu / gelu, dropout 0.2-0.5), a small input width or the real 768, a ragged zero-padded
batch (collate_fn), dropout masks, and compares logits, loss and every parameter gradient.  Not collected by pytest.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "golden"))
import torch  # noqa: E402
from weights import synth_state_dict  # noqa: E402

from oracle import cnnlstm_train_oracle as to  # noqa: E402
from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM  # noqa: E402

ZERO_GRAD = ("conv1.bias", "conv2.bias", "shortcut.0.bias", "attention_weights.bias")
first, count = int(sys.argv[1]), int(sys.argv[2])
worst_all = 0.0
for seed in range(first, first + count):
    rng = np.random.Generator(np.random.PCG64(seed))
    C = int(rng.choice([32, 64, 128]))
    H = int(rng.choice([64, 128]))
    D = int(rng.choice([16, 32, 64, 128, 768]))
    act = str(rng.choice(["silu", "gelu"]))
    B = int(rng.integers(2, 21))
    T = int(rng.integers(4, 160))
    p_rate = float(rng.uniform(0.2, 0.5))
    p_block = 0.2
    sd = synth_state_dict(D, C, H, seed)
    x = np.zeros((B, T, D), np.float32)
    for b in range(B):                                   # ragged lengths, right zero-padding, at least one full row
        n = T if b == 0 else int(rng.integers(2, T + 1))
        x[b, :n] = rng.standard_normal((n, D)).astype(np.float32)
    labels = rng.integers(0, 2, B)
    mk = to.make_masks(B, T, C, H, p_block, p_rate, seed + 7)
    m = CNNLSTM(input_dim=D, cnn_out_channels=C, lstm_hidden_dim=H, activation_fn=act, dropout_rate=p_rate)
    full = m.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    m.load_state_dict(full)
    m = m.to("cuda").train()
    t = lambda a: torch.from_numpy(a).to("cuda")                                  # noqa: E731
    m.forced_masks = {"res_block1": t(mk["res_block1"]), "res_block2": t(mk["res_block2"]), "lstm": [t(mk["lstm0"])], "fc": t(mk["fc"])}
    out = m(t(x))
    loss = torch.nn.CrossEntropyLoss()(out, t(labels))
    loss.backward()
    torch.cuda.synchronize()
    want = to.forward_backward(sd, x, labels, act, masks=mk, return_stages=True)
    # max_pool1d routes the gradient to the larger frame of each pair: a pair closer than float32 resolution is decided
    # by rounding, and with a non-monotonic activation the two routes differ materially (seed 9037: gap 6e-9 at 0.15)
    r1 = want["stages"]["res1"]
    pa, pb = r1[:, 0:2 * (T // 2):2], r1[:, 1:2 * (T // 2):2]
    gap = np.abs(pa - pb)
    near_tie = bool(np.any((gap > 0) & (gap < 2.4e-7 * np.maximum(np.abs(pa), 1e-3))))
    e_log = np.abs(out.detach().cpu().numpy() - want["logits"]).max() / max(np.abs(want["logits"]).max(), 1.0)
    worst, wk = 0.0, ""
    for k, p in m.named_parameters():
        if k.endswith(ZERO_GRAD):
            continue
        g = want["grads"][k]
        e = np.abs(p.grad.cpu().numpy() - g).max() / max(np.abs(g).max(), 1e-7)
        if e > worst:
            worst, wk = e, k
    if not near_tie:
        worst_all = max(worst_all, worst, e_log)
    flag = "" if max(worst, e_log) < 1e-4 else ("   (max-pool near-tie below float32 resolution: routing is precision-dependent)" if near_tie else "   <-- ABOVE 1e-4")
    print(f"seed {seed}: D={D} C={C} H={H} {act} B={B} T={T} p={p_rate:.2f}  logits {e_log:.2e}  loss {abs(loss.item() - want['loss']):.2e}  "
          f"worst grad {worst:.2e} ({wk}){flag}", flush=True)
print(f"worst over {count} cases (near-tie cases excluded): {worst_all:.3e}")
